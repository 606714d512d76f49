"""frame_check.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Teacher-forced check of a change-based network AS THE BENCH RUNS IT (fused launches, lazy pooling, call plans)
against the oracle's restatement of the reference's module state machines (oracle/cb_oracle.py: CBConv2d
.forward_normal conv2d.py:178-259, CBPoolMax2d.forward conv2d.py:49-78), layer by layer with the GPU's own layer
inputs, so that every layer's change list must match bit for bit and every state within the fp32 bar.

Only tests/ and __graft_entry__.smoke() import this module.

The product network (pycbinfer: CBConv2d, lazy CBPoolMax2d, CBConv2d, lazy CBPoolMax2d, CBConv2d, CBTail1x1 or a
dense 1x1 tail) is run one WHOLE frame at a time, exactly like bench.py's timed loop does.  Afterwards the oracle
twin -- in the REFERENCE's structure (conv, change-based pool, conv, change-based pool, conv, dense 1x1 tail;
sceneLabeling/modelLoader.py:62-78) -- is stepped layer by layer on what the GPU layers saw:
  * conv i: input = the frame / the oracle pool's output computed from the GPU's previous layer output;
    asserted: change list == GPU's list (bit-exact, order included), prevInput == GPU's (bit-exact: feedback
    refresh copies values), prevOutput within `tol`;
  * pool: the oracle's change-based pool on (GPU conv output, GPU change list); asserted equal to dense
    max-pooling of the GPU conv output (the pool is exact);
  * tail: dense conv1x1 -> ReLU -> conv1x1 (double accumulation) of the GPU's last conv output vs the network's
    output within `tol`.
"""
import numpy as np

from . import cb_oracle as orc


class BenchTwin(object):
    def __init__(self, pkg, net):
        import torch.nn as nn
        self.pkg = pkg
        kids = list(net.children())
        self.convs = [m for m in kids if type(m) is pkg.CBConv2d]
        self.pools = [m for m in kids if type(m) is pkg.CBPoolMax2d]
        assert len(self.convs) == 3 and len(self.pools) == 2, "scene-labeling experiment 5/6 structure expected"
        assert all(m.feedbackLoop and not m.finegrained for m in self.convs)
        self.oconvs = [orc.OracleCBConv2d(m.weight.detach().cpu().numpy(), m.bias.detach().cpu().numpy(),
                                          m.threshold, withReLU=m.withReLU, feedbackLoop=True,
                                          propChangeIndexes=True, copyInput=m.copyInput) for m in self.convs]
        self.opools = [orc.OracleCBPoolMax2d(ceil_mode=m.ceil_mode, propChangeIndexes=False) for m in self.pools]
        tail = kids[kids.index(self.convs[-1]) + 1:]
        if len(tail) == 1 and type(tail[0]) is pkg.CBTail1x1:
            t = tail[0]
            self.tail = [(t.weight1.detach().cpu().numpy(), t.bias1.detach().cpu().numpy(), True),
                         (t.weight2.detach().cpu().numpy(), t.bias2.detach().cpu().numpy(), False)]
        else:
            self.tail = []
            for i, m in enumerate(tail):
                if type(m) is nn.Conv2d:
                    relu = i + 1 < len(tail) and type(tail[i + 1]) is nn.ReLU
                    self.tail.append((m.weight.detach().cpu().numpy(), m.bias.detach().cpu().numpy(), relu))
        self.net = net
        self.frames = 0
        self.maxerr = 0.0

    def _gpu_list(self, m):
        ci = m.lastChangeIndexes()
        return ci.tensor().cpu().numpy().copy()

    def step(self, frame, tol=1e-4):
        """Run one frame through the product network and check it.  Returns the network output (torch)."""
        import torch
        with torch.no_grad():
            y = self.net(frame)
        x = frame.detach().cpu().numpy()
        stats = []
        for i, (m, o) in enumerate(zip(self.convs, self.oconvs)):
            idx_gpu = self._gpu_list(m)
            got = o.forward(x)
            assert isinstance(got, tuple)
            idx_o = got[2]
            assert idx_o.dtype == np.int32 and np.array_equal(idx_gpu, idx_o), \
                "frame %d, conv %d: change list differs (%d vs %d entries)" % (self.frames, i, idx_gpu.size,
                                                                               idx_o.size)
            assert np.array_equal(m.prevInput.cpu().numpy(), o.prevInput), \
                "frame %d, conv %d: feedback-refreshed state differs" % (self.frames, i)
            po = m.prevOutput.cpu().numpy()
            err = float(np.abs(po - o.prevOutput).max())
            assert err <= tol, "frame %d, conv %d: |prevOutput - oracle| = %.3e" % (self.frames, i, err)
            self.maxerr = max(self.maxerr, err)
            stats.append(int(idx_gpu.size))
            if i < len(self.opools):
                pooled = self.opools[i].forward(('changeIndexes', po, idx_gpu))
                assert np.array_equal(pooled, orc.maxpool_dense(po, self.pools[i].ceil_mode)), \
                    "frame %d, pool %d: change-based pooling != dense pooling" % (self.frames, i)
                x = pooled
            else:
                x = po
        for (w, b, relu) in self.tail:
            x = orc.conv2d_dense(x, w, b, relu=relu)
        err = float(np.abs(y.cpu().numpy() - x).max())
        assert err <= tol, "frame %d: |output - oracle tail| = %.3e" % (self.frames, err)
        self.maxerr = max(self.maxerr, err)
        self.frames += 1
        self.lastN = stats
        return y
