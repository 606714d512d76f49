"""Frame pipelining for ONE video sequence (execution level, results unchanged).

The layers of a converted network form a chain, but consecutive FRAMES only meet in each layer's own state:
layer i of frame t+1 needs layer i of frame t and layer i-1 of frame t+1, nothing else.  `FramePipeline` cuts
an nn.Sequential in two stages and runs stage 2 (the deep, expensive layers) of frame t on a side HIP stream
while stage 1 of frame t+1 runs on the caller's stream -- the small kernels, detection latencies and launch
ramps of one stage fill the idle CUs of the other, which is what several concurrent sequences do for
throughput (bench.py --sequences), here for a single one.  Frames stay in order, every module sees exactly the
calls it would see serially, so outputs and states are those of the serial execution.

The tensor handed from stage 1 to stage 2 is private to its frame (a lazily pooled tensor is materialised
densely at the cut, anything else is cloned), so stage 1 never has to wait for stage 2; its memory is tied
to the side stream with record_stream().  The reference has nothing of the kind (single stream, one blocking
sync per layer, conv2d_cg.py:202).
"""
import torch

from .conv2d import LazyPool
from .streams import side_stream


class FramePipeline(object):
    def __init__(self, net, cut, side_stream=None):
        """side_stream: the HIP stream of stage 2 (default: a new one on first use).  HIP maps streams onto
        its few hardware queues in creation order; a side stream that shares a queue with the caller's stream
        does not overlap with it, so a process that creates many streams should create this one early."""
        mods = list(net.children())
        assert 0 < cut < len(mods), "cut must leave at least one module on each side"
        self.stage1, self.stage2 = mods[:cut], mods[cut:]
        # a 1x1 tail that rides in its producer's second launch (CBConv2d._folded_tail) would be evaluated by stage 1
        # into a stage-2 module's state: where the cut separates the two, the tail keeps its own launch
        # (the switch belongs to THIS pipeline: close() -- or dropping the pipeline -- hands the folding back)
        self._unfolded = []
        for m in self.stage1:
            tail = m.__dict__.get('_fusedTail')
            if tail is not None and any(tail is k for k in self.stage2) and not m.__dict__.get('_noTailFold'):
                m.__dict__['_noTailFold'] = True
                self._unfolded.append(m)
        # likewise a producer whose launch would do the pooled change detection of a stage-2 layer
        # (pycbinfer.fuseDetectionIntoProducer): stage 2 receives a densely pooled private tensor and detects itself
        self._nofold = []
        for m in self.stage1:
            link = m.__dict__.get('_fusedNext')
            if link is not None and any(link[1] is k for k in self.stage2) and not m.__dict__.get('_noNextFold'):
                m.__dict__['_noNextFold'] = True
                self._nofold.append(m)
        self.side = side_stream
        self._done = None       # event: stage 2 of the most recently submitted frame

    def close(self):
        """Give the network back as it was: heads whose 1x1 tail this pipeline kept in a launch of its own fold it
        into their second launch again (the next frame goes through the general path once)."""
        for m in self._unfolded:
            m.__dict__.pop('_noTailFold', None)
        self._unfolded = []
        for m in getattr(self, '_nofold', []):
            m.__dict__.pop('_noNextFold', None)
        self._nofold = []

    def __del__(self):
        try:
            self.close()
        except Exception:       # pragma: no cover  (interpreter shutdown)
            pass

    def _private(self, h):
        """What stage 2 receives: a tensor (or tuple) no later frame's stage 1 will write to."""
        if isinstance(h, LazyPool):
            return h.tensor()                       # dense pooling: a fresh tensor per frame
        if isinstance(h, tuple):
            idx = h[2].clone() if hasattr(h[2], 'clone') else h[2]
            return (h[0], h[1].clone(), idx)
        return h.clone()

    @staticmethod
    def _handed(h):
        if not isinstance(h, tuple):
            return [h]
        out = [h[1]]
        idx = h[2]
        if isinstance(idx, torch.Tensor):
            out.append(idx)
        else:                       # ChangeIndexes (a clone: plain buffer + count)
            out += [idx.buffer, idx.count]
        return out

    def submit(self, frame):
        """Enqueue one frame; returns the network output, valid once wait() (or a sync) has passed."""
        cur = torch.cuda.current_stream(frame.device)
        if self.side is None:
            self.side = side_stream(frame.device)      # (probed: one that overlaps with the caller's)
        with torch.no_grad():
            h = frame
            for m in self.stage1:
                h = m(h)
            h = self._private(h)
            ready = torch.cuda.Event()
            ready.record(cur)
            # EVERY tensor handed over was allocated on the caller's stream and is read by stage 2 on the side
            # stream: the boundary tensor and, with a ('changeIndexes', tensor, indexes) tuple, the cloned index
            # buffer and its device-side count -- otherwise the next frame's stage 1 may be handed the same
            # memory by the caching allocator while stage 2 still reads it
            for t in self._handed(h):
                t.record_stream(self.side)
            self.side.wait_event(ready)
            with torch.cuda.stream(self.side):
                y = h
                for m in self.stage2:
                    y = m(y)
                self._done = torch.cuda.Event()
                self._done.record(self.side)
        return y

    def wait(self):
        """Make the caller's stream wait for everything submitted so far."""
        if self._done is not None:
            torch.cuda.current_stream().wait_event(self._done)

    def __call__(self, frame):
        y = self.submit(frame)
        self.wait()
        return y
