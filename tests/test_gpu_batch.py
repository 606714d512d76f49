"""-m gpu tests of pycbinfer.SequenceBatch: several independent sequences through one converted network with ONE
launch per step for all of them.  The bar: per sequence, outputs and every layer state are
BIT-IDENTICAL to running that sequence alone through its own copy of the network (same kernels, same tile
arithmetic; only the enumeration of the work items differs)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    import pycbinfer
    assert torch.cuda.is_available()
    return pycbinfer


def _nets(pkg, n, size):
    import bench
    from cbinfer_amd import workloads
    nets = []
    for _ in range(n):
        _, t = bench.build_bench_model()
        nets.append(t)
    return nets


def _states(pkg, net):
    out = []
    for m in net.children():
        if type(m) is pkg.CBConv2d:
            out.append((m.prevInput, m.prevOutput))
        elif type(m) is pkg.CBTail1x1:
            out.append((None, m.prevOutput))
    return out


@pytest.mark.parametrize("S,H,W", [(4, 160, 240), (3, 96, 160), (8, 64, 96), (4, 320, 480)])
def test_batch_is_bit_identical_to_independent_sequences(pkg, S, H, W):
    from cbinfer_amd import workloads
    nets = _nets(pkg, S, (H, W))
    batch = pkg.SequenceBatch(_nets(pkg, 1, (H, W))[0], S)
    # different change ratios per sequence, one of them static after its first frame, one changing a lot
    ratios = [0.10, 0.0, 0.5, 0.25, 0.10, 0.05, 0.15, 0.35][:S]
    blk = 32 if H % 32 == 0 and W % 32 == 0 else 16
    vids = [workloads.SyntheticVideo(H=H, W=W, ratio=r, block=blk, seed=40 + q) for q, r in enumerate(ratios)]
    with torch.no_grad():
        for t in range(6):
            frames = [v.frame if t == 0 else v.next() for v in vids]
            outs = batch([f.contiguous() for f in frames])
            for q in range(S):
                y = nets[q](frames[q])
                assert torch.equal(outs[q], y), (t, q)
                for (pi, po), (bi, bo) in zip(_states(pkg, nets[q]), batch.states(q)):
                    assert torch.equal(po, bo), (t, q)
                    if pi is not None:
                        assert torch.equal(pi, bi), (t, q)
    assert not batch.rangeExceeded()
    # the static sequence really had nothing to do, the others did
    if S > 1:
        assert batch.changeCounts(1) == [0, 0] and min(batch.changeCounts(0)) > 0


def test_batch_under_graph_replay_and_threshold_change(pkg):
    from cbinfer_amd import workloads
    S, H, W = 4, 160, 240
    nets = _nets(pkg, S, (H, W))
    shared = _nets(pkg, 1, (H, W))[0]
    batch = pkg.SequenceBatch(shared, S)
    vids = [workloads.SyntheticVideo(H=H, W=W, ratio=0.1, block=16, seed=70 + q) for q in range(S)]
    static = [v.frame.clone() for v in vids]
    with torch.no_grad():
        batch(static)
        for q in range(S):
            nets[q](vids[q].frame)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            batch(static)
        torch.cuda.current_stream().wait_stream(side)
        for q in range(S):
            nets[q](vids[q].frame)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            outs = batch(static)
        for q in range(S):
            nets[q](vids[q].frame)
        for t in range(4):
            for q in range(S):
                static[q].copy_(vids[q].next())
            g.replay()
            torch.cuda.synchronize()
            for q in range(S):
                assert torch.equal(outs[q], nets[q](vids[q].frame)), (t, q)
        # eager again, with a LOWERED threshold on the deep layers: the producer-mask shortcut must be off for a frame
        for net in nets + [shared]:
            convs = [m for m in net.children() if type(m) is pkg.CBConv2d]
            convs[1].threshold = convs[2].threshold = 0.01
        same = [v.frame for v in vids]
        outs = batch(same)
        for q in range(S):
            assert torch.equal(outs[q], nets[q](vids[q].frame)), q


def test_batch_refuses_what_it_cannot_run(pkg):
    from cbinfer_amd import workloads
    from cbinfer_amd._lib import CBinferError
    _, plain = workloads.sceneLabelingModels(experimentIdx=6, threshold=0.05)      # change-based pool launches
    with pytest.raises(CBinferError):
        pkg.SequenceBatch(plain, 2)([torch.rand(1, 3, 64, 96, device="cuda")] * 2)
    with pytest.raises(CBinferError):
        pkg.SequenceBatch(plain, 9)
    import bench
    _, net = bench.build_bench_model()
    b = pkg.SequenceBatch(net, 2)
    with pytest.raises(CBinferError):
        b([torch.rand(1, 3, 64, 96, device="cuda")])                               # one frame for two sequences


def test_batch_notices_written_weights(pkg):
    """Parameters written between steps (every layer kind: row-segment, split-state, the folded tail): the batch
    re-prepares its copies and keeps the sequences' states -- still bit-identical to independent networks whose
    parameters were written the same way."""
    from cbinfer_amd import workloads
    S, H, W = 2, 96, 160
    nets = _nets(pkg, S, (H, W))
    bnet = _nets(pkg, 1, (H, W))[0]
    batch = pkg.SequenceBatch(bnet, S)
    vids = [workloads.SyntheticVideo(H=H, W=W, ratio=0.15, block=16, seed=70 + q) for q in range(S)]

    def params(net):
        out = []
        for m in net.children():
            if type(m) is pkg.CBConv2d:
                out += [m.weight, m.bias]
            elif type(m) is pkg.CBTail1x1:
                out += [m.weight1, m.bias1, m.weight2, m.bias2]
        return out

    with torch.no_grad():
        for t in range(8):
            if t in (3, 5):
                for net in nets + [bnet]:
                    for k, p in enumerate(params(net)):
                        p.mul_(1.0 + 0.01 * (k + t))
            frames = [v.frame if t == 0 else v.next() for v in vids]
            outs = batch([f.contiguous() for f in frames])
            for q in range(S):
                assert torch.equal(outs[q], nets[q](frames[q])), (t, q)
                for (pi, po), (bi, bo) in zip(_states(pkg, nets[q]), batch.states(q)):
                    assert torch.equal(po, bo), (t, q)


def test_two_batches_on_two_streams(pkg):
    """What bench.py's `grouped_value` runs: two SequenceBatch groups of two sequences, each on its own stream, their
    steps enqueued back to back without any sync in between -- every sequence bit-identical to its own network."""
    from cbinfer_amd import workloads
    H, W, T = 160, 240, 8
    nets = _nets(pkg, 4, (H, W))
    groups = [(pkg.SequenceBatch(_nets(pkg, 1, (H, W))[0], 2), torch.cuda.Stream()) for _ in range(2)]
    vids = [workloads.SyntheticVideo(H=H, W=W, ratio=r, block=16, seed=90 + q)
            for q, r in enumerate((0.1, 0.3, 0.05, 0.2))]
    frames = [[v.frame] + [v.next() for _ in range(T - 1)] for v in vids]
    torch.cuda.synchronize()
    got = [[None] * T for _ in range(4)]
    with torch.no_grad():
        for t in range(T):
            for gi, (sb, st) in enumerate(groups):
                with torch.cuda.stream(st):
                    outs = sb([frames[2 * gi][t], frames[2 * gi + 1][t]])
                    for k in range(2):
                        got[2 * gi + k][t] = outs[k].clone()
        torch.cuda.synchronize()
        for q in range(4):
            for t in range(T):
                assert torch.equal(got[q][t], nets[q](frames[q][t])), (q, t)
