"""-m gpu tests of round 6's two forms of an fp16 frame (cb_split.hip: cbinfer_hsplit_forward_group):
(1) the change detection of a layer's CONSUMERS inside the producing layer's contraction launch -- the reference chains
    layers through their state tensors (conv2d.py:259: the returned tensor IS prevOutput; :180-186, :256-259), a consumer
    in copy mode (conv2d.py:234-236) compares that buffer value by value (cbconv2d_cg_half_backend.cu:24-35) --, and
(2) two layers of one geometry in one launch (the two branches of an OpenPose stage, PoseModel.py:122-137).
Both are execution forms: every buffer must equal, bit for bit, what the separate launches leave."""
import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    import pycbinfer
    assert torch.cuda.is_available()
    return pycbinfer


def video(rng, C, H, W, n, frac, blk=6):
    x = rng.standard_normal((1, C, H, W)).astype(np.float16)
    out = [x.copy()]
    for t in range(n - 1):
        x = x.copy()
        f = frac[t % len(frac)] if isinstance(frac, (list, tuple)) else frac
        if f >= 1.0:
            x = rng.standard_normal((1, C, H, W)).astype(np.float16)
        else:
            for _ in range(int(round(f * H * W / blk / blk))):
                y0, x0 = rng.integers(0, H - blk), rng.integers(0, W - blk)
                x[0, :, y0:y0 + blk, x0:x0 + blk] = rng.standard_normal((C, blk, blk)).astype(np.float16)
        out.append(x)
    return out


def chain(pkg, spec, seed, threshold=0.05, scale=None):
    """nn.Sequential of Conv2d(+ReLU) per (C_in, C_out, k), fp16, converted; kaiming weights so that every layer is alive."""
    torch.manual_seed(seed)
    layers = []
    for i, (ci, co, k) in enumerate(spec):
        conv = nn.Conv2d(ci, co, k, padding=k // 2)
        nn.init.kaiming_normal_(conv.weight, nonlinearity='relu')
        nn.init.normal_(conv.bias, std=0.1)
        layers += [conv, nn.ReLU()] if i < len(spec) - 1 else [conv]
    return pkg.convert(nn.Sequential(*layers).cuda().half().eval(), threshold=threshold)


SPECS = {
    # shallow producers on 64x64 and 128x128 tiles, a deep one (7x7 on 128: 98 k-stages), output channels off the
    # 64-channel grid (100 -> the consumer's records are padded to 128), 1x1 layers of two and eight k-stages
    "mixed": [(64, 128, 3), (128, 128, 7), (128, 100, 3), (100, 128, 3), (128, 512, 1), (512, 38, 1)],
    # deep producers only (3x3 on 512: 72 stages; 7x7 on 128), a 1x1 consumer behind a deep producer
    "deep": [(64, 512, 3), (512, 512, 3), (512, 128, 3), (128, 128, 7), (128, 128, 7), (128, 128, 1), (128, 19, 1)],
}


@pytest.mark.parametrize("name,H,W", [("mixed", 46, 81), ("deep", 23, 40), ("mixed", 92, 163)])
def test_consumer_detection_in_the_producing_launch_is_bit_identical(pkg, name, H, W, monkeypatch):
    """A chain of fp16 CBConv2d layers with the consumers' detection folded into the producers' launches
    (pycbinfer.fuseDetectionIntoProducer) against the same chain with every layer running its own detection launch:
    per frame and layer the change list, prevInput and prevOutput bit-identical -- over frames of 5-15 % change, a frame
    that changes EVERYTHING (the deep contractions then run unsplit: the detection rides in the contraction's own
    epilogue), and two idle frames -- and the folding really runs from the third frame on."""
    spec = SPECS[name]
    a = chain(pkg, spec, 3)
    b = chain(pkg, spec, 3)
    pkg.fuseDetectionIntoProducer(a)
    rng = np.random.default_rng(H * W)
    frac = [0.1, 0.05, 0.15, 1.0, 0.1, 0.0, 0.0, 0.08, 0.1]
    frames = video(rng, spec[0][0], H, W, 12, frac)
    ca = [m for m in a.modules() if type(m) is pkg.CBConv2d]
    cb = [m for m in b.modules() if type(m) is pkg.CBConv2d]
    folded = np.zeros(len(ca), dtype=int)
    with torch.no_grad():
        for t, f in enumerate(frames):
            x = torch.from_numpy(f).cuda()
            ya, yb = a(x), b(x)
            torch.cuda.synchronize()
            for i, (ma, mb) in enumerate(zip(ca, cb)):
                assert torch.equal(ma.lastChangeIndexes().tensor(), mb.lastChangeIndexes().tensor()), (t, i)
                assert torch.equal(ma.prevInput, mb.prevInput), (t, i)
                assert torch.equal(ma.prevOutput, mb.prevOutput), (t, i)
                hs = ma._work.get('hsplit')
                if hs is not None and hs['layer'][0].detect == 0:
                    folded[i] += 1
            assert torch.equal(ya, yb), t
    # every layer behind the first one had its detection done by its producer in the steady state
    assert folded[0] == 0 and all(n >= len(frames) - 3 for n in folded[1:]), folded
    assert all(m._plan is not None and m._plan.get('hsplit') for m in ca)
    assert all(m._work['hsplit']['layer'][0].detect == 1 for m in cb)
    # ... and something was there to detect: the layers are alive
    assert all(m.lastChangeIndexes().numel() > 0 for m in ca)


def test_two_consumers_with_their_own_thresholds(pkg):
    """One producer, two consumers of its output (the first layers of the two branches of an OpenPose stage,
    PoseModel.py:122-137) with different thresholds and filter sizes: both detections in the producer's launch
    (pycbinfer.linkConsumers) equal the consumers' own."""
    H, W = 46, 81

    def build(link):
        torch.manual_seed(11)
        prod = chain(pkg, [(64, 128, 3), (128, 128, 3)], 5)
        b1 = chain(pkg, [(128, 128, 3), (128, 38, 1)], 6, threshold=0.03)
        b2 = chain(pkg, [(128, 128, 7), (128, 19, 1)], 7, threshold=0.2)
        if link:
            for seq in (prod, b1, b2):
                pkg.fuseDetectionIntoProducer(seq)
            last = [m for m in prod.modules() if type(m) is pkg.CBConv2d][-1]
            pkg.linkConsumers(last, [next(iter(b1.children())), next(iter(b2.children()))])
        return prod, b1, b2
    A, B = build(True), build(False)
    rng = np.random.default_rng(5)
    frames = video(rng, 64, H, W, 8, 0.1)
    folded = [0, 0]
    with torch.no_grad():
        for t, f in enumerate(frames):
            x = torch.from_numpy(f).cuda()
            outs = []
            for prod, b1, b2 in (A, B):
                feat = prod(x)
                outs.append((b1(feat), b2(feat)))
            torch.cuda.synchronize()
            assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), t
            for k in (1, 2):
                ma, mb = next(iter(A[k].children())), next(iter(B[k].children()))
                assert torch.equal(ma.prevInput, mb.prevInput) and torch.equal(ma.prevOutput, mb.prevOutput), (t, k)
                assert torch.equal(ma.lastChangeIndexes().tensor(), mb.lastChangeIndexes().tensor()), (t, k)
                folded[k - 1] += int(ma._work['hsplit']['layer'][0].detect == 0)
    assert folded[0] >= 5 and folded[1] >= 5, folded
    last = [m for m in A[0].modules() if type(m) is pkg.CBConv2d][-1]
    assert last._work['hsplit']['layer'][0].nNext == 2


def test_branch_pairs_in_one_launch_are_bit_identical(pkg):
    """pycbinfer.BranchGroup: two branches of one geometry (PoseModel.py:122-137: 38 and 19 output maps) walked in
    lockstep, every layer pair ONE cbinfer_hsplit_forward_group call -- shallow 3x3, deep 7x7 (split along k: one reduce
    launch for both), 1x1 layers, last layers of different channel counts (both pad to 64 rows) -- against the same
    branches run one after the other: outputs, states and change lists bit-identical in every frame; the producers'
    folded detections work across the grouped launches too."""
    H, W = 46, 81

    def build():
        prod = chain(pkg, [(64, 128, 3), (128, 128, 3)], 5)
        spec = [(128, 128, 3), (128, 128, 7), (128, 128, 7), (128, 128, 1)]
        b1 = chain(pkg, spec + [(128, 38, 1)], 6, threshold=0.03)
        b2 = chain(pkg, spec + [(128, 19, 1)], 7, threshold=0.06)
        for seq in (prod, b1, b2):
            pkg.fuseDetectionIntoProducer(seq)
        last = [m for m in prod.modules() if type(m) is pkg.CBConv2d][-1]
        pkg.linkConsumers(last, [next(iter(b1.children())), next(iter(b2.children()))])
        return prod, b1, b2
    A, B = build(), build()
    group = pkg.BranchGroup([A[1], A[2]])
    calls = []
    from cbinfer_amd import _lib, branches
    real = _lib.C.cbinfer_hsplit_forward_group

    class Spy(object):      # (counts the grouped calls: the library object's attributes are read-only function pointers)
        def __getattr__(self, name):
            return getattr(_lib.C, name)

        def cbinfer_hsplit_forward_group(self, layers, n, *a):
            calls.append(n)
            return real(layers, n, *a)
    rng = np.random.default_rng(9)
    frames = video(rng, 64, H, W, 9, [0.1, 0.1, 0.05, 1.0, 0.1, 0.0, 0.1, 0.1])
    branches.C = Spy()
    try:
        with torch.no_grad():
            for t, f in enumerate(frames):
                x = torch.from_numpy(f).cuda()
                ya = group(A[0](x))
                fb = B[0](x)
                yb = [B[1](fb), B[2](fb)]
                torch.cuda.synchronize()
                for k in (0, 1):
                    assert torch.equal(ya[k], yb[k]), (t, k)
                    for ma, mb in zip(A[1 + k].children(), B[1 + k].children()):
                        assert torch.equal(ma.prevInput, mb.prevInput) and torch.equal(ma.prevOutput, mb.prevOutput), (t, k)
                        assert torch.equal(ma.lastChangeIndexes().tensor(), mb.lastChangeIndexes().tensor()), (t, k)
    finally:
        branches.C = _lib.C
    # from the second frame on (the first builds the call plans) every layer pair is one call of two layers
    assert calls == [2] * (5 * (len(frames) - 1)), calls
    assert all(m._work['hsplit']['layer'][0].detect == 0 for seq in (A[1], A[2]) for m in seq.children())


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_channel_concat_is_torch_cat(pkg, dtype):
    """pycbinfer.ChannelConcat (cbinfer_concat_channels: PoseModel.py:131's torch.cat(dim=1) as one library launch into a
    buffer that keeps its address) against torch.cat: OpenPose's 38 + 19 + 128 channels at an odd map size (block
    boundaries that are not 16-byte aligned), one source, four sources."""
    cc = pkg.ChannelConcat()
    for chans, (H, W) in (((38, 19, 128), (46, 81)), ((5,), (3, 7)), ((1, 2, 3, 4), (9, 5)), ((38, 19, 128), (46, 81))):
        ts = [torch.randn(1, c, H, W, device="cuda").to(dtype) for c in chans]
        out = cc(ts)
        torch.cuda.synchronize()
        assert torch.equal(out, torch.cat(ts, 1))
    a = cc([torch.ones(1, 3, 4, 4, device="cuda", dtype=dtype)] * 2)
    b = cc([torch.zeros(1, 3, 4, 4, device="cuda", dtype=dtype)] * 2)
    assert a.data_ptr() == b.data_ptr()                      # (the buffer is kept: the address a recorded program replays)
    with pytest.raises(Exception):
        cc([torch.ones(1, 3, 4, 4, device="cuda"), torch.ones(1, 3, 4, 5, device="cuda")])


def test_openpose_as_a_recorded_launch_program(pkg):
    """A converted OpenPose T=2 network (fp16, 96x160) in the library-only execution form -- consumers' detection in the
    producers' launches, branches grouped, the three pools change-based and folded into the detections, the stage inputs
    concatenated by the library -- replayed from a recorded launch program (pycbinfer.FrameProgram) against the same network
    run module by module: both outputs and every state tensor bit-identical over a walk with idle frames.  With dense
    nn.MaxPool2d / torch.cat in the network the recording refuses."""
    from cbinfer_amd import workloads
    from cbinfer_amd._lib import CBinferError
    H, W = 96, 160

    def build():
        net = workloads.convertOpenPose(workloads.OpenPoseModel(T=2, init='kaiming', groupedBranches=True).cuda().half(),
                                        threshold=0.05)
        return workloads.fuseOpenPoseDetections(net)
    vid = workloads.SyntheticVideo(H=H, W=W, ratio=0.10, block=16, seed=21)
    frames = [(f * (255.0 / 256.0) - 0.5).half().contiguous() for f in vid.frames(12)]
    frames = frames[:8] + [frames[7], frames[7]] + frames[8:]
    a, b = build(), build()
    with torch.no_grad():
        a(frames[0])
        with pytest.raises(CBinferError):      # (dense pools and torch.cat: operators a replay would not repeat)
            pkg.FrameProgram(a).record(frames[0])
        a, b = build(), build()
        for net in (a, b):
            pkg.insertCBPooling(net, cloneOutput=False)
            pkg.fusePoolingIntoDetection(net)
            net.libraryConcat = True
        for f in frames[:4]:
            a(f), b(f)
        prog = pkg.FrameProgram(a)
        ya, yb = prog.record(frames[4]), b(frames[4])
        assert len(prog.calls) <= 25
        for t, f in enumerate(frames[5:]):
            ya, yb = prog(f), b(f)
            torch.cuda.synchronize()
            assert all(torch.equal(u, v) for u, v in zip(ya, yb)), t
    for ta, tb in zip(pkg.getStateTensors(a), pkg.getStateTensors(b)):
        assert torch.equal(ta, tb)
    assert sum(1 for m in a.modules() if type(m) is pkg.CBConv2d) == 36
