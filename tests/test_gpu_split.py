"""-m gpu tests of the split-state kernels (cb_split.hip) through the C ABI: detection + refresh of both states,
the LDS-DMA contraction in BOTH arithmetics -- bf16 triples ("x3": f32-equivalent, the default since round 5) and f16
pairs (every test below runs once per arithmetic unless it says otherwise) --, the k-split and its invariance,
several sequences per launch.
Reference behaviour: cbconv2d_cg_backend.cu:40-81 (detection, feedback refresh), :138-197 + conv2d_cg.py:342-349
(gather, contraction, scatter) as restated by the oracle."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

FP32_TOL = 1e-4


@pytest.fixture(scope="module")
def lib():
    from cbinfer_amd import _lib
    assert torch.cuda.is_available()
    return _lib


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


ARITH = "x3"


@pytest.fixture(scope="module", autouse=True, params=["x3", "f16x2"])
def arith(request):
    """Every test of this module once per arithmetic of the split-state kernels (Layer picks it up)."""
    global ARITH
    ARITH = request.param
    yield request.param
    ARITH = "x3"


class Layer(object):
    """Buffers of one layer for nSeq sequences + the calls, as CBConv2d._forward_split makes them."""

    def __init__(self, lib, w, b, H, W, nSeq=1, pooled=None, arith=None):
        C_ = lib.C
        self.lib, self.nSeq, self.H, self.W = lib, nSeq, H, W
        self.K, self.C, self.kH, self.kW = w.shape
        self.w, self.b = dev(w), dev(b)
        self.x3 = x3 = (arith or ARITH) == "x3"
        K, Cc, kH, kW = w.shape
        if x3:      # bf16 triples: no scale (0.0 tells the frame functions which form the buffers hold)
            self.scale = 0.0
            self.wp = torch.empty(C_.cbinfer_split3_prepared_bytes(Cc, K, kH, kW), dtype=torch.uint8, device="cuda")
            lib.check(C_.cbinfer_split3_prep_weights(self.w.data_ptr(), self.wp.data_ptr(), K, Cc, kH, kW, H, W, None))
        else:
            wmax = float(np.abs(w).max())
            self.scale = 2.0 ** (13 - math.floor(math.log2(wmax)))
            self.wp = torch.empty(C_.cbinfer_split_prepared_bytes(Cc, K, kH, kW), dtype=torch.uint8, device="cuda")
            lib.check(C_.cbinfer_split_prep_weights(self.w.data_ptr(), self.wp.data_ptr(), K, Cc, kH, kW, H, W,
                                                     self.scale, None))
        words = C_.cbinfer_mask_words(H, W)
        wsb = C_.cbinfer_split_workspace_bytes(nSeq, Cc, H, W, K, kH, kW)
        self.ws = torch.zeros(wsb, dtype=torch.uint8, device="cuda") if wsb else None
        self.seqs = (lib.SplitSeq * nSeq)()
        self.state, self.S, self.masks, self.out, self.idx, self.cnt, self.copy = [], [], [], [], [], [], []
        self.flag = torch.zeros(1, dtype=torch.int32, device="cuda")
        self.pooled = pooled
        for q in range(nSeq):
            self.state.append(torch.full((1, Cc, H, W), float("inf"), device="cuda"))
            if x3:
                S = torch.empty(C_.cbinfer_split3_state_bytes(Cc, H, W, kH, kW), dtype=torch.uint8, device="cuda")
                lib.check(C_.cbinfer_split3_state_init(S.data_ptr(), Cc, H, W, kH, kW, None))
            else:
                S = torch.empty(C_.cbinfer_split_state_bytes(Cc, H, W, kH, kW), dtype=torch.uint8, device="cuda")
                lib.check(C_.cbinfer_split_state_init(S.data_ptr(), Cc, H, W, kH, kW, None))
            self.S.append(S)
            self.rebuild(q)
            self.masks.append(torch.zeros(C_.cbinfer_frame_mask_bytes(H, W) // 8, dtype=torch.int64, device="cuda"))
            self.out.append(torch.full((1, K, H, W), float("inf"), device="cuda"))
            self.idx.append(torch.zeros(H * W, dtype=torch.int32, device="cuda"))
            self.cnt.append(torch.zeros(1, dtype=torch.int32, device="cuda"))
            self.copy.append(torch.zeros(words, dtype=torch.int64, device="cuda"))
            s = self.seqs[q]
            s.state, s.splitState, s.frameMasks = self.state[q].data_ptr(), S.data_ptr(), self.masks[q].data_ptr()
            s.output, s.idxOut, s.countOut = self.out[q].data_ptr(), self.idx[q].data_ptr(), self.cnt[q].data_ptr()
            s.rangeFlag, s.maskCopy = self.flag.data_ptr(), self.copy[q].data_ptr()

    def rebuild(self, q=0):
        """The split state of sequence q made again from its f32 state."""
        C_, a = self.lib.C, (self.C, self.H, self.W, self.kH, self.kW)
        if self.x3:
            self.lib.check(C_.cbinfer_split3_state_rebuild(self.state[q].data_ptr(), self.S[q].data_ptr(), *a, None))
        else:
            self.lib.check(C_.cbinfer_split_state_rebuild(self.state[q].data_ptr(), self.S[q].data_ptr(), *a,
                                                          self.flag.data_ptr(), None))

    def frame(self, inputs, th, relu=False, force=0, prodMasks=None):
        C_ = self.lib.C
        for q, x in enumerate(inputs):
            self.seqs[q].input = x.data_ptr()
            self.seqs[q].producerMask = prodMasks[q].data_ptr() if prodMasks else None
        pH, pW = (inputs[0].shape[-2], inputs[0].shape[-1]) if self.pooled else (0, 0)
        self.lib.check(C_.cbinfer_split_detect(self.seqs, self.nSeq, int(bool(self.pooled)) | (8 if self.x3 else 0),
                                               pH, pW, self.C, self.H, self.W, self.kH, self.kW, th, None))
        self.lib.check(C_.cbinfer_split_conv(self.seqs, self.nSeq, self.wp.data_ptr(), self.b.data_ptr(), self.C,
                                             self.H, self.W, self.K, self.kH, self.kW, self.scale, int(relu),
                                             self.ws.data_ptr() if self.ws is not None else None, force, None))
        torch.cuda.synchronize()

    def list(self, q=0):
        return self.idx[q][:int(self.cnt[q].item())].cpu().numpy()


def block_video(rng, C, H, W, frames, frac, blk=8):
    x = rng.standard_normal((1, C, H, W)).astype(np.float32)
    out = [x.copy()]
    for _ in range(frames - 1):
        x = x.copy()
        for _ in range(max(1, int(frac * H * W / blk / blk))):
            y0, x0 = rng.integers(0, max(1, H - blk)), rng.integers(0, max(1, W - blk))
            bh, bw = min(blk, H - y0), min(blk, W - x0)
            x[0, :, y0:y0 + bh, x0:x0 + bw] = rng.standard_normal((C, bh, bw))
        out.append(x)
    return out


@pytest.mark.parametrize("C,K,kH,kW,H,W,frac", [
    (16, 64, 7, 7, 160, 240, 0.1), (64, 256, 7, 7, 80, 120, 0.1), (64, 256, 7, 7, 80, 120, 0.6),
    (16, 16, 3, 3, 37, 70, 0.3), (32, 40, 3, 5, 45, 67, 0.2), (64, 130, 5, 5, 33, 64, 0.15),
    (16, 64, 7, 7, 7, 9, 0.5), (32, 200, 7, 7, 21, 130, 0.05)])
def test_split_frames_vs_oracle(lib, oracle, C, K, kH, kW, H, W, frac):
    """Feedback-mode frames of one layer against the oracle's state machine (CBConv2d.forward_normal,
    conv2d.py:178-259): change list bit-exact incl. order, refreshed state bit-exact, outputs <= 1e-4, the mask
    copy equal to the dilated map; odd sizes, non-square filters, output channels off the tile grid."""
    rng = np.random.default_rng(C * 1000 + K + H)
    w = (rng.standard_normal((K, C, kH, kW)) / np.sqrt(C * kH * kW)).astype(np.float32)
    b = rng.standard_normal(K).astype(np.float32)
    L = Layer(lib, w, b, H, W)
    o = oracle.OracleCBConv2d(w, b, 0.1, withReLU=True, feedbackLoop=True, propChangeIndexes=True)
    for t, x in enumerate(block_video(rng, C, H, W, 5, frac)):
        # sub-threshold noise everywhere: must never trigger, and must not leak into the state either
        xn = (x + rng.uniform(-0.03, 0.03, x.shape)).astype(np.float32)
        L.frame([dev(xn)], 0.1, relu=True)
        got = o.forward(xn)
        assert np.array_equal(L.list(), got[2]), t
        assert np.array_equal(L.state[0].cpu().numpy(), o.prevInput), t
        err = np.abs(L.out[0].cpu().numpy() - o.prevOutput).max()
        assert err <= FP32_TOL, (t, err)
        words = L.copy[0].cpu().numpy().view(np.uint64)
        wpr = (W + 63) // 64
        bits = np.unpackbits(words.view(np.uint8), bitorder="little").reshape(H, wpr * 64)[:, :W]
        assert np.array_equal(bits.astype(np.int8), o.changeMap), t
    assert int(L.flag.item()) == 0
    # a repeated frame changes nothing and leaves an empty list
    L.frame([dev(xn)], 0.1, relu=True)
    assert L.list().size == 0


@pytest.mark.parametrize("C,K,H,W", [(64, 256, 40, 60), (16, 64, 80, 120), (32, 128, 33, 47)])
def test_split_arithmetic_accuracy(lib, oracle, C, K, H, W):
    """The f16-pair arithmetic (x 2^-4 = hi + lo 2^-11, three MFMA products per multiply, f32 accumulation)
    against the double-accumulated oracle on operands spanning many binades: every element within
    64 * 2^-24 * sum|a||b| -- the bound tests/test_gpu_ops.py::test_split_contraction_accuracy holds the bf16x3 form
    to -- and the worst relative error reported (per product the analysis gives <= 3 * 2^-22)."""
    rng = np.random.default_rng(C + K)
    x = (rng.standard_normal((1, C, H, W)) * np.exp(rng.uniform(-6, 3, (1, C, 1, 1)))).astype(np.float32)
    x[0, 0, :4] *= 1e-6                                     # values deep in the f16 subnormal range of hi AND lo
    w = (rng.standard_normal((K, C, 7, 7)) / np.sqrt(C * 49) * np.exp(rng.uniform(-4, 2, (K, 1, 1, 1)))).astype(np.float32)
    b = rng.standard_normal(K).astype(np.float32)
    L = Layer(lib, w, b, H, W)
    L.frame([dev(x)], 0.1)                                   # first frame: every pixel
    idx = np.arange(H * W, dtype=np.int32)
    assert np.array_equal(L.list(), idx)
    X = oracle.genXMatrix(x, idx, (7, 7))
    Y = oracle.matrixMult(X, w, b).T.reshape(1, K, H, W)
    mag = (np.abs(X).astype(np.float64) @ np.abs(w.reshape(K, -1)).astype(np.float64).T).T.reshape(1, K, H, W)
    err = np.abs(L.out[0].cpu().numpy().astype(np.float64) - Y)
    rel = (err / (mag + 1e-30)).max()
    print("%s C%d K%d: max |err| %.3g, relative to sum|a||b| %.3g (2^%.1f)" % (ARITH, C, K, err.max(), rel, np.log2(rel)))
    assert np.all(err <= 64 * 2.0 ** -24 * mag + 1e-30)
    assert rel <= 8 * 2.0 ** -22
    if L.x3:      # exact operands; what is left is one rounding per 16-k block sum + the dropped terms (< 2^-25 each)
        assert rel <= 2.0 ** -21


def test_split_is_invariant_under_the_k_split(lib, oracle):
    """A deep contraction is the left-to-right sum of four partial sums over fixed k-ranges whether four workgroups
    compute them (slabs + reduce launch) or one does, one after the other: bit-identical outputs."""
    rng = np.random.default_rng(5)
    C, K, H, W = 64, 256, 40, 60
    w = (rng.standard_normal((K, C, 7, 7)) / np.sqrt(C * 49)).astype(np.float32)
    b = rng.standard_normal(K).astype(np.float32)
    vid = block_video(rng, C, H, W, 4, 0.2)
    outs = {}
    for force in (1, 4):
        L = Layer(lib, w, b, H, W)
        for x in vid:
            L.frame([dev(x)], 0.05, relu=True, force=force)
        outs[force] = L.out[0].clone()
    assert torch.equal(outs[1], outs[4])
    Ld = Layer(lib, w, b, H, W)              # (the kernel's own choice: split while the tiles are few)
    for x in vid:
        Ld.frame([dev(x)], 0.05, relu=True)
    assert torch.equal(Ld.out[0], outs[1])


@pytest.mark.parametrize("C,K,H,W", [(16, 64, 64, 96), (64, 256, 40, 60)])
def test_split_sequences_in_one_launch(lib, oracle, C, K, H, W):
    """nSeq sequences per launch: each one bit-identical to its own single-sequence run, whatever the others do
    (an unchanged one, a fully changing one)."""
    rng = np.random.default_rng(9)
    w = (rng.standard_normal((K, C, 7, 7)) / np.sqrt(C * 49)).astype(np.float32)
    b = rng.standard_normal(K).astype(np.float32)
    S = 5
    vids = [block_video(rng, C, H, W, 4, f) for f in (0.1, 0.0, 1.0, 0.3, 0.02)]
    vids[1] = [vids[1][0]] * 4                               # a static sequence
    Lb = Layer(lib, w, b, H, W, nSeq=S)
    singles = [Layer(lib, w, b, H, W) for _ in range(S)]
    for t in range(4):
        Lb.frame([dev(v[t]) for v in vids], 0.05, relu=True)
        for q in range(S):
            singles[q].frame([dev(vids[q][t])], 0.05, relu=True)
            assert np.array_equal(Lb.list(q), singles[q].list()), (t, q)
            assert torch.equal(Lb.out[q], singles[q].out[0]), (t, q)
            assert torch.equal(Lb.state[q], singles[q].state[0]), (t, q)
            assert torch.equal(Lb.copy[q], singles[q].copy[0]), (t, q)
    assert Lb.list(1).size == 0


def test_split_pooled_detection_and_producer_mask(lib, oracle):
    """The 2x2 pool folded into the detection (odd sizes, floor mode), with and without the producer-mask
    shortcut: same list, same states as detection on the densely pooled tensor."""
    rng = np.random.default_rng(11)
    C, K, pH, pW = 16, 32, 45, 67
    H, W = pH // 2, pW // 2
    w = (rng.standard_normal((K, C, 3, 3)) / np.sqrt(C * 9)).astype(np.float32)
    b = rng.standard_normal(K).astype(np.float32)
    La, Lb, Lc = Layer(lib, w, b, H, W, pooled=True), Layer(lib, w, b, H, W), Layer(lib, w, b, H, W, pooled=True)
    vid = block_video(rng, C, pH, pW, 4, 0.1, blk=6)
    prev = None
    for t, x in enumerate(vid):
        xd = dev(x)
        pooled = torch.nn.functional.max_pool2d(xd, 2, 2)
        La.frame([xd], 0.1)
        Lb.frame([pooled.contiguous()], 0.1)
        # producer mask: the pre-pool pixels that differ from the previous frame (a superset is allowed)
        if prev is None:
            Lc.frame([xd], 0.1)
        else:
            ch = (x != prev).any(axis=1)[0]
            wpr = (pW + 63) // 64
            bits = np.zeros((pH, wpr * 64), np.uint8)
            bits[:, :pW] = ch
            pm = dev(np.packbits(bits, axis=1, bitorder="little").view(np.int64).reshape(-1))
            Lc.frame([xd], 0.1, prodMasks=[pm])
        prev = x
        for L in (La, Lc):
            assert np.array_equal(L.list(), Lb.list()), t
            assert torch.equal(L.state[0], Lb.state[0]) and torch.equal(L.out[0], Lb.out[0]), t


def test_split_state_rebuild_and_range_flag(lib, oracle):
    """A state written from outside (restored states, eval03.py:88-95) is re-split by cbinfer_split_state_rebuild;
    values beyond the arithmetic's range (|x| >= 2^20) raise the flag."""
    rng = np.random.default_rng(13)
    C, K, H, W = 16, 64, 24, 40
    w = (rng.standard_normal((K, C, 7, 7)) / np.sqrt(C * 49)).astype(np.float32)
    b = rng.standard_normal(K).astype(np.float32)
    L = Layer(lib, w, b, H, W)
    x0, x1 = [dev(v) for v in block_video(rng, C, H, W, 2, 0.2)]
    L.frame([x0], 0.1)
    # overwrite the state with another frame, re-split, and present a frame that differs in a few pixels only
    y = dev(rng.standard_normal((1, C, H, W)).astype(np.float32))
    L.state[0].copy_(y)
    L.rebuild(0)
    y2 = y.clone()
    y2[0, :, 5:9, 7:12] += 1.0
    L.frame([y2], 0.1)
    o = oracle.OracleCBConv2d(w, b, 0.1, feedbackLoop=True, propChangeIndexes=True)
    o.forward(x0.cpu().numpy())
    o.prevInput = y.cpu().numpy().copy()
    got = o.forward(y2.cpu().numpy())
    assert np.array_equal(L.list(), got[2])
    n = L.list()
    out = L.out[0].cpu().numpy().reshape(K, -1)[:, n]
    assert np.abs(out - o.prevOutput.reshape(K, -1)[:, n]).max() <= FP32_TOL
    assert int(L.flag.item()) == 0
    y3 = y2.clone()
    y3[0, 3, 0, 0] = 3e6
    L.frame([y3], 0.1)
    if not L.x3:
        assert int(L.flag.item()) == 1
        return
    # bf16 triples have f32's range: no flag, and the huge value is simply an operand
    assert int(L.flag.item()) == 0
    got = o.forward(y3.cpu().numpy())
    n = got[2]
    assert np.array_equal(L.list(), n)
    out = L.out[0].cpu().numpy().reshape(K, -1)[:, n].astype(np.float64)
    mag = _sum_abs(oracle, o.prevInput, w, n, 7, 7).T + np.abs(b)[:, None]
    assert np.all(np.abs(out - o.prevOutput.reshape(K, -1)[:, n]) <= 64 * 2.0 ** -24 * mag)
    # a finite value above the largest bf16 (FLT_MAX): the head of its triple is taken by truncation, not rounded to
    # inf -- the outputs stay finite and within the bound
    y4 = y3.clone()
    y4[0, 5, 10, 20] = 3.4028234e38
    y4[0, 6, 11, 21] = -3.39e38
    L.frame([y4], 0.1)
    got = o.forward(y4.cpu().numpy())
    n = got[2]
    assert np.array_equal(L.list(), n)
    out = L.out[0].cpu().numpy().reshape(K, -1)[:, n].astype(np.float64)
    assert np.isfinite(out).all()
    want = (oracle.genXMatrix(o.prevInput, n, (7, 7)).astype(np.float64) @ w.reshape(K, -1).astype(np.float64).T).T \
        + b.astype(np.float64)[:, None]
    mag = _sum_abs(oracle, o.prevInput, w, n, 7, 7).T + np.abs(b)[:, None]
    assert np.all(np.abs(out - want) <= 64 * 2.0 ** -24 * mag)


def test_split_module_reports_range_and_survives_state_restore(lib):
    import pycbinfer
    from cbinfer_amd import workloads
    import bench
    _, net = bench.build_bench_model()
    _, ref = bench.build_bench_model()
    vid = workloads.SyntheticVideo(H=96, W=160, ratio=0.1, block=16, seed=3)
    frames = vid.frames(6)
    with torch.no_grad():
        for f in frames[:3]:
            net(f), ref(f)
        # save / restore the state tensors as the reference's eval03.py:88-95 does
        saved = [t.clone() for t in pycbinfer.getStateTensors(net)]
        for f in frames[3:]:
            net(f)
        for t, s_ in zip(pycbinfer.getStateTensors(net), saved):
            t.copy_(s_)
        for f in frames[3:]:
            a, b_ = net(f), ref(f)
            assert torch.equal(a, b_)
    convs = [m for m in net.children() if type(m) is pycbinfer.CBConv2d]
    assert not convs[1].rangeExceeded() and not convs[2].rangeExceeded()


@pytest.mark.parametrize("nSeq,force,bias", [(1, 0, True), (1, 1, True), (1, 4, True), (3, 0, True), (3, 4, True),
                                             (1, 4, False), (2, 1, False)])
def test_split_tail_in_the_second_launch(lib, oracle, nSeq, force, bias):
    """cbinfer_split_forward_tail = cbinfer_split_forward followed by cbinfer_tail1x1, bit for bit (layer outputs,
    change lists, tail outputs), whether the contraction is split along k (the tail's columns come out of the slabs)
    or not (gathered from prevOutput), for one and for several sequences; and the tail agrees with a dense
    conv1x1 -> ReLU -> conv1x1 of the layer output (sceneLabeling/modelLoader.py:45-47) within the fp32 bar."""
    C_ = lib.C
    rng = np.random.default_rng(21 + nSeq + force)
    C, K, H, W, C1, C2 = 64, 256, 40, 60, 64, 8
    assert C_.cbinfer_split_tail_supported(C, K, 7, 7, C1, C2) == 1
    assert C_.cbinfer_split_tail_supported(16, 64, 7, 7, C1, C2) == 0       # (not a deep contraction)
    w = (rng.standard_normal((K, C, 7, 7)) / np.sqrt(C * 49)).astype(np.float32)
    b = rng.standard_normal(K).astype(np.float32)
    w1 = (rng.standard_normal((C1, K)) / np.sqrt(K)).astype(np.float32)
    b1 = rng.standard_normal(C1).astype(np.float32)
    w2 = (rng.standard_normal((C2, C1)) / np.sqrt(C1)).astype(np.float32)
    b2 = rng.standard_normal(C2).astype(np.float32)
    dw1, db1, dw2, db2 = dev(w1), dev(b1), dev(w2), dev(b2)
    w1p = torch.empty(C_.cbinfer_tail1x1_prepared_bytes(C1, K) // 4, device="cuda")
    lib.check(C_.cbinfer_tail1x1_prep(dw1.data_ptr(), w1p.data_ptr(), C1, K, None))
    vids = [block_video(rng, C, H, W, 4, f) for f in (0.15, 0.6, 0.02)[:nSeq]]
    ref, fused = Layer(lib, w, b, H, W, nSeq=nSeq), Layer(lib, w, b, H, W, nSeq=nSeq)
    if not bias:            # (a layer without bias: the C ABI takes a null pointer)
        class NoBias(object):
            @staticmethod
            def data_ptr():
                return None
        ref.b = fused.b = NoBias()
    tref = [torch.full((1, C2, H, W), float("inf"), device="cuda") for _ in range(nSeq)]
    tfus = [torch.full((1, C2, H, W), float("inf"), device="cuda") for _ in range(nSeq)]
    st = lib.SplitTail()
    st.w1Prepared, st.b1, st.w2, st.b2 = w1p.data_ptr(), db1.data_ptr(), dw2.data_ptr(), db2.data_ptr()
    st.C1, st.C2, st.relu1, st.relu2 = C1, C2, 1, 0
    for q in range(nSeq):
        st.output[q] = tfus[q].data_ptr()
    import ctypes
    for t in range(4):
        xs = [dev(v[t]) for v in vids]
        ref.frame(xs, 0.05, relu=True, force=force)
        for q in range(nSeq):
            lib.check(C_.cbinfer_tail1x1(ref.out[q].data_ptr(), ref.idx[q].data_ptr(), H * W, ref.cnt[q].data_ptr(),
                                         w1p.data_ptr(), db1.data_ptr(), dw2.data_ptr(), db2.data_ptr(),
                                         tref[q].data_ptr(), K, C1, C2, H, W, 1, 0, None))
        for q, x in enumerate(xs):
            fused.seqs[q].input = x.data_ptr()
            fused.seqs[q].producerMask = None
        lib.check(C_.cbinfer_split_forward_tail(fused.seqs, nSeq, 0, 0, 0, fused.wp.data_ptr(), fused.b.data_ptr(), C,
                                                H, W, K, 7, 7, 0.05, fused.scale, 1, fused.ws.data_ptr(), force,
                                                ctypes.pointer(st), None))
        torch.cuda.synchronize()
        for q in range(nSeq):
            assert np.array_equal(ref.list(q), fused.list(q)), (t, q)
            assert torch.equal(ref.out[q], fused.out[q]), (t, q)
            assert torch.equal(tref[q], tfus[q]), (t, q)
    for q in range(nSeq):
        y = fused.out[q]
        dense = torch.nn.functional.conv2d(torch.relu(torch.nn.functional.conv2d(
            y.double(), dw1.double().view(C1, K, 1, 1), db1.double())), dw2.double().view(C2, C1, 1, 1), db2.double())
        assert float((dense - tfus[q].double()).abs().max()) <= FP32_TOL


def _sum_abs(oracle, x, w, idx, kH, kW):
    """sum |a||b| of every output of the changed pixels `idx` (double)."""
    X = oracle.genXMatrix(x, idx, (kH, kW))
    return np.abs(X).astype(np.float64) @ np.abs(w.reshape(w.shape[0], -1)).astype(np.float64).T      # [N, K]


@pytest.mark.parametrize("C,K,H,W,nSeq,tail", [(16, 64, 24, 40, 1, False), (64, 256, 40, 60, 1, False),
                                               (64, 256, 40, 60, 2, True), (64, 256, 40, 60, 1, True)])
def test_split_range_flag_acts(lib, oracle, C, K, H, W, nSeq, tail):
    """a state value beyond the f16 pair's range (|x| >= 2^20) must never produce silent
    garbage.  The detection raises the layer's flag and the contraction launch of the SAME frame computes the layer
    from prevInput with plain f32 arithmetic (cbs_exact_tile): outputs against the double-accumulated oracle within the
    f32 chain's bound, in that frame and in the frames behind it (the flag is sticky), for one sequence and for two in
    one launch, with and without the fused 1x1 tail in the second launch (the unsplit form of that launch)."""
    import ctypes
    if ARITH == "x3":
        pytest.skip("the range flag belongs to the f16-pair form (bf16 triples have f32's range: "
                    "test_split_state_rebuild_and_range_flag covers a huge operand there)")
    C_ = lib.C
    rng = np.random.default_rng(C + K + nSeq)
    w = (rng.standard_normal((K, C, 7, 7)) / np.sqrt(C * 49)).astype(np.float32)
    b = rng.standard_normal(K).astype(np.float32)
    L = Layer(lib, w, b, H, W, nSeq=nSeq)
    os_ = [oracle.OracleCBConv2d(w, b, 0.1, withReLU=True, feedbackLoop=True, propChangeIndexes=True)
           for _ in range(nSeq)]
    C1, C2 = 64, 8
    if tail:
        w1 = (rng.standard_normal((C1, K)) / np.sqrt(K)).astype(np.float32)
        b1 = rng.standard_normal(C1).astype(np.float32)
        w2 = (rng.standard_normal((C2, C1)) / np.sqrt(C1)).astype(np.float32)
        b2 = rng.standard_normal(C2).astype(np.float32)
        dw1, db1, dw2, db2 = dev(w1), dev(b1), dev(w2), dev(b2)
        w1p = torch.empty(C_.cbinfer_tail1x1_prepared_bytes(C1, K) // 4, device="cuda")
        lib.check(C_.cbinfer_tail1x1_prep(dw1.data_ptr(), w1p.data_ptr(), C1, K, None))
        tout = [torch.full((1, C2, H, W), float("inf"), device="cuda") for _ in range(nSeq)]
        st = lib.SplitTail()
        st.w1Prepared, st.b1, st.w2, st.b2 = w1p.data_ptr(), db1.data_ptr(), dw2.data_ptr(), db2.data_ptr()
        st.C1, st.C2, st.relu1, st.relu2 = C1, C2, 1, 0
        for q in range(nSeq):
            st.output[q] = tout[q].data_ptr()

    def frame(xs):
        if not tail:
            return L.frame([dev(x) for x in xs], 0.1, relu=True)
        keep = [dev(x) for x in xs]
        for q, x in enumerate(keep):
            L.seqs[q].input, L.seqs[q].producerMask = x.data_ptr(), None
        lib.check(C_.cbinfer_split_forward_tail(L.seqs, nSeq, 0, 0, 0, L.wp.data_ptr(), L.b.data_ptr(), C, H, W, K, 7, 7,
                                                0.1, L.scale, 1, L.ws.data_ptr(), 0, ctypes.pointer(st), None))
        torch.cuda.synchronize()

    vids = [block_video(rng, C, H, W, 5, 0.15) for _ in range(nSeq)]
    for t in range(5):
        xs = [v[t].copy() for v in vids]
        if t >= 2:
            xs[0][0, 3, 5, 7] = 3.0e6 + t            # beyond 2^20 = 1.05e6, and changing from frame to frame
        if t == 3:
            xs[0][0, 1, 11, 13] = float(2 ** 20)      # the first value out of range, exactly
        frame(xs)
        assert int(L.flag.item()) == (1 if t >= 2 else 0), t
        for q in range(nSeq):
            got = os_[q].forward(xs[q])
            assert np.array_equal(L.list(q), got[2]), (t, q)
            assert np.array_equal(L.state[q].cpu().numpy(), os_[q].prevInput), (t, q)
            n = got[2]
            out = L.out[q].cpu().numpy().reshape(K, -1)[:, n].astype(np.float64)
            ref = os_[q].prevOutput.reshape(K, -1)[:, n].astype(np.float64)
            mag = _sum_abs(oracle, os_[q].prevInput, w, n, 7, 7).T + np.abs(b)[:, None]
            err = np.abs(out - ref)
            # the f32 fma chain's own bound, n_k * 2^-24 * sum|a||b| (every add behind the huge term is rounded at ITS
            # magnitude); where no huge value is in reach that is <= 1e-4
            assert np.all(err <= C * 49 * 2.0 ** -24 * mag + 1e-30), (t, q, float((err / mag).max()))
            small = mag.max(axis=0) < 1e3
            assert small.any() and err[:, small].max() <= FP32_TOL, (t, q)
            if tail:
                y = L.out[q].double()
                dense = torch.nn.functional.conv2d(torch.relu(torch.nn.functional.conv2d(
                    y, dw1.double().view(C1, K, 1, 1), db1.double())), dw2.double().view(C2, C1, 1, 1), db2.double())
                te = (dense - tout[q].double()).abs().reshape(C2, -1)[:, torch.from_numpy(n.astype(np.int64)).cuda()]
                ty = y.abs().reshape(K, -1)[:, torch.from_numpy(n.astype(np.int64)).cuda()].amax(dim=0)
                assert bool((te.amax(dim=0) <= 1e-4 * torch.clamp(ty, min=1.0)).all()), (t, q)


def test_split_forced_k_split_respects_the_workspace(lib, oracle):
    """forceSplit >= 4 on a first frame (every pixel changed) would need 600 partial tiles where the
    workspace holds 512 -- the kernel must fall back to the unsplit form (same bits) instead of writing past it."""
    rng = np.random.default_rng(31)
    C, K, H, W = 64, 256, 80, 120
    w = (rng.standard_normal((K, C, 7, 7)) / np.sqrt(C * 49)).astype(np.float32)
    b = rng.standard_normal(K).astype(np.float32)
    x = dev(rng.standard_normal((1, C, H, W)).astype(np.float32))
    outs = []
    for force in (4, 1):
        L = Layer(lib, w, b, H, W)
        guard = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")      # (allocated right behind the workspace)
        L.frame([x], 0.1, relu=True, force=force)
        assert int(guard.sum().item()) == 0
        outs.append(L.out[0].clone())
    assert torch.equal(outs[0], outs[1])
    # the capacity the kernel was told about is the workspace's
    cap = (lib.C.cbinfer_split_workspace_bytes(1, C, H, W, K, 7, 7) - 256) // (128 * 128 * 4)
    assert cap == 512 and (H * W + 127) // 128 * 2 * 4 == 600


@pytest.mark.parametrize("C,K,H,W", [(64, 256, 80, 120), (16, 64, 160, 240)])
def test_split_hostile_data_accuracy_fullsize(lib, oracle, C, K, H, W, capsys):
    """the two split-state layers of the bench network at FULL size on hostile
    data -- activations spanning 2^-20 ... 2^19 with ReLU-like sparsity, heavy-tailed weights -- against the
    double-accumulated oracle: the bf16-TRIPLE split-state kernel (the default, f32-equivalent), the f16-pair one, and
    the library's exact-f32 MFMA kernel (CB_F32: the f32 fma chain, what conv2d_cg.py:342-349's sgemm does) and
    rounds 1-2's bf16x3 kernel (CB_F32S) side by side, every pixel changed.  Reported: max |err| / sum|a||b| of each.
    Asserted: the triple form's error is NO LARGER than the exact f32 chain's on both layers (its operands are exact
    and it rounds once per 16-k block sum where the chain rounds once per product); the f16-pair form within
    4 * 2^-22 * sum|a||b| (22-23 significant bits per operand) and no worse than 16x the chain's."""
    if ARITH != "x3":
        pytest.skip("one run compares all arithmetics")
    from cbinfer_amd import conv2d_cg, _lib as L_
    rng = np.random.default_rng(C * 7 + K)
    mag_x = np.exp2(rng.uniform(-20, 19, (1, C, H, W)))
    x = (rng.standard_normal((1, C, H, W)) * mag_x).astype(np.float32)
    x = np.where(rng.uniform(size=x.shape) < 0.5, 0.0, np.abs(x)).astype(np.float32)      # ReLU-like: half zeros, >= 0
    x = np.minimum(x, np.float32(2.0 ** 19.9))
    w = (rng.standard_t(2.5, (K, C, 7, 7)) / np.sqrt(C * 49)).astype(np.float32)           # heavy tails
    b = rng.standard_normal(K).astype(np.float32)
    Ly = Layer(lib, w, b, H, W, arith="f16x2")
    Ly.frame([dev(x)], 0.0)
    assert int(Ly.flag.item()) == 0
    idx = np.arange(H * W, dtype=np.int32)
    assert np.array_equal(Ly.list(), idx)
    L3 = Layer(lib, w, b, H, W, arith="x3")
    L3.frame([dev(x)], 0.0)
    assert np.array_equal(L3.list(), idx)
    X = oracle.genXMatrix(x, idx, (7, 7)).astype(np.float64)
    wm = w.reshape(K, -1).astype(np.float64)
    Y = (X @ wm.T + b[None, :].astype(np.float64)).T.reshape(K, H * W)
    mag = (np.abs(X) @ np.abs(wm).T).T.reshape(K, H * W)
    sumw = np.abs(wm).sum(axis=1)[:, None]
    del X
    res = {}
    res["x3 split-state (bf16 triples)"] = np.abs(L3.out[0].cpu().numpy().reshape(K, -1).astype(np.float64) - Y)
    res["f16x2 split-state"] = np.abs(Ly.out[0].cpu().numpy().reshape(K, -1).astype(np.float64) - Y)
    xd, wd, bd = dev(x), dev(w), dev(b)
    idxd = torch.arange(H * W, dtype=torch.int32, device="cuda")
    for name, arith in (("exact f32 MFMA", L_.CB_F32), ("bf16x3", L_.CB_F32S)):
        out = torch.zeros(1, K, H, W, device="cuda")
        conv2d_cg.convChanged(xd, idxd, wd, bd, out, withReLU=False, arith=arith)
        torch.cuda.synchronize()
        res[name] = np.abs(out.cpu().numpy().reshape(K, -1).astype(np.float64) - Y)
    rel = {k: float((v / (mag + 1e-300)).max()) for k, v in res.items()}
    with capsys.disabled():
        print("\nhostile data %d->%d @%dx%d: max |err| / sum|a||b|: " % (C, K, H, W) +
              ", ".join("%s %.3g (2^%.1f), max |err| %.3g" % (k, rel[k], np.log2(rel[k]), res[k].max())
                        for k in res))
    import json
    import os
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/hostile_accuracy_%d_%d.json" % (C, K), "w") as f:
        json.dump({"layer": "%d->%d 7x7 @%dx%d" % (C, K, H, W), "rel_to_sum_abs": rel,
                   "max_abs_err": {k: float(v.max()) for k, v in res.items()},
                   "max_sum_abs": float(mag.max())}, f)
    split = res["f16x2 split-state"]
    assert np.all(split <= 4 * 2.0 ** -22 * mag + 1e-30)
    assert rel["f16x2 split-state"] <= 16 * max(rel["exact f32 MFMA"], 2.0 ** -24)
    # the f32-equivalent form: no worse than the f32 fma chain itself
    assert rel["x3 split-state (bf16 triples)"] <= rel["exact f32 MFMA"], rel
    assert res["x3 split-state (bf16 triples)"].max() <= res["exact f32 MFMA"].max() * 1.05 + 1e-30


@pytest.mark.parametrize("H,W,K,k2,frac", [(45, 67, 64, 3, 0.1), (160, 240, 64, 7, 0.1), (38, 130, 32, 5, 0.3),
                                            (64, 64, 16, 3, 0.02)])
def test_split_next_detection_in_window_order(lib, oracle, H, W, K, k2, frac):
    """cbinfer_split_conv_next (round 6): the producer's contraction in pooling-window order + the pooled change
    detection of the layer behind the 2x2 pool in its epilogue, against the separate launches (cbinfer_split_conv, then
    the consumer's cbinfer_split_detect with the pool folded in): the producer's outputs and change list (row-major, as
    the reference orders it), the consumer's f32 state, its split copy, its list and its outputs -- all bit for bit."""
    if ARITH != "x3":
        pytest.skip("window order exists for the bf16-triple arithmetic")
    C_ = lib.C
    rng = np.random.default_rng(H * 7 + W + K)
    Cin, K2 = 16, 48
    H2, W2 = H // 2, W // 2
    w1 = (rng.standard_normal((K, Cin, 7, 7)) / np.sqrt(Cin * 49)).astype(np.float32)
    b1 = rng.standard_normal(K).astype(np.float32)
    w2 = (rng.standard_normal((K2, K, k2, k2)) / np.sqrt(K * k2 * k2)).astype(np.float32)
    b2 = rng.standard_normal(K2).astype(np.float32)
    Pa, Pb = Layer(lib, w1, b1, H, W), Layer(lib, w1, b1, H, W)
    Ca, Cb = Layer(lib, w2, b2, H2, W2, pooled=True), Layer(lib, w2, b2, H2, W2, pooled=True)
    nd = lib.NextDetect()
    nd.state, nd.splitState, nd.frameMasks = Cb.state[0].data_ptr(), Cb.S[0].data_ptr(), Cb.masks[0].data_ptr()
    nd.rangeFlag, nd.H, nd.W, nd.kH, nd.kW, nd.threshold, nd.arith = Cb.flag.data_ptr(), H2, W2, k2, k2, 0.05, 1
    import ctypes
    assert C_.cbinfer_split_next_supported(Cin, K, 7, 7, H, W, ctypes.pointer(nd)) == 1
    for t, x in enumerate(block_video(rng, Cin, H, W, 6, frac)):
        xd = dev((x + rng.uniform(-0.03, 0.03, x.shape)).astype(np.float32))
        # separate launches
        Pa.frame([xd], 0.1, relu=True)
        Ca.frame([Pa.out[0]], 0.05, relu=True)
        # folded: the producer's detection, its contraction + the consumer's detection, the consumer's contraction alone
        Pb.seqs[0].input, Pb.seqs[0].producerMask = xd.data_ptr(), None
        lib.check(C_.cbinfer_split_detect(Pb.seqs, 1, 8, 0, 0, Cin, H, W, 7, 7, 0.1, None))
        lib.check(C_.cbinfer_split_conv_next(Pb.seqs, 1, Pb.wp.data_ptr(), Pb.b.data_ptr(), Cin, H, W, K, 7, 7, 0.0, 1,
                                             None, ctypes.pointer(nd), None))
        Cb.seqs[0].input, Cb.seqs[0].producerMask = Pb.out[0].data_ptr(), None
        lib.check(C_.cbinfer_split_conv(Cb.seqs, 1, Cb.wp.data_ptr(), Cb.b.data_ptr(), K, H2, W2, K2, k2, k2, 0.0, 1,
                                        Cb.ws.data_ptr() if Cb.ws is not None else None, 0, None))
        torch.cuda.synchronize()
        assert np.array_equal(Pa.list(), Pb.list()), t
        assert torch.equal(Pa.out[0], Pb.out[0]) and torch.equal(Pa.state[0], Pb.state[0]), t
        assert torch.equal(Pa.copy[0], Pb.copy[0]), t
        assert np.array_equal(Ca.list(), Cb.list()), t
        assert torch.equal(Ca.state[0], Cb.state[0]), t
        assert torch.equal(Ca.S[0], Cb.S[0]), t
        assert torch.equal(Ca.out[0], Cb.out[0]) and torch.equal(Ca.copy[0], Cb.copy[0]), t
        assert len(Cb.list()) > 0 or t > 0


@pytest.mark.parametrize("order", ["window", "pixel"])
def test_split_contraction_carries_a_side_refresh(lib, oracle, order):
    """cbinfer_split_conv_next_refresh / cbinfer_split_conv_refresh (round 6): the feedback refresh of ANOTHER layer's state on
    the contraction's workgroups without a work item -- against cbinfer_refresh_state in a launch of its own, bit for bit, and
    the contraction's own results untouched by it; a frame with few tiles (idle workgroups do it) and one with more tiles than
    workgroups (everybody does a slice)."""
    if ARITH != "x3":
        pytest.skip("the side job rides on the bf16-triple instances")
    import ctypes
    C_ = lib.C
    rng = np.random.default_rng(17)
    Cin, K, K2, H, W, k2 = 16, 64, 32, 160, 240, 3
    H2, W2 = H // 2, W // 2
    w1 = (rng.standard_normal((K, Cin, 7, 7)) / np.sqrt(Cin * 49)).astype(np.float32)
    b1 = rng.standard_normal(K).astype(np.float32)
    w2 = (rng.standard_normal((K2, K, k2, k2)) / np.sqrt(K * k2 * k2)).astype(np.float32)
    b2 = rng.standard_normal(K2).astype(np.float32)
    Pa, Pb = Layer(lib, w1, b1, H, W), Layer(lib, w1, b1, H, W)
    Cb = Layer(lib, w2, b2, H2, W2, pooled=True)
    nd = lib.NextDetect()
    nd.state, nd.splitState, nd.frameMasks = Cb.state[0].data_ptr(), Cb.S[0].data_ptr(), Cb.masks[0].data_ptr()
    nd.rangeFlag, nd.H, nd.W, nd.kH, nd.kW, nd.threshold, nd.arith = Cb.flag.data_ptr(), H2, W2, k2, k2, 0.05, 1
    sC, sH, sW = 3, 97, 131
    sa = torch.zeros((sC, sH, sW), device="cuda")
    sb = torch.zeros((sC, sH, sW), device="cuda")
    side = lib.SideRefresh()
    assert C_.cbinfer_split_refresh_supported(Cin, K, 7, 7, H, W) == 1
    for t, (x, frac) in enumerate(zip(block_video(rng, Cin, H, W, 6, 0.1), (1.0, 0.1, 0.1, 0.9, 0.1, 0.05))):
        if t == 3:
            x = (x + rng.standard_normal(x.shape)).astype(np.float32)      # (every pixel changes: more tiles than workgroups)
        xd = dev(x)
        fr = dev((rng.standard_normal((sC, sH, sW)) * (rng.random((1, sH, sW)) < 0.3)).astype(np.float32))
        for P in (Pa, Pb):
            P.seqs[0].input, P.seqs[0].producerMask = xd.data_ptr(), None
            lib.check(C_.cbinfer_split_detect(P.seqs, 1, 8, 0, 0, Cin, H, W, 7, 7, 0.1, None))
        side.frame, side.state, side.C, side.H, side.W, side.threshold = fr.data_ptr(), sb.data_ptr(), sC, sH, sW, 0.25
        if order == "window":
            lib.check(C_.cbinfer_split_conv_next(Pa.seqs, 1, Pa.wp.data_ptr(), Pa.b.data_ptr(), Cin, H, W, K, 7, 7, 0.0, 1,
                                                 None, ctypes.pointer(nd), None))
            Cb.masks[0].zero_()
            lib.check(C_.cbinfer_split_conv_next_refresh(Pb.seqs, 1, Pb.wp.data_ptr(), Pb.b.data_ptr(), Cin, H, W, K, 7, 7,
                                                         0.0, 1, None, ctypes.pointer(nd), ctypes.pointer(side), None))
            Cb.masks[0].zero_()
        else:
            lib.check(C_.cbinfer_split_conv(Pa.seqs, 1, Pa.wp.data_ptr(), Pa.b.data_ptr(), Cin, H, W, K, 7, 7, 0.0, 1, None,
                                            0, None))
            lib.check(C_.cbinfer_split_conv_refresh(Pb.seqs, 1, Pb.wp.data_ptr(), Pb.b.data_ptr(), Cin, H, W, K, 7, 7, 0.0,
                                                    1, None, ctypes.pointer(side), None))
        lib.check(C_.cbinfer_refresh_state(fr.data_ptr(), sa.data_ptr(), sC, sH, sW, 0.25, None))
        torch.cuda.synchronize()
        assert torch.equal(sa, sb), t
        assert torch.equal(Pa.out[0], Pb.out[0]) and np.array_equal(Pa.list(), Pb.list()), t
        # what the refresh means (cbconv2d_cg_backend.cu:74-80): a pixel of which some channel differs by more than the
        # threshold holds the frame's values, every other pixel is within the threshold of them
        assert float((sb - fr).abs().max()) <= 0.25
