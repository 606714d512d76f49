"""-m gpu tests of the row-pair kernel (cb_rowpair.hip) through the C ABI and at module level: the fused a5..a8 path of
a layer of few channels (cbconv2d_cg_backend.cu:138-197, conv2d_cg.py:342-349) against the oracle, and the NEXT layer's
pooled change detection folded into the same launch (CBPoolMax2d conv2d.py:49-78 + changeDetection with
updateInputState cbconv2d_cg_backend.cu:40-81) against the separate detection launch, bit for bit."""
import ctypes
import math

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

FP32_TOL = 1e-4


@pytest.fixture(scope="module")
def lib():
    from cbinfer_amd import _lib
    assert torch.cuda.is_available()
    return _lib


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def pack_mask(m):
    H, W = m.shape
    wpr = (W + 63) // 64
    bits = np.zeros((H, wpr * 64), np.uint8)
    bits[:, :W] = m
    return np.packbits(bits, axis=1, bitorder="little").view(np.int64).reshape(-1)


@pytest.mark.parametrize("C,K,k,H,W", [(3, 16, 7, 64, 96), (3, 16, 7, 37, 131), (4, 16, 3, 45, 70), (1, 9, 5, 20, 64),
                                       (3, 16, 7, 320, 480)])
def test_rowpairs_contraction_vs_oracle(lib, oracle, C, K, k, H, W):
    """cbinfer_conv_changed_rowpairs (no folding) at the pixels of a random dilated change mask: outputs against the
    double-accumulated oracle (genXMatrix + matrixMult + updateOutput) <= 1e-4 at the masked pixels, every other
    output untouched, the mask zeroed, its copy kept, the arrival counter back at zero; odd sizes, a last row without
    a partner, full and nearly empty words."""
    C_ = lib.C
    rng = np.random.default_rng(C * 100 + K + H)
    assert C_.cbinfer_rowpairs_supported(C, K, k, k, H, W) == 1
    w = (rng.standard_normal((K, C, k, k)) / np.sqrt(C * k * k)).astype(np.float32)
    b = rng.standard_normal(K).astype(np.float32)
    x = rng.standard_normal((1, C, H, W)).astype(np.float32)
    m = np.zeros((H, W), np.int8)
    for _ in range(max(2, H * W // 900)):
        y0, x0 = rng.integers(0, H), rng.integers(0, W)
        m[y0:y0 + rng.integers(1, 12), x0:x0 + rng.integers(1, 40)] = 1
    m[H - 1, :] = 1                                # the last row (alone in its pair when H is odd), a full word
    m[0, 0] = m[0, W - 1] = 1
    idx = np.flatnonzero(m.reshape(-1)).astype(np.int32)
    wp = torch.empty(C_.cbinfer_rowconv_prepared_bytes(C, K, k, k), dtype=torch.uint8, device="cuda")
    lib.check(C_.cbinfer_rowconv_prep_weights(dev(w).data_ptr(), wp.data_ptr(), K, C, k, k, None))
    words = C_.cbinfer_mask_words(H, W)
    bits, copy = dev(pack_mask(m)), torch.zeros(words, dtype=torch.int64, device="cuda")
    ctl = torch.zeros(words, dtype=torch.int32, device="cuda")
    out0 = rng.standard_normal((1, K, H, W)).astype(np.float32)
    out, xd, bd = dev(out0), dev(x), dev(b)
    for relu in (0, 1):
        bits.copy_(dev(pack_mask(m)))
        out.copy_(dev(out0))
        lib.check(C_.cbinfer_conv_changed_rowpairs(xd.data_ptr(), bits.data_ptr(), ctl.data_ptr(), copy.data_ptr(),
                                                   wp.data_ptr(), bd.data_ptr(), out.data_ptr(), C, H, W, K, k, k, relu,
                                                   None, None))
        torch.cuda.synchronize()
        X = oracle.genXMatrix(x, idx, (k, k))
        Y = oracle.matrixMult(X, w, b)
        ref = oracle.updateOutput(Y.T.copy(), idx, out0.copy(), withReLU=bool(relu))
        got = out.cpu().numpy()
        assert np.abs(got - ref).max() <= FP32_TOL
        keep = np.ones(H * W, bool)
        keep[idx] = False
        assert np.array_equal(got.reshape(K, -1)[:, keep], out0.reshape(K, -1)[:, keep])
        assert int(bits.abs().sum().item()) == 0 and int(ctl.abs().sum().item()) == 0
        assert np.array_equal(copy.cpu().numpy(), pack_mask(m))
    # an empty mask: nothing happens (and the kernel exits cleanly)
    lib.check(C_.cbinfer_conv_changed_rowpairs(xd.data_ptr(), bits.data_ptr(), ctl.data_ptr(), copy.data_ptr(),
                                               wp.data_ptr(), bd.data_ptr(), out.data_ptr(), C, H, W, K, k, k, 0, None,
                                               None))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), got) and int(copy.abs().sum().item()) == 0


@pytest.mark.parametrize("x3", [True, False])
@pytest.mark.parametrize("H,W,ceil,k2", [(64, 96, False, 7), (45, 67, False, 7), (45, 67, True, 3), (90, 200, False, 5)])
def test_rowpairs_folded_detection_equals_separate_launch(lib, oracle, H, W, ceil, k2, x3):
    """The next layer's pooled change detection inside the row-pair launch against cbinfer_split_detect (pooled, with
    the producer's mask) fed the same outputs: the next layer's f32 state, its pre-split copy, its frame mask and its
    range flag bit-identical over a sequence of frames; floor and ceil pooling of odd maps; the next layer's records as
    bf16 triples (x3: no range flag) and as f16 pairs."""
    C_ = lib.C
    rng = np.random.default_rng(H + W)
    C, K, k = 3, 16, 7
    H2, W2 = ((H + 1) // 2, (W + 1) // 2) if ceil else (H // 2, W // 2)
    w = (rng.standard_normal((K, C, k, k)) / np.sqrt(C * k * k)).astype(np.float32)
    b = rng.standard_normal(K).astype(np.float32)
    wp = torch.empty(C_.cbinfer_rowconv_prepared_bytes(C, K, k, k), dtype=torch.uint8, device="cuda")
    lib.check(C_.cbinfer_rowconv_prep_weights(dev(w).data_ptr(), wp.data_ptr(), K, C, k, k, None))
    bd = dev(b)
    words, words2 = C_.cbinfer_mask_words(H, W), C_.cbinfer_mask_words(H2, W2)

    class Side(object):          # one copy of everything per variant (folded / separate)
        def __init__(self):
            self.state = torch.full((1, C, H, W), float("inf"), device="cuda")
            self.out = torch.full((1, K, H, W), float("inf"), device="cuda")
            self.bits = torch.zeros(words, dtype=torch.int64, device="cuda")
            self.ctl = torch.zeros(words, dtype=torch.int32, device="cuda")
            self.copy = torch.zeros(words, dtype=torch.int64, device="cuda")
            self.state2 = torch.full((1, K, H2, W2), float("inf"), device="cuda")
            self.flag = torch.zeros(1, dtype=torch.int32, device="cuda")
            if x3:
                self.S2 = torch.empty(C_.cbinfer_split3_state_bytes(K, H2, W2, k2, k2), dtype=torch.uint8,
                                      device="cuda")
                lib.check(C_.cbinfer_split3_state_init(self.S2.data_ptr(), K, H2, W2, k2, k2, None))
                lib.check(C_.cbinfer_split3_state_rebuild(self.state2.data_ptr(), self.S2.data_ptr(), K, H2, W2, k2,
                                                          k2, None))
            else:
                self.S2 = torch.empty(C_.cbinfer_split_state_bytes(K, H2, W2, k2, k2), dtype=torch.uint8,
                                      device="cuda")
                lib.check(C_.cbinfer_split_state_init(self.S2.data_ptr(), K, H2, W2, k2, k2, None))
                lib.check(C_.cbinfer_split_state_rebuild(self.state2.data_ptr(), self.S2.data_ptr(), K, H2, W2, k2,
                                                         k2, self.flag.data_ptr(), None))
            self.mask2 = torch.zeros(C_.cbinfer_frame_mask_bytes(H2, W2) // 8, dtype=torch.int64, device="cuda")

    fo, se = Side(), Side()
    nd = lib.NextDetect()
    nd.state, nd.splitState, nd.frameMasks = fo.state2.data_ptr(), fo.S2.data_ptr(), fo.mask2.data_ptr()
    nd.rangeFlag, nd.H, nd.W, nd.kH, nd.kW, nd.threshold = fo.flag.data_ptr(), H2, W2, k2, k2, 0.07
    nd.arith = 1 if x3 else 0
    seq = (lib.SplitSeq * 1)()
    x = rng.standard_normal((1, C, H, W)).astype(np.float32)
    total2 = 0
    for t in range(6):
        x = x.copy()
        if t > 0:
            for _ in range(max(1, H * W // 1500)):
                y0, x0 = rng.integers(0, H - 4), rng.integers(0, W - 4)
                x[0, :, y0:y0 + rng.integers(2, 9), x0:x0 + rng.integers(2, 30)] = rng.standard_normal((C, 1, 1))
            if t == 4:
                x[0, 0, H // 2, W // 2] = 1.0e9         # an output beyond the next layer's f16-pair range -> its flag
        xd = dev(x)
        # folded: detection + (contraction + next layer's detection)
        lib.check(C_.cbinfer_cbconv2d_forward_rowpairs(xd.data_ptr(), fo.state.data_ptr(), fo.out.data_ptr(),
                                                       fo.bits.data_ptr(), fo.ctl.data_ptr(), fo.copy.data_ptr(),
                                                       wp.data_ptr(), bd.data_ptr(), C, H, W, K, k, k, 0.05, 1,
                                                       ctypes.pointer(nd), None))
        # separate: the same launch without folding, then the next layer's own pooled detection with the producer mask
        lib.check(C_.cbinfer_cbconv2d_forward_rowpairs(xd.data_ptr(), se.state.data_ptr(), se.out.data_ptr(),
                                                       se.bits.data_ptr(), se.ctl.data_ptr(), se.copy.data_ptr(),
                                                       wp.data_ptr(), bd.data_ptr(), C, H, W, K, k, k, 0.05, 1, None,
                                                       None))
        s = seq[0]
        s.input, s.state, s.splitState = se.out.data_ptr(), se.state2.data_ptr(), se.S2.data_ptr()
        s.frameMasks, s.rangeFlag = se.mask2.data_ptr(), se.flag.data_ptr()
        s.producerMask = se.copy.data_ptr() if t > 0 else None
        lib.check(C_.cbinfer_split_detect(seq, 1, 1 | (8 if x3 else 0), H, W, K, H2, W2, k2, k2, 0.07, None))
        torch.cuda.synchronize()
        assert torch.equal(fo.out, se.out) and torch.equal(fo.state, se.state) and torch.equal(fo.copy, se.copy), t
        assert torch.equal(fo.state2, se.state2), t
        assert torch.equal(fo.mask2, se.mask2), t
        assert torch.equal(fo.S2, se.S2), t
        assert int(fo.flag.item()) == int(se.flag.item()) == (1 if t >= 4 and not x3 else 0), t
        # ... and against the oracle: pooled outputs vs the refreshed state, strict >, dilation
        n2 = int(torch.count_nonzero(fo.mask2[:words2]).item())
        total2 += n2
        fo.mask2.zero_(), se.mask2.zero_()         # (the next layer's contraction would)
    assert total2 > 0
    pooled = torch.nn.functional.max_pool2d(fo.out, 2, 2, ceil_mode=ceil)
    # feedback state: every pooled pixel within the threshold of its state
    assert float((pooled - fo.state2).abs().max()) <= 0.07


def test_bench_network_with_and_without_the_folded_detection(lib):
    """Module level, the bench configuration: the 3->16 layer's row-pair launch doing the 16->64 layer's pooled
    detection (pycbinfer.fuseDetectionIntoProducer, the default of bench.build_bench_model) against the same network
    with the separate detection launch: bit-identical outputs and layer states over a walk that includes a change of
    the consumer's threshold, a restored state (eval03.py:88-95) and clearMemory -- the frames that must fall back to
    the separate launch -- and the fold really runs in between."""
    import pycbinfer as pkg
    import bench
    _, fold = bench.build_bench_model()
    _, sep = bench.build_bench_model(fuse_detect=False)
    cf = [m for m in fold.children() if type(m) is pkg.CBConv2d]
    cs = [m for m in sep.children() if type(m) is pkg.CBConv2d]
    assert cf[0].__dict__.get('_fusedNext') is not None and cs[0].__dict__.get('_fusedNext') is None
    frames = bench.bench_video(99).frames(16)
    folded = []
    with torch.no_grad():
        for t, f in enumerate(frames):
            if t == 6:
                cf[1].threshold = cs[1].threshold = 0.08
            if t == 9:
                saved = [[s.clone() for s in pkg.getStateTensors(n)] for n in (fold, sep)]
            if t == 11:
                for n, sv in zip((fold, sep), saved):
                    for s, v in zip(pkg.getStateTensors(n), sv):
                        s.copy_(v)
            if t == 13:
                pkg.clearMemory(fold), pkg.clearMemory(sep)
            ya, yb = fold(f), sep(f)
            tok = getattr(cf[0].lastChangeIndexes(), 'nextDetect', None)
            folded.append(tok is not None)
            assert torch.equal(ya, yb), t
            for a, b in zip(cf, cs):
                assert torch.equal(a.prevInput, b.prevInput) and torch.equal(a.prevOutput, b.prevOutput), t
    # frame 0 allocates, frame 1 is the consumer's first shortcut-less frame; threshold change at 6, restore at 11,
    # clearMemory at 13
    assert folded[2:6] == [True] * 4 and folded[6] is False and folded[7:11] == [True] * 4, folded
    assert folded[11] is False and folded[12] is True and folded[13] is False and folded[15] is True, folded
    assert cf[0]._plan is not None and cf[0]._plan.get('pairs') and cf[0]._plan.get('nextToken') is not None
    assert cs[0]._plan is not None and cs[0]._plan.get('pairs') and cs[0]._plan.get('nextToken') is None


def test_rowpair_network_tracks_the_row_segment_kernel_over_many_frames(lib, monkeypatch):
    """Regression test of round 4's sporadic failure (workgroups that read their inputs a few microseconds into the
    launch computed whole units from wrong operands while the mask words' zeroing stores were issued at the start of
    the workgroup): 100 frames of the bench walk through the bench network on the row-pair kernel -- with the folded
    detection and without it -- against the same network on round 2's row-segment kernel (CBINFER_NO_ROWPAIRS=1;
    another summation order, so <= 1e-4 instead of bit for bit), every layer's state after every frame."""
    import pycbinfer as pkg
    import bench
    nets = {}
    monkeypatch.setenv("CBINFER_NO_ROWPAIRS", "0")
    _, nets["fold"] = bench.build_bench_model()
    _, nets["sep"] = bench.build_bench_model(fuse_detect=False)
    _, nets["rows"] = bench.build_bench_model(fuse_detect=False)
    convs = {k: [m for m in n.children() if type(m) is pkg.CBConv2d] for k, n in nets.items()}
    frames = bench.bench_video(1234).frames(2 + 32)
    walk = frames[2:]
    seq = frames[:2] + [walk[bench.pingpong(i, len(walk))] for i in range(100)]
    with torch.no_grad():
        for t, f in enumerate(seq):
            ya, yb = nets["fold"](f), nets["sep"](f)
            monkeypatch.setenv("CBINFER_NO_ROWPAIRS", "1")
            yr = nets["rows"](f)
            monkeypatch.setenv("CBINFER_NO_ROWPAIRS", "0")
            assert torch.equal(ya, yb), t
            for a, b, r in zip(convs["fold"], convs["sep"], convs["rows"]):
                assert torch.equal(a.prevOutput, b.prevOutput) and torch.equal(a.prevInput, b.prevInput), t
                assert float((a.prevOutput - r.prevOutput).abs().max()) <= FP32_TOL, t
                assert float((a.prevInput - r.prevInput).abs().max()) <= 2 * FP32_TOL, t
    assert convs["fold"][0]._plan.get('pairs') and convs["fold"][0]._plan.get('nextToken') is not None
    assert convs["rows"][0]._plan is not None and not convs["rows"][0]._plan.get('pairs')


@pytest.mark.parametrize("H,W,th", [(64, 96, 0.05), (45, 67, 0.05), (90, 200, 0.2), (320, 480, 0.05), (33, 130, -1.0),
                                    (8, 10, 0.05), (6, 70, 0.05)])
def test_rowpairs_with_their_own_detection(lib, H, W, th):
    """cbinfer_conv_rowpairs_detect + cbinfer_refresh_state (round 6: the layer's own change detection inside the row-pair
    launch, the state refreshed behind it) against cbinfer_cbconv2d_forward_rowpairs (detection launch + row pairs): outputs,
    refreshed state, the frame's dilated mask and -- with the next layer's pooled detection folded in -- that layer's state,
    split copy and frame mask, bit for bit over a sequence of frames; odd sizes, a map narrower than a mask word's multiple,
    threshold -1 (every pixel changes)."""
    C_ = lib.C
    rng = np.random.default_rng(H * 3 + W)
    C, K, k, k2 = 3, 16, 7, 7
    H2, W2 = H // 2, W // 2
    w = (rng.standard_normal((K, C, k, k)) / np.sqrt(C * k * k)).astype(np.float32)
    b = rng.standard_normal(K).astype(np.float32)
    wp = torch.empty(C_.cbinfer_rowconv_prepared_bytes(C, K, k, k), dtype=torch.uint8, device="cuda")
    lib.check(C_.cbinfer_rowconv_prep_weights(dev(w).data_ptr(), wp.data_ptr(), K, C, k, k, None))
    bd = dev(b)
    words = C_.cbinfer_mask_words(H, W)

    class Side(object):
        def __init__(self):
            self.state = torch.zeros((1, C, H, W), device="cuda")
            self.out = torch.zeros((1, K, H, W), device="cuda")
            self.bits = torch.zeros(words, dtype=torch.int64, device="cuda")
            self.ctl = torch.zeros(words, dtype=torch.int32, device="cuda")
            self.copy = torch.zeros(words, dtype=torch.int64, device="cuda")
            self.state2 = torch.zeros((1, K, H2, W2), device="cuda")
            self.S2 = torch.empty(C_.cbinfer_split3_state_bytes(K, H2, W2, k2, k2), dtype=torch.uint8, device="cuda")
            lib.check(C_.cbinfer_split3_state_init(self.S2.data_ptr(), K, H2, W2, k2, k2, None))
            lib.check(C_.cbinfer_split3_state_rebuild(self.state2.data_ptr(), self.S2.data_ptr(), K, H2, W2, k2, k2, None))
            self.mask2 = torch.zeros(C_.cbinfer_frame_mask_bytes(H2, W2) // 8, dtype=torch.int64, device="cuda")
            self.nd = lib.NextDetect()
            nd = self.nd
            nd.state, nd.splitState, nd.frameMasks = self.state2.data_ptr(), self.S2.data_ptr(), self.mask2.data_ptr()
            nd.rangeFlag, nd.H, nd.W, nd.kH, nd.kW, nd.threshold, nd.arith = None, H2, W2, k2, k2, 0.07, 1

    a, d = Side(), Side()
    x = rng.standard_normal((1, C, H, W)).astype(np.float32)
    changed = 0
    for t in range(6):
        x = x.copy()
        if t > 0:
            for _ in range(max(1, H * W // 1500)):
                y0, x0 = rng.integers(0, H - 4), rng.integers(0, W - 4)
                x[0, :, y0:y0 + rng.integers(2, 9), x0:x0 + rng.integers(2, 30)] = rng.standard_normal((C, 1, 1))
            x += rng.uniform(-0.02, 0.02, x.shape).astype(np.float32)      # (sub-threshold noise must not leak in)
        xd = dev(x)
        lib.check(C_.cbinfer_cbconv2d_forward_rowpairs(xd.data_ptr(), a.state.data_ptr(), a.out.data_ptr(),
                                                       a.bits.data_ptr(), a.ctl.data_ptr(), a.copy.data_ptr(),
                                                       wp.data_ptr(), bd.data_ptr(), C, H, W, K, k, k, th, 1,
                                                       ctypes.pointer(a.nd), None))
        lib.check(C_.cbinfer_conv_rowpairs_detect(xd.data_ptr(), d.state.data_ptr(), d.out.data_ptr(), d.copy.data_ptr(),
                                                  wp.data_ptr(), bd.data_ptr(), C, H, W, K, k, k, th, 1,
                                                  ctypes.pointer(d.nd), None))
        lib.check(C_.cbinfer_refresh_state(xd.data_ptr(), d.state.data_ptr(), C, H, W, th, None))
        torch.cuda.synchronize()
        assert torch.equal(a.copy, d.copy), t
        assert torch.equal(a.out, d.out), t
        assert torch.equal(a.state, d.state), t
        assert torch.equal(a.state2, d.state2) and torch.equal(a.S2, d.S2) and torch.equal(a.mask2, d.mask2), t
        changed += int(torch.count_nonzero(d.copy).item())
        a.mask2.zero_(), d.mask2.zero_()
    assert changed > 0
