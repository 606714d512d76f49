"""-m gpu parity tests of the HIP ops (through the C ABI) against the CPU oracle, the golden fixtures
emitted by the reference, and -- where oracle/_ref was built -- the reference's own GPU kernels.

Bars: bit-exact for change masks, index lists and every pure data-movement op (gather, scatter, pool);
|err| <= 1e-4 (north-star fp32 tolerance) for anything that went through the contraction.
"""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

FP32_TOL = 1e-4


@pytest.fixture(scope="module")
def cb():
    import cbinfer_amd
    from cbinfer_amd import conv2d_cg, conv2d_fg
    assert torch.cuda.is_available()
    return cbinfer_amd, conv2d_cg, conv2d_fg


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def rand_case(rng, C, H, W, frac, th=0.1, blocks=False):
    inp = rng.standard_normal((1, C, H, W)).astype(np.float32)
    prev = inp.copy()
    if blocks:
        n = max(1, int(frac * H * W / 64))
        for _ in range(n):
            y0, x0 = rng.integers(0, max(1, H - 8)), rng.integers(0, max(1, W - 8))
            prev[0, :, y0:y0 + 8, x0:x0 + 8] += rng.uniform(1, 2)
    else:
        m = rng.random((H, W)) < frac
        c = rng.integers(0, C, (H, W))
        for ch in range(C):
            prev[0, ch][m & (c == ch)] += 1.0
    # sub-threshold noise everywhere: must never trigger
    prev += (rng.uniform(-0.4, 0.4, prev.shape) * th).astype(np.float32)
    return inp, prev


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden",
                                                               "ops_case*.npz"))))
def test_golden_ops(cb, case):
    _, cg, _ = cb
    d = dict(np.load(case))
    filt = tuple(int(v) for v in d["filtSize"])
    th = float(d["threshold"])
    inp, prev = dev(d["input"]), dev(d["prevInput"])
    cm = cg.changeDetection(inp, prev.clone(), filt, th)
    assert np.array_equal(cm.cpu().numpy(), d["changeMap"])
    cm1 = cg.changeDetection(inp, prev.clone(), (1, 1), th)
    assert np.array_equal(cm1.cpu().numpy(), d["changeMap1x1"])
    assert np.array_equal(cg.changePropagation(cm1, filt).cpu().numpy(), d["changeMap"])
    idx = cg.changeIndexesExtr(cm)
    assert idx.dtype == torch.int32
    assert np.array_equal(idx.cpu().numpy(), d["changeIndexes"])
    X = cg.genXMatrix(inp, idx, filt)
    assert np.array_equal(X.cpu().numpy(), d["X"])
    w, b = dev(d["weight"]), dev(d["bias"])
    Y = cg.matrixMult(X, w, b)
    np.testing.assert_allclose(Y.cpu().numpy(), d["Y"], rtol=0, atol=FP32_TOL)
    Yt = cg.matrixMult(X, w, b, transposeOut=True)
    assert torch.equal(Yt, Y.t().contiguous())
    out = cg.updateOutput(dev(d["Y"]).t(), idx, dev(d["prevOutput"]), withReLU=False)
    assert np.array_equal(out.cpu().numpy(), d["out_plain"])
    out = cg.updateOutput(dev(d["Y"]).t(), idx, dev(d["prevOutput"]), withReLU=True)
    assert np.array_equal(out.cpu().numpy(), d["out_relu"])
    # fused gather -> MFMA -> scatter == the four-op chain
    for relu, key in ((False, "out_plain"), (True, "out_relu")):
        fused = cg.convChanged(inp, idx, w, b, dev(d["prevOutput"]), withReLU=relu)
        np.testing.assert_allclose(fused.cpu().numpy(), d[key], rtol=0, atol=FP32_TOL)


def test_kats(cb, golden_dir):
    _, cg, _ = cb
    k = dict(np.load(os.path.join(golden_dir, "kat_genTestData.npz")))
    rng = np.random.default_rng(0)
    inp = rng.standard_normal(tuple(k["shape"])).astype(np.float32)
    prev = inp.copy()
    for c, y, x, dl in k["points"]:
        prev[0, int(c), int(y), int(x)] += np.float32(dl)
    cm = cg.changeDetection(dev(inp), dev(prev), (3, 3), 0.1)
    assert cg.changeIndexesExtr(cm).cpu().tolist() == k["changeIndexes"].tolist()
    k2 = dict(np.load(os.path.join(golden_dir, "kat_changeIndexesExtr.npz")))
    cmk = torch.zeros(tuple(k2["shape"]), dtype=torch.int8)
    for y, x in k2["points"]:
        cmk[y, x] = 1
    assert cg.changeIndexesExtr(cmk.cuda()).cpu().tolist() == [259, 765, 1277, 1779, 1783, 6127]


@pytest.mark.parametrize("C,H,W,filt,frac,blocks", [
    (3, 320, 480, (7, 7), 0.10, True),      # scene-labeling L1
    (16, 160, 240, (7, 7), 0.10, True),     # L2
    (64, 80, 120, (7, 7), 0.10, True),      # L3
    (256, 80, 120, (1, 1), 0.10, False),    # L4
    (5, 37, 131, (3, 5), 0.05, False),      # ragged: odd sizes, non-square filter
    (1, 1, 1, (3, 3), 1.0, False),          # single pixel
    (2, 7, 300, (9, 9), 0.02, False),       # wider halo than the common cases
    (4, 46, 81, (7, 7), 0.30, True),        # OpenPose stage resolution
])
@pytest.mark.parametrize("update", [False, True])
def test_detection_and_indexes(cb, oracle, C, H, W, filt, frac, blocks, update):
    """mask, feedback-updated state and index list: bit-exact vs the oracle (and vs the reference's
    own kernel when oracle/_ref exists)."""
    _, cg, _ = cb
    rng = np.random.default_rng(C * 1000 + H)
    inp, prev = rand_case(rng, C, H, W, frac, blocks=blocks)
    st_o = prev.copy()
    cm_o = oracle.changeDetection(inp, st_o, filt, 0.1, updateInputState=update)
    st_g = dev(prev)
    cm_g = cg.changeDetection(dev(inp), st_g, filt, 0.1, updateInputState=update)
    assert np.array_equal(cm_g.cpu().numpy(), cm_o)
    assert np.array_equal(st_g.cpu().numpy(), st_o)
    idx = cg.changeIndexesExtr(cm_g)
    assert np.array_equal(idx.cpu().numpy(), oracle.changeIndexesExtr(cm_o))
    # bit-mask form + its compaction (what the frame pipeline uses)
    from cbinfer_amd._lib import C as lib, check, stream_ptr
    bits = torch.zeros(lib.cbinfer_mask_words(H, W), dtype=torch.int64, device="cuda")
    st_b = dev(prev)
    check(lib.cbinfer_change_detection_bits(dev(inp).data_ptr(), st_b.data_ptr(), bits.data_ptr(), W, H,
                                            C, (filt[0] - 1) // 2, (filt[1] - 1) // 2, 0.1, int(update),
                                            0, stream_ptr()))
    idx_b = torch.empty(H * W, dtype=torch.int32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    mp = torch.empty(H, W, dtype=torch.int8, device="cuda")
    check(lib.cbinfer_compact_bits(bits.data_ptr(), W, H, idx_b.data_ptr(), cnt.data_ptr(), None,
                                   mp.data_ptr(), stream_ptr()))
    n = int(cnt.item())
    assert n == idx.numel()
    assert torch.equal(idx_b[:n], idx)
    assert np.array_equal(mp.cpu().numpy(), cm_o)
    assert np.array_equal(st_b.cpu().numpy(), st_o)
    import refgpu
    if refgpu.available():
        st_r = dev(prev)
        cm_r = refgpu.changeDetection(dev(inp), st_r, filt, 0.1, update)
        assert torch.equal(cm_r, cm_g)
        assert torch.equal(st_r, st_g)


def test_detection_edge_semantics(cb):
    _, cg, _ = cb
    inp = torch.zeros(1, 1, 4, 4, device="cuda")
    st = inp.clone()
    st[0, 0, 1, 1] = 0.5
    assert cg.changeDetection(inp, st.clone(), (1, 1), 0.5).sum().item() == 0     # strict >
    assert cg.changeDetection(inp, st.clone(), (1, 1), 0.49).sum().item() == 1
    assert cg.changeDetection(inp, torch.full_like(inp, float("inf")), (1, 1), 0.1).all()
    assert cg.changeDetection(torch.full_like(inp, float("nan")), st.clone(), (1, 1), 0.1).sum().item() == 0
    # empty map -> empty index list
    assert cg.changeIndexesExtr(torch.zeros(5, 7, dtype=torch.int8, device="cuda")).numel() == 0
    # full map -> every index, ascending
    full = cg.changeIndexesExtr(torch.ones(129, 254, dtype=torch.int8, device="cuda"))
    assert torch.equal(full.cpu(), torch.arange(129 * 254, dtype=torch.int32))


def test_detection_half(cb, oracle):
    _, cg, _ = cb
    rng = np.random.default_rng(11)
    inp = rng.standard_normal((1, 8, 30, 70)).astype(np.float16)
    prev = inp.copy()
    m = rng.random((30, 70)) < 0.05
    prev[0, 3][m] += np.float16(0.5)
    prev += (rng.uniform(-0.2, 0.2, prev.shape)).astype(np.float16) * np.float16(0.1)  # incl. near-ties
    for update in (False, True):
        st_o = prev.copy()
        cm_o = oracle.changeDetection_half(inp, st_o, (3, 3), 0.01, updateInputState=update)
        st_g = dev(prev)
        cm_g = cg.changeDetection(dev(inp), st_g, (3, 3), 0.01, updateInputState=update, useHalf=True)
        assert np.array_equal(cm_g.cpu().numpy(), cm_o)
        assert np.array_equal(st_g.cpu().numpy(), st_o)


@pytest.mark.parametrize("C,K,H,W,filt,frac", [
    (3, 16, 320, 480, (7, 7), 0.10),
    (16, 64, 160, 240, (7, 7), 0.10),
    (64, 256, 80, 120, (7, 7), 0.10),
    (256, 64, 80, 120, (1, 1), 0.10),
    (64, 8, 80, 120, (1, 1), 0.10),
    (5, 7, 23, 41, (3, 5), 0.2),
    (3, 33, 19, 67, (3, 3), 1.0),           # K just above one MFMA tile, every pixel changed
    (4, 65, 21, 70, (5, 5), 0.5),           # K just above the 64-row workgroup tile
    (2, 100, 9, 33, (1, 1), 1.0),           # K between tiles, 1x1 filter, narrow map
])
def test_gather_gemm_scatter_fullsize(cb, oracle, C, K, H, W, filt, frac):
    """BASELINE-size layers: X exact, Y and the fused kernel within 1e-4 of the oracle (double
    accumulation), scatter exact; plus agreement with the reference's own GPU kernels."""
    _, cg, _ = cb
    rng = np.random.default_rng(K)
    inp, prev = rand_case(rng, C, H, W, frac, blocks=True)
    w = (rng.standard_normal((K, C) + filt) / np.sqrt(C * filt[0] * filt[1])).astype(np.float32)
    b = rng.standard_normal(K).astype(np.float32)
    cm = oracle.changeDetection(inp, prev.copy(), filt, 0.1)
    idx_o = oracle.changeIndexesExtr(cm)
    idx = dev(idx_o)
    X = cg.genXMatrix(dev(inp), idx, filt)
    X_o = oracle.genXMatrix(inp, idx_o, filt)
    assert np.array_equal(X.cpu().numpy(), X_o)
    Y = cg.matrixMult(X, dev(w), dev(b), transposeOut=True)
    Y_o = oracle.matrixMult(X_o, w, b)
    np.testing.assert_allclose(Y.cpu().numpy(), Y_o.T, rtol=0, atol=FP32_TOL)
    po = rng.standard_normal((1, K, H, W)).astype(np.float32)
    for relu in (False, True):
        out_o = oracle.updateOutput(np.ascontiguousarray(Y_o.T), idx_o, po.copy(), withReLU=relu)
        out = cg.updateOutput(dev(np.ascontiguousarray(Y_o.T)), idx, dev(po), withReLU=relu)
        assert np.array_equal(out.cpu().numpy(), out_o)
        fused = cg.convChanged(dev(inp), idx, dev(w), dev(b), dev(po), withReLU=relu)
        np.testing.assert_allclose(fused.cpu().numpy(), out_o, rtol=0, atol=FP32_TOL)
    import refgpu
    if refgpu.available() and idx.numel():
        assert torch.equal(refgpu.genXMatrix(dev(inp), idx, filt), X)
        assert torch.equal(refgpu.updateOutput(Y, idx, dev(po), True), cg.updateOutput(Y, idx, dev(po), True))


def test_device_side_count(cb, oracle):
    """A ChangeIndexes (capacity buffer + device count) drives gather/GEMM/scatter without a sync."""
    _, cg, _ = cb
    rng = np.random.default_rng(2)
    inp, prev = rand_case(rng, 4, 40, 70, 0.1)
    w = rng.standard_normal((6, 4, 3, 3)).astype(np.float32) * 0.2
    b = rng.standard_normal(6).astype(np.float32)
    cm = cg.changeDetection(dev(inp), dev(prev), (3, 3), 0.1)
    ci = cg.changeIndexesExtrAsync(cm)
    assert isinstance(ci, cg.ChangeIndexes) and ci.buffer.numel() == 40 * 70
    po = rng.standard_normal((1, 6, 40, 70)).astype(np.float32)
    out_async = cg.convChanged(dev(inp), ci, dev(w), dev(b), dev(po), withReLU=True)
    out_sync = cg.convChanged(dev(inp), ci.tensor(), dev(w), dev(b), dev(po), withReLU=True)
    assert torch.equal(out_async, out_sync)
    idx_o = oracle.changeIndexesExtr(cm.cpu().numpy())
    Y_o = oracle.matrixMult(oracle.genXMatrix(inp, idx_o, (3, 3)), w, b)
    out_o = oracle.updateOutput(np.ascontiguousarray(Y_o.T), idx_o, po.copy(), withReLU=True)
    np.testing.assert_allclose(out_async.cpu().numpy(), out_o, rtol=0, atol=FP32_TOL)


@pytest.mark.parametrize("C,H,W,ceil", [(16, 320, 480, False), (64, 160, 240, False), (3, 11, 15, False),
                                        (3, 11, 15, True), (5, 184, 327, False)])
def test_maxpool(cb, oracle, C, H, W, ceil):
    _, cg, _ = cb
    rng = np.random.default_rng(H)
    x = rng.standard_normal((1, C, H, W)).astype(np.float32)
    oh, ow = ((H - 1) // 2 + 1, (W - 1) // 2 + 1) if ceil else (H // 2, W // 2)
    chg = np.sort(rng.choice(H * W, max(1, H * W // 10), replace=False)).astype(np.int32)
    st_o = rng.standard_normal((1, C, oh, ow)).astype(np.float32)
    st_g = dev(st_o)
    oracle.maxPool2d(x, st_o, chg, guardOutput=True)
    cg.maxPool2d(dev(x), st_g, dev(chg), (2, 2), (2, 2))
    assert np.array_equal(st_g.cpu().numpy(), st_o)
    # all pixels changed => plain max pooling
    full = torch.full((1, C, oh, ow), float("inf"), device="cuda")
    cg.maxPool2d(dev(x), full, torch.arange(H * W, dtype=torch.int32, device="cuda"), (2, 2), (2, 2))
    ref = torch.nn.functional.max_pool2d(dev(x), 2, 2, ceil_mode=ceil)
    assert torch.equal(full, ref)
    import refgpu
    if refgpu.available() and H % 2 == 0 and W % 2 == 0:   # the reference writes out of bounds otherwise
        st_r = dev(st_o * 0 + 7)
        st_m = st_r.clone()
        refgpu.maxPool2d(dev(x), st_r, dev(chg))
        cg.maxPool2d(dev(x), st_m, dev(chg), (2, 2), (2, 2))
        assert torch.equal(st_r, st_m)


def test_maxpool_half(cb):
    _, cg, _ = cb
    x = torch.randn(1, 6, 20, 30, device="cuda").half()
    out = torch.full((1, 6, 10, 15), float("inf"), device="cuda", dtype=torch.float16)
    cg.maxPool2d(x, out, torch.arange(600, dtype=torch.int32, device="cuda"), (2, 2), (2, 2), useHalf=True)
    assert torch.equal(out, torch.nn.functional.max_pool2d(x.float(), 2, 2).half())


def test_finegrained(cb, oracle, golden_dir):
    _, _, fg = cb
    # the reference's own test vector (conv2d_fg.py:98-150): error < 1e-6 against dense
    for name in ("fg_test1.npz", "fg_case1.npz"):
        d = dict(np.load(os.path.join(golden_dir, name)))
        th = float(d["threshold"])
        out = fg.cbconvFG(dev(d["input"]), dev(d["prevInput"]), dev(d["prevOutput"]), dev(d["weight"]), th)
        np.testing.assert_allclose(out.cpu().numpy(), d["output"], rtol=0, atol=FP32_TOL)
        if "outputRef" in d:
            assert np.abs(out.cpu().numpy() - d["outputRef"]).max() < 1e-6
        det = fg.cbconvFG_deterministic(dev(d["input"]), dev(d["prevInput"]), dev(d["prevOutput"]),
                                        dev(d["weight"]), th)
        np.testing.assert_allclose(det.cpu().numpy(), d["output"], rtol=0, atol=FP32_TOL)
        # host path of the same entry point (cbconv2d_fg_backend.cu:81-112)
        cpu = fg.cbconvFG(torch.from_numpy(d["input"]), torch.from_numpy(d["prevInput"]),
                          torch.from_numpy(d["prevOutput"].copy()), torch.from_numpy(d["weight"]), th)
        np.testing.assert_allclose(cpu.numpy(), d["output"], rtol=0, atol=1e-5)
    # scene-labeling L2-shaped random case vs the oracle and the reference's kernels
    rng = np.random.default_rng(9)
    inp = rng.standard_normal((1, 16, 40, 60)).astype(np.float32)
    prev = inp + ((rng.random(inp.shape) < 0.03) * rng.standard_normal(inp.shape)).astype(np.float32)
    w = (rng.standard_normal((24, 16, 7, 7)) * 0.05).astype(np.float32)
    po = oracle.conv2d_dense(prev, w, None)
    diffs_g, cm_g = fg.changeDetectionFG(dev(inp), dev(prev), 0.2, zeroUnchanged=True)
    diffs_o, cm_o = oracle.changeDetectionFG(inp, prev, 0.2)
    assert np.array_equal(cm_g.cpu().numpy(), cm_o)          # mask: bit-exact
    assert np.array_equal(diffs_g.cpu().numpy(), diffs_o)
    coords = np.nonzero(cm_o.reshape(-1))[0].astype(np.int64)
    out_o = oracle.updateOutputFG(diffs_o, w, po.copy(), coords)
    out_g = fg.updateOutputFG(diffs_g, dev(w), dev(po), dev(coords))
    np.testing.assert_allclose(out_g.cpu().numpy(), out_o, rtol=0, atol=FP32_TOL)
    import refgpu
    if refgpu.available():
        d_r, cm_r = refgpu.changeDetectionFG(dev(inp), dev(prev), 0.2)
        assert torch.equal(cm_r, cm_g)
        out_r = refgpu.updateOutputFG(d_r, dev(w), dev(po), dev(coords))
        np.testing.assert_allclose(out_g.cpu().numpy(), out_r.cpu().numpy(), rtol=0, atol=FP32_TOL)


def test_contraction_half(cb, oracle):
    """fp16 path (cg_half): fp16 operands, f32 accumulation, result rounded to fp16.  The reference
    pins no tolerance for it; the bar here is 2 fp16 ulp of the largest |output|
    (= 2 * 2^-10 relative) against the double-accumulated oracle on the same fp16 inputs."""
    _, cg, _ = cb
    rng = np.random.default_rng(4)
    C, K, H, W, filt = 32, 48, 46, 81, (3, 3)
    inp = rng.standard_normal((1, C, H, W)).astype(np.float16)
    w = (rng.standard_normal((K, C) + filt) / np.sqrt(C * 9)).astype(np.float16)
    b = rng.standard_normal(K).astype(np.float16)
    idx_o = np.sort(rng.choice(H * W, 700, replace=False)).astype(np.int32)
    X = cg.genXMatrix(dev(inp), dev(idx_o), filt, useHalf=True)
    X_o = oracle.genXMatrix(inp.astype(np.float32), idx_o, filt)
    assert np.array_equal(X.cpu().numpy().astype(np.float32), X_o)
    Y_o = oracle.matrixMult(X_o, w.astype(np.float32), b.astype(np.float32))
    tol = 2 * 2.0 ** -10 * float(np.abs(Y_o).max())
    Y = cg.matrixMult(X, dev(w), dev(b))
    np.testing.assert_allclose(Y.float().cpu().numpy(), Y_o, rtol=0, atol=tol)
    po = torch.zeros(1, K, H, W, device="cuda", dtype=torch.float16)
    fused = cg.convChanged(dev(inp), dev(idx_o), dev(w), dev(b), po, withReLU=True)
    got = fused.float().cpu().numpy().reshape(K, -1)[:, idx_o]
    np.testing.assert_allclose(got, np.maximum(Y_o.T, 0), rtol=0, atol=tol)


def test_c_abi_rejects_bad_arguments(cb):
    from cbinfer_amd._lib import C as lib, CBinferError, check
    with pytest.raises(CBinferError):
        check(lib.cbinfer_change_detection(None, None, None, 4, 4, 1, 0, 0, 0.1, 0, 0, None))
    with pytest.raises(CBinferError):
        check(lib.cbinfer_conv_changed(None, None, 1, None, None, None, None, 1, 1, 1, 1, 1, 1, 0, 0, None,
                                       0, None, 0, None))


@pytest.mark.parametrize("size", [(8, 8, False), (7, 9, False), (7, 9, True), (160, 240, False), (37, 130, True)])
def test_pool_change_indexes_vs_oracle(oracle, size):
    """8f-4: index list of the pooled map from the input list, bit-exact incl. order, device-side count."""
    from cbinfer_amd import conv2d_cg as cg
    H, W, ceil = size
    oH, oW = ((H - 1) // 2 + 1, (W - 1) // 2 + 1) if ceil else (H // 2, W // 2)
    g = torch.Generator().manual_seed(H * 1000 + W)
    for frac in (0.0, 0.03, 0.4, 1.0):
        m = torch.rand(H, W, generator=g) < frac
        idx = torch.nonzero(m.view(-1)).view(-1).int()
        want = oracle.poolChangeIndexes(idx.numpy(), (H, W), (oH, oW))
        # exact-length tensor and capacity buffer + device count
        got = cg.poolChangeIndexes(idx.cuda(), (H, W), (oH, oW))
        assert got.tensor().cpu().numpy().tolist() == want.tolist()
        buf = torch.zeros(H * W, dtype=torch.int32)
        buf[:idx.numel()] = idx
        ci = cg.ChangeIndexes(buf.cuda(), torch.tensor([idx.numel()], dtype=torch.int32).cuda())
        got = cg.poolChangeIndexes(ci, (H, W), (oH, oW))
        assert got.tensor().cpu().numpy().tolist() == want.tolist()
