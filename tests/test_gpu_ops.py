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


# ------------------------------------------------------------------------------------------------
# the reference-signature compat shims, executed on the GPU
# ------------------------------------------------------------------------------------------------
def _compat_case(dtype, seed=31):
    rng = np.random.default_rng(seed)
    C, H, W = 6, 37, 83
    inp = rng.standard_normal((1, C, H, W)).astype(np.float32)
    prev = inp.copy()
    ys, xs = rng.integers(0, H, 40), rng.integers(0, W, 40)
    prev[0, rng.integers(0, C, 40), ys, xs] += rng.choice([-1.0, 1.0], 40).astype(np.float32)
    prev[0, 0, 0, 0] += 1.0
    prev[0, C - 1, H - 1, W - 1] -= 1.0
    return inp.astype(dtype), prev.astype(dtype)


@pytest.mark.parametrize("symbol", ["changeDetection", "changePropagation", "genXMatrix", "updateOutput",
                                    "maxPool2d"])
def test_compat_cg_shim_matches_reference_kernels(oracle, symbol):
    """cbinfer_amd/compat/cbconv2d_cg_backend_<machine>.so called with the reference's own argument lists
    (conv2d_cg.py:6-38: six geometry ints, bool* map) must give what the reference's kernels give
    (oracle/_ref, compiled from the reference's .cu) -- and what the oracle says when _ref is absent."""
    import refgpu
    so = refgpu.compat("cg")
    ref = refgpu.available()
    inp, prev = _compat_case(np.float32)
    filt, th = (5, 3), 0.3
    if symbol == "changeDetection":
        for update in (False, True):
            st = dev(prev)
            cm = refgpu.changeDetection(dev(inp), st, filt, th, update, so=so)
            st_o = prev.copy()
            cm_o = oracle.changeDetection(inp, st_o, filt, th, updateInputState=update)
            assert np.array_equal(cm.cpu().numpy(), cm_o) and np.array_equal(st.cpu().numpy(), st_o)
            if ref:
                st_r = dev(prev)
                assert torch.equal(cm, refgpu.changeDetection(dev(inp), st_r, filt, th, update))
                assert torch.equal(st, st_r)
        return
    cm_o = oracle.changeDetection(inp, prev.copy(), (1, 1), th)
    if symbol == "changePropagation":
        out = refgpu.changePropagation(dev(cm_o), filt, so=so)
        assert np.array_equal(out.cpu().numpy(), oracle.changePropagation(cm_o, filt))
        if ref:
            assert torch.equal(out, refgpu.changePropagation(dev(cm_o), filt))
        return
    idx_o = oracle.changeIndexesExtr(oracle.changePropagation(cm_o, filt))
    if symbol == "genXMatrix":
        X = refgpu.genXMatrix(dev(inp), dev(idx_o), filt, so=so)
        assert np.array_equal(X.cpu().numpy(), oracle.genXMatrix(inp, idx_o, filt))
        if ref:
            assert torch.equal(X, refgpu.genXMatrix(dev(inp), dev(idx_o), filt))
    elif symbol == "updateOutput":
        rng = np.random.default_rng(5)
        K, (H, W) = 7, inp.shape[-2:]
        Yt = rng.standard_normal((K, idx_o.size)).astype(np.float32)
        base = rng.standard_normal((1, K, H, W)).astype(np.float32)
        for relu in (False, True):
            out = refgpu.updateOutput(dev(Yt), dev(idx_o), dev(base), relu, so=so)
            assert np.array_equal(out.cpu().numpy(), oracle.updateOutput(Yt, idx_o, base.copy(), relu))
            if ref:
                assert torch.equal(out, refgpu.updateOutput(dev(Yt), dev(idx_o), dev(base), relu))
    else:
        C, H, W = inp.shape[-3:]
        even = inp[:, :, :H - 1, :W - 1].copy()          # 36 x 82: the reference kernel has no output guard
        idx_e = oracle.changeIndexesExtr((np.random.default_rng(6).random((H - 1, W - 1)) < 0.2).astype(np.int8))
        base = np.full((1, C, (H - 1) // 2, (W - 1) // 2), np.inf, np.float32)
        out = refgpu.maxPool2d(dev(even), dev(base), dev(idx_e), so=so)
        assert np.array_equal(out.cpu().numpy(), oracle.maxPool2d(even, base.copy(), idx_e))
        if ref:
            assert torch.equal(out, refgpu.maxPool2d(dev(even), dev(base), dev(idx_e)))


@pytest.mark.parametrize("symbol", ["changeDetection", "changePropagation", "genXMatrix", "updateOutput",
                                    "maxPool2d"])
def test_compat_cg_half_shim_matches_oracle(oracle, symbol):
    """The half library of the reference reuses the float* header (conv2d_cg.py:46-50): same symbols on
    fp16 buffers.  cbconv2d_cg_half_backend.cu cannot be built here, so the checker is the oracle's
    restatement of it."""
    import refgpu
    so = refgpu.compat("cg_half")
    inp, prev = _compat_case(np.float16)
    filt, th = (3, 5), 0.3
    if symbol == "changeDetection":
        for update in (False, True):
            st = dev(prev)
            cm = refgpu.changeDetection(dev(inp), st, filt, th, update, so=so)
            st_o = prev.copy()
            cm_o = oracle.changeDetection_half(inp, st_o, filt, th, updateInputState=update)
            assert np.array_equal(cm.cpu().numpy(), cm_o) and np.array_equal(st.cpu().numpy(), st_o)
        return
    cm_o = oracle.changeDetection_half(inp, prev.copy(), (1, 1), th)
    if symbol == "changePropagation":
        out = refgpu.changePropagation(dev(cm_o), filt, so=so)
        assert np.array_equal(out.cpu().numpy(), oracle.changePropagation(cm_o, filt))
        return
    idx_o = oracle.changeIndexesExtr(oracle.changePropagation(cm_o, filt))
    if symbol == "genXMatrix":
        X = refgpu.genXMatrix(dev(inp), dev(idx_o), filt, so=so)
        assert np.array_equal(X.cpu().numpy(), oracle.genXMatrix_half(inp, idx_o, filt))
    elif symbol == "updateOutput":
        rng = np.random.default_rng(5)
        K, (H, W) = 7, inp.shape[-2:]
        Yt = rng.standard_normal((K, idx_o.size)).astype(np.float16)
        base = rng.standard_normal((1, K, H, W)).astype(np.float16)
        for relu in (False, True):
            out = refgpu.updateOutput(dev(Yt), dev(idx_o), dev(base), relu, so=so)
            assert np.array_equal(out.cpu().numpy(), oracle.updateOutput_half(Yt, idx_o, base.copy(), relu))
    else:
        C, H, W = inp.shape[-3:]
        idx_e = oracle.changeIndexesExtr((np.random.default_rng(6).random((H, W)) < 0.2).astype(np.int8))
        base = np.full((1, C, H // 2, W // 2), np.inf, np.float16)
        out = refgpu.maxPool2d(dev(inp), dev(base), dev(idx_e), so=so)      # odd sizes: guarded here
        assert np.array_equal(out.cpu().numpy(), oracle.maxPool2d_half(inp, base.copy(), idx_e))


@pytest.mark.parametrize("symbol", ["changeDetectionFG", "updateOutputFG", "conv2d_fg_cpu"])
def test_compat_fg_shim_matches_reference_kernels(oracle, golden_dir, symbol):
    """cbconv2d_fg_backend shim with the reference's argument lists (conv2d_fg.py:14-29: const long*
    coordinates, geometry ints ignored)."""
    import refgpu
    so = refgpu.compat("fg")
    ref = refgpu.available()
    rng = np.random.default_rng(12)
    inp = rng.standard_normal((1, 5, 23, 41)).astype(np.float32)
    prev = inp + ((rng.random(inp.shape) < 0.05) * rng.standard_normal(inp.shape)).astype(np.float32)
    w = (rng.standard_normal((9, 5, 3, 5)) * 0.1).astype(np.float32)
    th = 0.15
    d_o, cm_o = oracle.changeDetectionFG(inp, prev, th)
    if symbol == "changeDetectionFG":
        d, cm = refgpu.changeDetectionFG(dev(inp), dev(prev), th, so=so)
        assert np.array_equal(cm.cpu().numpy(), cm_o)
        assert np.array_equal(d.cpu().numpy()[cm_o != 0], d_o[cm_o != 0])
        if ref:
            d_r, cm_r = refgpu.changeDetectionFG(dev(inp), dev(prev), th)
            assert torch.equal(cm, cm_r) and torch.equal(d[cm_r != 0], d_r[cm_r != 0])
    elif symbol == "updateOutputFG":
        coords = np.nonzero(cm_o.reshape(-1))[0].astype(np.int64).reshape(-1, 1)
        po = oracle.conv2d_dense(prev, w, None)
        out = refgpu.updateOutputFG(dev(d_o), dev(w), dev(po), dev(coords), so=so)
        np.testing.assert_allclose(out.cpu().numpy(), oracle.updateOutputFG(d_o, w, po.copy(), coords),
                                   rtol=0, atol=FP32_TOL)
        if ref:
            out_r = refgpu.updateOutputFG(dev(d_o), dev(w), dev(po), dev(coords))
            np.testing.assert_allclose(out.cpu().numpy(), out_r.cpu().numpy(), rtol=0, atol=FP32_TOL)
    else:
        d = dict(np.load(os.path.join(golden_dir, "fg_case1.npz")))
        out = torch.from_numpy(d["prevOutput"].copy())
        refgpu.conv2d_fg_cpu(torch.from_numpy(d["input"]), torch.from_numpy(d["prevInput"]), out,
                             torch.from_numpy(d["weight"]), float(d["threshold"]), so=so)
        np.testing.assert_allclose(out.numpy(), d["output"], rtol=0, atol=1e-5)


def test_fg_coordinate_extraction_and_list_scatter(cb, oracle):
    """a11: per-value coordinate extraction on the device (ascending int32 list + device count) equals
    numpy.nonzero of the oracle's change tensor, and the scatter driven by it (no host sync) equals the
    int64-list form and the oracle."""
    _, _, fg = cb
    rng = np.random.default_rng(19)
    inp = rng.standard_normal((1, 16, 40, 60)).astype(np.float32)
    prev = inp + ((rng.random(inp.shape) < 0.04) * rng.standard_normal(inp.shape)).astype(np.float32)
    w = (rng.standard_normal((24, 16, 7, 7)) * 0.05).astype(np.float32)
    th = 0.2
    diffs, cm = fg.changeDetectionFG(dev(inp), dev(prev), th)
    d_o, cm_o = oracle.changeDetectionFG(inp, prev, th)
    coords = fg.changeCoordsExtrFG(cm)
    expect = np.nonzero(cm_o.reshape(-1))[0]
    assert coords.buffer.dtype == torch.int32
    assert np.array_equal(coords.tensor().cpu().numpy(), expect.astype(np.int32))
    po = oracle.conv2d_dense(prev, w, None)
    out_list = fg.updateOutputFG(diffs, dev(w), dev(po), coords)
    out_o = oracle.updateOutputFG(d_o, w, po.copy(), expect.astype(np.int64))
    np.testing.assert_allclose(out_list.cpu().numpy(), out_o, rtol=0, atol=FP32_TOL)
    # empty change set: nothing is touched
    d0, cm0 = fg.changeDetectionFG(dev(inp), dev(inp), th)
    c0 = fg.changeCoordsExtrFG(cm0)
    assert c0.numel() == 0
    out0 = fg.updateOutputFG(d0, dev(w), dev(po), c0)
    assert np.array_equal(out0.cpu().numpy(), po)


def test_fg_frame_kernels_vs_oracle(cb, oracle):
    """cbinfer_cbconv2d_forward_fg (per-value detection + accumulating self-compacting contraction) against
    the oracle's forward_fg arithmetic: delta tensor and touched-pixel list bit-exact, output <= 1e-4,
    the relu'd copy consistent, prevInput refreshed exactly where input and state differ."""
    from cbinfer_amd._lib import C as lib, check, ptr
    from cbinfer_amd import conv2d_cg as cg
    rng = np.random.default_rng(23)
    for (C, K, H, W, k, frac) in [(16, 24, 40, 60, 7, 0.03), (3, 16, 64, 96, 7, 0.2), (64, 70, 20, 30, 3, 0.01)]:
        inp = rng.standard_normal((1, C, H, W)).astype(np.float32)
        prev = inp.copy()
        sel = rng.random(inp.shape) < frac
        prev[sel] += rng.standard_normal(int(sel.sum())).astype(np.float32)
        tiny = rng.random(inp.shape) < 0.05          # sub-threshold drift: absorbed into the state, not computed
        prev[tiny & ~sel] += 1e-3
        w = (rng.standard_normal((K, C, k, k)) / np.sqrt(C * k * k)).astype(np.float32)
        th = 0.1
        po = oracle.conv2d_dense(prev, w, None)
        d_o, cm_o = oracle.changeDetectionFG(inp, prev, th)
        coords = np.nonzero(cm_o.reshape(-1))[0].astype(np.int64)
        out_o = oracle.updateOutputFG(d_o, w, po.copy(), coords)
        touched_o = oracle.changeIndexesExtr(oracle.changePropagation(cm_o[0].max(axis=0), (k, k)))
        x, st, out = dev(inp), dev(prev), dev(po)
        relu = torch.relu(out)
        delta = torch.full_like(x, 7.0)
        bits = torch.zeros(lib.cbinfer_frame_mask_bytes(H, W) // 8 + 1, dtype=torch.int64, device="cuda")
        idx = torch.empty(H * W, dtype=torch.int32, device="cuda")
        cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
        wp = cg.prepWeights(dev(w), H, W)
        ws = cg.newConvWorkspace(x.device)
        for rep in range(2):        # second call: input == state -> nothing changes, masks were re-zeroed
            check(lib.cbinfer_cbconv2d_forward_fg(ptr(x), ptr(st), ptr(delta), ptr(out), ptr(relu), ptr(bits),
                                                  ptr(idx), ptr(cnt), ptr(wp), C, H, W, K, k, k, th, 1, ptr(ws),
                                                  0, None))
            torch.cuda.synchronize()
            if rep == 0:
                assert np.array_equal(delta.cpu().numpy(), np.where(cm_o != 0, d_o, 0).astype(np.float32))
                assert np.array_equal(idx[:int(cnt.item())].cpu().numpy(), touched_o)
            else:
                assert int(cnt.item()) == 0 and float(delta.abs().max()) == 0.0
            np.testing.assert_allclose(out.cpu().numpy(), out_o, rtol=0, atol=FP32_TOL)
            assert torch.equal(relu, torch.relu(out))
            assert torch.equal(st, x)


def test_tail1x1_vs_numpy(cb, oracle):
    """cbinfer_tail1x1: conv1x1 -> ReLU -> conv1x1 at listed pixels only (double-accumulated numpy reference,
    1e-4), other pixels untouched, device-side count honoured, foreign-resolution entries dropped."""
    from cbinfer_amd._lib import C as lib, check, ptr
    rng = np.random.default_rng(29)
    for (C0, C1, C2, H, W, n) in [(256, 64, 8, 80, 120, 3458), (20, 12, 5, 12, 16, 50), (128, 128, 38, 46, 81, 700),
                                  (7, 3, 1, 5, 9, 45)]:
        x = rng.standard_normal((1, C0, H, W)).astype(np.float32)
        w1 = (rng.standard_normal((C1, C0)) / np.sqrt(C0)).astype(np.float32)
        b1 = rng.standard_normal(C1).astype(np.float32)
        w2 = (rng.standard_normal((C2, C1)) / np.sqrt(C1)).astype(np.float32)
        b2 = rng.standard_normal(C2).astype(np.float32)
        idx = np.sort(rng.choice(H * W, n, replace=False)).astype(np.int32)
        xs = x.reshape(C0, -1)[:, idx].astype(np.float64)
        h = np.maximum(w1.astype(np.float64) @ xs + b1[:, None], 0)
        y = (w2.astype(np.float64) @ h + b2[:, None]).astype(np.float32)
        wp = torch.empty(lib.cbinfer_tail1x1_prepared_bytes(C1, C0) // 4, device="cuda")
        check(lib.cbinfer_tail1x1_prep(ptr(dev(w1)), ptr(wp), C1, C0, None))
        out = torch.full((1, C2, H, W), 123.0, device="cuda")
        cap = H * W
        buf = torch.full((cap,), H * W + 5, dtype=torch.int32, device="cuda")   # garbage beyond the count
        buf[:n] = dev(idx)
        cnt = torch.tensor([n], dtype=torch.int32, device="cuda")
        dw2, db1, db2, dx = dev(w2), dev(b1), dev(b2), dev(x)
        check(lib.cbinfer_tail1x1(ptr(dx), ptr(buf), cap, ptr(cnt), ptr(wp), ptr(db1), ptr(dw2), ptr(db2),
                                  ptr(out), C0, C1, C2, H, W, 1, 0, None))
        got = out.cpu().numpy().reshape(C2, -1)
        np.testing.assert_allclose(got[:, idx], y, rtol=0, atol=FP32_TOL)
        rest = np.setdiff1d(np.arange(H * W), idx)
        assert np.all(got[:, rest] == 123.0)
        # host count, list containing an out-of-map entry: dropped, nothing written through it
        out2 = torch.full((1, C2, H, W), 123.0, device="cuda")
        lst = torch.cat([dev(idx[:10]), torch.tensor([H * W + 3], dtype=torch.int32, device="cuda")])
        check(lib.cbinfer_tail1x1(ptr(dx), ptr(lst), 11, None, ptr(wp), ptr(db1), ptr(dw2), ptr(db2),
                                  ptr(out2), C0, C1, C2, H, W, 1, 0, None))
        got2 = out2.cpu().numpy().reshape(C2, -1)
        np.testing.assert_allclose(got2[:, idx[:10]], y[:, :10], rtol=0, atol=FP32_TOL)
        assert int((got2 != 123.0).sum()) == C2 * 10


@pytest.mark.parametrize("C,K,kH,kW,H,W,frac", [
    (3, 16, 7, 7, 64, 96, 0.05), (16, 64, 7, 7, 40, 60, 0.03), (5, 20, 3, 5, 37, 83, 0.02),
    (1, 1, 3, 3, 9, 130, 0.1), (4, 33, 5, 3, 21, 64, 0.02), (16, 64, 7, 7, 160, 240, 0.02),
    (3, 16, 7, 7, 320, 480, 0.01), (7, 16, 1, 7, 18, 200, 0.05)])
def test_rowconv_vs_oracle(cb, oracle, C, K, kH, kW, H, W, frac):
    """Row-segment contraction (cbinfer_conv_changed_rows) behind the single-mask detection: outputs at the
    changed pixels <= 1e-4 from the double-accumulated dense convolution of the state (incl. image borders,
    partial last mask word, K not a multiple of 16, C not a multiple of 4), every other output untouched,
    maskCopy == the frame's mask (its compaction = the oracle's index list), mask and arrival counters left
    zero; a second launch on the emptied mask changes nothing."""
    from cbinfer_amd._lib import C as lib, check, ptr
    assert lib.cbinfer_rowconv_supported(C, K, kH, kW)
    rng = np.random.default_rng(C * 131 + K)
    x, st_np = rand_case(rng, C, H, W, frac, th=0.1, blocks=True)
    w = (rng.standard_normal((K, C, kH, kW)) / np.sqrt(C * kH * kW)).astype(np.float32)
    b = rng.standard_normal(K).astype(np.float32)
    st_o = st_np.copy()
    cm_o = oracle.changeDetection(x, st_o, (kH, kW), 0.1, updateInputState=True)
    idx_o = oracle.changeIndexesExtr(cm_o)
    assert idx_o.size > 0
    dense = oracle.conv2d_dense(st_o, w, b, relu=True)
    words = lib.cbinfer_mask_words(H, W)
    bits = torch.zeros(words, dtype=torch.int64, device="cuda")
    arrive = torch.zeros(words, dtype=torch.int32, device="cuda")
    copy = torch.full((words,), -1, dtype=torch.int64, device="cuda")
    wq = torch.empty(lib.cbinfer_rowconv_prepared_bytes(C, K, kH, kW), dtype=torch.uint8, device="cuda")
    dw, db = dev(w), dev(b)
    check(lib.cbinfer_rowconv_prep_weights(ptr(dw), ptr(wq), K, C, kH, kW, None))
    dx, st = dev(x), dev(st_np)
    out = torch.full((1, K, H, W), 77.0, device="cuda")
    check(lib.cbinfer_change_detection_bits(ptr(dx), ptr(st), ptr(bits), W, H, C, (kH - 1) // 2, (kW - 1) // 2,
                                            0.1, 1, 0, None))
    mask = bits.clone()
    check(lib.cbinfer_conv_changed_rows(ptr(st), ptr(bits), ptr(arrive), ptr(copy), ptr(wq), ptr(db), ptr(out),
                                        C, H, W, K, kH, kW, 1, None))
    torch.cuda.synchronize()
    assert np.array_equal(st.cpu().numpy(), st_o)
    assert torch.equal(copy, mask) and int(bits.abs().sum()) == 0 and int(arrive.abs().sum()) == 0
    idx = torch.empty(H * W, dtype=torch.int32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    check(lib.cbinfer_compact_bits(ptr(copy), W, H, ptr(idx), ptr(cnt), None, None, None))
    assert np.array_equal(idx[:int(cnt.item())].cpu().numpy(), idx_o)
    got = out.cpu().numpy().reshape(K, -1)
    np.testing.assert_allclose(got[:, idx_o], dense.reshape(K, -1)[:, idx_o], rtol=0, atol=FP32_TOL)
    rest = np.setdiff1d(np.arange(H * W), idx_o)
    assert np.all(got[:, rest] == 77.0)
    before = out.clone()
    check(lib.cbinfer_conv_changed_rows(ptr(st), ptr(bits), ptr(arrive), ptr(copy), ptr(wq), ptr(db), ptr(out),
                                        C, H, W, K, kH, kW, 1, None))
    torch.cuda.synchronize()
    assert torch.equal(out, before) and int(copy.abs().sum()) == 0


@pytest.mark.parametrize("C,K,H,W,filt,frac", [(64, 256, 80, 120, (7, 7), 0.3), (16, 64, 160, 240, (7, 7), 0.1),
                                               (185, 128, 46, 81, (7, 7), 0.2), (5, 20, 33, 70, (3, 5), 0.5),
                                               (3, 16, 64, 96, (7, 7), 0.2),
                                               # the 128 x 128 form off its comfortable shapes: three row tiles
                                               # (no XCD-aware order), 129 and 1 changed pixels, every pixel
                                               (9, 384, 20, 33, (3, 3), 0.9), (64, 128, 17, 40, (5, 5), 0.19),
                                               (40, 256, 12, 31, (7, 7), 0.003), (12, 128, 31, 67, (3, 7), 1.0)])
def test_split_contraction_accuracy(cb, oracle, C, K, H, W, filt, frac):
    """CB_F32S: the fused contraction with every f32 operand split into three bf16 terms and six cross
    products on the bf16 MFMA.  Against the double-accumulated oracle it must meet the fp32 bar (1e-4) with a
    wide margin -- the dropped terms are below 2^-24 of each product -- also for operands spanning many
    binades; and it must agree with the exact-f32 kernel to a few f32 ulp of the accumulated magnitude."""
    from cbinfer_amd._lib import CB_F32S
    _, cg, _ = cb
    rng = np.random.default_rng(C + K)
    inp = (rng.standard_normal((1, C, H, W)) * np.exp(rng.uniform(-6, 3, (1, C, 1, 1)))).astype(np.float32)
    w = (rng.standard_normal((K, C) + filt) / np.sqrt(C * filt[0] * filt[1])).astype(np.float32)
    b = rng.standard_normal(K).astype(np.float32)
    idx_o = np.sort(rng.choice(H * W, int(frac * H * W), replace=False)).astype(np.int32)
    X_o = oracle.genXMatrix(inp, idx_o, filt)
    Y_o = oracle.matrixMult(X_o, w, b).T                     # [K, N], accumulated in double
    mag = np.abs(X_o).astype(np.float64) @ np.abs(w.reshape(K, -1)).astype(np.float64).T   # sum |a||b|
    res = {}
    for name, arith in (("split", CB_F32S), ("exact", None)):
        po = torch.zeros(1, K, H, W, device="cuda")
        cg.convChanged(dev(inp), dev(idx_o), dev(w), dev(b), po, withReLU=False, arith=arith)
        res[name] = po.cpu().numpy().reshape(K, -1)[:, idx_o]
    err_s = np.abs(res["split"] - Y_o)
    err_e = np.abs(res["exact"] - Y_o)
    print("C%d K%d: max |err| split %.3g, exact f32 %.3g; relative to sum|a||b| %.3g / %.3g"
          % (C, K, err_s.max(), err_e.max(), (err_s / mag.T).max(), (err_e / mag.T).max()))
    assert err_s.max() <= FP32_TOL
    # per-element bound: a few 2^-24 of the accumulated magnitude (each product is off by < 3 * 2^-24, plus
    # the f32 roundings of the running sum) -- i.e. no worse than the exact chain's own rounding error bound
    assert np.all(err_s <= 64 * 2.0 ** -24 * mag.T + 1e-30)


@pytest.mark.parametrize("C,K,kH,kW,H,W,frac", [
    (16, 64, 7, 7, 40, 60, 0.03), (64, 256, 7, 7, 20, 30, 0.05), (5, 20, 3, 5, 37, 83, 0.02),
    (8, 16, 3, 3, 9, 130, 0.1), (33, 70, 5, 3, 21, 64, 0.02), (16, 64, 7, 7, 160, 240, 0.02),
    (64, 256, 7, 7, 80, 120, 0.02), (3, 130, 2, 7, 18, 200, 0.05)])
def test_blockconv_vs_oracle(cb, oracle, C, K, kH, kW, H, W, frac):
    """Patch-staged contraction (cbinfer_conv_changed_blocks, bf16x3 arithmetic) behind the single-mask
    detection: same checks as test_rowconv_vs_oracle -- outputs at the changed pixels <= 1e-4 from the
    double-accumulated dense convolution of the state (image borders, partial last mask word, odd row count
    against the 2-row units, C not a multiple of 8, K not a multiple of 64), everything else untouched, mask
    copy / clearing protocol."""
    from cbinfer_amd._lib import C as lib, check, ptr
    assert lib.cbinfer_blockconv_supported(C, K, kH, kW)
    rng = np.random.default_rng(C * 131 + K)
    x, st_np = rand_case(rng, C, H, W, frac, th=0.1, blocks=True)
    w = (rng.standard_normal((K, C, kH, kW)) / np.sqrt(C * kH * kW)).astype(np.float32)
    b = rng.standard_normal(K).astype(np.float32)
    st_o = st_np.copy()
    cm_o = oracle.changeDetection(x, st_o, (kH, kW), 0.1, updateInputState=True)
    idx_o = oracle.changeIndexesExtr(cm_o)
    assert idx_o.size > 0
    dense = oracle.conv2d_dense(st_o, w, b, relu=True)
    words = lib.cbinfer_mask_words(H, W)
    bits = torch.zeros(words, dtype=torch.int64, device="cuda")
    arrive = torch.zeros(words, dtype=torch.int32, device="cuda")
    copy = torch.full((words,), -1, dtype=torch.int64, device="cuda")
    wq = torch.empty(lib.cbinfer_blockconv_prepared_bytes(C, K, kH, kW), dtype=torch.uint8, device="cuda")
    dw, db = dev(w), dev(b)
    check(lib.cbinfer_blockconv_prep_weights(ptr(dw), ptr(wq), K, C, kH, kW, None))
    dx, st = dev(x), dev(st_np)
    out = torch.full((1, K, H, W), 77.0, device="cuda")
    check(lib.cbinfer_change_detection_bits(ptr(dx), ptr(st), ptr(bits), W, H, C, (kH - 1) // 2, (kW - 1) // 2,
                                            0.1, 1, 0, None))
    mask = bits.clone()
    check(lib.cbinfer_conv_changed_blocks(ptr(st), ptr(bits), ptr(arrive), ptr(copy), ptr(wq), ptr(db), ptr(out),
                                          C, H, W, K, kH, kW, 1, None))
    torch.cuda.synchronize()
    assert torch.equal(copy, mask) and int(bits.abs().sum()) == 0 and int(arrive.abs().sum()) == 0
    got = out.cpu().numpy().reshape(K, -1)
    err = np.abs(got[:, idx_o] - dense.reshape(K, -1)[:, idx_o]).max()
    assert err <= FP32_TOL, err
    rest = np.setdiff1d(np.arange(H * W), idx_o)
    assert np.all(got[:, rest] == 77.0)
    before = out.clone()
    check(lib.cbinfer_conv_changed_blocks(ptr(st), ptr(bits), ptr(arrive), ptr(copy), ptr(wq), ptr(db), ptr(out),
                                          C, H, W, K, kH, kW, 1, None))
    torch.cuda.synchronize()
    assert torch.equal(out, before) and int(copy.abs().sum()) == 0


@pytest.mark.parametrize("blocks,C,K,kH,kW,H,W,frac", [
    (0, 3, 16, 7, 7, 64, 96, 0.05), (0, 5, 12, 3, 5, 33, 70, 0.1), (1, 16, 64, 7, 7, 40, 60, 0.05),
    (1, 33, 70, 5, 3, 21, 64, 0.05), (1, 8, 16, 3, 3, 9, 130, 0.2)])
def test_fg_frame_on_masked_contractions(cb, oracle, blocks, C, K, kH, kW, H, W, frac):
    """Fine-grained frame on the mask-driven contractions (cbinfer_cbconv2d_forward_fg_masked: per-value
    detection into the delta tensor + single mask, then out += conv(W, delta) on the row-segment / patch-staged
    kernel with the accumulating epilogue): delta and state exact against the oracle's per-value detection, the
    output within the fp32 bar of out0 + conv(W, delta) accumulated in double, its relu'd copy consistent,
    untouched pixels bit-identical, mask copy = the dilated touched-pixel mask, mask and counters cleared."""
    from cbinfer_amd._lib import C as lib, check, ptr
    assert (lib.cbinfer_blockconv_supported if blocks else lib.cbinfer_rowconv_supported)(C, K, kH, kW)
    rng = np.random.default_rng(C * 17 + K + blocks)
    x, st_np = rand_case(rng, C, H, W, frac, th=0.1, blocks=True)
    w = (rng.standard_normal((K, C, kH, kW)) / np.sqrt(C * kH * kW)).astype(np.float32)
    d = x - st_np
    delta_o = np.where(np.abs(d) > 0.1, d, 0).astype(np.float32)
    out0 = rng.standard_normal((1, K, H, W)).astype(np.float32)
    want = out0.astype(np.float64) + oracle.conv2d_dense(delta_o, w, np.zeros(K, np.float32), relu=False)
    words = lib.cbinfer_mask_words(H, W)
    bits = torch.zeros(words, dtype=torch.int64, device="cuda")
    arrive = torch.zeros(words, dtype=torch.int32, device="cuda")
    copy = torch.full((words,), -1, dtype=torch.int64, device="cuda")
    if blocks:
        wq = torch.empty(lib.cbinfer_blockconv_prepared_bytes(C, K, kH, kW), dtype=torch.uint8, device="cuda")
        check(lib.cbinfer_blockconv_prep_weights(ptr(dev(w)), ptr(wq), K, C, kH, kW, None))
    else:
        wq = torch.empty(lib.cbinfer_rowconv_prepared_bytes(C, K, kH, kW), dtype=torch.uint8, device="cuda")
        check(lib.cbinfer_rowconv_prep_weights(ptr(dev(w)), ptr(wq), K, C, kH, kW, None))
    dx, st = dev(x), dev(st_np)
    delta = torch.full_like(dx, 123.0)
    out, relu = dev(out0), torch.full((1, K, H, W), -5.0, device="cuda")
    check(lib.cbinfer_cbconv2d_forward_fg_masked(blocks, ptr(dx), ptr(st), ptr(delta), ptr(out), ptr(relu), ptr(bits),
                                                 ptr(arrive), ptr(copy), ptr(wq), C, H, W, K, kH, kW, 0.1, 1, None))
    torch.cuda.synchronize()
    assert np.array_equal(delta.cpu().numpy(), delta_o)
    assert np.array_equal(st.cpu().numpy(), x)                       # prev <- in wherever they differ
    assert int(bits.abs().sum()) == 0 and int(arrive.abs().sum()) == 0
    # touched pixels: any-channel |d| > th, dilated by the filter support
    touched = oracle.changeDetection(x, st_np.copy(), (kH, kW), 0.1).reshape(-1).astype(bool)
    got = out.cpu().numpy()
    scale = max(1.0, float(np.abs(want).max()))
    assert np.abs(got - want).max() <= FP32_TOL * scale
    g2 = got.reshape(K, -1)
    assert np.array_equal(g2[:, ~touched], out0.reshape(K, -1)[:, ~touched])
    r2 = relu.cpu().numpy().reshape(K, -1)
    assert np.array_equal(r2[:, touched], np.maximum(g2[:, touched], 0)) and np.all(r2[:, ~touched] == -5.0)
    # the mask copy holds exactly the touched pixels
    idx = torch.empty(H * W, dtype=torch.int32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    check(lib.cbinfer_compact_bits(ptr(copy), W, H, ptr(idx), ptr(cnt), None, None, None))
    n = int(cnt.item())
    assert np.array_equal(idx[:n].cpu().numpy(), np.flatnonzero(touched).astype(np.int32))


# ------------------------------------------------------------------------------------------------
# fp16 (cg_half) against fixtures derived from the REFERENCE (its python twins on CPU half tensors,
# tests/golden/gen_golden.py::gen_half; reference kernels: cbconv2d_cg_half_backend.cu:10-237)
# ------------------------------------------------------------------------------------------------
HALF_ULPS = 2.0     # bar; the fixtures need <= 1 (tests/test_oracle_golden.py explains the magnitude)


def half_tol(ref, bias=None):
    m = float(np.abs(ref.astype(np.float64)).max())
    if bias is not None:
        m += float(np.abs(bias.astype(np.float64)).max())
    return HALF_ULPS * 2.0 ** (np.floor(np.log2(max(m, 2.0 ** -14))) - 10)


@pytest.mark.parametrize("case", sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden",
                                                               "ops_half_case*.npz"))))
def test_golden_ops_half(cb, case):
    """The HIP cg_half ops on the reference-derived fp16 fixtures: mask, index list, X and scatter bit-exact
    (incl. differences a few half ulps either side of the threshold), contraction within HALF_ULPS."""
    _, cg, _ = cb
    d = dict(np.load(case))
    filt = tuple(int(v) for v in d["filtSize"])
    th = float(d["threshold"])
    inp, prev = dev(d["input"]), dev(d["prevInput"])
    assert inp.dtype == torch.float16
    cm = cg.changeDetection(inp, prev.clone(), filt, th, useHalf=True)
    assert np.array_equal(cm.cpu().numpy(), d["changeMap"])
    idx = cg.changeIndexesExtr(cm)
    assert np.array_equal(idx.cpu().numpy(), d["changeIndexes"])
    X = cg.genXMatrix(inp, idx, filt, useHalf=True)
    assert np.array_equal(X.cpu().numpy(), d["X"])
    w, b = dev(d["weight"]), dev(d["bias"])
    tol = half_tol(d["Y"], d["bias"])
    Y = cg.matrixMult(X, w, b)
    assert Y.dtype == torch.float16
    err = np.abs(Y.cpu().numpy().astype(np.float64) - d["Y"].astype(np.float64)).max()
    assert err <= tol, (err, tol)
    out = cg.updateOutput(dev(d["Y"]).t(), idx, dev(d["prevOutput"]), withReLU=False, useHalf=True)
    assert np.array_equal(out.cpu().numpy(), d["out_plain"])
    out = cg.updateOutput(dev(d["Y"]).t(), idx, dev(d["prevOutput"]), withReLU=True, useHalf=True)
    assert np.array_equal(out.cpu().numpy(), d["out_relu"])
    for relu, key in ((False, "out_plain"), (True, "out_relu")):
        fused = cg.convChanged(inp, idx, w, b, dev(d["prevOutput"]), withReLU=relu)
        err = np.abs(fused.cpu().numpy().astype(np.float64) - d[key].astype(np.float64)).max()
        assert err <= tol, (key, err, tol)
    # feedback refresh at the pre-dilation changed pixels only (cg_half.cu:68-76), against the fixture's inputs
    st = prev.clone()
    cg.changeDetection(inp, st, filt, th, updateInputState=True, useHalf=True)
    cm1 = cg.changeDetection(inp, prev.clone(), (1, 1), th, useHalf=True).bool()
    sel = cm1[None, None].expand_as(st)
    assert torch.equal(st[sel], inp[sel]) and torch.equal(st[~sel], prev[~sel])
