"""Pin the CPU oracle (oracle/cb_oracle.{c,py}) against the reference's own known-answer vectors and
the golden fixtures emitted by the reference's pure-torch ops (tests/golden/gen_golden.py).

The python twin of the change predicate uses >= where the CUDA kernels use > (SURVEY 8c trap 1); the
fixtures contain no exact ties, so both comparison modes must reproduce them bit-exactly.
"""
import glob
import os

import numpy as np
import pytest


def _load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name), allow_pickle=False))


def test_kat_genTestData(oracle, golden_dir):
    # reference: conv2d_cg.py:84-97 (genTestData) -> 15 dilated indices, independent of the seed
    k = _load(golden_dir, "kat_genTestData.npz")
    expected = [3, 4, 5, 303, 304, 305, 2703, 2704, 2705, 3003, 3004, 3005, 3303, 3304, 3305]
    assert k["changeIndexes"].tolist() == expected
    rng = np.random.default_rng(0)
    inp = rng.standard_normal(tuple(k["shape"])).astype(np.float32)
    prev = inp.copy()
    for c, y, x, d in k["points"]:
        prev[0, int(c), int(y), int(x)] += np.float32(d)
    for cmp in (oracle.CMP_GT, oracle.CMP_GE):
        cm = oracle.changeDetection(inp, prev.copy(), (3, 3), 0.1, cmp=cmp)
        assert oracle.changeIndexesExtr(cm).tolist() == expected


def test_kat_changeIndexesExtr(oracle, golden_dir):
    # reference: conv2d_cg.py:215-236
    k = _load(golden_dir, "kat_changeIndexesExtr.npz")
    assert k["changeIndexes"].tolist() == [259, 765, 1277, 1779, 1783, 6127]
    cm = np.zeros(tuple(k["shape"]), np.int8)
    for y, x in k["points"]:
        cm[y, x] = 1
    assert oracle.changeIndexesExtr(cm).tolist() == k["changeIndexes"].tolist()


@pytest.mark.parametrize("case", sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden",
                                                               "ops_case*.npz"))))
def test_ops_against_reference_python(oracle, case):
    d = dict(np.load(case))
    filt = tuple(int(v) for v in d["filtSize"])
    th = float(d["threshold"])
    for cmp in (oracle.CMP_GT, oracle.CMP_GE):
        cm = oracle.changeDetection(d["input"], d["prevInput"].copy(), filt, th, cmp=cmp)
        assert np.array_equal(cm, d["changeMap"])
        cm1 = oracle.changeDetection(d["input"], d["prevInput"].copy(), (1, 1), th, cmp=cmp)
        assert np.array_equal(cm1, d["changeMap1x1"])
    if "propagated" in d:
        assert np.array_equal(oracle.changePropagation(d["changeMap1x1"], filt), d["propagated"])
    # gather-form dilation of the 1x1 map == fused detection (changePropagation_test1, :179-195)
    assert np.array_equal(oracle.changePropagation(d["changeMap1x1"], filt), d["changeMap"])
    idx = oracle.changeIndexesExtr(d["changeMap"])
    assert np.array_equal(idx, d["changeIndexes"])
    X = oracle.genXMatrix(d["input"], idx, filt)
    assert np.array_equal(X, d["X"])                       # data movement: exact
    for accMode in (0, 1):
        Y = oracle.matrixMult(X, d["weight"], d["bias"], accMode=accMode)
        np.testing.assert_allclose(Y, d["Y"], rtol=0, atol=1e-4)   # north-star fp32 tolerance
    Yt = np.ascontiguousarray(d["Y"].T)
    assert np.array_equal(oracle.updateOutput(Yt, idx, d["prevOutput"].copy(), False), d["out_plain"])
    assert np.array_equal(oracle.updateOutput(Yt, idx, d["prevOutput"].copy(), True), d["out_relu"])


def test_feedback_update_only_touches_changed_pixels(oracle):
    # cbconv2d_cg_backend.cu:74-80: state is refreshed at PRE-dilation changed pixels only
    rng = np.random.default_rng(3)
    inp = rng.standard_normal((1, 4, 10, 12)).astype(np.float32)
    state = inp + rng.uniform(-0.01, 0.01, inp.shape).astype(np.float32)   # sub-threshold drift
    state[0, 2, 4, 5] += 1.0
    before = state.copy()
    cm = oracle.changeDetection(inp, state, (3, 3), 0.1, updateInputState=True)
    assert cm.sum() == 9
    changed = np.zeros((10, 12), bool)
    changed[4, 5] = True
    assert np.array_equal(state[0][:, changed], inp[0][:, changed])
    assert np.array_equal(state[0][:, ~changed], before[0][:, ~changed])


def test_strict_vs_inclusive_threshold(oracle):
    # SURVEY 8c trap 1: |d| == th is "changed" for the python twin (>=) but not for CUDA (>)
    inp = np.zeros((1, 1, 4, 4), np.float32)
    state = inp.copy()
    state[0, 0, 1, 1] = 0.5
    assert oracle.changeDetection(inp, state.copy(), (1, 1), 0.5, cmp=oracle.CMP_GT).sum() == 0
    assert oracle.changeDetection(inp, state.copy(), (1, 1), 0.5, cmp=oracle.CMP_GE).sum() == 1
    # +inf initial state => everything changed; NaN never compares true (trap 2)
    st = np.full_like(inp, np.inf)
    assert oracle.changeDetection(inp, st, (1, 1), 0.1).all()
    nan_in = np.full_like(inp, np.nan)
    assert oracle.changeDetection(nan_in, state.copy(), (1, 1), 0.1).sum() == 0


@pytest.mark.parametrize("name", ["seq_default", "seq_prop1x1", "seq_nocopy", "seq_k3"])
def test_module_sequences_against_reference(oracle, golden_dir, name):
    """Whole converted scene-labeling-shaped net, 4 frames, as run by the reference on CPU."""
    d = _load(golden_dir, name + ".npz")
    k = int(d["k"])
    th = float(d["threshold"])
    W = lambda i: (d["param_%d.weight" % i], d["param_%d.bias" % i])
    prop = name == "seq_prop1x1"
    if prop:
        assert bool(d["ref_propChangeIndexesOf1x1_is_noop"])   # reference quirk, __init__.py:73
    convs = [
        oracle.OracleCBConv2d(*W(0), th, withReLU=True, cmp=oracle.CMP_GE),
        oracle.OracleCBConv2d(*W(3), th, withReLU=True, cmp=oracle.CMP_GE),
        oracle.OracleCBConv2d(*W(6), th, withReLU=True, cmp=oracle.CMP_GE, propChangeIndexes=prop),
        oracle.OracleCBConv2d(*W(8), th, withReLU=True, cmp=oracle.CMP_GE, propChangeIndexes=prop),
        oracle.OracleCBConv2d(*W(10), th, withReLU=False, cmp=oracle.CMP_GE),
    ]
    for m in convs:
        m.copyInput = name != "seq_nocopy"
    net = oracle.OracleSequential([convs[0], oracle.OracleMaxPool2d(), convs[1],
                                   oracle.OracleMaxPool2d(), convs[2], convs[3], convs[4]])
    assert d["childNames"].tolist() == ['0', '2', '3', '5', '6', '8', '10']
    for t in range(4):
        y = net.forward(d["frame%d" % t])
        for li, m in enumerate(convs):
            key = "cm%d_l%d" % (t, li)
            if key in d:
                assert np.array_equal(m.changeMap, d[key]), (t, li)
            np.testing.assert_allclose(m.prevOutput, d["prevOutput%d_l%d" % (t, li)], rtol=0,
                                       atol=1e-4)
        np.testing.assert_allclose(y, d["out%d" % t], rtol=0, atol=1e-4)
    assert k in (3, 7)


def test_fg_against_compiled_reference(oracle, golden_dir):
    # cbconvFG_test1 (conv2d_fg.py:98-150): error < 1e-6 vs dense; and a random th>0 case, both
    # produced by the reference's conv2d_fg_cpu compiled from its own source (oracle/_ref).
    for name in ("fg_test1.npz", "fg_case1.npz"):
        d = _load(golden_dir, name)
        th = float(d["threshold"])
        got = oracle.conv2d_fg_cpu(d["input"], d["prevInput"], d["prevOutput"].copy(), d["weight"], th)
        np.testing.assert_allclose(got, d["output"], rtol=0, atol=1e-5)
        if "outputRef" in d:
            assert np.abs(got - d["outputRef"]).max() < 1e-6
        # GPU-form restatement (detect -> nonzero -> scatter-add) agrees except for exact ties
        diffs, cm = oracle.changeDetectionFG(d["input"], d["prevInput"], th, cmp=oracle.CMP_GE)
        coords = np.nonzero(cm.reshape(-1))[0]
        got2 = oracle.updateOutputFG(diffs, d["weight"], d["prevOutput"].copy(), coords)
        np.testing.assert_allclose(got2, d["output"], rtol=0, atol=1e-4)


def test_pool_matches_dense_when_indexes_cover_changes(oracle):
    # ground truth for change-based pooling (no reference test exists): wherever the index
    # list covers every changed pixel, outputState == max_pool2d(full tensor)
    rng = np.random.default_rng(5)
    for (H, W, ceil) in [(12, 16, False), (11, 15, False), (11, 15, True)]:
        x0 = rng.standard_normal((1, 3, H, W)).astype(np.float32)
        pool = oracle.OracleCBPoolMax2d(ceil_mode=ceil)
        allidx = np.arange(H * W, dtype=np.int32)
        y0 = pool.forward(('changeIndexes', x0, allidx))
        assert np.array_equal(y0, oracle.maxpool_dense(x0, ceil))
        x1 = x0.copy()
        chg = rng.choice(H * W, 17, replace=False).astype(np.int32)
        chg.sort()
        x1.reshape(3, -1)[:, chg] += 2.0
        y1 = pool.forward(('changeIndexes', x1, chg))
        assert np.array_equal(y1, oracle.maxpool_dense(x1, ceil))


def test_half_predicate(oracle):
    # cbconv2d_cg_half_backend.cu:24-29: compare in half precision after one rounding of the diff
    inp = np.zeros((1, 2, 4, 6), np.float16)
    st = inp.copy()
    st[0, 1, 2, 3] = np.float16(0.1001)       # rounds to 0.10009765625 > th16(0.1)=0.0999755859375
    st[0, 0, 1, 1] = np.float16(0.0999)       # == th16 after rounding -> not strictly greater
    cm = oracle.changeDetection_half(inp, st.copy(), (1, 1), 0.1)
    assert cm[2, 3] == 1 and cm[1, 1] == 0 and cm.sum() == 1


# ------------------------------------------------------------------------------------------------------
# fp16 (cg_half): fixtures from the reference's python twins / CBConv2d.forward_normal on CPU HALF tensors
# (tests/golden/gen_golden.py::gen_half).  They pin the oracle's restatement of cbconv2d_cg_half_backend.cu
# :10-237 (oracle.changeDetection_half, genXMatrix_half, matrixMult_half, updateOutput_half,
# OracleCBConv2dHalf): masks, lists and moved data bit-exact; the contraction within HALF_ULPS fp16 ulps of
# the layer's largest |output| + |bias|: torch's CPU half matmul rounds the product sum (a value of up to that
# magnitude) and then the biased sum to half, the oracle -- like the HIP kernel -- rounds once, so where bias and
# product sum cancel the reference's own result is off by an ulp of the LARGER operand.  The fixtures need <= 1.
# ------------------------------------------------------------------------------------------------------
HALF_ULPS = 2.0


def half_tol(ref, bias=None):
    """HALF_ULPS fp16 ulps at the magnitude max|ref| + max|bias|."""
    m = float(np.abs(ref.astype(np.float64)).max())
    if bias is not None:
        m += float(np.abs(bias.astype(np.float64)).max())
    return HALF_ULPS * 2.0 ** (np.floor(np.log2(max(m, 2.0 ** -14))) - 10)


@pytest.mark.parametrize("case", sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden",
                                                               "ops_half_case*.npz"))))
def test_half_ops_against_reference_python(oracle, case):
    d = dict(np.load(case))
    assert d["input"].dtype == np.float16 and d["Y"].dtype == np.float16
    filt = tuple(int(v) for v in d["filtSize"])
    th = float(d["threshold"])
    for cmp in (oracle.CMP_GT, oracle.CMP_GE):      # no exact ties in the fixtures: both modes agree
        cm = oracle.changeDetection_half(d["input"], d["prevInput"].copy(), filt, th, cmp=cmp)
        assert np.array_equal(cm, d["changeMap"])
    idx = oracle.changeIndexesExtr(cm)
    assert np.array_equal(idx, d["changeIndexes"])
    X = oracle.genXMatrix_half(d["input"], idx, filt)
    assert np.array_equal(X, d["X"])
    Y = oracle.matrixMult_half(X, d["weight"], d["bias"])
    err = np.abs(Y.astype(np.float64) - d["Y"].astype(np.float64)).max()
    assert err <= half_tol(d["Y"], d["bias"]), (err, half_tol(d["Y"], d["bias"]))
    Yt = np.ascontiguousarray(d["Y"].T)
    assert np.array_equal(oracle.updateOutput_half(Yt, idx, d["prevOutput"].copy(), False), d["out_plain"])
    assert np.array_equal(oracle.updateOutput_half(Yt, idx, d["prevOutput"].copy(), True), d["out_relu"])
    # the feedback refresh only touches the pre-dilation changed pixels (cg_half.cu:68-76)
    st = d["prevInput"].copy()
    cm1 = oracle.changeDetection_half(d["input"], d["prevInput"].copy(), (1, 1), th)
    oracle.changeDetection_half(d["input"], st, filt, th, updateInputState=True)
    sel = np.broadcast_to(cm1.astype(bool)[None, None], st.shape)
    assert np.array_equal(st[sel], d["input"][sel]) and np.array_equal(st[~sel], d["prevInput"][~sel])


@pytest.mark.parametrize("name", ["seq_half", "seq_half_k3"])
def test_half_module_sequence_against_reference(oracle, golden_dir, name):
    """OracleCBConv2dHalf, layer by layer on the inputs the reference's own layers saw (prevInput after a
    copyInput frame IS the layer's input), 4 frames: change maps bit-exact, states within HALF_ULPS."""
    d = _load(golden_dir, name + ".npz")
    k = int(d["k"])
    convs = [0, 3, 6, 8, 10]
    layers = []
    for li, ci in enumerate(convs):
        for cmp in (oracle.CMP_GT, oracle.CMP_GE):
            layers.append((li, cmp, oracle.OracleCBConv2dHalf(d["param_%d.weight" % ci], d["param_%d.bias" % ci],
                                                             float(d["threshold"]), withReLU=li < 4, cmp=cmp)))
    worst = 0.0
    for t in range(4):
        for li, cmp, o in layers:
            x = d["prevInput%d_l%d" % (t, li)]
            assert x.dtype == np.float16
            out = o.forward(x)
            assert np.array_equal(o.changeMap, d["cm%d_l%d" % (t, li)]), (t, li, cmp)
            ref = d["prevOutput%d_l%d" % (t, li)]
            err = np.abs(out.astype(np.float64) - ref.astype(np.float64)).max()
            tol = half_tol(ref, o.bias)
            worst = max(worst, err / (tol / HALF_ULPS))
            assert err <= tol, (t, li, err, tol)
    assert k in (3, 7) and worst <= 1.0, worst      # what the fixtures actually need: one ulp at that magnitude
