#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ FROM THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference); the outputs (*.npz: inputs and expected
outputs, no reference source) are committed and travel to the GPU box.

How the reference is run here -- three harness-side shims, no reference file touched:
  1. a stub `cffi` module whose FFI().dlopen() returns a dummy (the CUDA .so files do not exist);
  2. torch.nn.functional.Variable = identity (torch-0.3 API used by the reference);
  3. Tensor.fill_ maps the fp32-overflowing literals (1e1000, -1e100) to +-inf as torch 0.3 did.
The reference's CPU code path (`*_python` ops, CBConv2d.forward_normal on CPU tensors,
pycbinfer.convert) is then executed unmodified.  The compiled reference (oracle/_ref, built by
oracle/Makefile from the reference .cu files) supplies conv2d_fg_cpu for the fine-grained fixtures.

Usage:  python tests/golden/gen_golden.py [--only-half]
"""
import ctypes
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def import_reference():
    class _Dummy:
        def __getattr__(self, name):
            raise RuntimeError("CUDA backend not available in the golden generator")

    class _FFI:
        def cdef(self, *_a, **_k):
            pass

        def dlopen(self, *_a, **_k):
            return _Dummy()

        def cast(self, *_a, **_k):
            raise RuntimeError("no native calls in the golden generator")

        def dlclose(self, *_a, **_k):
            pass

    cffi = types.ModuleType("cffi")
    cffi.FFI = _FFI
    sys.modules["cffi"] = cffi
    F.Variable = lambda x, *a, **k: x
    _orig_fill = torch.Tensor.fill_

    def _fill(self, v):
        if isinstance(v, float) and abs(v) > 3.4e38 and v == v:
            v = float("inf") if v > 0 else float("-inf")
        return _orig_fill(self, v)

    torch.Tensor.fill_ = _fill
    sys.path.insert(0, REF)
    import pycbinfer  # the REFERENCE package
    assert pycbinfer.__file__.startswith(REF)
    return pycbinfer


def np32(t):
    return t.detach().cpu().numpy().astype(np.float32)


def np16(t):
    assert t.dtype == torch.float16
    return t.detach().cpu().numpy().copy()      # (a copy: the module states are updated in place by later frames)


# ------------------------------------------------------------------------------------------------------
# fp16 (cg_half) fixtures.  The reference's half backend (cbconv2d_cg_half_backend.cu) is unbuildable here
# (<cuda_fp16.h>), but its Python side is dtype-agnostic: the `*_python` twins and CBConv2d.forward_normal run
# on CPU HALF tensors of this torch build with ONE more harness-side shim -- changeDetection_python feeds a
# float change map and a ones-filter made by `input.new(...)` (half here) to F.conv2d, which today's torch
# refuses for mixed dtypes: F.conv2d is wrapped to cast the filter to the map's dtype (the filter is all
# ones and the map 0/1, so the dilation is exact in any float type).  Everything else executes unmodified:
#   predicate   (input - prevInput).abs().ge(threshold) on half tensors: ONE rounding of the difference to
#               half, threshold converted to half -- the arithmetic of __hsub / __hgt in cg_half.cu:24-29
#               (ties |d| == th16 excepted: the twin uses >=, the kernel >; the fixtures hold none -- the
#               oracle tests check both comparison modes)
#   gather      genXMatrix_python: data movement
#   contraction matrixMult_python: torch CPU half matmul (+ in-place half bias add)
#   scatter     updateOutput_python: clamp_(0, inf) on half
# ------------------------------------------------------------------------------------------------------
def gen_half(ref):
    from pycbinfer import conv2d_cg as rcg
    _conv2d = F.conv2d
    refused = []

    def conv2d_mixed(inp, w, *a, **k):
        if inp.dtype != w.dtype:
            w = w.to(inp.dtype)
        return _conv2d(inp, w, *a, **k)

    # does today's torch refuse the twin as is?  (recorded in the fixture)
    try:
        rcg.changeDetection_python(torch.zeros(1, 1, 4, 4).half(), torch.ones(1, 1, 4, 4).half(), (3, 3), 0.1)
    except Exception as e:      # noqa
        refused.append("changeDetection_python: F.conv2d(float map, half ones-filter) -> %s" % type(e).__name__)
    F.conv2d = conv2d_mixed
    rcg.F.conv2d = conv2d_mixed

    g = torch.Generator().manual_seed(4321)
    cases = []
    for ci_, (C, H, W, kH, kW, K, th, nchg) in enumerate([
            (3, 12, 17, 1, 1, 4, 0.10, 9),
            (5, 14, 19, 3, 3, 6, 0.10, 7),
            (4, 16, 21, 7, 7, 5, 0.25, 5),
            (2, 9, 64, 3, 3, 3, 0.05, 11),
            (3, 11, 65, 7, 7, 4, 0.05, 6),
            (8, 10, 13, 3, 1, 4, 0.10, 4),
    ]):
        inp = torch.randn(1, C, H, W, generator=g).half()
        prev = inp.clone()
        ys = torch.randint(0, H, (nchg,), generator=g).tolist() + [0, H - 1, 0, H - 1]
        xs = torch.randint(0, W, (nchg,), generator=g).tolist() + [0, W - 1, W - 1, 0]
        cs = torch.randint(0, C, (nchg + 4,), generator=g).tolist()
        for c, y, x in zip(cs, ys, xs):
            prev[0, c, y, x] += (1.0 if (y + x) % 2 else -3.0)
        for _ in range(5):      # sub-threshold perturbations that must NOT trigger
            c = int(torch.randint(0, C, (1,), generator=g)); y = int(torch.randint(0, H, (1,), generator=g))
            x = int(torch.randint(0, W, (1,), generator=g))
            if prev[0, c, y, x] == inp[0, c, y, x]:
                prev[0, c, y, x] += th * 0.5
        # a few differences within a few half ulps of the threshold, on both sides (never exactly on it)
        th16 = torch.tensor(th).half()
        for side in (-3, -1, 1, 3):
            c = int(torch.randint(0, C, (1,), generator=g)); y = int(torch.randint(0, H, (1,), generator=g))
            x = int(torch.randint(0, W, (1,), generator=g))
            d = torch.tensor(float(th16) * (1 + side * 2.0 ** -10)).half()
            old = prev[0, c, y, x].clone()
            prev[0, c, y, x] = (inp[0, c, y, x].float() + d.float()).half()
            if (inp[0, c, y, x] - prev[0, c, y, x]).abs() == th16:      # rounded onto the threshold: not this one
                prev[0, c, y, x] = old
        assert not ((inp - prev).abs() == th16).any(), "tie in op case %d: change the seed" % ci_
        weight = (torch.randn(K, C, kH, kW, generator=g) * 0.3).half()
        bias = torch.randn(K, generator=g).half()
        cm = rcg.changeDetection_python(inp, prev, (kH, kW), th)
        idx = rcg.changeIndexesExtr_python(cm)
        X = rcg.genXMatrix_python(inp, idx, (kH, kW))
        Y = rcg.matrixMult_python(X, weight, bias)
        assert X.dtype == torch.float16 and Y.dtype == torch.float16
        prevOut = torch.randn(1, K, H, W, generator=g).half()
        o_plain = rcg.updateOutput_python(Y.transpose(0, 1).clone(), idx, prevOut.clone(), withReLU=False)
        o_relu = rcg.updateOutput_python(Y.transpose(0, 1).clone(), idx, prevOut.clone(), withReLU=True)
        np.savez_compressed(
            os.path.join(HERE, "ops_half_case%d.npz" % ci_), input=np16(inp), prevInput=np16(prev),
            threshold=np.float32(th), filtSize=np.array([kH, kW]), weight=np16(weight), bias=np16(bias),
            changeMap=cm.numpy().astype(np.int8).reshape(H, W), changeIndexes=idx.numpy().astype(np.int32),
            X=np16(X), Y=np16(Y), prevOutput=np16(prevOut), out_plain=np16(o_plain), out_relu=np16(o_relu),
            torch_refused=np.array(refused))
        cases.append((C, H, W, kH, kW, int(idx.numel())))
    print("half op cases (C,H,W,kH,kW,N):", cases, "| refused as is:", refused)

    # module level: the reference's convert()-ed net in half on CPU, 4 frames
    import contextlib
    import io

    def make_net(seed, chans=(3, 4, 6, 8, 6, 4), k=7):
        torch.manual_seed(seed)
        c0, c1, c2, c3, c4, c5 = chans
        return nn.Sequential(
            nn.Conv2d(c0, c1, k, padding=k // 2), nn.ReLU(), nn.MaxPool2d(2, 2),
            nn.Conv2d(c1, c2, k, padding=k // 2), nn.ReLU(), nn.MaxPool2d(2, 2),
            nn.Conv2d(c2, c3, k, padding=k // 2), nn.ReLU(),
            nn.Conv2d(c3, c4, 1), nn.ReLU(),
            nn.Conv2d(c4, c5, 1)).eval()

    gg = torch.Generator().manual_seed(199)
    H, W, blk = 24, 32, 6
    f = torch.rand(1, 3, H, W, generator=gg).half()
    frames = [f.clone()]
    for t in range(1, 4):
        f = f.clone()
        for _ in range(2):
            y0 = int(torch.randint(0, H - blk + 1, (1,), generator=gg))
            x0 = int(torch.randint(0, W - blk + 1, (1,), generator=gg))
            f[:, :, y0:y0 + blk, x0:x0 + blk] = torch.rand(1, 3, blk, blk, generator=gg).half()
        frames.append(f)
    for name, k in (("seq_half", 7), ("seq_half_k3", 3)):
        base = make_net(0, k=k).half()
        with contextlib.redirect_stdout(io.StringIO()):
            cb = ref.convert(base, threshold=0.02)
        cbmods = [m for m in cb.modules() if type(m) is ref.CBConv2d]
        for m in cbmods:
            m.saveChangeMap = True
        d = dict(threshold=np.float32(0.02), childNames=np.array([n for n, _ in cb.named_children()]),
                 k=np.int64(k), torch_refused=np.array(refused))
        for n_, p_ in base.state_dict().items():
            d["param_" + n_] = np16(p_)
        ref.clearMemory(cb)
        with torch.no_grad():
            for t, fr in enumerate(frames):
                y = cb(fr.clone())
                assert y.dtype == torch.float16
                d["frame%d" % t] = np16(fr)
                d["out%d" % t] = np16(y)
                for li, m in enumerate(cbmods):
                    d["cm%d_l%d" % (t, li)] = m.changeMap.numpy().astype(np.int8).reshape(m.changeMap.shape[-2:])
                    d["prevOutput%d_l%d" % (t, li)] = np16(m.prevOutput)
                    d["prevInput%d_l%d" % (t, li)] = np16(m.prevInput)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
        print(name, "children", d["childNames"].tolist(), "changed px per layer, last frame:",
              [int(d["cm3_l%d" % li].sum()) for li in range(len(cbmods))])
    F.conv2d = _conv2d
    rcg.F.conv2d = _conv2d


def main():
    ref = import_reference()
    from pycbinfer import conv2d_cg as rcg
    out = {}
    if "--only-half" in sys.argv:      # (leaves the fp32 fixtures of round 1 untouched)
        gen_half(ref)
        return

    # ------------------------------------------------------------------ KAT 1: genTestData
    # conv2d_cg.py:84-97 stimulus (without .cuda()); 15 dilated indices, seed independent.
    torch.manual_seed(7)
    inp = torch.randn(1, 16, 400, 300)
    prev = inp.clone()
    for (c, y, x, d) in [(0, 0, 4, 1.00), (1, 6, 9, 0.05), (2, 10, 4, -11.00), (1, 6, 19, -0.05)]:
        prev[0, c, y, x] += d
    cm = rcg.changeDetection_python(inp, prev, (3, 3), 0.1)
    idx = rcg.changeIndexesExtr_python(cm)
    np.savez_compressed(os.path.join(HERE, "kat_genTestData.npz"),
                        points=np.array([(0, 0, 4, 1.00), (1, 6, 9, 0.05), (2, 10, 4, -11.00),
                                         (1, 6, 19, -0.05)], dtype=np.float64),
                        shape=np.array([1, 16, 400, 300]), threshold=0.1, filtSize=np.array([3, 3]),
                        changeIndexes=idx.numpy().astype(np.int32))
    print("KAT genTestData idx:", idx.tolist())

    # ------------------------------------------------------------------ KAT 2: changeIndexesExtr_test1
    cmk = torch.zeros(129, 254, dtype=torch.int8)
    pts = [(3, 3), (7, 5), (5, 7), (7, 1), (1, 5), (24, 31)]
    for (y, x) in pts:
        cmk[y][x] = 1
    ci = rcg.changeIndexesExtr_python(cmk)
    np.savez_compressed(os.path.join(HERE, "kat_changeIndexesExtr.npz"),
                        shape=np.array([129, 254]), points=np.array(pts),
                        changeIndexes=ci.numpy().astype(np.int32))
    print("KAT changeIndexesExtr:", ci.tolist())

    # ------------------------------------------------------------------ per-op fixtures
    # random inputs (no exact |d|==th ties: the python twin uses >=, the CUDA path > -- SURVEY 8c)
    g = torch.Generator().manual_seed(1234)
    cases = []
    for ci_, (C, H, W, kH, kW, K, th, nchg) in enumerate([
            (3, 12, 17, 1, 1, 4, 0.10, 9),
            (5, 14, 19, 3, 3, 6, 0.10, 7),
            (4, 16, 21, 7, 7, 5, 0.25, 5),
            (2, 9, 64, 3, 3, 3, 0.05, 11),     # a full 64-wide row (wave-width boundary)
            (3, 11, 65, 7, 7, 4, 0.05, 6),     # W = 65: one pixel past a wave
            (6, 10, 13, 3, 1, 4, 0.10, 4),     # non-square filter
    ]):
        inp = torch.randn(1, C, H, W, generator=g)
        prev = inp.clone()
        # point changes incl. the four corners/borders
        ys = torch.randint(0, H, (nchg,), generator=g).tolist() + [0, H - 1, 0, H - 1]
        xs = torch.randint(0, W, (nchg,), generator=g).tolist() + [0, W - 1, W - 1, 0]
        cs = torch.randint(0, C, (nchg + 4,), generator=g).tolist()
        for c, y, x in zip(cs, ys, xs):
            prev[0, c, y, x] += (1.0 if (y + x) % 2 else -3.0)
        # a few sub-threshold perturbations that must NOT trigger
        for _ in range(5):
            c = int(torch.randint(0, C, (1,), generator=g)); y = int(torch.randint(0, H, (1,), generator=g))
            x = int(torch.randint(0, W, (1,), generator=g))
            if prev[0, c, y, x] == inp[0, c, y, x]:
                prev[0, c, y, x] += th * 0.5
        weight = torch.randn(K, C, kH, kW, generator=g) * 0.3
        bias = torch.randn(K, generator=g)
        cm = rcg.changeDetection_python(inp, prev, (kH, kW), th)
        cm1 = rcg.changeDetection_python(inp, prev, (1, 1), th)
        prop = rcg.changePropagation_python(cm1.clone(), (kH, kW)) if kH == kW else None
        idx = rcg.changeIndexesExtr_python(cm)
        X = rcg.genXMatrix_python(inp, idx, (kH, kW))
        Y = rcg.matrixMult_python(X, weight, bias)
        prevOut = torch.randn(1, K, H, W, generator=g)
        o_plain = rcg.updateOutput_python(Y.transpose(0, 1).clone(), idx, prevOut.clone(), withReLU=False)
        o_relu = rcg.updateOutput_python(Y.transpose(0, 1).clone(), idx, prevOut.clone(), withReLU=True)
        d = dict(input=np32(inp), prevInput=np32(prev), threshold=np.float32(th),
                 filtSize=np.array([kH, kW]), weight=np32(weight), bias=np32(bias),
                 changeMap=cm.numpy().astype(np.int8).reshape(H, W),
                 changeMap1x1=cm1.numpy().astype(np.int8).reshape(H, W),
                 changeIndexes=idx.numpy().astype(np.int32), X=np32(X), Y=np32(Y),
                 prevOutput=np32(prevOut), out_plain=np32(o_plain), out_relu=np32(o_relu))
        if prop is not None:
            d["propagated"] = prop.numpy().astype(np.int8).reshape(H, W)
        np.savez_compressed(os.path.join(HERE, "ops_case%d.npz" % ci_), **d)
        cases.append((C, H, W, kH, kW, int(idx.numel())))
    print("op cases (C,H,W,kH,kW,N):", cases)

    # ------------------------------------------------------------------ module-level sequences
    # scene-labeling-shaped net (SURVEY 8a note 1) with small channel counts, converted by the
    # reference's pycbinfer.convert(); CPU => feedbackLoop False and plain nn.MaxPool2d (8c trap 4).
    def make_net(seed, chans=(3, 4, 6, 8, 6, 4), k=7):
        torch.manual_seed(seed)
        c0, c1, c2, c3, c4, c5 = chans
        return nn.Sequential(
            nn.Conv2d(c0, c1, k, padding=k // 2), nn.ReLU(), nn.MaxPool2d(2, 2),
            nn.Conv2d(c1, c2, k, padding=k // 2), nn.ReLU(), nn.MaxPool2d(2, 2),
            nn.Conv2d(c2, c3, k, padding=k // 2), nn.ReLU(),
            nn.Conv2d(c3, c4, 1), nn.ReLU(),
            nn.Conv2d(c4, c5, 1)).eval()

    def make_frames(seed, T, H, W, blk):
        gg = torch.Generator().manual_seed(seed)
        f = torch.rand(1, 3, H, W, generator=gg)
        frames = [f.clone()]
        for t in range(1, T):
            f = f.clone()
            for _ in range(2):
                y0 = int(torch.randint(0, H - blk + 1, (1,), generator=gg))
                x0 = int(torch.randint(0, W - blk + 1, (1,), generator=gg))
                f[:, :, y0:y0 + blk, x0:x0 + blk] = torch.rand(1, 3, blk, blk, generator=gg)
            frames.append(f)
        return frames

    import contextlib
    import io
    for name, variant in [("seq_default", {}), ("seq_prop1x1", {"prop1x1": True}),
                          ("seq_nocopy", {"copyInput": False}), ("seq_k3", {"k": 3})]:
        k = variant.get("k", 7)
        base = make_net(0, k=k)
        with contextlib.redirect_stdout(io.StringIO()):
            cb = ref.convert(base, threshold=0.02)
            if variant.get("prop1x1"):
                cb = ref.propChangeIndexesOf1x1(cb)
        cbmods = [m for m in cb.modules() if type(m) is ref.CBConv2d]
        if variant.get("prop1x1"):
            # reference quirk: propChangeIndexesOf1x1 compares the kernel_size TUPLE with the LIST
            # [1,1] (__init__.py:73), so it never enables anything; the apps set the flags by hand
            # (sceneLabeling/modelLoader.py:43-44, experiment 1).  Record the quirk, then do as the apps.
            quirk_noop = not any(m.propChangeIndexes for m in cbmods)
            cbmods[2].propChangeIndexes = True
            cbmods[3].propChangeIndexes = True
        for m in cbmods:
            m.saveChangeMap = True
            if "copyInput" in variant:
                m.copyInput = variant["copyInput"]
        names = [n for n, _ in cb.named_children()]
        frames = make_frames(99, 4, 24, 32, 6)
        d = dict(threshold=np.float32(0.02), childNames=np.array(names), k=np.int64(k))
        if variant.get("prop1x1"):
            d["ref_propChangeIndexesOf1x1_is_noop"] = np.bool_(quirk_noop)
        for i, (n_, p_) in enumerate(base.state_dict().items()):
            d["param_" + n_] = np32(p_)
        ref.clearMemory(cb)
        with torch.no_grad():
            for t, fr in enumerate(frames):
                # the reference aliases its input when copyInput=False: hand it a private copy
                y = cb(fr.clone())
                d["frame%d" % t] = np32(fr)
                d["out%d" % t] = np32(y)
                d["dense%d" % t] = np32(base(fr))
                for li, m in enumerate(cbmods):
                    if hasattr(m, "changeMap") and m.changeMap is not None and not (
                            variant.get("prop1x1") and li in (3, 4)):
                        d["cm%d_l%d" % (t, li)] = m.changeMap.numpy().astype(np.int8).reshape(
                            m.changeMap.shape[-2:])
                    d["prevOutput%d_l%d" % (t, li)] = np32(m.prevOutput)
        d["repr"] = np.array(repr(cb))
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
        print(name, "children", names, "max|cb-dense| last frame",
              float(np.abs(d["out3"] - d["dense3"]).max()))

    # ------------------------------------------------------------------ fine-grained (compiled ref)
    fg_so = os.path.join(REPO, "oracle", "_ref", "cbconv2d_fg_backend.so")
    os.environ.setdefault("OMP_NUM_THREADS", "1")   # the reference's omp-for races for ni>1
    fg = ctypes.CDLL(fg_so)

    def ref_fg_cpu(inp, prev, outp, w, th):
        a = [np.ascontiguousarray(t, dtype=np.float32) for t in (inp, prev, outp, w)]
        K, C, kh, kw = a[3].shape
        H, W = a[0].shape[-2:]
        fg.conv2d_fg_cpu(*[ctypes.c_void_p(t.ctypes.data) for t in a], ctypes.c_float(th),
                         ctypes.c_int(K), ctypes.c_int(C), ctypes.c_int(H), ctypes.c_int(W),
                         ctypes.c_int(kh), ctypes.c_int(kw))
        return a[2]

    # cbconvFG_test1 stimulus (conv2d_fg.py:98-123), random block drawn here with a seed
    torch.manual_seed(5)
    inp = torch.zeros(1, 2, 9, 9)
    prev = inp.clone()
    inp[0, 0, 1, 1] = 1.0
    inp[0, 0, 1, 2] = 3.2
    inp[0, 0, 0, 5] = 1.5
    inp[0, 1, 5:, 7:] = torch.randint(0, 100, (4, 2)).float() - 50
    prev[0, 1, 7, 5] = 4.0
    w = torch.full((3, 2, 3, 3), 3.0)
    outRef = F.conv2d(inp, w, padding=1)
    prevOut = F.conv2d(prev, w, padding=1)
    got = ref_fg_cpu(np32(inp), np32(prev), np32(prevOut).copy(), np32(w), 0.0)
    err = float(np.abs(got - np32(outRef)).max())
    assert err < 1e-6, err
    np.savez_compressed(os.path.join(HERE, "fg_test1.npz"), input=np32(inp), prevInput=np32(prev),
                        weight=np32(w), prevOutput=np32(prevOut), output=got, outputRef=np32(outRef),
                        threshold=np.float32(0.0))
    # random fine-grained case, threshold > 0 (values below threshold are dropped)
    g = torch.Generator().manual_seed(77)
    inp = torch.randn(1, 3, 10, 12, generator=g)
    prev = inp + (torch.rand(1, 3, 10, 12, generator=g) < 0.2).float() * torch.randn(1, 3, 10, 12, generator=g)
    w = torch.randn(4, 3, 3, 3, generator=g)
    prevOut = F.conv2d(prev, w, padding=1)
    got = ref_fg_cpu(np32(inp), np32(prev), np32(prevOut).copy(), np32(w), 0.3)
    np.savez_compressed(os.path.join(HERE, "fg_case1.npz"), input=np32(inp), prevInput=np32(prev),
                        weight=np32(w), prevOutput=np32(prevOut), output=got, threshold=np.float32(0.3))
    print("fg fixtures ok (cbconvFG_test1 err %.1e)" % err)
    gen_half(ref)


if __name__ == "__main__":
    main()
