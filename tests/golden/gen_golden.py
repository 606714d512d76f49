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

Usage:  python tests/golden/gen_golden.py
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


def main():
    ref = import_reference()
    from pycbinfer import conv2d_cg as rcg
    out = {}

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


if __name__ == "__main__":
    main()
