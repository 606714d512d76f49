#!/usr/bin/env python3
"""Diagnostic: per-workgroup phase time stamps of the fp32 contraction kernel (layer-3 shape).
Needs the stamp build of the library:  make -C cbinfer_amd/csrc EXTRA=-DCB_STAMP  (then rebuild without
it before testing/benchmarking).  Prints, over the workgroups, min/mean/max of the time each phase ended
and the shader clock measured inside the kernel."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbinfer_amd import conv2d_cg as cg, _lib
from tools.bench_conv import LAYERS, blocks_list
raw = ctypes.CDLL(_lib.LIB_PATH)
C, K, k, H, W = LAYERS[int(sys.argv[1]) if len(sys.argv) > 1 else 2]
gen = torch.Generator().manual_seed(0)
x = torch.randn(1, C, H, W, device="cuda"); w = torch.randn(K, C, k, k, device="cuda") / (C*k*k)**0.5
ARITH = _lib.CB_F32S if os.environ.get('STAMP_SPLIT', '1') != '0' else None
b = torch.randn(K, device="cuda"); out = torch.zeros(1, K, H, W, device="cuda"); wp = cg.prepWeights(w, H, W, arith=ARITH)
for ratio in ([float(a) for a in sys.argv[2:]] or [0.1, 0.36, 1.0]):
    idx = blocks_list(H, W, ratio, 8, gen)
    for _ in range(5):
        cg.convChanged(x, idx, w, b, out, withReLU=True, weightsPrepared=wp, arith=ARITH)
    torch.cuda.synchronize()
    buf = np.zeros(1024 * 8, dtype=np.uint64)
    raw.cbinfer_debug_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(buf.nbytes))
    st = buf.reshape(1024, 8)[:512].astype(np.int64)
    st = st[st[:, 0] > 0]          # workgroups of this launch's grid
    t0 = st[:, 0].min()
    us = (st - t0) / 100.0
    names = ['entry', 'prologue done', 'item setup done', 'stage loop done', 'KS reduce done', 'ticket done', 'epilogue done', 'exit']
    print("ratio %.2f N=%d" % (ratio, idx.numel()))
    for i, n in enumerate(names):
        col = us[:, i]; valid = st[:, i] >= t0
        if valid.sum() == 0: continue
        print("  %-18s min %7.2f  mean %7.2f  max %7.2f us (%d wgs)" % (n, col[valid].min(), col[valid].mean(), col[valid].max(), valid.sum()))
    ok2 = (st[:, 2] >= t0) & (st[:, 3] >= st[:, 2])
    if ok2.sum():
        d = (st[ok2, 3] - st[ok2, 2]) / 100.0
        print("  stage loop of the first item: min %.2f mean %.2f max %.2f us over %d wgs" % (d.min(), d.mean(), d.max(), ok2.sum()))
    sc = np.zeros(2 * 24 * 4, dtype=np.uint64)
    if hasattr(raw, 'cbinfer_debug_stage_clocks') and os.environ.get('CBINFER_X3_WIDE', '1') != '0':
        raw.cbinfer_debug_stage_clocks(sc.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(sc.nbytes))
        sc = sc.reshape(2, 24, 4).astype(np.int64)
        for wv, nm in ((0, 'wave 0 (store, load, MFMA)'), (1, 'wave 8 (MFMA, store, load)')):
            rows = [r for r in sc[wv] if r[0] > 0]
            if len(rows) > 3:
                r = np.array(rows[1:-1])
                seg = np.diff(r, axis=1).mean(0)
                per = np.diff(r[:, 0]).mean() if len(r) > 1 else 0
                print("  %s: segments %s cycles; stage period %.0f cycles (%d stages)" % (nm, [int(x) for x in seg], per, len(r)))
    clk = np.zeros(1024 * 2, dtype=np.uint64)
    raw.cbinfer_debug_clocks(clk.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(clk.nbytes))
    ck = clk.reshape(1024, 2)[:len(st)].astype(np.int64)
    dcyc = ck[:, 1] - ck[:, 0]; dus = (st[:, 7] - st[:, 0]) / 100.0
    ok = dus > 5
    print("  shader clock: s_memtime ticks per us  min %.0f mean %.0f max %.0f" % ((dcyc[ok] / dus[ok]).min(), (dcyc[ok] / dus[ok]).mean(), (dcyc[ok] / dus[ok]).max()))
