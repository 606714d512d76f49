#!/usr/bin/env python3
"""Per-phase timing of the row-segment kernel's workgroups from in-kernel time stamps (library built with
make EXTRA=-DCB_ROW_STAMP).  Phases: 0 entry, 1 mask word known, 2 row table, 3 patch staged, 4 k-loop done,
5 k-parts reduced, 6 outputs stored."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbinfer_amd import conv2d_cg as cg  # noqa: E402
from cbinfer_amd._lib import C as lib, check, ptr, LIB_PATH  # noqa: E402

raw = ctypes.CDLL(LIB_PATH)


def main():
    gen = torch.Generator().manual_seed(0)
    for (C, K, k, H, W, blk, ratio) in [(3, 16, 7, 320, 480, 32, 0.10), (16, 64, 7, 160, 240, 16, 0.10)]:
        x = torch.randn(1, C, H, W, device="cuda")
        w = torch.randn(K, C, k, k, device="cuda") / (C * k * k) ** 0.5
        b = torch.randn(K, device="cuda")
        out = torch.zeros(1, K, H, W, device="cuda")
        cm = torch.zeros(H, W, dtype=torch.int8)
        gy, gx = H // blk, W // blk
        cells = torch.randperm(gy * gx, generator=gen)[:max(1, int(round(ratio * gy * gx)))]
        for c in cells.tolist():
            y0, x0 = (c // gx) * blk, (c % gx) * blk
            cm[max(0, y0 - 3):y0 + blk + 3, max(0, x0 - 3):x0 + blk + 3] = 1
        cm = cm.cuda()
        words = lib.cbinfer_mask_words(H, W)
        wpr = lib.cbinfer_mask_words_per_row(W)
        pad = torch.zeros(H, wpr * 64, dtype=torch.int64, device="cuda")
        pad[:, :W] = cm.long()
        mask = (pad.view(H, wpr, 64) << torch.arange(64, device="cuda")).sum(-1).view(-1).contiguous()
        bits = torch.zeros(words, dtype=torch.int64, device="cuda")
        arrive = torch.zeros(words, dtype=torch.int32, device="cuda")
        copy = torch.zeros(words, dtype=torch.int64, device="cuda")
        wq = torch.empty(lib.cbinfer_rowconv_prepared_bytes(C, K, k, k), dtype=torch.uint8, device="cuda")
        check(lib.cbinfer_rowconv_prep_weights(ptr(w), ptr(wq), K, C, k, k, None))

        def go():
            bits.copy_(mask)
            check(lib.cbinfer_conv_changed_rows(ptr(x), ptr(bits), ptr(arrive), ptr(copy), ptr(wq), ptr(b), ptr(out),
                                                C, H, W, K, k, k, 1, None))
        for _ in range(5):
            go()
        torch.cuda.synchronize()
        raw.cbinfer_debug_row_stamps(None, 0, 1)
        torch.cuda.synchronize()
        go()
        torch.cuda.synchronize()
        buf = np.zeros(8192 * 8, dtype=np.uint64)
        raw.cbinfer_debug_row_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(buf.nbytes), 0)
        st = buf.reshape(8192, 8).astype(np.int64)
        ran = st[:, 0] > 0
        act = ran & (st[:, 6] > 0)
        t0 = st[ran, 0].min()
        print("%d->%d @%dx%d: %d workgroups started, %d active; first start -> last start %.2f us, kernel span %.2f us"
              % (C, K, H, W, ran.sum(), act.sum(), (st[ran, 0].max() - t0) / 100.0,
                 (st[act, 6].max() - t0) / 100.0))
        a = st[act]
        names = ["mask word", "row table", "staging", "k-loop", "reduce", "store"]
        for i, nm in enumerate(names):
            d = (a[:, i + 1] - a[:, i]) / 100.0
            print("   %-10s median %.2f us   p90 %.2f   max %.2f" % (nm, np.median(d), np.percentile(d, 90), d.max()))
        life = (a[:, 6] - a[:, 0]) / 100.0
        print("   lifetime   median %.2f us   p90 %.2f   max %.2f ; active start spread %.2f us" % (
            np.median(life), np.percentile(life, 90), life.max(), (a[:, 0].max() - a[:, 0].min()) / 100.0))
        e = st[ran & ~act]
        if len(e):
            d = (e[:, 1][e[:, 1] > 0] - e[:, 0][e[:, 1] > 0]) / 100.0
            print("   (half-exits with stamp 1: %d, mask wait median %.2f)" % (len(d), np.median(d) if len(d) else 0))


def occupancy():
    raw.cbinfer_debug_row_occupancy.restype = ctypes.c_int
    for lds in (0, 12032, 16384, 32768):
        print("resident workgroups per CU (256 threads, %5d B dynamic LDS): %d"
              % (lds, raw.cbinfer_debug_row_occupancy(256, ctypes.c_long(lds))))


def batch(S):
    """the row-segment launch of a SequenceBatch step of S sequences of the bench workload"""
    import bench
    import pycbinfer
    _, net = bench.build_bench_model()
    sb = pycbinfer.SequenceBatch(net, S)
    vids = [bench.bench_video(1234 + 7919 * q) for q in range(S)]
    walk = [[v.frame] + [v.next() for _ in range(11)] for v in vids]
    with torch.no_grad():
        for i in range(11):
            sb([w[i] for w in walk])
        torch.cuda.synchronize()
        raw.cbinfer_debug_row_stamps(None, 0, 1)
        torch.cuda.synchronize()
        sb([w[11] for w in walk])
        torch.cuda.synchronize()
    buf = np.zeros(8192 * 8, dtype=np.uint64)
    raw.cbinfer_debug_row_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(buf.nbytes), 0)
    st = buf.reshape(8192, 8).astype(np.int64)
    ran = st[:, 0] > 0
    act = ran & (st[:, 6] > 0)
    t0 = st[ran, 0].min()
    print("S=%d: %d workgroups stamped (of %d), %d active; first start -> last start %.2f us, span to the last store %.2f us"
          % (S, ran.sum(), 2560 * S, act.sum(), (st[ran, 0].max() - t0) / 100.0, (st[act, 6].max() - t0) / 100.0))
    a = st[act]
    for i, nm in enumerate(["mask word", "row table", "staging", "k-loop", "reduce", "store"]):
        d = (a[:, i + 1] - a[:, i]) / 100.0
        print("   %-10s median %.2f us   p90 %.2f   max %.2f" % (nm, np.median(d), np.percentile(d, 90), d.max()))
    life = (a[:, 6] - a[:, 0]) / 100.0
    print("   lifetime   median %.2f us   p90 %.2f   max %.2f" % (np.median(life), np.percentile(life, 90), life.max()))
    # start times by workgroup index: how fast does the dispatcher walk the grid?
    idx = np.nonzero(ran)[0]
    for lo in range(0, 8192, 1024):
        sel = ran[lo:lo + 1024]
        if sel.any():
            ts = (st[lo:lo + 1024][sel, 0] - t0) / 100.0
            print("   workgroups %4d..%4d start at %.2f .. %.2f us" % (lo, lo + 1023, ts.min(), ts.max()))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "batch":
        occupancy()
        batch(int(sys.argv[2]))
    else:
        main()
