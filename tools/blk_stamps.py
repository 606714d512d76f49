#!/usr/bin/env python3
"""Per-phase timing of the patch-staged kernel's workgroups from in-kernel stamps (make EXTRA=-DCB_BLK_STAMP)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbinfer_amd._lib import C as lib, check, ptr, LIB_PATH  # noqa: E402

raw = ctypes.CDLL(LIB_PATH)


def main():
    gen = torch.Generator().manual_seed(0)
    shapes = [(64, 256, 7, 80, 120, 8, 0.01), (64, 256, 7, 80, 120, 8, 0.10)]
    if len(sys.argv) > 1 and sys.argv[1] == 'l2':      # the 16->64 layer at the bench's change ratio
        shapes = [(16, 64, 7, 160, 240, 16, 0.10), (16, 64, 7, 160, 240, 16, 0.14)]
    for (C, K, k, H, W, blk, ratio) in shapes:
        CH = (C + 7) // 8
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
        wq = torch.empty(lib.cbinfer_blockconv_prepared_bytes(C, K, k, k), dtype=torch.uint8, device="cuda")
        check(lib.cbinfer_blockconv_prep_weights(ptr(w), ptr(wq), K, C, k, k, None))

        def go():
            bits.copy_(mask)
            check(lib.cbinfer_conv_changed_blocks(ptr(x), ptr(bits), ptr(arrive), ptr(copy), ptr(wq), ptr(b), ptr(out),
                                                  C, H, W, K, k, k, 1, None))
        for _ in range(5):
            go()
        torch.cuda.synchronize()
        raw.cbinfer_debug_blk_stamps(None, 0, 1)
        torch.cuda.synchronize()
        go()
        torch.cuda.synchronize()
        buf = np.zeros(2048 * 32, dtype=np.uint64)
        raw.cbinfer_debug_blk_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(buf.nbytes), 0)
        st = buf.reshape(2048, 32).astype(np.int64)
        ran = st[:, 0] > 0
        act = ran & (st[:, 15] > 0)
        t0 = st[ran, 0].min()
        print("N=%d: %d workgroups started, %d active; start spread %.2f us, span %.2f us" % (
            int(cm.sum()), ran.sum(), act.sum(), (st[ran, 0].max() - t0) / 100.0, (st[act, 15].max() - t0) / 100.0))
        a = st[act]
        # when do the active workgroups start and end (deciles, us after the first workgroup's entry)
        q = [0, 10, 25, 50, 75, 90, 100]
        print("   active workgroups: start deciles %s | end deciles %s" % (
            " ".join("%.1f" % ((np.percentile(a[:, 0], x) - t0) / 100.0) for x in q),
            " ".join("%.1f" % ((np.percentile(a[:, 15], x) - t0) / 100.0) for x in q)))
        idle = st[ran & ~act]
        if len(idle):
            print("   inactive workgroups: %d, lifetime median %.2f us, last start %.1f us" % (
                len(idle), float(np.median(idle[:, 15] - idle[:, 0])) / 100.0 if (idle[:, 15] > 0).any() else -1,
                (idle[:, 0].max() - t0) / 100.0))
        med = lambda v: float(np.median(v)) / 100.0
        print("   multiplier: prologue %.2f | wait first chunk %.2f | per chunk %s | loop tail %.2f | store %.2f | lifetime %.2f (max %.2f)" % (
            med(a[:, 1] - a[:, 0]), med(a[:, 2] - a[:, 1]),
            " ".join("%.2f" % med(a[:, 3 + i] - a[:, 2 + i]) for i in range(min(CH, 8) - 1)),
            med(a[:, 14] - a[:, 1 + min(CH, 8)]), med(a[:, 15] - a[:, 14]), med(a[:, 15] - a[:, 0]),
            float((a[:, 15] - a[:, 0]).max()) / 100.0))
        print("   stager: start %.2f after entry | chunk staged at (rel. to its start) %s" % (
            med(a[:, 16] - a[:, 0]), " ".join("%.2f" % med(a[:, 17 + i] - a[:, 16]) for i in range(min(CH, 8)))))


if __name__ == "__main__":
    main()
