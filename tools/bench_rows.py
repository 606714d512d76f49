#!/usr/bin/env python3
"""Stand-alone timing of the row-segment contraction (cbinfer_conv_changed_rows) on the scene-labeling L1/L2
shapes with the change pattern of the bench (32x32 input blocks -> dilated), next to the list kernel.
CBINFER_ROW_DBG ablations are applied by the library (set in the environment before the process starts)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbinfer_amd import conv2d_cg as cg  # noqa: E402
from cbinfer_amd._lib import C as lib, check, ptr  # noqa: E402


def ev(fn, reps=60):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


def main():
    gen = torch.Generator().manual_seed(0)
    for (C, K, k, H, W, blk, ratio) in [(3, 16, 7, 320, 480, 32, 0.10), (16, 64, 7, 160, 240, 16, 0.10)]:
        x = torch.randn(1, C, H, W, device="cuda")
        w = torch.randn(K, C, k, k, device="cuda") / (C * k * k) ** 0.5
        b = torch.randn(K, device="cuda")
        out = torch.zeros(1, K, H, W, device="cuda")
        # changed blocks on the block grid, dilated by the filter support
        cm = torch.zeros(H, W, dtype=torch.int8)
        gy, gx = H // blk, W // blk
        cells = torch.randperm(gy * gx, generator=gen)[:max(1, int(round(ratio * gy * gx)))]
        for c in cells.tolist():
            y0, x0 = (c // gx) * blk, (c % gx) * blk
            cm[max(0, y0 - 3):y0 + blk + 3, max(0, x0 - 3):x0 + blk + 3] = 1
        cm = cm.cuda()
        idx = cg.changeIndexesExtr(cm)
        N = idx.numel()
        words = lib.cbinfer_mask_words(H, W)
        # bit mask from the byte map
        wpr = lib.cbinfer_mask_words_per_row(W)
        pad = torch.zeros(H, wpr * 64, dtype=torch.int64, device="cuda")
        pad[:, :W] = cm.long()
        mask = (pad.view(H, wpr, 64) << torch.arange(64, device="cuda")).sum(-1).view(-1).contiguous()
        bits = torch.zeros(words, dtype=torch.int64, device="cuda")
        arrive = torch.zeros(words, dtype=torch.int32, device="cuda")
        copy = torch.zeros(words, dtype=torch.int64, device="cuda")
        wq = torch.empty(lib.cbinfer_rowconv_prepared_bytes(C, K, k, k), dtype=torch.uint8, device="cuda")
        check(lib.cbinfer_rowconv_prep_weights(ptr(w), ptr(wq), K, C, k, k, None))
        wp = cg.prepWeights(w, H, W)
        ws = cg.newConvWorkspace(x.device)
        t_fill = ev(lambda: bits.copy_(mask))
        t_rows = ev(lambda: (bits.copy_(mask), check(lib.cbinfer_conv_changed_rows(
            ptr(x), ptr(bits), ptr(arrive), ptr(copy), ptr(wq), ptr(b), ptr(out), C, H, W, K, k, k, 1, None)))) - t_fill
        cnt = torch.tensor([N], dtype=torch.int32, device="cuda")
        t_list = ev(lambda: check(lib.cbinfer_conv_changed(
            ptr(x), ptr(idx), N, ptr(cnt), ptr(wp), ptr(b), ptr(out), C, H, W, K, k, k, 1, 0, None, 0, ptr(ws), 0,
            None)))
        nz = int((mask != 0).sum())
        print("%d->%d k%d @%dx%d  N=%d (%.1f%%)  non-zero words %d of %d | rows kernel %.1f us | list kernel %.1f us | "
              "dbg=%s" % (C, K, k, H, W, N, 100.0 * N / (H * W), nz, words, t_rows, t_list,
                          os.environ.get("CBINFER_ROW_DBG", "0")), flush=True)


if __name__ == "__main__":
    main()
