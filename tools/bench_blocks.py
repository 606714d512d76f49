#!/usr/bin/env python3
"""Stand-alone timing of the patch-staged contraction (cbinfer_conv_changed_blocks) on the scene-labeling
L2/L3 shapes with the bench's change pattern, next to the list kernel (exact f32 and bf16x3)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbinfer_amd import conv2d_cg as cg  # noqa: E402
from cbinfer_amd._lib import C as lib, check, ptr, CB_F32S  # noqa: E402
from tools.bench_rows import ev  # noqa: E402


def main():
    gen = torch.Generator().manual_seed(0)
    for (C, K, k, H, W, blk, ratio) in [(16, 64, 7, 160, 240, 16, 0.10), (64, 256, 7, 80, 120, 8, 0.10),
                                        (64, 256, 7, 80, 120, 8, 0.01), (64, 256, 7, 80, 120, 8, 1.0)]:
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
        idx = cg.changeIndexesExtr(cm)
        N = idx.numel()
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
        t_fill = ev(lambda: bits.copy_(mask))
        t_blk = ev(lambda: (bits.copy_(mask), check(lib.cbinfer_conv_changed_blocks(
            ptr(x), ptr(bits), ptr(arrive), ptr(copy), ptr(wq), ptr(b), ptr(out), C, H, W, K, k, k, 1, None)))) - t_fill
        cnt = torch.tensor([N], dtype=torch.int32, device="cuda")
        ws = cg.newConvWorkspace(x.device)
        res = {}
        for name, ar in (("exact", 0), ("split", CB_F32S)):
            wp = cg.prepWeights(w, H, W, arith=ar)
            res[name] = ev(lambda: check(lib.cbinfer_conv_changed(
                ptr(x), ptr(idx), N, ptr(cnt), ptr(wp), ptr(b), ptr(out), C, H, W, K, k, k, 1, 0, None, 0, ptr(ws), ar,
                None)))
        fl = 2.0 * N * C * k * k * K
        print("%d->%d k%d @%dx%d N=%d (%.0f%%): block kernel %.1f us (%.0f TFLOP/s) | list f32 %.1f | list bf16x3 %.1f"
              % (C, K, k, H, W, N, 100.0 * N / (H * W), t_blk, fl / t_blk / 1e6, res["exact"], res["split"]), flush=True)


if __name__ == "__main__":
    main()
