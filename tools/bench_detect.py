#!/usr/bin/env python3
"""Stand-alone timing of the pooled change detection with and without the producer's mask."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbinfer_amd._lib import C as lib, check, ptr  # noqa: E402
from tools.bench_rows import ev  # noqa: E402


def main():
    gen = torch.Generator().manual_seed(0)
    for (C, pH, pW, blk, ratio) in [(16, 320, 480, 32, 0.10), (64, 160, 240, 16, 0.10)]:
        H, W = pH // 2, pW // 2
        pre = torch.randn(1, C, pH, pW, device="cuda")
        state = torch.nn.functional.max_pool2d(pre, 2, 2).contiguous()
        cm = torch.zeros(pH, pW, dtype=torch.int8)
        gy, gx = pH // blk, pW // blk
        cells = torch.randperm(gy * gx, generator=gen)[:max(1, int(round(ratio * gy * gx)))]
        for c in cells.tolist():
            y0, x0 = (c // gx) * blk, (c % gx) * blk
            cm[max(0, y0 - 3):y0 + blk + 3, max(0, x0 - 3):x0 + blk + 3] = 1
        cm = cm.cuda()
        pre2 = pre.clone()
        pre2[:, :, cm.bool()] += 1.0          # the producer rewrote exactly these pixels
        wpr = lib.cbinfer_mask_words_per_row(pW)
        pad = torch.zeros(pH, wpr * 64, dtype=torch.int64, device="cuda")
        pad[:, :pW] = cm.long()
        pmask = (pad.view(pH, wpr, 64) << torch.arange(64, device="cuda")).sum(-1).view(-1).contiguous()
        bits = torch.zeros(lib.cbinfer_mask_words(H, W), dtype=torch.int64, device="cuda")
        res = {}
        for name, pm in (("full scan", None), ("producer mask", pmask)):
            st = state.clone()

            def go():
                check(lib.cbinfer_change_detection_bits_pooled(ptr(pre2), pH, pW, ptr(pm), ptr(st), ptr(bits), W, H, C,
                                                               3, 3, 0.05, 0, None))
            res[name] = ev(go)
            res[name + " bits"] = bits.clone()
            bits.zero_()
        same = torch.equal(res["full scan bits"], res["producer mask bits"])
        print("pooled detect C%d %dx%d -> %dx%d: full scan %.2f us | with producer mask %.2f us | same mask: %s" % (
            C, pH, pW, H, W, res["full scan"], res["producer mask"], same))


if __name__ == "__main__":
    main()
