#!/usr/bin/env python3
"""Stand-alone timing of the fused list contraction on the scene-labeling L2/L3 shapes: exact f32 vs bf16x3
split (CBINFER_CONV_DBG ablations applied by the library)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbinfer_amd import conv2d_cg as cg  # noqa: E402
from cbinfer_amd._lib import C as lib, check, ptr, CB_F32S  # noqa: E402
from tools.bench_rows import ev  # noqa: E402


def main():
    gen = torch.Generator().manual_seed(0)
    for (C, K, k, H, W, blk, ratio) in [(16, 64, 7, 160, 240, 16, 0.10), (64, 256, 7, 80, 120, 8, 0.10)]:
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
        idx = cg.changeIndexesExtr(cm.cuda())
        N = idx.numel()
        cnt = torch.tensor([N], dtype=torch.int32, device="cuda")
        ws = cg.newConvWorkspace(x.device)
        res = {}
        for name, ar in (("exact", 0), ("split", CB_F32S)):
            wp = cg.prepWeights(w, H, W, arith=ar)
            res[name] = ev(lambda: check(lib.cbinfer_conv_changed(
                ptr(x), ptr(idx), N, ptr(cnt), ptr(wp), ptr(b), ptr(out), C, H, W, K, k, k, 1, 0, None, 0, ptr(ws), ar,
                None)))
        print("%d->%d k%d @%dx%d N=%d (%.0f%%): exact f32 %.1f us | bf16x3 %.1f us | dbg=%s" % (
            C, K, k, H, W, N, 100.0 * N / (H * W), res["exact"], res["split"], os.environ.get("CBINFER_CONV_DBG", "0")),
            flush=True)


if __name__ == "__main__":
    main()
