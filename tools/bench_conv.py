#!/usr/bin/env python3
"""Tuning aid: time the fused gather->MFMA->scatter kernel (cbinfer_conv_changed) on the
scene-labeling layer shapes for several change ratios and tile configurations
(CBINFER_CONV_CFG = 100*WM + 10*WN + KS).  Prints one line per (layer, ratio, cfg)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbinfer_amd import conv2d_cg as cg  # noqa: E402

LAYERS = [(3, 16, 7, 320, 480), (16, 64, 7, 160, 240), (64, 256, 7, 80, 120), (256, 64, 1, 80, 120)]
if os.environ.get('CBINFER_BENCH_LAYER'):
    LAYERS = [LAYERS[int(os.environ['CBINFER_BENCH_LAYER'])]]


def time_ms(fn, reps=30):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def blocks_list(H, W, ratio, blk, gen):
    """flat indices of a union of blk x blk cells covering `ratio` of the map (ascending)."""
    gh, gw = H // blk, W // blk
    n = max(1, int(round(ratio * gh * gw)))
    cells = torch.randperm(gh * gw, generator=gen)[:n]
    m = torch.zeros(H, W, dtype=torch.bool)
    for c in cells.tolist():
        y0, x0 = (c // gw) * blk, (c % gw) * blk
        m[y0:y0 + blk, x0:x0 + blk] = True
    return torch.nonzero(m.view(-1)).view(-1).int().cuda()


def main():
    cfgs = [int(c) for c in sys.argv[1:]] or [0]
    gen = torch.Generator().manual_seed(0)
    for (C, K, k, H, W) in LAYERS:
        dt = torch.float16 if os.environ.get("CBINFER_BENCH_HALF") else torch.float32
        x = torch.randn(1, C, H, W, device="cuda").to(dt)
        w = (torch.randn(K, C, k, k, device="cuda") / (C * k * k) ** 0.5).to(dt)
        b = torch.randn(K, device="cuda").to(dt)
        out = torch.zeros(1, K, H, W, device="cuda", dtype=dt)
        wp = cg.prepWeights(w, H, W)
        for ratio in (0.1, 0.2, 0.36, 1.0):
            idx = blocks_list(H, W, ratio, 8, gen)
            N = idx.numel()
            flops = 2.0 * N * C * k * k * K
            for cfg in cfgs:
                narrow = K <= 32
                if cfg and narrow and cfg // 100 != 1:      # K <= 32 runs the 32-row tiles only
                    continue
                if cfg:
                    os.environ["CBINFER_CONV_CFG"] = str(cfg)
                else:
                    os.environ.pop("CBINFER_CONV_CFG", None)
                buf = torch.zeros(H * W, dtype=torch.int32, device="cuda")
                buf[:N] = idx
                ci = cg.ChangeIndexes(buf, torch.tensor([N], dtype=torch.int32, device="cuda"))
                try:
                    ms = time_ms(lambda: cg.convChanged(x, idx, w, b, out, withReLU=True, weightsPrepared=wp))
                    ms_cap = time_ms(lambda: cg.convChanged(x, ci, w, b, out, withReLU=True, weightsPrepared=wp))
                except Exception as e:  # config not valid for this shape
                    print("conv %d->%d k%d %dx%d N=%d cfg=%d: %s" % (C, K, k, H, W, N, cfg, e))
                    continue
                print("conv %3d->%3d k%d %3dx%3d ratio=%.2f N=%6d cfg=%3d: %8.2f us  %6.2f TFLOP/s | worst-case grid %8.2f us"
                      % (C, K, k, H, W, ratio, N, cfg, ms * 1e3, flops / ms / 1e9, ms_cap * 1e3), flush=True)


if __name__ == "__main__":
    main()
