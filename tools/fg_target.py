#!/usr/bin/env python3
"""Profiling target for the fine-grained path (BASELINE config 3, "fine-grained + CBPoolMax2d"): experiment 7 at
480x320 with its two pools change-based (pycbinfer.insertCBPooling) and folded into the fine-grained detections
(fusePoolingIntoDetection), 10 % of the pixels re-drawn per frame in 16x16 blocks, in-place execution form, eager
launches (every kernel its own dispatch).  Prints frames/s and the per-layer touched-pixel counts.
FG_DENSE_POOLS=1: the experiment as the reference's loader builds it (nn.MaxPool2d)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pycbinfer  # noqa: E402
from cbinfer_amd import workloads  # noqa: E402


def main():
    ratio = float(sys.argv[1]) if len(sys.argv) > 1 else 0.10
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    base, test = workloads.sceneLabelingModels(experimentIdx=7, threshold=0.05)
    cbs = [m for m in test.modules() if type(m) is pycbinfer.CBConv2d]
    for m in cbs:
        m.fgInPlace = True
    if os.environ.get("FG_DENSE_POOLS", "0") != "1":
        pycbinfer.insertCBPooling(test, cloneOutput=False)
    pycbinfer.fuseTail1x1(test)
    pycbinfer.fusePoolingIntoDetection(test)
    vid = workloads.SyntheticVideo(H=320, W=480, ratio=ratio, block=16, seed=7)
    frames = vid.frames(64)
    with torch.no_grad():
        for f in frames[:8]:
            test(f)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            j = i % 110
            test(frames[8 + (j if j < 56 else 110 - j)])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print("fine-grained experiment 7 + CBPoolMax2d (folded), in-place form, %.0f %% change: %.0f frames/s"
          % (100 * vid.ratio, steps / dt))
    for m in cbs:
        K, C, kH, kW = m.weight.shape
        # (a mask-driven fine-grained frame leaves the mask of touched output pixels; its list and count are made
        #  when somebody asks -- lastChangeIndexes().count does)
        ci = m.lastChangeIndexes()
        n = int(ci.count.item()) if ci is not None else int(m._work['count'].item())
        H, W = m.prevOutput.shape[-2:]
        print("  conv %d->%d k%d @%dx%d: %d touched output pixels (%.0f %%), algorithmic %.1f MFLOP (2*N*C*k*k*K)"
              % (C, K, kH, H, W, n, 100.0 * n / (H * W), 2e-6 * n * C * kH * kW * K))


if __name__ == "__main__":
    main()
