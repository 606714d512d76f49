#!/usr/bin/env python3
"""Profiling target for BASELINE config 4: OpenPose T=2, 368x654, fp16 (cg_half path), all 36 convs converted, 10 % of
the input re-drawn per frame in 16x16 blocks, eager launches (every kernel its own dispatch).  Prints frames/s of the
change-based and the dense network and the per-layer change ratios.  usage: pose_target.py [feedback]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pycbinfer  # noqa: E402
from cbinfer_amd import workloads  # noqa: E402


def main():
    feedback = len(sys.argv) > 1 and sys.argv[1] == "feedback"
    H, W = 368, 654
    vid = workloads.SyntheticVideo(H=H, W=672, ratio=0.10, block=16, seed=3)
    frames = [(f[:, :, :, :W] * (255.0 / 256.0) - 0.5).half().contiguous() for f in vid.frames(40)]
    test = workloads.convertOpenPose(workloads.OpenPoseModel(T=2).cuda().half(), threshold=0.02, feedbackLoop=feedback)
    base = workloads.OpenPoseModel(T=2).cuda().half()
    with torch.no_grad():
        for f in frames[:6]:
            test(f)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for f in frames[6:]:
            test(f)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        for f in frames[:3]:
            base(f)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for f in frames[6:]:
            base(f)
        torch.cuda.synchronize()
        dd = time.perf_counter() - t1
    n = len(frames) - 6
    print("OpenPose T=2 %dx%d fp16%s: change-based %.0f frames/s (%.1f us per frame), dense %.0f frames/s, eager"
          % (H, W, ", feedback mode" if feedback else "", n / dt, 1e6 * dt / n, n / dd))
    rs = []
    for m in test.modules():
        if type(m) is pycbinfer.CBConv2d and m.lastChangeIndexes() is not None:
            ci = m.lastChangeIndexes()
            K, C, kH, kW = m.weight.shape
            r = ci.numel() / float(ci.size[0] * ci.size[1])
            rs.append(r)
            print("  conv %3d->%3d k%d @%dx%d: %5.1f %% of the pixels recomputed, %.1f MFLOP"
                  % (C, K, kH, ci.size[0], ci.size[1], 100 * r, 2e-6 * ci.numel() * C * kH * kW * K))
    print("mean post-dilation ratio over %d layers: %.1f %%" % (len(rs), 100 * sum(rs) / max(1, len(rs))))


if __name__ == "__main__":
    main()
