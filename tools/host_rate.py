#!/usr/bin/env python3
"""Host-side enqueue time per frame of the bench network (eager) against the GPU time per frame."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
base, test = bench.build_bench_model()
frames = bench.bench_video(1234).frames(64)
with torch.no_grad():
    for f in frames[:8]:
        test(f)
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        n = 0
        for k in range(10):
            for i in range(8, 64):
                test(frames[i if k % 2 == 0 else 71 - i]); n += 1
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("frames %d: host enqueue %.1f us/frame, until GPU idle %.1f us/frame" % (n, 1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n))
    # per-module host cost
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for i in range(8, 64):
        test(frames[i])
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
