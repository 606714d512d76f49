#!/usr/bin/env python3
"""Kernel durations (in-process kernel trace, bench.traced_kernel_durations) of a SequenceBatch step of S sequences of
the bench workload.  usage: batch_trace.py [S ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import pycbinfer  # noqa: E402

for S in [int(a) for a in sys.argv[1:]] or [4, 8]:
    _, net = bench.build_bench_model()
    sb = pycbinfer.SequenceBatch(net, S)
    vids = [bench.bench_video(1234 + 7919 * q) for q in range(S)]
    walk = [v.frames(2 + 32) for v in vids]
    with torch.no_grad():
        for i in range(2):
            sb([w[i] for w in walk])

        def step(i):
            with torch.no_grad():
                sb([w[2 + bench.pingpong(i, 32)] for w in walk])
        got = bench.traced_kernel_durations(step, 40)
    if got[0] is None:
        print("S=%d: %s" % (S, got[1]))
        continue
    k, busy, span = got
    print("S=%d: busy %.1f us per step (%.1f us per frame), span %.1f us per step -> %.0f frames/s if back to back"
          % (S, busy, busy / S, span, 1e6 * S / busy))
    for n, d in sorted(k.items(), key=lambda x: -x[1]["avg_us"] * x[1]["launches_per_frame"]):
        print("   %7.2f us x %.0f  %s" % (d["avg_us"], d["launches_per_frame"], n[:100]))
