#!/usr/bin/env python3
"""Per-module time inside the eager bench frame by HIP events (one module bracketed per pass, like
bench.inframe_conv_times but at module granularity: a CBConv2d = its detection + contraction launches)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

base, test = bench.build_bench_model()
frames = bench.bench_video(1234).frames(40)
mods = list(test.children())
with torch.no_grad():
    for f in frames[:4]:
        test(f)
    torch.cuda.synchronize()
    res = []
    for which in range(len(mods) + 1):
        pairs = []
        for it, f in enumerate(frames[4:]):
            x = f
            e0 = e1 = None
            if which == len(mods):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
            for i, m in enumerate(mods):
                if i == which:
                    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                    e0.record()
                x = m(x)
                if i == which:
                    e1.record()
            if which == len(mods):
                e1.record()
            pairs.append((e0, e1))
        torch.cuda.synchronize()
        ts = sorted(1e3 * a.elapsed_time(b) for a, b in pairs[4:])
        res.append(ts[len(ts) // 2])
    for m, t in zip(mods + ["whole frame"], res):
        print("%8.1f us  %s" % (t, m if isinstance(m, str) else repr(m)[:90]))
    print("sum of modules %.1f us" % sum(res[:-1]))
