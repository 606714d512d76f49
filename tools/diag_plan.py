#!/usr/bin/env python3
"""Diagnostic: how often do the modules of the bench network leave their call plans (the per-frame fast path)
during a bench run?  usage: diag_plan.py [bench.py arguments]"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cbinfer_amd import conv2d  # noqa: E402

counts = collections.Counter()
for name in ("_forward_split", "_run_split_plan", "_forward_pooled", "_forward_fused", "forward_normal"):
    orig = getattr(conv2d.CBConv2d, name)

    def wrap(self, *a, _o=orig, _n=name, **k):
        r = _o(self, *a, **k)
        counts[_n + (" -> None" if r is None else "")] += 1
        return r
    setattr(conv2d.CBConv2d, name, wrap)
otail = conv2d.CBTail1x1.forward


def tail(self, inp):
    counts["tail forward" + (" (done by producer)" if getattr(inp[2], 'tailDone', None) is self else " (own launch)")] += 1
    return otail(self, inp)


conv2d.CBTail1x1.forward = tail
bench.main()
for k, v in sorted(counts.items()):
    print("%8d  %s" % (v, k), file=sys.stderr)
