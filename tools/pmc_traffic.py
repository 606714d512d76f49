#!/usr/bin/env python3
"""Median FETCH_SIZE / WRITE_SIZE (KB as rocprofv3 reports them) per kernel and per value cluster from the
pmc_fetch/ and pmc_write/ passes of tools/collect_profiles.sh.  The same contraction kernel serves several
layers; its dispatches separate cleanly into clusters by traffic (L2 < L3)."""
import csv
import glob
import os
import statistics
import sys
from collections import defaultdict

root = sys.argv[1]


def load(sub, counter):
    by = defaultdict(list)
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                by[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return by


def clusters(v):
    v = sorted(v)
    out, cur = [], [v[0]]
    for x in v[1:]:
        if x > cur[-1] * 1.3 + 8 and len(cur) >= 3:
            out.append(cur)
            cur = [x]
        else:
            cur.append(x)
    out.append(cur)
    return [c for c in out if len(c) >= 5]


fetch, write = load("pmc_fetch", "FETCH_SIZE"), load("pmc_write", "WRITE_SIZE")
for name in sorted(set(fetch) | set(write)):
    if not ("cb_" in name):
        continue
    short = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:80]
    print(short)
    for label, d in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
        if name in d:
            for c in clusters(d[name]):
                print("    %-10s n=%4d median %10.2f KB  (min %.1f max %.1f)" % (label, len(c), statistics.median(c), c[0], c[-1]))
