#!/usr/bin/env python3
"""Per-kernel duration distribution of a rocprofv3 --kernel-trace CSV directory: min / quartiles / max and count.
usage: trace_hist.py <dir> [name filter]"""
import csv
import glob
import sys
import collections

import numpy as np

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
by = collections.defaultdict(list)
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if flt in n:
            by[n.replace("void ", "").replace("(anonymous namespace)::", "")[:60]].append(
                (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    v = np.array(v)
    if len(v) < 5:
        continue
    print("%-60s n=%5d min %6.1f p25 %6.1f p50 %6.1f p75 %6.1f max %6.1f us" % (
        n, len(v), v.min(), np.percentile(v, 25), np.median(v), np.percentile(v, 75), v.max()))
