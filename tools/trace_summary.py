#!/usr/bin/env python3
"""Per-kernel in-situ durations from a rocprofv3 --kernel-trace CSV: groups dispatches by kernel name and,
within a name, by duration cluster (the same contraction kernel serves several layers).
usage: trace_summary.py <dir-or-csv> [skip_first_n_dispatches]"""
import csv
import glob
import os
import sys
from collections import defaultdict

path = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[skip:]
by = defaultdict(list)
for r in rows:
    by[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in by.values())
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
print("dispatches %d  kernel time %.0f us  span %.0f us  (busy %.0f%%)" % (len(rows), tot, span, 100 * tot / span))
for name, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    # split into clusters where consecutive sorted durations jump by > 35 %
    clusters, cur = [], [v[0]]
    for x in v[1:]:
        if x > cur[-1] * 1.35 and len(cur) >= 3:
            clusters.append(cur)
            cur = [x]
        else:
            cur.append(x)
    clusters.append(cur)
    short = name[:90]
    print("%6.1f%%  %s" % (100 * sum(v) / tot, short))
    for c in clusters:
        print("          n=%5d  avg %8.2f us  min %8.2f  max %8.2f" % (len(c), sum(c) / len(c), c[0], c[-1]))
