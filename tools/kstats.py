#!/usr/bin/env python3
"""Print a rocprofv3 *kernel_stats.csv (name, calls, average us, total us) restricted to the library's kernels.
usage: kstats.py <dir or csv> [all]"""
import csv
import glob
import os
import re
import sys

p = sys.argv[1]
f = p if p.endswith(".csv") else glob.glob(os.path.join(p, "**", "*kernel_stats.csv"), recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if len(sys.argv) < 3 and not re.search(r"cb[shp]?_|cbs::|cbp::", n):
        continue
    n = re.sub(r"\(.*", "", n).replace("void ", "").replace("(anonymous namespace)::", "")
    print("%5d x %8.2f us = %9.1f us  %s" % (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3, n[:90]))
