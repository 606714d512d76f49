#!/usr/bin/env python3
"""Sum the counter rows of kernels matching a name over the per-pass CSVs pmc_passes.sh wrote.
usage: pmc_summarize.py OUTDIR [kernel-substring]"""
import csv
import glob
import sys
from collections import defaultdict

out = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "cb_mfma"
tot, cnt = defaultdict(float), defaultdict(int)
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        if pat in row["Kernel_Name"]:
            tot[row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[row["Counter_Name"]] += 1
for k in sorted(tot):
    print("%-40s %16.0f per launch (%d rows)" % (k, tot[k] / max(cnt[k], 1), cnt[k]))
