#!/usr/bin/env python3
"""Median FETCH_SIZE / WRITE_SIZE (KB as rocprofv3 reports them) per kernel from the pmc_fetch/ and pmc_write/
passes of tools/collect_profiles.sh.  Prints a text summary and, with --json OUT, writes the file bench.py reads
for `roofline.traffic` (keyed by the bench's layer labels, stamped with the hash of the kernel sources the
passes ran on).  In the bench workload every layer has a contraction kernel of its own: cb_rowconv (3->16),
cbs_conv_kernel<64,64,...> (16->64), cbs_conv_kernel<128,128,...> + its reduce launch (64->256), so no clustering
by value is needed."""
import csv
import glob
import json
import os
import statistics
import sys
from collections import defaultdict

root = sys.argv[1]
out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
commit = sys.argv[sys.argv.index("--commit") + 1] if "--commit" in sys.argv else "?"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load(sub, counter):
    by = defaultdict(list)
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                by[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return by


fetch, write = load("pmc_fetch", "FETCH_SIZE"), load("pmc_write", "WRITE_SIZE")
L1, L2, L3, TAIL = ("conv 3->16 k7 @320x480", "conv 16->64 k7 @160x240", "conv 64->256 k7 @80x120",
                    "tail 1x1 256->64->8 @80x120")
# the kernel of every layer in the bench workload (first pattern that matches), by the bench's layer label
LAYER = [("cbp_rowpair_kernel", L1), ("cb_rowconv_f32_kernel", L1),
         ("cbs_conv_kernel<64, 64", L2), ("cb_blockconv_kernel", L2),
         ("cbs_conv_kernel<128, 128", L3), ("cb_mfma_f32_kernel<4, 2, 2, 1, 2, true", L3),
         ("cb_mfma_f32_kernel<2, 4, 2, 1, 2, true", L3), ("cb_mfma_f32_kernel<2, 2, 2, 1, 2, true", L3),
         ("cb_tail1x1_kernel", TAIL)]
# second launch of the same contraction (summed into the layer's entry)
ADDS = [("cbs_reduce_tail_kernel", L3), ("cbs_reduce_kernel", L3), ("cb_splitk_reduce_kernel<2, true", L3)]
table = {}
extra = {}
for name in sorted(set(fetch) | set(write)):
    if "cb_" not in name and "cbs" not in name and "cbp" not in name:
        continue
    short = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void cbs::", "cbs::").replace("void cbp::", "cbp::")[:90]
    # steady state: drop the first dispatches (100 %-change first frame, priming)
    f = fetch.get(name, [])[3:] or fetch.get(name, [])
    w = write.get(name, [])[3:] or write.get(name, [])
    fm = statistics.median(f) if f else 0.0
    wm = statistics.median(w) if w else 0.0
    print("%-92s n=%4d  FETCH_SIZE median %10.2f KB   WRITE_SIZE median %10.2f KB" % (short, max(len(f), len(w)), fm, wm))
    for pat, label in LAYER:
        if pat in name and label not in table:
            table[label] = {"kernel": short, "fetch_kb": fm, "write_kb": wm,
                            "bytes_per_launch": int((fm + wm) * 1024),
                            "bytes_per_launch_fetch_x2": int((2 * fm + wm) * 1024)}
    for prio, (pat, label) in enumerate(ADDS):      # (the first pattern of ADDS that occurs wins)
        if pat in name and (label not in extra or prio < extra[label][0]):
            extra[label] = (prio, short, fm, wm)
for label, (_, short, fm, wm) in extra.items():      # second launch of the same contraction: add its bytes
    if label in table:
        t = table[label]
        t["kernel"] += " + " + short
        t["fetch_kb"] += fm
        t["write_kb"] += wm
        t["bytes_per_launch"] = int((t["fetch_kb"] + t["write_kb"]) * 1024)
        t["bytes_per_launch_fetch_x2"] = int((2 * t["fetch_kb"] + t["write_kb"]) * 1024)
if out_json:
    import bench
    table["_note"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, counters + kernel trace only; "
                      "tools/collect_profiles.sh) over `python3 bench.py --mode eager --steps 40 --warmup 5 "
                      "--no-cpu-baseline --no-dense --multi 0`, MI355X.  Per-dispatch medians in KB as reported "
                      "(x1024 = bytes).  FETCH_SIZE counts L2->fabric read requests: bytes still resident in the "
                      "XCD's L2 are not counted, and on gfx950 it under-reports wide 16-B/lane streaming reads by 2x "
                      "(MI355X_MICROARCH.md, HBM section): `bytes_per_launch` is the raw sum, "
                      "`bytes_per_launch_fetch_x2` the upper bound.")
    table["kernel_source_sha256"] = bench.kernel_source_hash()
    table["commit"] = commit
    json.dump(table, open(out_json, "w"), indent=1)
    print("wrote", out_json)
