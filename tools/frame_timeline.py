#!/usr/bin/env python3
"""Steady-state frames from a rocprofv3 --kernel-trace CSV of the bench: the dispatches are cut into frames (a
frame starts with the detection in front of the first layer's row-segment contraction, or with that contraction
when it detects itself), frames are grouped by
their sequence of kernel names -- the timed loop and the in-frame measurement passes issue different sequences --
and for every common sequence the medians of durations and of the idle gaps between consecutive kernels are
printed.  Under the profiler an eager run is host-bound (the gaps are the host's), so the durations are the
figures to read.
usage: frame_timeline.py <dir-or-csv>"""
import collections
import csv
import glob
import os
import statistics
import sys

path = sys.argv[1]
files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    return n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace(
        "void cbs::", "cbs::").replace("void cbp::", "cbp::")


# (round 6: a row-pair launch that detects the layer's changes itself -- cbp_rowpair_kernel<7, 7, true> -- IS the frame's first)
starts = [(i if "cbp_rowpair_kernel<7, 7, true>" in r["Kernel_Name"] else i - 1) for i, r in enumerate(rows)
          if ("cb_rowconv_f32_kernel" in r["Kernel_Name"] or "cbp_rowpair_kernel" in r["Kernel_Name"]) and i > 0]
groups = collections.defaultdict(list)
for a, b in zip(starts[:-1], starts[1:]):
    fr = rows[a:b]
    if 4 <= len(fr) <= 12:
        groups[tuple(short(r["Kernel_Name"]).split("(")[0] for r in fr)].append(fr)
for sig, frames in sorted(groups.items(), key=lambda kv: -len(kv[1]))[:3]:
    if len(frames) < 20:
        continue
    frames = frames[len(frames) // 5:]      # (drop the warm-up fifth)
    print("%d frames of %d kernels" % (len(frames), len(sig)))
    tot = []
    for k in range(len(sig)):
        dur = [(int(f[k]["End_Timestamp"]) - int(f[k]["Start_Timestamp"])) / 1e3 for f in frames]
        gap = [(int(f[k]["Start_Timestamp"]) - int(f[k - 1]["End_Timestamp"])) / 1e3 for f in frames] if k else [0.0]
        print("  gap %6.2f us | %7.2f us  %s" % (statistics.median(gap), statistics.median(dur),
                                               short(frames[0][k]["Kernel_Name"])[:80]))
        tot.append(statistics.median(dur))
    print("  sum of kernel durations %.1f us" % sum(tot))
