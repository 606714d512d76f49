#!/usr/bin/env python3
"""One steady-state frame from a rocprofv3 --kernel-trace CSV of the bench (eager): start offsets, durations
and the idle gaps between consecutive kernels, medians over many frames.
usage: frame_timeline.py <dir-or-csv>"""
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
rows = rows[len(rows) // 2:]          # second half: the timed region / in-frame measurement
# a frame starts with the detection in front of the first layer's row-segment contraction
starts = [i - 1 for i, r in enumerate(rows) if "cb_rowconv_f32_kernel" in r["Kernel_Name"] and i > 0]
frames = []
for a, b in zip(starts[:-1], starts[1:]):
    fr = rows[a:b]
    if 5 <= len(fr) <= 12:
        frames.append(fr)
lens = statistics.mode([len(f) for f in frames])
frames = [f for f in frames if len(f) == lens]
print("%d frames of %d kernels" % (len(frames), lens))
tot = []
for k in range(lens):
    dur = [(int(f[k]["End_Timestamp"]) - int(f[k]["Start_Timestamp"])) / 1e3 for f in frames]
    gap = [(int(f[k]["Start_Timestamp"]) - int(f[k - 1]["End_Timestamp"])) / 1e3 for f in frames] if k else [0.0]
    name = frames[0][k]["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void cbs::", "cbs::")
    print("  gap %6.2f us | %7.2f us  %s" % (statistics.median(gap), statistics.median(dur), name[:80]))
    tot.append(statistics.median(dur) + statistics.median(gap))
period = [(int(b[0]["Start_Timestamp"]) - int(a[0]["Start_Timestamp"])) / 1e3 for a, b in zip(frames[:-1], frames[1:])]
print("sum of kernels+gaps %.1f us; frame period median %.1f us" % (sum(tot), statistics.median(period)))
