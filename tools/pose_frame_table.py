#!/usr/bin/env python3
"""One steady-state frame of tools/pose_target.py from a rocprofv3 --kernel-trace CSV: the launches in order with their
durations (device timestamps), and totals by kernel.  usage: pose_frame_table.py <dir with *_kernel_trace.csv> [frame]"""
import collections
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# a frame starts with the detection launch of the 3-channel first layer: the only layer on the list kernels since round 6
# (cb_detect_kernel; every other layer's detection is cbh_detect_kernel or rides in its producer's launch)
det = [i for i, r in enumerate(rows) if "cb_detect_kernel" in r["Kernel_Name"]]
# (default: a frame of the timed walk -- 25 frames before the last one; the threshold calibration runs in front of it)
frame = int(sys.argv[2]) if len(sys.argv) > 2 else len(det) - 25
a, b = det[frame], det[frame + 1]
t0 = int(rows[a]["Start_Timestamp"])
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows[a:b]:
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:70]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg[name][0] += 1
    agg[name][1] += d
    print("%8.1f %7.2f  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, d, name))
print("frame span %.1f us (serialised by the tracer), %d launches" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3, b - a))
for n, (c, d) in sorted(agg.items(), key=lambda x: -x[1][1]):
    print("%4d x %7.2f us = %8.1f us  %s" % (c, d / c, d, n))
