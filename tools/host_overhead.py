#!/usr/bin/env python3
"""Diagnostic: host enqueue time vs total time per frame of the eager (non-graph) launch mode on the headline
configuration, plus a cProfile of the Python side.  Eager is GPU-bound as long as enqueue < total."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pycbinfer
from cbinfer_amd import workloads
base, test = workloads.sceneLabelingModels(experimentIdx=6, threshold=0.05)
for m in test.modules():
    if type(m) is pycbinfer.CBPoolMax2d: m.cloneOutput = False
pycbinfer.fusePoolingIntoDetection(test)
vid = workloads.SyntheticVideo(H=320, W=480, ratio=0.1, block=32, seed=1234)
frames = vid.frames(230)
with torch.no_grad():
    for f in frames[:30]: test(f)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in frames[30:230]: test(f)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print("enqueue %.1f us/frame, total %.1f us/frame" % ((t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6))
import cProfile, pstats
pr = cProfile.Profile()
with torch.no_grad():
    pr.enable()
    for f in frames[30:130]: test(f)
    pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
