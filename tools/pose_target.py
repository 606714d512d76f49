#!/usr/bin/env python3
"""Profiling target for BASELINE config 4: OpenPose T=2, 368x654, fp16 (cg_half path), all 36 convs converted, 10 % of
the input re-drawn per frame in 16x16 blocks, eager launches (every kernel its own dispatch).  Prints frames/s of the
change-based and the dense network and the per-layer change ratios.
LIVE network (default since round 5): variance-preserving random weights (workloads.OpenPoseModel(init='kaiming')) and
per-layer thresholds that give every layer a post-dilation change ratio of 10 % on this video
(workloads.calibrateChangeRatio) -- all 36 layers recompute; POSE_INIT=default POSE_TH=0.02 is rounds 1-4's artefact
(nn.Conv2d's default init, one threshold: the change dies behind the fifth conv).
usage: pose_target.py [feedback]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pycbinfer  # noqa: E402
from cbinfer_amd import workloads  # noqa: E402


def main():
    feedback = len(sys.argv) > 1 and sys.argv[1] == "feedback"
    H, W = 368, 654
    vid = workloads.SyntheticVideo(H=H, W=672, ratio=0.10, block=16, seed=3)

    def prep(f):        # PoseDetector.py:72: x * 255/256 - 0.5, fp16
        return (f[:, :, :, :W] * (255.0 / 256.0) - 0.5).half().contiguous()
    init = os.environ.get("POSE_INIT", "kaiming")
    conc = os.environ.get("POSE_CONCURRENT", "0") == "1"      # the two branches of every stage on two HIP streams
    grp = os.environ.get("POSE_GROUPED", "1") == "1" and not conc      # ... or in lockstep, one launch per layer pair
    test = workloads.convertOpenPose(workloads.OpenPoseModel(T=2, init=init, concurrentBranches=conc,
                                                             groupedBranches=grp).cuda().half(),
                                     threshold=float(os.environ.get("POSE_TH", "0.02")), feedbackLoop=feedback)
    if os.environ.get("POSE_NOFOLD", "0") != "1":      # (round 6: the consumers' detection inside the producers' launches)
        workloads.fuseOpenPoseDetections(test)
    base = workloads.OpenPoseModel(T=2, init=init, concurrentBranches=conc).cuda().half()
    if "POSE_TH" not in os.environ:
        # (calibrated on the running video; the timed walk continues it, so the network is in its steady state)
        workloads.calibrateChangeRatio(test, lambda: prep(vid.next()), target=float(os.environ.get("POSE_TARGET", "0.10")))
    if os.environ.get("POSE_POOLS", "0") == "1":       # the three VGG pools change-based and folded into the detections
        pycbinfer.insertCBPooling(test, cloneOutput=False)      # (behind the calibration: its hooks read tensor inputs)
        pycbinfer.fusePoolingIntoDetection(test)
    frames = [prep(vid.frame)] + [prep(vid.next()) for _ in range(49)]
    frames, fresh = frames[:40], frames[40:]      # (the last ten: for the per-layer ratios, behind the timed walk)
    with torch.no_grad():
        for f in frames[:6]:
            test(f)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for f in frames[6:]:
            test(f)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        for f in frames[:3]:
            base(f)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for f in frames[6:]:
            base(f)
        torch.cuda.synchronize()
        dd = time.perf_counter() - t1
    n = len(frames) - 6
    print("OpenPose T=2 %dx%d fp16%s: change-based %.0f frames/s (%.1f us per frame), dense %.0f frames/s, eager"
          % (H, W, ", feedback mode" if feedback else "", n / dt, 1e6 * dt / n, n / dd))
    if os.environ.get("POSE_PROGRAM", "0") == "1":      # the frame as a recorded launch program (pycbinfer.FrameProgram)
        test.libraryConcat = True
        more = [prep(vid.next()) for _ in range(46)]
        with torch.no_grad():
            for f in more[:4]:
                test(f)
            prog = pycbinfer.FrameProgram(test)
            prog.record(more[4])
            for f in more[5:10]:
                prog(f)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for f in more[10:]:
                prog(f)
            torch.cuda.synchronize()
            dg = time.perf_counter() - t0
        print("  recorded launch program (%d library calls per frame): %.0f frames/s (%.1f us per frame)"
              % (len(prog.calls), 36 / dg, 1e6 * dg / 36))
        fresh = [prep(vid.next()) for _ in range(10)]
    if os.environ.get("POSE_GRAPH", "0") == "1":      # the same walk replayed from a hipGraph (one frame per graph)
        import bench
        more = [prep(vid.next()) for _ in range(40)]
        r = bench.FrameRunner(test, more[0], "graph")
        r.prime(more[:2])
        for f in more[2:6]:
            r.step(f)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for f in more[6:]:
            r.step(f)
        torch.cuda.synchronize()
        dg = time.perf_counter() - t0
        print("  replayed from a hipGraph: %.0f frames/s (%.1f us per frame)" % (34 / dg, 1e6 * dg / 34))
        r.graph = None
        fresh = [prep(vid.next()) for _ in range(10)]
    # per-layer change ratios: the mean over ten more frames of the walk (untimed: reading a list length is a sync)
    convs = [m for m in test.modules() if type(m) is pycbinfer.CBConv2d]
    counts = [0.0] * len(convs)
    with torch.no_grad():
        # (FRESH frames, forward: walking back over the stored ones under-states the change behind thresholded layers
        #  -- a pixel that was refreshed one frame ago jumps less than one that has been stale for ten)
        for f in fresh:
            test(f)
            for i, m in enumerate(convs):
                counts[i] += m.lastChangeIndexes().numel() / 10.0
    rs, flops = [], []
    for m, n in zip(convs, counts):
        if m.lastChangeIndexes() is not None:
            ci = m.lastChangeIndexes()
            K, C, kH, kW = m.weight.shape
            r = n / float(ci.size[0] * ci.size[1])
            rs.append(r)
            flops.append(2.0 * n * C * kH * kW * K)
            print("  conv %3d->%3d k%d @%dx%d: %5.1f %% of the pixels recomputed, %.1f MFLOP, threshold %.4g%s"
                  % (C, K, kH, ci.size[0], ci.size[1], 100 * r, 1e-6 * flops[-1], float(m.threshold),
                     (", " + m._plan['fn'].__name__) if getattr(m, '_plan', None) and m._plan.get('fn') is not None else ""))
    nf = sum(1 for m in convs if (m._work or {}).get('hsplit') and m._work['hsplit']['layer'][0].detect == 0)
    print("%d of %d layers had their change detection done by their producer's launch" % (nf, len(convs)))
    print("mean post-dilation ratio over %d layers: %.1f %%; recomputed work %.2f GFLOP per frame of %.1f dense"
          % (len(rs), 100 * sum(rs) / max(1, len(rs)), 1e-9 * sum(flops), 1e-9 * workloads.openPoseDenseOps(2, H, W)))


if __name__ == "__main__":
    main()
