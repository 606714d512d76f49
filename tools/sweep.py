#!/usr/bin/env python3
"""BASELINE.json configs[2] and [3] on one MI355X: frames/s of the change-based scene-labeling net over
change ratios (coarse-grained experiment 6; fine-grained experiment 7 in its three execution forms: the
default fused frame, the in-place fused frame and the reference-structured atomic scatter) and of the
OpenPose T=2 net in fp16 (cg_half path), each next to the dense
network timed the same way.  Prints markdown tables."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import pycbinfer  # noqa: E402
from cbinfer_amd import workloads  # noqa: E402


def measure(model, frames, steps, warm, mode):
    runner = bench.FrameRunner(model, frames[0], mode)
    runner.prime(frames[:2])
    for f in frames[2:2 + warm]:
        runner.step(f)
    seq = frames[2 + warm:2 + warm + steps]
    dt = bench.timed_loop(runner, seq, len(seq), lambda: None)
    return len(seq) / dt


def layer_ratios(model):
    out = []
    for m in model.modules():
        if type(m) is pycbinfer.CBConv2d and m.lastChangeIndexes() is not None:
            ci = m.lastChangeIndexes()
            out.append(ci.numel() / float(ci.size[0] * ci.size[1]))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--skip-fg", action="store_true")
    ap.add_argument("--skip-pose", action="store_true")
    ap.add_argument("--skip-sweep", action="store_true", help="only the OpenPose part")
    args = ap.parse_args()
    n = 2 + args.warmup + args.steps

    print("## scene labeling 480x320 fp32, 16x16 re-drawn blocks (config 3 sweep); arithmetic of the split-state "
          "kernels: %s\n" % bench.ARITH_TEXT[bench.split_arith()])
    print("| change | dense f/s | CG exp6 f/s | speed-up | post-dilation ratio per CB layer | "
          "FG exp7 f/s (default: fresh tensors, eager) | FG exp7 in-place f/s (best of graph/eager) | "
          "FG in-place speed-up | FG in-place + CBPoolMax2d f/s | FG exp7 atomic scatter f/s (eager) |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    if not args.skip_sweep:      # throw-away measurement: clocks, allocator and MIOpen find are cold at first
        _b, _t = workloads.sceneLabelingModels(experimentIdx=6, threshold=0.05)
        _f = workloads.SyntheticVideo(H=320, W=480, ratio=0.05, block=16, seed=5).frames(n)
        measure(_b, _f, args.steps, args.warmup, "eager")
        measure(_t, _f, args.steps, args.warmup, "eager")
        del _b, _t, _f
    for ratio in (() if args.skip_sweep else (0.01, 0.02, 0.05, 0.10, 0.20, 0.30, 0.50)):
        vid = workloads.SyntheticVideo(H=320, W=480, ratio=ratio, block=16, seed=7)
        frames = vid.frames(n)
        base, cg = workloads.sceneLabelingModels(experimentIdx=6, threshold=0.05)
        for m in cg.modules():            # the bench's execution options (same results)
            if type(m) is pycbinfer.CBPoolMax2d:
                m.cloneOutput = False
        pycbinfer.fuseTail1x1(cg)
        pycbinfer.fusePoolingIntoDetection(cg)
        # both launch forms for both networks, the better one counts (bench.py --mode auto)
        dense = max(measure(base, frames, args.steps, args.warmup, m) for m in ("graph", "eager"))
        fcg = max(measure(cg, frames, args.steps, args.warmup, m) for m in ("graph", "eager"))
        ratios = ", ".join("%.0f%%" % (100 * r) for r in layer_ratios(cg))
        ffg = ffd = ffa = ffp = float("nan")
        if not args.skip_fg:
            _, fg = workloads.sceneLabelingModels(experimentIdx=7, threshold=0.05)
            ffg = measure(fg, frames, args.steps, args.warmup, "eager")
            _, fd = workloads.sceneLabelingModels(experimentIdx=7, threshold=0.05)
            for m in fd.modules():
                if type(m) is pycbinfer.CBConv2d:
                    m.fgInPlace = True
            pycbinfer.fuseTail1x1(fd)
            ffd = max(measure(fd, frames, args.steps, args.warmup, m) for m in ("graph", "eager"))
            # BASELINE.json configs[2] word for word: fine-grained convs + CBPoolMax2d (the fine-grained head hands on
            # the pixels it touched -- an extension: the reference's forward_fg hands on no indexes, conv2d.py:160-176)
            _, fp = workloads.sceneLabelingModels(experimentIdx=7, threshold=0.05)
            for m in fp.modules():
                if type(m) is pycbinfer.CBConv2d:
                    m.fgInPlace = True
            pycbinfer.insertCBPooling(fp, cloneOutput=False)
            pycbinfer.fuseTail1x1(fp)
            pycbinfer.fusePoolingIntoDetection(fp)      # (the pools in the fine-grained detections: no launch of their own)
            ffp = max(measure(fp, frames, args.steps, args.warmup, m) for m in ("graph", "eager"))
            _, fa = workloads.sceneLabelingModels(experimentIdx=7, threshold=0.05)
            for m in fa.modules():
                if type(m) is pycbinfer.CBConv2d:
                    m.atomicFG = True
            nfg = min(len(frames), 2 + 2 + 12)
            ffa = measure(fa, frames[:nfg], 12, 2, "eager")
        print("| %.0f%% | %.0f | %.0f | %.2fx | %s | %.0f | %.0f | %.2fx | %.0f | %.1f |" %
              (100 * vid.ratio, dense, fcg, fcg / dense, ratios, ffg, ffd, ffd / dense, ffp, ffa), flush=True)

    if not args.skip_pose:
        print("\n## OpenPose T=2 368x654 fp16, coarse-grained (config 4), 10 % of the input re-drawn per frame in 16x16 "
              "blocks -- LIVE network (bench.openpose_config: variance-preserving random weights, per-layer thresholds "
              "calibrated to ~10 % post-dilation change in the running network, fresh consecutive frames)\n")

        def pmeasure(model, frames, mode, steps=30, warm=5):
            return measure(model, frames, steps, warm, mode)
        c = bench.openpose_config(args, pmeasure)
        print("| dense f/s | CB f/s (%s) | speed-up | every layer scanning its input | feedback mode (own thresholds; mean "
              "ratio) | + change-based pools folded into the detections | mean / min / max post-dilation ratio over 36 "
              "layers | recomputed GFLOP per frame (dense %.1f) |" % (c["cb_launch"], c["dense_ops_per_frame"] / 1e9))
        print("|---|---|---|---|---|---|---|---|")
        print("| %.1f | %.1f | %.2fx | %.1f (%.2fx) | %.1f (%.2fx; %.0f %%) | %.1f (%.2fx) | %.1f / %.1f / %.1f %% | %.1f |"
              % (c["dense_fps"], c["cb_fps"], c["speedup"], c["cb_unchained_fps"], c["unchained_speedup"],
                 c["cb_feedback_mode_fps"], c["feedback_speedup"], 100 * c["feedback_mode_mean_ratio"],
                 c["cb_with_change_based_pools_fps"], c["change_based_pools_speedup"],
                 100 * c["mean_post_dilation_ratio"], 100 * c["min_ratio"], 100 * c["max_ratio"],
                 c["recomputed_gflop_per_frame"]))
        print("\nper layer (ratio, threshold, library call):")
        for r in c["per_layer"]:
            print("  %-22s %5.1f %%  th %.4g  %s" % (r["layer"], 100 * r["ratio"], r["threshold"], r["path"]))
        rf = c.get("roofline") or {}
        if "contractions" in rf:
            print("\ncontraction launches: %.1f GFLOP in %.0f us = %.1f TFLOP/s = %.3f of the dense f16 MFMA peak; "
                  "detections: %.0f MB in %.0f us = %.0f GB/s = %.2f of HBM"
                  % (rf["contractions"]["flops_per_frame"] / 1e9, rf["contractions"]["us_per_frame"],
                     rf["contractions"]["achieved"], rf["contractions"]["frac"], rf["detections"]["bytes_per_frame"] / 1e6,
                     rf["detections"]["us_per_frame"], rf["detections"]["achieved"], rf["detections"]["frac"]))

if __name__ == "__main__":
    main()
