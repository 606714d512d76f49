#!/usr/bin/env python3
"""How the launch form bounds the frame rate of the bench network: eager launches, one frame per replayed hipGraph,
K frames per graph with the staging copies on a side stream (bench.MultiFrameRunner), and -- as a bound, not a valid
measurement -- K frames per graph without any staging (every replay re-reads the buffers' old frames: no change, idle
frames are cheaper, so the walk is made of two ALTERNATING captured K-frame sequences that really differ)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def rate(runner, frames, n):
    for i in range(40):
        runner.step(frames[bench.pingpong(i, len(frames))])
    if hasattr(runner, "flush"):
        runner.flush()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(40, 40 + n):
        runner.step(frames[bench.pingpong(i, len(frames))])
    if hasattr(runner, "flush"):
        runner.flush()
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0)


def main():
    vid = bench.bench_video(1234)
    allf = vid.frames(2 + 64)
    n = 1920
    # the literal 20-step region of the driver's run (barrier + synchronize on both sides): what the launch form costs
    # when the stream's queue starts empty
    def short(runner, frames, k=20, reps=20):
        best = 0.0
        for rep in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(k):
                runner.step(frames[bench.pingpong(100 + rep * k + i, len(frames))])
            if hasattr(runner, "flush"):
                runner.flush()
            torch.cuda.synchronize()
            best = max(best, k / (time.perf_counter() - t0))
        return best
    for mode in ["eager", "program", "graph", "graph4", "graph16", "eager", "program"]:
        _, net = bench.build_bench_model()
        if mode == "program":
            r = bench.ProgramRunner(net, allf[0])
        elif mode in ("eager", "graph"):
            r = bench.FrameRunner(net, allf[0], mode)
        else:
            r = bench.MultiFrameRunner(net, allf[0], int(mode[5:]))
        r.prime(allf[:2])
        print("%-8s %.0f frames/s; best 20-step region %.0f frames/s" % (mode, rate(r, allf[2:], n), short(r, allf[2:])),
              flush=True)
        del r, net
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
