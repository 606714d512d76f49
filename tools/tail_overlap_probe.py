#!/usr/bin/env python3
"""Feasibility probe (not a measurement of the product path): the 1x1 tail of frame t on a side stream while the first
launches of frame t+1 run -- the tail as a launch of its own (CBINFER_NO_TAILFOLD=1), four frames per captured hipGraph so
that the host is out of the picture, forked / joined inside the graph -- against the same four-frame graph on one stream."""
import os
import sys
import time

os.environ["CBINFER_NO_TAILFOLD"] = "1"
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

K = 4


def main():
    vid = bench.bench_video(1234)
    allf = vid.frames(2 + 64)
    walk = allf[2:]
    for form in ("serial", "overlap", "serial", "overlap"):
        _, net = bench.build_bench_model()
        mods = list(net.children())
        head, tail = mods[:-1], mods[-1]
        cap, side = torch.cuda.Stream(), torch.cuda.Stream()
        bufs = [walk[0].clone() for _ in range(K)]
        with torch.no_grad():
            for f in allf[:2]:
                net(f)

            def frames_on(cap_stream):
                ev_prev = None
                for b in bufs:
                    x = b
                    for m in head[:-1]:
                        x = m(x)
                    if form == "overlap" and ev_prev is not None:
                        cap_stream.wait_event(ev_prev)      # (the previous frame's tail has read the last layer's list)
                    x = head[-1](x)
                    if form == "serial":
                        tail(x)
                        continue
                    ev1 = torch.cuda.Event()
                    ev1.record(cap_stream)
                    side.wait_event(ev1)
                    with torch.cuda.stream(side):
                        tail(x)
                    ev_prev = torch.cuda.Event()
                    ev_prev.record(side)
                if form == "overlap":
                    cap_stream.wait_stream(side)
            cap.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(cap):
                frames_on(cap)      # (plans for this stream)
                frames_on(cap)
            torch.cuda.current_stream().wait_stream(cap)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=cap):
                frames_on(cap)

            def step(i):
                for k, b in enumerate(bufs):
                    b.copy_(walk[bench.pingpong(K * i + k, len(walk))])
                g.replay()
            for i in range(20):
                step(i)
            torch.cuda.synchronize()
            n = 500
            t0 = time.perf_counter()
            for i in range(20, 20 + n):
                step(i)
            torch.cuda.synchronize()
            print("%-8s %.0f frames/s (%d frames per graph)" % (form, K * n / (time.perf_counter() - t0), K), flush=True)


if __name__ == "__main__":
    main()
