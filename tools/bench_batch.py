#!/usr/bin/env python3
"""Throughput of pycbinfer.SequenceBatch on the bench workload: S sequences per launch, eager and as one graph.
usage: bench_batch.py [S ...]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, pycbinfer

for S in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    _, net = bench.build_bench_model()
    batch = pycbinfer.SequenceBatch(net, S)
    vids = [bench.bench_video(1234 + 7919 * q) for q in range(S)]
    walk = [[v.frame] + [v.next() for _ in range(15)] for v in vids]
    with torch.no_grad():
        for i in range(4):
            batch([w[i] for w in walk])
        torch.cuda.synchronize()
        res = {}
        # eager
        n = 0
        t0 = time.perf_counter()
        for rep in range(40):
            for i in list(range(4, 16)) + list(range(14, 4, -1)):
                batch([w[i] for w in walk]); n += 1
        torch.cuda.synchronize()
        res['eager'] = S * n / (time.perf_counter() - t0)
        # graph
        static = [w[4].clone() for w in walk]
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            batch(static)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            batch(static)
        n = 0
        t0 = time.perf_counter()
        for rep in range(40):
            for i in list(range(4, 16)) + list(range(14, 4, -1)):
                for q in range(S):
                    static[q].copy_(walk[q][i])
                g.replay(); n += 1
        torch.cuda.synchronize()
        res['graph'] = S * n / (time.perf_counter() - t0)
    print("S=%d: eager %.0f frames/s (%.1f us per step), graph %.0f frames/s; changed px %s" % (
        S, res['eager'], 1e6 * S / res['eager'], res['graph'], batch.changeCounts(0)), flush=True)
