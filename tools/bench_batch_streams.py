#!/usr/bin/env python3
"""Throughput of G SequenceBatch groups of B sequences each, every group on its own stream (eager launches from one
host thread, round-robin over the groups): does overlapping one group's latency-bound launches with another
group's beat one batch of G*B?   usage: bench_batch_streams.py G B"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, pycbinfer

G, B = int(sys.argv[1]), int(sys.argv[2])
groups = []
for g in range(G):
    _, net = bench.build_bench_model()
    groups.append(dict(batch=pycbinfer.SequenceBatch(net, B), stream=torch.cuda.Stream(),
                       walk=[[v.frame] + [v.next() for _ in range(15)]
                             for v in [bench.bench_video(1234 + 7919 * (g * B + q)) for q in range(B)]]))
torch.cuda.synchronize()
with torch.no_grad():
    for i in range(4):
        for gr in groups:
            with torch.cuda.stream(gr['stream']):
                gr['batch']([w[i] for w in gr['walk']])
    torch.cuda.synchronize()
    n = 0
    t0 = time.perf_counter()
    for rep in range(40):
        for i in list(range(4, 16)) + list(range(14, 4, -1)):
            for gr in groups:
                with torch.cuda.stream(gr['stream']):
                    gr['batch']([w[i] for w in gr['walk']])
            n += 1
    torch.cuda.synchronize()
    fps = G * B * n / (time.perf_counter() - t0)
print("%d groups x %d sequences: %.0f frames/s (%.1f us per step of all)" % (G, B, fps, 1e6 * G * B / fps))
# the same with every group's step captured as a hipGraph (static input buffers, replay on the group's stream)
with torch.no_grad():
    for gr in groups:
        gr['static'] = [w[4].clone() for w in gr['walk']]
        torch.cuda.synchronize()
        with torch.cuda.stream(gr['stream']):
            gr['batch'](gr['static'])
        torch.cuda.synchronize()
        gr['graph'] = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr['graph'], stream=gr['stream']):
            gr['batch'](gr['static'])
    torch.cuda.synchronize()
    n = 0
    t0 = time.perf_counter()
    for rep in range(40):
        for i in list(range(4, 16)) + list(range(14, 4, -1)):
            for gr in groups:
                with torch.cuda.stream(gr['stream']):
                    for q in range(B):
                        gr['static'][q].copy_(gr['walk'][q][i])
                    gr['graph'].replay()
            n += 1
    torch.cuda.synchronize()
    fps = G * B * n / (time.perf_counter() - t0)
print("   ... as graphs: %.0f frames/s" % fps)
