#!/usr/bin/env python3
"""GPU time of every library call of a SequenceBatch step (HIP events, one call bracketed per pass; the ~5 us an
event pair costs is included).  usage: batch_events.py [S]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, pycbinfer
from cbinfer_amd import batch as B

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
_, net = bench.build_bench_model()
sb = pycbinfer.SequenceBatch(net, S)
vids = [bench.bench_video(1234 + 7919 * q) for q in range(S)]
walk = [[v.frame] + [v.next() for _ in range(39)] for v in vids]
if os.environ.get('STATIC') == '1':      # no change at all: what the launches cost when nothing is to be done
    walk = [[w[0]] * 40 for w in walk]
names = ["cbinfer_change_detection_bits_batched", "cbinfer_conv_changed_rows_batched", "cbinfer_split_forward", "cbinfer_split_forward_tail",
         "cbinfer_tail1x1_batched"]
orig = {n: getattr(B.C, n) for n in names}
state = dict(which=-1, idx=0, pairs=[])


class Proxy(object):
    def __getattr__(self, n):
        f = getattr(orig_C, n)
        if n not in orig:
            return f

        def call(*a):
            mine = state['idx'] == state['which']
            if mine:
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
            r = f(*a)
            if mine:
                e1.record(); state['pairs'].append((e0, e1, n))
            state['idx'] += 1
            return r
        return call


orig_C = B.C
B.C = Proxy()
with torch.no_grad():
    for i in range(4):
        sb([w[i] for w in walk])
    torch.cuda.synchronize()
    ncalls = 4
    tot = 0.0
    for which in list(range(ncalls)) + [-2]:
        state['which'], state['pairs'] = which, []
        frames_t = []
        for i in range(4, 40):
            state['idx'] = 0
            if which == -2:
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
            sb([w[i] for w in walk])
            if which == -2:
                e1.record(); frames_t.append((e0, e1))
        torch.cuda.synchronize()
        if which == -2:
            ts = sorted(1e3 * a.elapsed_time(b) for a, b in frames_t[4:])
            print("%8.1f us  whole step (S=%d)" % (ts[len(ts) // 2], S))
        else:
            ts = sorted(1e3 * a.elapsed_time(b) for a, b, _ in state['pairs'][4:])
            print("%8.1f us  call %d: %s" % (ts[len(ts) // 2], which, state['pairs'][0][2]))
            tot += ts[len(ts) // 2]
    print("sum of calls %.1f us" % tot)
