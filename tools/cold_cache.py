#!/usr/bin/env python3
"""Diagnostic: the fused contraction timed one launch at a time (events around a single launch, so the
~10 us dispatch cost is included) with hot caches, after a 64 MB memset (L2 flushed), after a 512 MB memset
(Infinity Cache flushed) and after another kernel rewrote its input.  Finding (round 1): cold caches do
not slow it down -- the difference between stand-alone and in-frame durations is per-launch cost."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbinfer_amd import conv2d_cg as cg
from tools.bench_conv import LAYERS, blocks_list
def timed(fn, pre, reps=40):
    tot = 0.0
    for _ in range(reps):
        pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / reps * 1e3
gen = torch.Generator().manual_seed(0)
big = torch.empty(64 << 20, dtype=torch.uint8, device='cuda')
huge = torch.empty(512 << 20, dtype=torch.uint8, device='cuda')
for li, ratio in ((1, 0.22), (2, 0.36), (0, 0.13)):
    C, K, k, H, W = LAYERS[li]
    x = torch.randn(1, C, H, W, device="cuda"); w = torch.randn(K, C, k, k, device="cuda") / (C*k*k)**0.5
    b = torch.randn(K, device="cuda"); out = torch.zeros(1, K, H, W, device="cuda"); wp = cg.prepWeights(w, H, W)
    idx = blocks_list(H, W, ratio, 8, gen)
    fn = lambda: cg.convChanged(x, idx, w, b, out, withReLU=True, weightsPrepared=wp)
    hot = timed(fn, lambda: None)
    l2cold = timed(fn, lambda: big.zero_())
    allcold = timed(fn, lambda: huge.zero_())
    # only the gather source rewritten (as the detection does) / only the weights cold is not separable here
    xs = timed(fn, lambda: x.add_(0.0))
    print("layer %d N=%d: hot %.1f us | L2 flushed (64 MB memset) %.1f | MALL flushed (512 MB) %.1f | input rewritten by another kernel %.1f" % (li, idx.numel(), hot, l2cold, allcold, xs))
