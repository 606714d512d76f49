#!/usr/bin/env python3
"""Profiling target: N launches of the fused contraction kernel on ONE layer shape / change ratio, so
that rocprofv3 --pmc rows of cb_mfma_* all describe the same work.
usage: pmc_target.py [layer 0..3] [ratio] [launches]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbinfer_amd import conv2d_cg as cg  # noqa: E402
from cbinfer_amd._lib import CB_F32S  # noqa: E402
from tools.bench_conv import LAYERS, blocks_list  # noqa: E402


def main():
    layer = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    ratio = float(sys.argv[2]) if len(sys.argv) > 2 else 0.36
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    C, K, k, H, W = LAYERS[layer]
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(1, C, H, W, device="cuda")
    w = torch.randn(K, C, k, k, device="cuda") / (C * k * k) ** 0.5
    b = torch.randn(K, device="cuda")
    out = torch.zeros(1, K, H, W, device="cuda")
    arith = CB_F32S if os.environ.get('PMC_SPLIT', '1') != '0' else None   # the frame's arithmetic
    wp = cg.prepWeights(w, H, W, arith=arith)
    idx = blocks_list(H, W, ratio, 8, gen)
    for _ in range(reps):
        cg.convChanged(x, idx, w, b, out, withReLU=True, weightsPrepared=wp, arith=arith)
    torch.cuda.synchronize()
    print("layer", LAYERS[layer], "N", idx.numel(), "launches", reps)


if __name__ == "__main__":
    main()
