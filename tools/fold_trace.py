#!/usr/bin/env python3
"""Diagnostic: the library calls of every frame of the bench network run eagerly (which frames fold which detection)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cbinfer_amd import _lib  # noqa: E402

base, test = bench.build_bench_model()
frames = bench.bench_video(1234).frames(int(sys.argv[1]) if len(sys.argv) > 1 else 14)
with torch.no_grad():
    for i, f in enumerate(frames):
        rec = []
        _lib._RECORDING[0] = rec
        test(f)
        _lib._RECORDING[0] = None
        print(i, [fn.__name__.replace("cbinfer_", "") for fn, _ in rec])
torch.cuda.synchronize()
