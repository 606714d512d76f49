#!/usr/bin/env python3
"""Diagnostic (round 6): the 16 -> 64 contraction of the scene-labeling geometry (160 x 240, 7x7) at ~10 % change, in pixel
order + the consumer's separate pooled detection against window order with that detection folded in
(cbinfer_split_conv_next): microseconds per launch by events on the null stream, the same frames for both."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from cbinfer_amd import _lib as lib  # noqa: E402
import test_gpu_split as T  # noqa: E402

T.ARITH = "x3"
C_ = lib.C
H, W, K, k2, Cin, K2 = 160, 240, 64, 7, 16, 256
H2, W2 = H // 2, W // 2
rng = np.random.default_rng(5)
w1 = (rng.standard_normal((K, Cin, 7, 7)) / np.sqrt(Cin * 49)).astype(np.float32)
b1 = rng.standard_normal(K).astype(np.float32)
w2 = (rng.standard_normal((K2, K, k2, k2)) / np.sqrt(K * k2 * k2)).astype(np.float32)
b2 = rng.standard_normal(K2).astype(np.float32)
Pa, Pb = T.Layer(lib, w1, b1, H, W), T.Layer(lib, w1, b1, H, W)
Ca, Cb = T.Layer(lib, w2, b2, H2, W2, pooled=True), T.Layer(lib, w2, b2, H2, W2, pooled=True)
nd = lib.NextDetect()
nd.state, nd.splitState, nd.frameMasks = Cb.state[0].data_ptr(), Cb.S[0].data_ptr(), Cb.masks[0].data_ptr()
nd.rangeFlag, nd.H, nd.W, nd.kH, nd.kW, nd.threshold, nd.arith = Cb.flag.data_ptr(), H2, W2, k2, k2, 0.05, 1
frames = [T.dev(x) for x in T.block_video(rng, Cin, H, W, 40, float(os.environ.get('WIN_FRAC', '0.07')),
                                          blk=int(os.environ.get('WIN_BLK', '32')))]
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
ta, tb, td = [], [], []
for t, xd in enumerate(frames):
    for P in (Pa, Pb):
        P.seqs[0].input, P.seqs[0].producerMask = xd.data_ptr(), None
        lib.check(C_.cbinfer_split_detect(P.seqs, 1, 8, 0, 0, Cin, H, W, 7, 7, 0.1, None))
    ev[0].record()
    lib.check(C_.cbinfer_split_conv(Pa.seqs, 1, Pa.wp.data_ptr(), Pa.b.data_ptr(), Cin, H, W, K, 7, 7, 0.0, 1, None, 0, None))
    ev[1].record()
    Ca.seqs[0].input, Ca.seqs[0].producerMask = Pa.out[0].data_ptr(), None
    lib.check(C_.cbinfer_split_detect(Ca.seqs, 1, 1 | 8, H, W, K, H2, W2, k2, k2, 0.05, None))
    ev[2].record()
    lib.check(C_.cbinfer_split_conv_next(Pb.seqs, 1, Pb.wp.data_ptr(), Pb.b.data_ptr(), Cin, H, W, K, 7, 7, 0.0, 1, None,
                                         ctypes.pointer(nd), None))
    ev[3].record()
    for Cx in (Ca, Cb):      # (the consumers' contractions consume and zero the masks)
        Cx.seqs[0].input = (Pa if Cx is Ca else Pb).out[0].data_ptr()
        lib.check(C_.cbinfer_split_conv(Cx.seqs, 1, Cx.wp.data_ptr(), Cx.b.data_ptr(), K, H2, W2, K2, k2, k2, 0.0, 1,
                                        Cx.ws.data_ptr(), 0, None))
    torch.cuda.synchronize()
    if t >= 8:
        ta.append(ev[0].elapsed_time(ev[1]) * 1e3), td.append(ev[1].elapsed_time(ev[2]) * 1e3)
        tb.append(ev[2].elapsed_time(ev[3]) * 1e3)
n = int(Pa.cnt[0].item())
print("changed pixels of the last frame: %d of %d; pixel order %.1f us + separate detection %.1f us = %.1f; window order "
      "with the detection %.1f us (medians of %d frames, event to event)" % (
          n, H * W, np.median(ta), np.median(td), np.median(ta) + np.median(td), np.median(tb), len(ta)))
