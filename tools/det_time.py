#!/usr/bin/env python3
"""Diagnostic (round 6): the 3 -> 16 layer of the scene-labeling geometry (320 x 480, 7x7) with the 16 -> 64 layer's pooled
detection folded in -- detection launch + row pairs against row pairs with their own detection (+ the state refresh as a
launch of its own): microseconds by events on the null stream, the same frames for both."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from cbinfer_amd import _lib as lib  # noqa: E402
import test_gpu_split as T  # noqa: E402

C_ = lib.C
H, W, C, K, k, k2 = 320, 480, 3, 16, 7, 7
H2, W2 = H // 2, W // 2
rng = np.random.default_rng(5)
w = (rng.standard_normal((K, C, k, k)) / np.sqrt(C * k * k)).astype(np.float32)
b = rng.standard_normal(K).astype(np.float32)
wp = torch.empty(C_.cbinfer_rowconv_prepared_bytes(C, K, k, k), dtype=torch.uint8, device="cuda")
lib.check(C_.cbinfer_rowconv_prep_weights(T.dev(w).data_ptr(), wp.data_ptr(), K, C, k, k, None))
bd = T.dev(b)
words = C_.cbinfer_mask_words(H, W)


class Side(object):
    def __init__(self):
        self.state = torch.zeros((1, C, H, W), device="cuda")
        self.out = torch.zeros((1, K, H, W), device="cuda")
        self.bits = torch.zeros(words, dtype=torch.int64, device="cuda")
        self.ctl = torch.zeros(words, dtype=torch.int32, device="cuda")
        self.copy = torch.zeros(words, dtype=torch.int64, device="cuda")
        self.state2 = torch.zeros((1, K, H2, W2), device="cuda")
        self.S2 = torch.empty(C_.cbinfer_split3_state_bytes(K, H2, W2, k2, k2), dtype=torch.uint8, device="cuda")
        lib.check(C_.cbinfer_split3_state_init(self.S2.data_ptr(), K, H2, W2, k2, k2, None))
        lib.check(C_.cbinfer_split3_state_rebuild(self.state2.data_ptr(), self.S2.data_ptr(), K, H2, W2, k2, k2, None))
        self.mask2 = torch.zeros(C_.cbinfer_frame_mask_bytes(H2, W2) // 8, dtype=torch.int64, device="cuda")
        nd = self.nd = lib.NextDetect()
        nd.state, nd.splitState, nd.frameMasks = self.state2.data_ptr(), self.S2.data_ptr(), self.mask2.data_ptr()
        nd.rangeFlag, nd.H, nd.W, nd.kH, nd.kW, nd.threshold, nd.arith = None, H2, W2, k2, k2, 0.07, 1


a, d = Side(), Side()
frames = [T.dev(x) for x in T.block_video(rng, C, H, W, 40, float(os.environ.get("DET_FRAC", "0.10")), blk=32)]
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
ta, tb, tr = [], [], []
for t, xd in enumerate(frames):
    ev[0].record()
    lib.check(C_.cbinfer_cbconv2d_forward_rowpairs(xd.data_ptr(), a.state.data_ptr(), a.out.data_ptr(), a.bits.data_ptr(),
                                                   a.ctl.data_ptr(), a.copy.data_ptr(), wp.data_ptr(), bd.data_ptr(), C, H, W,
                                                   K, k, k, 0.05, 1, ctypes.pointer(a.nd), None))
    ev[1].record()
    lib.check(C_.cbinfer_conv_rowpairs_detect(xd.data_ptr(), d.state.data_ptr(), d.out.data_ptr(), d.copy.data_ptr(),
                                              wp.data_ptr(), bd.data_ptr(), C, H, W, K, k, k, 0.05, 1, ctypes.pointer(d.nd),
                                              None))
    ev[2].record()
    lib.check(C_.cbinfer_refresh_state(xd.data_ptr(), d.state.data_ptr(), C, H, W, 0.05, None))
    ev[3].record()
    torch.cuda.synchronize()
    a.mask2.zero_(), d.mask2.zero_()
    if t >= 8:
        ta.append(ev[0].elapsed_time(ev[1]) * 1e3), tb.append(ev[1].elapsed_time(ev[2]) * 1e3)
        tr.append(ev[2].elapsed_time(ev[3]) * 1e3)
n = int(torch.count_nonzero(torch.ones(1)).item())
bits = d.copy.cpu().numpy().view(np.uint64)
npx = int(sum(bin(int(v)).count("1") for v in bits))
print("changed pixels of the last frame: %d of %d; detection launch + row pairs %.1f us; row pairs with their own detection "
      "%.1f us (+ the refresh as a launch of its own %.1f us) (medians of %d frames, event to event)" % (
          npx, H * W, np.median(ta), np.median(tb), np.median(tr), len(ta)))
