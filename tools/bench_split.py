#!/usr/bin/env python3
"""Stand-alone timing of the split-state contraction (cbinfer_split_conv) on the scene-labeling L2/L3 shapes with
the change pattern of the bench (blocks, dilated), one or several sequences per launch.  The kernel consumes
its frame mask (two alternating masks + parity), so both halves are refilled before every launch; the refill
is timed alone and subtracted.  CBINFER_SPLIT_DBG ablations need a -DCBS_DBG build (tools/split_dbg_run.sh).
CBINFER_ARITH=f16x2 times the f16-pair form (default: x3, bf16 triples).
usage: bench_split.py [nSeq] [forceSplit]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbinfer_amd import conv2d_cg as cg  # noqa: E402
from cbinfer_amd import _lib  # noqa: E402
from cbinfer_amd._lib import C as lib, check, ptr  # noqa: E402


def ev(fn, reps=60):
    """Mean duration of fn() in microseconds over `reps` back-to-back calls (HIP events on the current stream)."""
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


def main():
    nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    force = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    x3 = os.environ.get("CBINFER_ARITH", "x3") == "x3"
    gen = torch.Generator().manual_seed(0)
    for (C, K, k, H, W, blk, ratio) in [(16, 64, 7, 160, 240, 16, 0.10), (64, 256, 7, 80, 120, 8, 0.10)]:
        w = torch.randn(K, C, k, k, device="cuda") / (C * k * k) ** 0.5
        b = torch.randn(K, device="cuda")
        import math
        if x3:
            scale = 0.0
            wp = torch.empty(lib.cbinfer_split3_prepared_bytes(C, K, k, k), dtype=torch.uint8, device="cuda")
            check(lib.cbinfer_split3_prep_weights(ptr(w), ptr(wp), K, C, k, k, H, W, None))
        else:
            scale = 2.0 ** (13 - math.floor(math.log2(float(w.abs().max()))))
            wp = torch.empty(lib.cbinfer_split_prepared_bytes(C, K, k, k), dtype=torch.uint8, device="cuda")
            check(lib.cbinfer_split_prep_weights(ptr(w), ptr(wp), K, C, k, k, H, W, scale, None))
        words = lib.cbinfer_mask_words(H, W)
        seqs = (_lib.SplitSeq * nseq)()
        keep, fills, Ns = [], [], []
        for q in range(nseq):
            x = torch.randn(1, C, H, W, device="cuda")
            flag = torch.zeros(1, dtype=torch.int32, device="cuda")
            if x3:
                S = torch.empty(lib.cbinfer_split3_state_bytes(C, H, W, k, k), dtype=torch.uint8, device="cuda")
                check(lib.cbinfer_split3_state_init(ptr(S), C, H, W, k, k, None))
                check(lib.cbinfer_split3_state_rebuild(ptr(x), ptr(S), C, H, W, k, k, None))
            else:
                S = torch.empty(lib.cbinfer_split_state_bytes(C, H, W, k, k), dtype=torch.uint8, device="cuda")
                check(lib.cbinfer_split_state_init(ptr(S), C, H, W, k, k, None))
                check(lib.cbinfer_split_state_rebuild(ptr(x), ptr(S), C, H, W, k, k, ptr(flag), None))
            cm = torch.zeros(H, W, dtype=torch.int8)
            gy, gx = H // blk, W // blk
            cells = torch.randperm(gy * gx, generator=gen)[:max(1, int(round(ratio * gy * gx)))]
            for c in cells.tolist():
                y0, x0 = (c // gx) * blk, (c % gx) * blk
                cm[max(0, y0 - 3):y0 + blk + 3, max(0, x0 - 3):x0 + blk + 3] = 1
            idx = cg.changeIndexesExtr(cm.cuda())
            Ns.append(idx.numel())
            # the bit mask of the map, row-padded
            wpr = lib.cbinfer_mask_words_per_row(W)
            bits = torch.zeros(H, wpr * 64, dtype=torch.int64)
            bits[:, :W] = cm.long()
            weights = (torch.ones(64, dtype=torch.int64) << torch.arange(64, dtype=torch.int64))
            mword = (bits.view(H, wpr, 64) * weights).sum(-1).view(-1).cuda()     # (bit 63 wraps to the sign: fine)
            fm = torch.zeros(lib.cbinfer_frame_mask_bytes(H, W) // 8, dtype=torch.int64, device="cuda")
            both = torch.cat([mword, mword])
            out = torch.zeros(1, K, H, W, device="cuda")
            lst = torch.zeros(H * W, dtype=torch.int32, device="cuda")
            cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
            cp = torch.zeros(words, dtype=torch.int64, device="cuda")
            s = seqs[q]
            s.input, s.state, s.splitState, s.frameMasks = ptr(x), ptr(x), ptr(S), ptr(fm)
            s.output, s.idxOut, s.countOut, s.rangeFlag, s.maskCopy = ptr(out), ptr(lst), ptr(cnt), ptr(flag), ptr(cp)
            keep.append((x, S, flag, fm, both, out, lst, cnt, cp))
            fills.append((fm, both, 2 * words))
        ws = torch.zeros(max(lib.cbinfer_split_workspace_bytes(nseq, C, H, W, K, k, k), 8), dtype=torch.uint8, device="cuda")

        def fill():
            for fm, both, n in fills:
                fm[:n].copy_(both)

        def run():
            fill()
            check(lib.cbinfer_split_conv(seqs, nseq, ptr(wp), ptr(b), C, H, W, K, k, k, scale, 1, ptr(ws), force,
                                         None))
        t_fill = ev(fill)
        t = max(ev(run) - t_fill, 1e-6)
        N = sum(Ns)
        print("%s %d->%d k%d @%dx%d  %d seq, N=%d (%.0f%%): %.1f us  = %.1f TFLOP/s f32-equivalent | dbg=%s force=%d"
              % ("x3" if x3 else "f16x2", C, K, k, H, W, nseq, N, 100.0 * N / (H * W * nseq), t,
                 2.0 * N * C * k * k * K / t / 1e6,
                 os.environ.get("CBINFER_SPLIT_DBG", "0"), force), flush=True)


if __name__ == "__main__":
    main()
