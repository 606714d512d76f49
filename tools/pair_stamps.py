#!/usr/bin/env python3
"""Per-phase timing of the row-pair kernel's workgroups (cb_rowpair.hip) inside the bench frame, from in-kernel time
stamps (library built with make EXTRA=-DCBP_STAMP; tools/pair_stamp_run.sh).  Phases of a workgroup's first non-empty
unit: 0 entry, 1 mask words known, 2 requests issued, 3 patch staged, 4 k-loop + stores done, 5 pooled detection
decided, 6 unit done, 7 kernel exit."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cbinfer_amd._lib import LIB_PATH  # noqa: E402

raw = ctypes.CDLL(LIB_PATH)


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    _, net = bench.build_bench_model()
    if S > 1:      # a SequenceBatch step of S sequences (usage: pair_stamps.py S)
        import pycbinfer
        sb = pycbinfer.SequenceBatch(net, S)
        walk = [bench.bench_video(1234 + 7919 * q).frames(14) for q in range(S)]
        with torch.no_grad():
            for i in range(12):
                sb([w[i] for w in walk])
            torch.cuda.synchronize()
            raw.cbinfer_debug_pair_stamps(None, 0, 1)
            torch.cuda.synchronize()
            sb([w[12] for w in walk])
            torch.cuda.synchronize()
    else:
        frames = bench.bench_video(1234).frames(14)
        with torch.no_grad():
            for f in frames[:12]:
                net(f)
            torch.cuda.synchronize()
            raw.cbinfer_debug_pair_stamps(None, 0, 1)
            torch.cuda.synchronize()
            net(frames[12])
            torch.cuda.synchronize()
    buf = np.zeros(4096 * 8, dtype=np.uint64)
    raw.cbinfer_debug_pair_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(buf.nbytes), 0)
    st = buf.reshape(4096, 8).astype(np.int64)
    ran = st[:, 0] > 0
    act = ran & (st[:, 3] > 0)
    t0 = st[ran, 0].min()
    print("%d workgroups, %d with a unit; first start -> last start %.2f us; last exit %.2f us after the first start"
          % (ran.sum(), act.sum(), (st[ran, 0].max() - t0) / 100.0, (st[ran, 7].max() - t0) / 100.0))
    a = st[act]
    names = ["mask words", "requests", "staging", "k-loop+store", "pool decide", "refresh+OR"]
    for i, nm in enumerate(names):
        ok = (a[:, i + 1] > 0) & (a[:, i] > 0)
        if ok.any():
            d = (a[ok, i + 1] - a[ok, i]) / 100.0
            print("   %-12s median %.2f us   p90 %.2f   max %.2f  (%d)" % (nm, np.median(d), np.percentile(d, 90), d.max(),
                                                                      ok.sum()))
    life = (a[:, 7] - a[:, 0]) / 100.0
    print("   lifetime     median %.2f us   p90 %.2f   max %.2f" % (np.median(life), np.percentile(life, 90), life.max()))
    e = st[ran & ~act]
    if len(e):
        print("   workgroups without work: lifetime median %.2f us, max %.2f" % (
            np.median((e[:, 7] - e[:, 0]) / 100.0), ((e[:, 7] - e[:, 0]) / 100.0).max()))
    starts = (st[ran, 0] - t0) / 100.0
    print("   start times: median %.2f, p90 %.2f, max %.2f us" % (np.median(starts), np.percentile(starts, 90), starts.max()))
    ends = (a[:, 7] - t0) / 100.0
    print("   exit times of working workgroups: median %.2f, p90 %.2f, max %.2f us" % (
        np.median(ends), np.percentile(ends, 90), ends.max()))


if __name__ == "__main__":
    main()
