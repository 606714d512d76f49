#!/usr/bin/env python3
"""Diagnostic: per-workgroup phase stamps of the split-state contraction (needs the -DCBS_STAMP build:
tools/split_stamp_run.sh).  usage: split_stamps.py [nSeq] [forceSplit]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbinfer_amd import _lib  # noqa: E402
import tools.bench_split as bs  # noqa: E402

raw = ctypes.CDLL(_lib.LIB_PATH)
NAMES = ['entry', 'list lengths known', 'item set up', 'ring primed', 'stage loop done', 'epilogue done']
FINE = {8: 'mask words in LDS', 9: 'pixel lookup done (thread 0)', 10: 'lookup barrier passed', 11: 'DMA offsets ready',
        12: 'window order: outputs stored', 13: 'window order: verdicts known', 14: 'window order: states refreshed'}


def report(clear=True):
    torch.cuda.synchronize()
    buf = np.zeros(4 * 2048 * 16, dtype=np.uint64)
    raw.cbinfer_debug_split_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(buf.nbytes), 0)
    for cfg, name in ((0, "64-row tile"), (1, "128-row tile")):
        st = buf.reshape(4, 2048, 16)[cfg].astype(np.int64)
        st = st[st[:, 0] > 0]
        if len(st):
            print(" ", name)
            report_one(st)
    st = buf.reshape(4, 2048, 16)[2].astype(np.int64)
    st = st[st[:, 0] > 0]
    if len(st):
        print("  reduce + tail launch: %d workgroups stamped" % len(st))
        t0 = st[:, 0].min()
        names = ['entry', 'launch info known', 'slab columns arrived', 'X tile written', 'X tile complete (barrier)',
                 'first layer done', 'hidden tile complete (barrier)', 'group done']
        for i, n in enumerate(names):
            v = st[:, i] >= t0
            if v.sum():
                us = (st[v, i] - t0) / 100.0
                print("    %-32s min %7.2f  mean %7.2f  max %7.2f us (%d wgs)" % (n, us.min(), us.mean(), us.max(), v.sum()))
    st = buf.reshape(4, 2048, 16)[3].astype(np.int64)
    st = st[st[:, 0] > 0]
    if len(st):
        print("  pooled detection (the LAST one of the frame): %d workgroups stamped" % len(st))
        t0 = st[:, 0].min()
        names = ['entry', 'producer mask says changed', 'values loaded, ballots merged', 'states refreshed, tile in LDS',
                 'records written', 'mask words ORed (end)']
        # (two launches of the frame write here, the second over the first's first entries: per workgroup, from its
        #  own entry; the launch span from the entries of the last launch = the largest cluster of entry times)
        for i, n in enumerate(names):
            v = st[:, i] >= st[:, 0]
            v &= st[:, i] > 0
            if v.sum() and i:
                us = (st[v, i] - st[v, 0]) / 100.0
                print("    %-32s min %7.2f  mean %7.2f  max %7.2f us after the workgroup's own entry (%d wgs)"
                      % (n, us.min(), us.mean(), us.max(), v.sum()))
        ent = np.sort(st[:, 0])
        late = ent[ent > ent[-1] - 1500]
        print("    entries of the last launch: %d workgroups over %.2f us" % (len(late), (late[-1] - late[0]) / 100.0))
    if clear:
        raw.cbinfer_debug_split_stamps(None, 0, 1)


def report_one(st):
    if len(st) == 0:
        return
    t0 = st[:, 0].min()
    print("  %d workgroups stamped" % len(st))
    for i, n in enumerate(NAMES):
        v = st[:, i] >= t0
        if v.sum() == 0:
            continue
        us = (st[v, i] - t0) / 100.0
        print("  %-20s min %7.2f  mean %7.2f  max %7.2f us (%d wgs)" % (n, us.min(), us.mean(), us.max(), v.sum()))
    for i, n in sorted(FINE.items()):
        v = st[:, i] >= t0
        if v.sum():
            us = (st[v, i] - t0) / 100.0
            print("    %-28s min %7.2f  mean %7.2f  max %7.2f us (%d wgs)" % (n, us.min(), us.mean(), us.max(), v.sum()))
    ok = (st[:, 4] > st[:, 3]) & (st[:, 3] >= t0)
    if ok.sum():
        d = (st[ok, 4] - st[ok, 3]) / 100.0
        cyc = (st[ok, 7] - st[ok, 6]).astype(np.float64)
        tot = (st[ok, 4] - st[ok, 0]) / 100.0
        print("  stage loop of the first item: min %.2f mean %.2f max %.2f us over %d wgs; shader clock ~%.0f MHz"
              % (d.min(), d.mean(), d.max(), ok.sum(), np.median(cyc / np.maximum(tot, 0.01))))


def report_by_xcd():
    """Stage-loop time of the 128-row tile's workgroups by blockIdx % 8 -- for the bench's 64->256 launch that is
    (k-slice, row tile) and the XCD the workgroup runs on."""
    torch.cuda.synchronize()
    buf = np.zeros(4 * 2048 * 16, dtype=np.uint64)
    raw.cbinfer_debug_split_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(buf.nbytes), 0)
    st = buf.reshape(4, 2048, 16)[1].astype(np.int64)
    idx = np.nonzero((st[:, 0] > 0) & (st[:, 4] > st[:, 3]))[0]
    if len(idx) == 0:
        return
    t0 = st[idx, 0].min()
    for x in range(8):
        sel = idx[idx % 8 == x]
        if len(sel):
            loop = (st[sel, 4] - st[sel, 3]) / 100.0
            end = (st[sel, 5] - t0) / 100.0
            print("  blockIdx %% 8 = %d: %3d wgs, loop mean %.2f max %.2f us, done mean %.2f max %.2f us"
                  % (x, len(sel), loop.mean(), loop.max(), end.mean(), end.max()))


_ev = bs.ev


def ev_once(fn, reps=3):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    raw.cbinfer_debug_split_stamps(None, 0, 1)
    fn()
    report()
    return 1.0


bs.ev = ev_once
if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "frame":
        # the stamps of the LAST frame of a bench-like eager run: the kernels as they run inside the frame
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        base, test = bench.build_bench_model()
        frames = bench.bench_video(1234).frames(24)
        with torch.no_grad():
            for f in frames[:-1]:
                test(f)
            torch.cuda.synchronize()
            raw.cbinfer_debug_split_stamps(None, 0, 1)
            test(frames[-1])
        report(clear=False)
        report_by_xcd()
    elif len(sys.argv) > 1 and sys.argv[1] == "batch":
        # the same for a SequenceBatch step of S sequences
        import bench
        import pycbinfer
        S = int(sys.argv[2]) if len(sys.argv) > 2 else 4
        _, net = bench.build_bench_model()
        sb = pycbinfer.SequenceBatch(net, S)
        vids = [bench.bench_video(1234 + 7919 * q) for q in range(S)]
        walk = [[v.frame] + [v.next() for _ in range(23)] for v in vids]
        with torch.no_grad():
            for i in range(23):
                sb([w[i] for w in walk])
            torch.cuda.synchronize()
            raw.cbinfer_debug_split_stamps(None, 0, 1)
            sb([w[23] for w in walk])
        report()
    else:
        bs.main()
