"""Independent branches of a network issued as ONE launch per layer pair.

The two branches of an OpenPose stage (poseDetection/openPose/PoseModel.py:122-137: `model{t}_1` and `model{t}_2`, both fed
the same tensor, independent until the concatenation) are chains of layers of the same geometry -- 128 -> 128 7x7, ... -- with
different weights.  Run one after the other (the reference; `OpenPoseModel.forward`) they are two dependent chains of small
launches; `BranchGroup` walks them in lockstep and hands each PAIR of fp16 CBConv2d layers to the library as one
`cbinfer_hsplit_forward_group` call: the work items of both layers in one persistent grid (own weights, states, masks, lists
and consumers each), one reduce launch for both.  Everything a module's own forward does around the library call -- chain
bookkeeping, the detection folded into the producer, the published change count -- is done per module as before
(CBConv2d._prepare_hsplit / _finish_hsplit), so the modules' states, lists and outputs are exactly those of the separate
calls (tests/test_gpu_hgroup.py: bit-identical), and any pair that is not two fp16 split-state layers with valid call plans
and one geometry simply runs module by module.
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib
from ._lib import C, check
from .conv2d import CBConv2d


def _kp(K):
    bm = 64 if K <= 64 else 128
    return (K + bm - 1) // bm * bm, bm


class BranchGroup(nn.Module):
    """nn.Module holding the branches (nn.Sequential each); forward(x) -> [branch(x) for branch in branches]."""

    def __init__(self, branches):
        super(BranchGroup, self).__init__()
        self.branches = nn.ModuleList(branches)
        self.__dict__['_pairs'] = {}

    def _pair_state(self, mods, geom):
        """Workspace and argument array of one layer pair (slabs of a deep contraction for both layers)."""
        key = tuple(id(m) for m in mods)
        st = self.__dict__['_pairs'].get(key)
        if st is None or st['geom'] != geom:
            Cin, H, W, kH, kW = geom[3:8]
            K = max(m.weight.size(0) for m in mods)
            nbytes = C.cbinfer_hsplit_group_workspace_bytes(len(mods), Cin, H, W, K, kH, kW)
            dev = mods[0].weight.device
            st = dict(geom=geom, layers=(_lib.HalfLayer * len(mods))(),
                      ws=torch.zeros(nbytes, dtype=torch.uint8, device=dev) if nbytes > 0 else None)
            self.__dict__['_pairs'][key] = st
        return st

    def _grouped(self, mods, inputs):
        """One library call for the layers `mods` (same position of every branch) if every one of them has a valid fp16
        split-state call plan for its input and they share one geometry; returns their outputs, or None (nothing done)."""
        if len(mods) < 2 or len(mods) > _lib.HGROUP_MAX:
            return None
        srcs, geom = [], None
        for m, x in zip(mods, inputs):
            plan = m.__dict__.get('_plan') if type(m) is CBConv2d else None
            if plan is None or not plan.get('hsplit') or m.finegrained:
                return None
            if m._forward_pre_hooks or m._forward_hooks:      # (somebody watches this module's forward: it is called)
                return None
            src = m._plan_source(x)
            if src is None:
                return None
            a = plan['args']      # (layers, 1, pooled, pH, pW, C, H, W, kH, kW, feedback, ws, stream)
            g = tuple(a[2:11]) + (a[12],) + _kp(m.weight.size(0))
            if geom is None:
                geom = g
            elif g != geom:
                return None
            srcs.append(src)
        st = self._pair_state(mods, geom)
        toks = []
        for q, (m, x, src) in enumerate(zip(mods, inputs, srcs)):
            m._note_upstream(x)
            toks.append(m._prepare_hsplit(m._plan, src, m._buffers))
            ctypes.memmove(ctypes.byref(st['layers'][q]), ctypes.byref(m._plan['layer']), ctypes.sizeof(_lib.HalfLayer))
            st['layers'][q].upstreamCount = None      # (one layer's idle frame must not end the launch for the others)
        a = mods[0]._plan['args']
        check(C.cbinfer_hsplit_forward_group(st['layers'], len(mods), a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10],
                                             st['ws'].data_ptr() if st['ws'] is not None else None, a[12]))
        return [m._finish_hsplit(m._plan, tk, m._buffers) for m, tk in zip(mods, toks)]

    def forward(self, x):
        kids = [list(b.children()) for b in self.branches]
        n = len(kids[0])
        if any(len(k) != n for k in kids):
            return [b(x) for b in self.branches]
        xs = [x] * len(kids)
        for pos in range(n):
            mods = [k[pos] for k in kids]
            out = self._grouped(mods, xs)
            xs = out if out is not None else [m(v) for m, v in zip(mods, xs)]
        return xs
