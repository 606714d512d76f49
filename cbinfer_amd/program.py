"""A frame of a converted network as a recorded LAUNCH PROGRAM (execution level, results unchanged).

A change-based network's frame is a handful of small dependent launches (six for the scene-labeling net); issued module by
module from Python the host spends ~10 us per launch on bookkeeping that is the same every frame -- chain tags, tokens, plan
checks --, which is as long as some of the kernels run.  A replayed hipGraph removes the host but costs the GPU ~1 us per kernel
node and ~9 us per graph launch on this stack (replay loses to eager launches).  `FrameProgram` is the third
form: it RECORDS the library calls one eager frame makes (function + argument objects, cbinfer_amd/_lib.py) and replays them
as plain stream launches -- the same kernels, the same order, the same arguments, with the frame's input pointer patched in --
from a loop of a few ctypes calls.  The contract is a captured graph's: the network's buffers, flags and thresholds must not
change between record and replay (the module-level decisions of the recorded frame are frozen), module bookkeeping
(lastChangeIndexes, chain tags) is not advanced by a replay, and the network must consist of library calls only -- CBConv2d,
lazily folded CBPoolMax2d, CBTail1x1, ChannelConcat -- since a torch operator in between cannot be recorded (the recording
watches the operators that run, through a TorchDispatchMode, and refuses a frame that ran anything but views).  Re-record (`record(frame)`) after anything changed.  Outputs and states are those of the eager network, bit for
bit (tests/test_gpu_modules.py, __graft_entry__.smoke())."""
import ctypes

import torch
import torch.nn as nn
from torch.utils._python_dispatch import TorchDispatchMode

from . import _lib
from ._lib import CBinferError, check
from .conv2d import CBConv2d, CBPoolMax2d, CBTail1x1


# torch operators a recorded frame may run: they launch nothing and allocate nothing (views and aliases of existing tensors)
_HARMLESS = ('detach', 'alias', 'view', '_unsafe_view', 'reshape', 'slice', 'select', 'unsqueeze', 'squeeze', 'expand',
             'as_strided', 'transpose', 't', 'permute', 'contiguous', '_reshape_alias', 'lift_fresh', 'is_same_size',
             'sym_size', 'sym_stride', 'sym_numel', 'size', 'stride', 'numel', 'dim')


class _OperatorWatch(TorchDispatchMode):
    """Notes every torch operator that runs while a frame is recorded and is not a pure view: such an operator launches or
    allocates, and a replay would not repeat it."""

    def __init__(self):
        super(_OperatorWatch, self).__init__()
        self.seen = []

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = getattr(getattr(func, 'overloadpacket', func), '__name__', str(func))
        if name not in _HARMLESS:
            self.seen.append(name)
        return func(*args, **(kwargs or {}))


class FrameProgram(object):
    def __init__(self, model):
        """model: any callable of one frame whose forward, in the steady state, consists of library calls only -- an
        nn.Sequential of CBConv2d / lazily folded CBPoolMax2d / CBTail1x1, or e.g. a converted OpenPose network with
        change-based pools folded into the detections and pycbinfer.ChannelConcat between its stages.  Whether that holds
        is checked when a frame is recorded."""
        mods = list(model.modules()) if isinstance(model, nn.Module) else []
        if any(type(m) is CBConv2d and (m.syncIndexes or m.gatherComputationStats or m.saveChangeMap or
                                        (m.finegrained and not m.fgInPlace) or
                                        not (m.feedbackLoop or m.copyInput or m.finegrained)) for m in mods):
            raise CBinferError("FrameProgram: every CBConv2d must run a sync-free frame on buffers of its own")
        self.model = model
        self.calls, self.patches, self.out, self.stream = None, None, None, None

    def record(self, frame):
        """Run one eager frame (a steady-state one: the network has seen a few frames) and keep its launch sequence."""
        if not (torch.is_tensor(frame) and frame.is_cuda and frame.is_contiguous()):
            raise CBinferError("FrameProgram: the frame must be a contiguous device tensor")
        calls = []
        watch = _OperatorWatch()
        _lib._RECORDING[0] = calls
        try:
            with torch.no_grad(), watch:
                out = self.model(frame)
        finally:
            _lib._RECORDING[0] = None
        if not calls:
            raise CBinferError("FrameProgram: the frame made no library call")
        if watch.seen:
            raise CBinferError("FrameProgram: the frame ran torch operators a replay would not repeat (%s): the network "
                               "must consist of library calls only -- CBConv2d, CBPoolMax2d folded into the detections "
                               "(pycbinfer.insertCBPooling + fusePoolingIntoDetection), CBTail1x1, pycbinfer.ChannelConcat "
                               "-- and be in its steady state (a first frame allocates)"
                               % ", ".join(sorted(set(watch.seen))))
        # where the frame's address went: integer arguments, and pointer fields of argument structures
        addr, patches = frame.data_ptr(), []
        for ci, (fn, args) in enumerate(calls):
            for ai, a in enumerate(args):
                if isinstance(a, int) and not isinstance(a, bool) and a == addr:
                    patches.append(('arg', ci, ai))
                elif isinstance(a, (ctypes.Array, ctypes.Structure)) or hasattr(a, 'contents'):
                    objs = list(a) if isinstance(a, ctypes.Array) else [a.contents if hasattr(a, 'contents') else a]
                    for o in objs:
                        for name, ctype in getattr(o, '_fields_', ()):
                            if ctype is ctypes.c_void_p and getattr(o, name) == addr:
                                patches.append(('field', o, name))
        if not patches:
            raise CBinferError("FrameProgram: the frame's address does not appear in the recorded calls")
        self.calls = [(fn.raw, list(args)) for fn, args in calls]
        self.patches, self.out = patches, out
        self.stream = _lib.raw_stream(frame.device.index)
        self.shape, self.dtype, self.device = tuple(frame.shape), frame.dtype, frame.device
        return out

    def __call__(self, frame):
        if self.calls is None:
            return self.record(frame)
        if (tuple(frame.shape) != self.shape or frame.dtype != self.dtype or frame.device != self.device or
                not frame.is_contiguous() or _lib.raw_stream(frame.device.index) != self.stream):
            raise CBinferError("FrameProgram: frame shape, dtype, device or stream differ from the recorded frame's")
        addr = frame.data_ptr()
        for p in self.patches:
            if p[0] == 'arg':
                self.calls[p[1]][1][p[2]] = addr
            else:
                setattr(p[1], p[2], addr)
        for raw, args in self.calls:
            st = raw(*args)
            if st != 0:
                check(st)
        return self.out
