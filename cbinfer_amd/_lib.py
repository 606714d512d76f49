"""ctypes binding of libcbinfer_hip.so (the C ABI declared in include/cbinfer_hip.h).

Replaces the reference's cffi ABI-mode dlopen at import time (pycbinfer/conv2d_cg.py:40-50,
conv2d_fg.py:13-32).  There is no fallback: if the HIP library has not been built, importing the
package fails, exactly as the reference fails when its CUDA .so files are missing.
"""
import ctypes
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libcbinfer_hip.so")

CB_F32 = 0
CB_F16 = 1
CB_F32S = 2     # f32 tensors, contraction as bf16x3 split products on the bf16 MFMA

_vp, _i, _l, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float


class SplitSeq(ctypes.Structure):
    """cbSplitSeq of include/cbinfer_hip.h: the buffers of ONE sequence for the split-state entry points."""
    _fields_ = [("input", _vp), ("state", _vp), ("splitState", _vp), ("frameMasks", _vp), ("producerMask", _vp),
                ("output", _vp), ("idxOut", _vp), ("countOut", _vp), ("rangeFlag", _vp), ("maskCopy", _vp),
                ("delta", _vp), ("reluOut", _vp)]


_sp = ctypes.POINTER(SplitSeq)


class TailSeq(ctypes.Structure):
    """cbTailSeq of include/cbinfer_hip.h."""
    _fields_ = [("input", _vp), ("changeList", _vp), ("countDev", _vp), ("output", _vp)]


_tp = ctypes.POINTER(TailSeq)


class SplitTail(ctypes.Structure):
    """cbSplitTail of include/cbinfer_hip.h: the fused 1x1 tail behind a split-state layer."""
    _fields_ = [("w1Prepared", ctypes.c_void_p), ("b1", ctypes.c_void_p), ("w2", ctypes.c_void_p),
                ("b2", ctypes.c_void_p), ("C1", ctypes.c_int), ("C2", ctypes.c_int), ("relu1", ctypes.c_int),
                ("relu2", ctypes.c_int), ("output", ctypes.c_void_p * 8)]


_stp = ctypes.POINTER(SplitTail)


class NextDetect(ctypes.Structure):
    """cbNextDetect of include/cbinfer_hip.h: the next layer's detection state for the row-pair kernel."""
    _fields_ = [("state", _vp), ("splitState", _vp), ("frameMasks", _vp), ("rangeFlag", _vp), ("H", _i), ("W", _i),
                ("kH", _i), ("kW", _i), ("threshold", _f), ("arith", _i)]


_ndp = ctypes.POINTER(NextDetect)


class SideRefresh(ctypes.Structure):
    """cbSideRefresh of include/cbinfer_hip.h: another layer's feedback refresh carried by a contraction launch."""
    _fields_ = [("frame", ctypes.c_void_p), ("state", ctypes.c_void_p), ("C", ctypes.c_int), ("H", ctypes.c_int),
                ("W", ctypes.c_int), ("threshold", ctypes.c_float)]


_srp = ctypes.POINTER(SideRefresh)


class PairSeq(ctypes.Structure):
    """cbPairSeq of include/cbinfer_hip.h."""
    _fields_ = [("state", _vp), ("output", _vp), ("bits", _vp), ("maskCopy", _vp), ("nextState", _vp),
                ("nextSplitState", _vp), ("nextFrameMasks", _vp), ("nextRangeFlag", _vp)]


_psp = ctypes.POINTER(PairSeq)
_vpp = ctypes.POINTER(ctypes.c_void_p)

HGROUP_MAX, HNEXT_MAX = 2, 2      # CBINFER_HGROUP_MAX, CBINFER_HNEXT_MAX


class HalfNext(ctypes.Structure):
    """cbHalfNext of include/cbinfer_hip.h: a CONSUMER of an fp16 layer's output whose detection rides in its launch."""
    _fields_ = [("state", _vp), ("pixelState", _vp), ("frameMasks", _vp), ("kH", _i), ("kW", _i), ("threshold", _f),
                ("reserved", _i)]


class HalfLayer(ctypes.Structure):
    """cbHalfLayer of include/cbinfer_hip.h: one fp16 layer of a cbinfer_hsplit_forward_group call."""
    _fields_ = [("upstreamCount", _vp), ("input", _vp), ("producerMask", _vp), ("state", _vp), ("pixelState", _vp),
                ("frameMasks", _vp), ("output", _vp), ("idxOut", _vp), ("countOut", _vp), ("maskCopy", _vp),
                ("prepared", _vp), ("bias", _vp), ("K", _i), ("threshold", _f), ("relu", _i), ("detect", _i),
                ("nNext", _i), ("reserved", _i), ("next", HalfNext * HNEXT_MAX)]


_hlp = ctypes.POINTER(HalfLayer)

_SIGNATURES = {
    # name: (restype, [argtypes])
    "cbinfer_abi_version": (_i, []),
    "cbinfer_status_string": (ctypes.c_char_p, [_i]),
    "cbinfer_mask_words_per_row": (_i, [_i]),
    "cbinfer_mask_words": (_l, [_i, _i]),
    "cbinfer_weights_kpad": (_i, [_i]),
    "cbinfer_weights_ckkpad": (_i, [_i, _i]),
    "cbinfer_change_detection": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _i, _vp]),
    "cbinfer_change_detection_bits": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _i, _vp]),
    "cbinfer_change_propagation": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "cbinfer_change_indexes_extr": (_i, [_vp, _l, _vp, _vp, _vp, _vp]),
    "cbinfer_compact_bits": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "cbinfer_gen_x_matrix": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "cbinfer_prepared_weights_bytes": (_l, [_i, _i, _i, _i, _i]),
    "cbinfer_prep_weights": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "cbinfer_conv_workspace_bytes": (_l, []),
    "cbinfer_matrix_mult": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp]),
    "cbinfer_update_output": (_i, [_vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _vp]),
    "cbinfer_conv_changed": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i,
                                  _vp, _l, _vp, _i, _vp]),
    "cbinfer_cbconv2d_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i,
                                      _i, _f, _i, _i, _i, _i, _i, _vp, _i, _i, _vp]),
    "cbinfer_cbconv2d_forward_after": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f,
                                            _i, _i, _i, _vp, _i, _vp]),
    "cbinfer_change_detection_frame_after": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _i, _vp]),
    "cbinfer_conv_changed_from_mask_after": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i,
                                                  _i, _vp, _i, _vp]),
    "cbinfer_frame_mask_bytes": (_l, [_i, _i]),
    "cbinfer_frame_mask_max_words": (_i, []),
    "cbinfer_change_detection_frame": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _i, _vp]),
    "cbinfer_change_detection_frame_pooled": (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "cbinfer_cbconv2d_forward_pooled": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i,
                                             _i, _i, _i, _f, _i, _vp, _i, _vp]),
    "cbinfer_conv_changed_from_mask": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i,
                                            _i, _vp, _i, _vp]),
    "cbinfer_max_pool2d": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "cbinfer_pool_change_indexes": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _vp]),
    "cbinfer_dilate_change_indexes": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp]),
    "cbinfer_change_detection_fg": (_i, [_vp, _vp, _vp, _vp, _l, _f, _i, _vp]),
    "cbinfer_update_output_fg": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _l, _vp]),
    "cbinfer_update_output_fg_list": (_i, [_vp, _vp, _vp, _vp, _l, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "cbinfer_change_detection_fg_frame": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "cbinfer_conv_accumulate_from_mask": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i,
                                               _vp, _i, _vp]),
    "cbinfer_cbconv2d_forward_fg": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i,
                                         _i, _f, _i, _vp, _i, _vp]),
    "cbinfer_change_detection_fg_bits": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "cbinfer_conv_accumulate_rows": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "cbinfer_conv_accumulate_blocks": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "cbinfer_cbconv2d_forward_fg_masked": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i,
                                                _i, _i, _i, _f, _i, _vp]),
    "cbinfer_tail1x1_max_hidden": (_i, []),
    "cbinfer_tail1x1_supported": (_i, [_i, _i, _i]),
    "cbinfer_tail1x1_prepared_bytes": (_l, [_i, _i]),
    "cbinfer_tail1x1_prep": (_i, [_vp, _vp, _i, _i, _vp]),
    "cbinfer_tail1x1": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "cbinfer_rowconv_supported": (_i, [_i, _i, _i, _i]),
    "cbinfer_rowconv_prepared_bytes": (_l, [_i, _i, _i, _i]),
    "cbinfer_rowconv_prep_weights": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "cbinfer_conv_changed_rows": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "cbinfer_change_detection_bits_pooled": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "cbinfer_cbconv2d_forward_rows": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i,
                                           _i, _i, _i, _f, _i, _i, _i, _vp]),
    "cbinfer_cbconv2d_forward_blocks": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i,
                                             _i, _i, _i, _i, _f, _i, _i, _i, _vp]),
    "cbinfer_blockconv_supported": (_i, [_i, _i, _i, _i]),
    "cbinfer_blockconv_prepared_bytes": (_l, [_i, _i, _i, _i]),
    "cbinfer_blockconv_prep_weights": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "cbinfer_conv_changed_blocks": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "cbinfer_change_detection_bits_batched": (_i, [_vpp, _vpp, _vpp, _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "cbinfer_conv_changed_rows_batched": (_i, [_vpp, _vpp, _vpp, _vpp, _vpp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i,
                                               _vp]),
    "cbinfer_tail1x1_batched": (_i, [_tp, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "cbinfer_split_supported": (_i, [_i, _i, _i, _i]),
    "cbinfer_split_max_sequences": (_i, []),
    "cbinfer_split_max_mask_words": (_l, [_i]),
    "cbinfer_split_state_bytes": (_l, [_i, _i, _i, _i, _i]),
    "cbinfer_split_prepared_bytes": (_l, [_i, _i, _i, _i]),
    "cbinfer_split_workspace_bytes": (_l, [_i, _i, _i, _i, _i, _i, _i]),
    "cbinfer_split_prep_weights": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "cbinfer_split_state_init": (_i, [_vp, _i, _i, _i, _i, _i, _vp]),
    "cbinfer_split_state_rebuild": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "cbinfer_split3_state_bytes": (_l, [_i, _i, _i, _i, _i]),
    "cbinfer_split3_prepared_bytes": (_l, [_i, _i, _i, _i]),
    "cbinfer_split3_prep_weights": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "cbinfer_split3_state_init": (_i, [_vp, _i, _i, _i, _i, _i, _vp]),
    "cbinfer_split3_state_rebuild": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "cbinfer_split_detect": (_i, [_sp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "cbinfer_split_conv": (_i, [_sp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _i, _vp]),
    "cbinfer_split_forward": (_i, [_sp, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _f, _i, _vp, _vp]),
    "cbinfer_split_next_supported": (_i, [_i, _i, _i, _i, _i, _i, _ndp]),
    "cbinfer_split_conv_next": (_i, [_sp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _ndp, _vp]),
    "cbinfer_split_refresh_supported": (_i, [_i, _i, _i, _i, _i, _i]),
    "cbinfer_split_conv_refresh": (_i, [_sp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _srp, _vp]),
    "cbinfer_split_conv_next_refresh": (_i, [_sp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _ndp, _srp, _vp]),
    "cbinfer_split_forward_next": (_i, [_sp, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _f, _i, _vp, _ndp,
                                        _vp]),
    "cbinfer_split_forward_fg": (_i, [_sp, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _f, _vp, _vp]),
    "cbinfer_hsplit_supported": (_i, [_i, _i, _i, _i]),
    "cbinfer_hsplit_max_mask_words": (_l, [_i]),
    "cbinfer_hsplit_state_bytes": (_l, [_i, _i, _i, _i, _i]),
    "cbinfer_hsplit_prepared_bytes": (_l, [_i, _i, _i, _i]),
    "cbinfer_hsplit_workspace_bytes": (_l, [_i, _i, _i, _i, _i, _i]),
    "cbinfer_hsplit_prep_weights": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "cbinfer_hsplit_state_init": (_i, [_vp, _i, _i, _i, _i, _i, _vp]),
    "cbinfer_hsplit_state_rebuild": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "cbinfer_hsplit_forward": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i,
                                    _i, _i, _i, _f, _i, _i, _vp, _vp]),
    "cbinfer_hsplit_group_workspace_bytes": (_l, [_i, _i, _i, _i, _i, _i, _i]),
    "cbinfer_hsplit_forward_group": (_i, [_hlp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "cbinfer_split_forward_tail": (_i, [_sp, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _f, _i, _vp, _i,
                                        _stp, _vp]),
    "cbinfer_split_forward_fg_tail": (_i, [_sp, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _f, _f, _vp, _stp, _vp]),
    "cbinfer_split_tail_supported": (_i, [_i, _i, _i, _i, _i, _i]),
    "cbinfer_split_conv_tail": (_i, [_sp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _i, _stp, _vp]),
    "cbinfer_rowpairs_supported": (_i, [_i, _i, _i, _i, _i, _i]),
    "cbinfer_conv_changed_rowpairs": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _ndp, _vp]),
    "cbinfer_conv_changed_rowpairs_batched": (_i, [_psp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _ndp, _vp]),
    "cbinfer_conv_rowpairs_detect": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _ndp, _vp]),
    "cbinfer_refresh_state": (_i, [_vp, _vp, _i, _i, _i, _f, _vp]),
    "cbinfer_cbconv2d_forward_rowpairs": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i,
                                               _ndp, _vp]),
    "cbinfer_frame_mask_copy_offset": (_l, [_i, _i]),
    "cbinfer_concat_channels": (_i, [_vpp, ctypes.POINTER(ctypes.c_int32), _i, _vp, _l, _i, _vp]),
    "cbinfer_conv2d_fg_cpu": (None, [_vp, _vp, _vp, _vp, _f, _i, _i, _i, _i, _i, _i]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "cbinfer_amd: %s not found. Build the HIP kernels first (python -c 'import "
            "__graft_entry__ as g; g.build()' or make -C cbinfer_amd/csrc). There is no CPU or "
            "PyTorch fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)     # AttributeError if the ABI is incomplete
        fn.restype = res
        fn.argtypes = args
    if lib.cbinfer_abi_version() != 11:
        raise ImportError("cbinfer_amd: libcbinfer_hip.so ABI version mismatch")
    return lib


class _Fn(object):
    """One entry point of the library.  Calls go straight to the ctypes function; while a `FrameProgram` records
    (cbinfer_amd/program.py) every call is also noted -- function and argument objects -- so that the frame's launch
    sequence can be replayed later without the modules around it."""
    __slots__ = ('raw', '__name__', 'launcher')

    def __init__(self, raw, name, launcher):
        # launcher: the function enqueues work on a stream (int status, last argument the stream) -- as opposed to the host
        # helpers (sizes, capability queries), which a recorded frame does not need to repeat
        self.raw, self.__name__, self.launcher = raw, name, launcher

    def __call__(self, *args):
        rec = _RECORDING[0]
        if rec is not None and self.launcher:
            rec.append((self, args))
        return self.raw(*args)


_RECORDING = [None]


class _Lib(object):
    def __init__(self, lib):
        self._cdll = lib
        for name in _SIGNATURES:
            res, args = _SIGNATURES[name]
            setattr(self, name, _Fn(getattr(lib, name), name, res is _i and bool(args) and args[-1] is _vp))


C = _Lib(_load())


class CBinferError(RuntimeError):
    pass


def check(status):
    """Raise on a non-zero launcher status (the reference launchers return void and never check)."""
    if status != 0:
        raise CBinferError("libcbinfer_hip: %s (status %d)" %
                           (C.cbinfer_status_string(status).decode(), status))


def dtype_code(t):
    import torch
    if t.dtype == torch.float32:
        return CB_F32
    if t.dtype == torch.float16:
        return CB_F16
    raise TypeError("cbinfer_amd supports float32 and float16 tensors, got %s" % t.dtype)


def stream_ptr(t=None):
    import torch
    return torch.cuda.current_stream(t.device if t is not None else None).cuda_stream


def raw_stream(device_index):
    """Raw handle of torch's current stream on a device (the cheap form used on the per-frame fast path)."""
    import torch
    try:
        return torch._C._cuda_getCurrentRawStream(device_index)
    except AttributeError:      # pragma: no cover  (older torch)
        return torch.cuda.current_stream(device_index).cuda_stream


def ptr(t):
    return t.data_ptr() if t is not None else None


def require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise CBinferError(
                "cbinfer_amd runs on HIP devices only: got a CPU tensor. (The reference's pure-torch "
                "CPU twins are restated in oracle/ for testing, they are not part of the product.)")
