"""TEST INFRASTRUCTURE: ctypes calls with the REFERENCE's launcher signatures (the cffi cdef of
conv2d_cg.py:6-38 and conv2d_fg.py:14-29: six launch-geometry ints, bool* map, const long* coordinates)
and the launch geometry the reference's Python wrappers compute (conv2d_cg.py:67-68,106-107,166-167,
244-247,296-299; conv2d_fg.py:55-56).  By default they go to the reference's own GPU kernels, compiled
from /root/reference/pycbinfer/cbconv2d_{cg,fg}_backend.cu by oracle/Makefile into oracle/_ref/ (the .so
travels to the GPU box, the sources do not) and serve as a checker in -m gpu tests; with `so=` they go to
any other library exporting the same symbols -- the product's compat shims
(cbinfer_amd/compat/cbconv2d_*_backend_<machine>.so), which is how those are tested."""
import ctypes
import os

import torch

_REF_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")


def available():
    return os.path.exists(os.path.join(_REF_DIR, "cbconv2d_cg_backend.so"))


_cg = _fg = None


def cg():
    global _cg
    if _cg is None:
        _cg = ctypes.CDLL(os.path.join(_REF_DIR, "cbconv2d_cg_backend.so"))
    return _cg


def fg():
    global _fg
    if _fg is None:
        _fg = ctypes.CDLL(os.path.join(_REF_DIR, "cbconv2d_fg_backend.so"))
    return _fg


def compat(kind):
    """CDLL of the product's reference-signature shim: kind in {'cg', 'cg_half', 'fg'}."""
    import platform
    path = os.path.join(os.path.dirname(_REF_DIR), os.pardir, "cbinfer_amd", "compat",
                        "cbconv2d_%s_backend_%s.so" % (kind, platform.machine()))
    return ctypes.CDLL(os.path.abspath(path))


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


_i = ctypes.c_int


def changeDetection(inp, prev, filt, th, update=False, so=None):
    C, H, W = inp.shape[-3:]
    cm = torch.zeros(H, W, dtype=torch.int8, device=inp.device)
    torch.cuda.synchronize()
    (so or cg()).changeDetection(_i(1), _i(1), _i((H * W - 1) // 128 + 1), _i(1), _i(1), _i(128), _p(inp),
                         _p(prev), _p(cm), _i(W), _i(H), _i(C), _i((filt[0] - 1) // 2),
                         _i((filt[1] - 1) // 2), ctypes.c_float(th), ctypes.c_bool(update))
    torch.cuda.synchronize()
    return cm


def changePropagation(cm, filt, so=None):
    H, W = cm.shape[-2:]
    out = torch.empty_like(cm)
    torch.cuda.synchronize()
    (so or cg()).changePropagation(_i(1), _i(1), _i((H * W - 1) // 128 + 1), _i(1), _i(1), _i(128), _p(cm),
                           _p(out), _i(W), _i(H), _i((filt[0] - 1) // 2), _i((filt[1] - 1) // 2))
    torch.cuda.synchronize()
    return out


def genXMatrix(inp, idx, filt, so=None):
    C, H, W = inp.shape[-3:]
    kH, kW = filt
    N = idx.numel()
    X = torch.empty(N, C * kH * kW, dtype=inp.dtype, device=inp.device)
    threadZ = 128 // (kH * kW)
    torch.cuda.synchronize()
    (so or cg()).genXMatrix(_i(1), _i(1), _i((N - 1) // threadZ + 1), _i(kH), _i(threadZ), _i(kW), _p(X),
                    _p(inp), _p(idx), _i(kW), _i(kH), _i(C), _i(W), _i(H), _i(N))
    torch.cuda.synchronize()
    return X


def updateOutput(Yt, idx, out, relu, so=None):
    K, H, W = out.shape[-3:]
    N = idx.numel()
    Yt = Yt.contiguous()
    torch.cuda.synchronize()
    (so or cg()).updateOutput(_i(1), _i(1), _i((N * K - 1) // 1024 + 1), _i(1), _i(1), _i(1024), _p(Yt), _p(out),
                      _p(idx), _i(H * W), _i(N), _i(K), ctypes.c_bool(relu))
    torch.cuda.synchronize()
    return out


def maxPool2d(inp, out, idx, so=None):
    C, H, W = inp.shape[-3:]
    oh, ow = out.shape[-2:]
    N = idx.numel()
    torch.cuda.synchronize()
    (so or cg()).maxPool2d(_i((N - 1) // 64 + 1), _i(64), _p(inp), _p(out), _p(idx), _i(N), _i(C), _i(H), _i(W),
                   _i(oh), _i(ow), _i(2), _i(2))
    torch.cuda.synchronize()
    return out


def changeDetectionFG(inp, prev, th, so=None):
    diffs = torch.zeros_like(inp)
    cm = torch.zeros(inp.shape, dtype=torch.int8, device=inp.device)
    torch.cuda.synchronize()
    (so or fg()).changeDetectionFG(_p(inp), _p(prev), _p(diffs), _p(cm), _i(inp.numel()), ctypes.c_float(th))
    torch.cuda.synchronize()
    return diffs, cm


def updateOutputFG(diffs, weight, out, coords, so=None):
    K, C, kH, kW = weight.shape
    H, W = out.shape[-2:]
    N = coords.shape[0]
    torch.cuda.synchronize()
    (so or fg()).updateOutputFG(_i(1), _i(1), _i((N - 1) // 128 + 1), _i(1), _i(1), _i(128), _p(diffs),
                        _p(weight), _p(out), _p(coords), _i(K), _i(C), _i(H), _i(W), _i(kH), _i(kW),
                        _i(N))
    torch.cuda.synchronize()
    return out


def conv2d_fg_cpu(inp, prev, out, weight, th, so=None):
    """HOST tensors (cbconv2d_fg_backend.cu:81-112)."""
    K, C, kH, kW = weight.shape
    H, W = inp.shape[-2:]
    (so or fg()).conv2d_fg_cpu(_p(inp), _p(prev), _p(out), _p(weight), ctypes.c_float(th), _i(K), _i(C),
                               _i(H), _i(W), _i(kH), _i(kW))
    return out
