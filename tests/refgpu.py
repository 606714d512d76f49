"""TEST INFRASTRUCTURE: ctypes access to the reference's own GPU kernels, compiled from
/root/reference/pycbinfer/cbconv2d_{cg,fg}_backend.cu by oracle/Makefile into oracle/_ref/ (the .so
travels to the GPU box, the sources do not).  Launch geometry is what the reference's Python wrappers
compute (conv2d_cg.py:67-68,106-107,166-167,244-247,296-299; conv2d_fg.py:55-56).  Used only as a
checker in -m gpu tests."""
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


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


_i = ctypes.c_int


def changeDetection(inp, prev, filt, th, update=False):
    C, H, W = inp.shape[-3:]
    cm = torch.zeros(H, W, dtype=torch.int8, device=inp.device)
    torch.cuda.synchronize()
    cg().changeDetection(_i(1), _i(1), _i((H * W - 1) // 128 + 1), _i(1), _i(1), _i(128), _p(inp),
                         _p(prev), _p(cm), _i(W), _i(H), _i(C), _i((filt[0] - 1) // 2),
                         _i((filt[1] - 1) // 2), ctypes.c_float(th), ctypes.c_bool(update))
    torch.cuda.synchronize()
    return cm


def changePropagation(cm, filt):
    H, W = cm.shape[-2:]
    out = torch.empty_like(cm)
    torch.cuda.synchronize()
    cg().changePropagation(_i(1), _i(1), _i((H * W - 1) // 128 + 1), _i(1), _i(1), _i(128), _p(cm),
                           _p(out), _i(W), _i(H), _i((filt[0] - 1) // 2), _i((filt[1] - 1) // 2))
    torch.cuda.synchronize()
    return out


def genXMatrix(inp, idx, filt):
    C, H, W = inp.shape[-3:]
    kH, kW = filt
    N = idx.numel()
    X = torch.empty(N, C * kH * kW, dtype=inp.dtype, device=inp.device)
    threadZ = 128 // (kH * kW)
    torch.cuda.synchronize()
    cg().genXMatrix(_i(1), _i(1), _i((N - 1) // threadZ + 1), _i(kH), _i(threadZ), _i(kW), _p(X),
                    _p(inp), _p(idx), _i(kW), _i(kH), _i(C), _i(W), _i(H), _i(N))
    torch.cuda.synchronize()
    return X


def updateOutput(Yt, idx, out, relu):
    K, H, W = out.shape[-3:]
    N = idx.numel()
    Yt = Yt.contiguous()
    torch.cuda.synchronize()
    cg().updateOutput(_i(1), _i(1), _i((N * K - 1) // 1024 + 1), _i(1), _i(1), _i(1024), _p(Yt), _p(out),
                      _p(idx), _i(H * W), _i(N), _i(K), ctypes.c_bool(relu))
    torch.cuda.synchronize()
    return out


def maxPool2d(inp, out, idx):
    C, H, W = inp.shape[-3:]
    oh, ow = out.shape[-2:]
    N = idx.numel()
    torch.cuda.synchronize()
    cg().maxPool2d(_i((N - 1) // 64 + 1), _i(64), _p(inp), _p(out), _p(idx), _i(N), _i(C), _i(H), _i(W),
                   _i(oh), _i(ow), _i(2), _i(2))
    torch.cuda.synchronize()
    return out


def changeDetectionFG(inp, prev, th):
    diffs = torch.zeros_like(inp)
    cm = torch.zeros(inp.shape, dtype=torch.int8, device=inp.device)
    torch.cuda.synchronize()
    fg().changeDetectionFG(_p(inp), _p(prev), _p(diffs), _p(cm), _i(inp.numel()), ctypes.c_float(th))
    torch.cuda.synchronize()
    return diffs, cm


def updateOutputFG(diffs, weight, out, coords):
    K, C, kH, kW = weight.shape
    H, W = out.shape[-2:]
    N = coords.shape[0]
    torch.cuda.synchronize()
    fg().updateOutputFG(_i(1), _i(1), _i((N - 1) // 128 + 1), _i(1), _i(1), _i(128), _p(diffs),
                        _p(weight), _p(out), _p(coords), _i(K), _i(C), _i(H), _i(W), _i(kH), _i(kW),
                        _i(N))
    torch.cuda.synchronize()
    return out
