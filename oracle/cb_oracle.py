"""cb_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy front-end of the plain-C CPU restatement (oracle/cb_oracle.c) of CBinfer's change-based
convolution path, plus numpy restatements of the cg_half backend's ops, of the operation-count
statistics (compStats) and of the module-level state machines (CBConv2d.forward_normal / forward_fg,
CBPoolMax2d.forward) in fp32 and fp16 (OracleCBConv2dHalf / OracleCBPoolMax2dHalf).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  The
product package never does.  Parity status: PINNED (see the header of cb_oracle.c).

Reference citations are relative to /root/reference/pycbinfer.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libcb_oracle.so")

CMP_GT = 0  # CUDA kernels: strict >
CMP_GE = 1  # python twin (conv2d_cg.py:126) and conv2d_fg_cpu (fg.cu:93): >=


def build(force=False):
    """Compile the C oracle with gcc (seconds)."""
    src = os.path.join(_HERE, "cb_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.orc_changeIndexesExtr.restype = ctypes.c_int
        _lib.orc_num_threads.restype = ctypes.c_int
    return _lib


def _p(a):
    return ctypes.c_void_p(a.ctypes.data)


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


def num_threads():
    return int(lib().orc_num_threads())


# ----------------------------------------------------------------------------------------------
# op level (fp32)
# ----------------------------------------------------------------------------------------------
def changeDetection(inp, state, filtSize, threshold, updateInputState=False, cmp=CMP_GT):
    """cbconv2d_cg_backend.cu:40-100.  `state` is updated in place when updateInputState.
    Returns the int8 [H,W] change map."""
    assert inp.shape == state.shape and inp.ndim == 4 and inp.shape[0] == 1
    assert inp.dtype == np.float32 and state.dtype == np.float32
    assert inp.flags.c_contiguous and state.flags.c_contiguous
    _, C, H, W = inp.shape
    cm = np.zeros((H, W), dtype=np.int8)
    lib().orc_changeDetection(_p(inp), _p(state), _p(cm), ctypes.c_int(W), ctypes.c_int(H),
                              ctypes.c_int(C), ctypes.c_int((filtSize[0] - 1) // 2),
                              ctypes.c_int((filtSize[1] - 1) // 2), ctypes.c_float(threshold),
                              ctypes.c_int(int(updateInputState)), ctypes.c_int(cmp))
    return cm


def changePropagation(changeMap, filtSize):
    """cbconv2d_cg_backend.cu:101-136."""
    cm = np.ascontiguousarray(changeMap.reshape(changeMap.shape[-2:]), dtype=np.int8)
    H, W = cm.shape
    out = np.zeros_like(cm)
    lib().orc_changePropagation(_p(cm), _p(out), ctypes.c_int(W), ctypes.c_int(H),
                                ctypes.c_int((filtSize[0] - 1) // 2),
                                ctypes.c_int((filtSize[1] - 1) // 2))
    return out


def changeIndexesExtr(changeMap):
    """conv2d_cg.py:200-209."""
    cm = np.ascontiguousarray(changeMap, dtype=np.int8).reshape(-1)
    idx = np.empty(cm.size, dtype=np.int32)
    n = lib().orc_changeIndexesExtr(_p(cm), ctypes.c_long(cm.size), _p(idx))
    return idx[:n].copy()


def genXMatrix(inp, changeIndexes, filtSize):
    """cbconv2d_cg_backend.cu:138-173."""
    inp = _f32(inp)
    _, C, H, W = inp.shape
    kH, kW = filtSize
    idx = np.ascontiguousarray(changeIndexes, dtype=np.int32)
    X = np.empty((idx.size, C * kH * kW), dtype=np.float32)
    if idx.size:
        lib().orc_genXMatrix(_p(X), _p(inp), _p(idx), ctypes.c_int(kW), ctypes.c_int(kH),
                             ctypes.c_int(C), ctypes.c_int(W), ctypes.c_int(H),
                             ctypes.c_int(idx.size))
    return X


def matrixMult(X, weight, bias, accMode=0):
    """conv2d_cg.py:342-349: Y[N,K] = X . W.view(K,-1)^T + bias."""
    X = _f32(X)
    K = weight.shape[0]
    Wm = _f32(weight).reshape(K, -1)
    b = _f32(bias)
    N, Ckk = X.shape
    assert Wm.shape[1] == Ckk
    Y = np.empty((N, K), dtype=np.float32)
    if N:
        lib().orc_matrixMult(_p(X), _p(Wm), _p(b), _p(Y), ctypes.c_int(N), ctypes.c_int(Ckk),
                             ctypes.c_int(K), ctypes.c_int(accMode))
    return Y


def updateOutput(Yt, changeIndexes, prevOutput, withReLU=False):
    """cbconv2d_cg_backend.cu:175-197.  Yt is [K,N]; prevOutput [1,K,H,W] is updated in place."""
    Yt = _f32(Yt)
    idx = np.ascontiguousarray(changeIndexes, dtype=np.int32)
    K, H, W = prevOutput.shape[-3:]
    assert prevOutput.dtype == np.float32 and prevOutput.flags.c_contiguous
    assert Yt.shape == (K, idx.size)
    if idx.size:
        lib().orc_updateOutput(_p(Yt), _p(prevOutput), _p(idx), ctypes.c_int(H * W),
                               ctypes.c_int(idx.size), ctypes.c_int(K), ctypes.c_int(int(withReLU)))
    return prevOutput


def maxPool2d(inp, outputState, changeIndexes, guardOutput=True):
    """cbconv2d_cg_backend.cu:199-240; outputState [1,C,oh,ow] updated in place."""
    inp = _f32(inp)
    idx = np.ascontiguousarray(changeIndexes, dtype=np.int32)
    C, H, W = inp.shape[-3:]
    oh, ow = outputState.shape[-2:]
    assert outputState.dtype == np.float32 and outputState.flags.c_contiguous
    if not guardOutput:
        # the unguarded reference form is only defined when no window falls outside the output
        assert (H - 1) // 2 < oh and (W - 1) // 2 < ow
    if idx.size:
        lib().orc_maxPool2d(_p(inp), _p(outputState), _p(idx), ctypes.c_int(idx.size),
                            ctypes.c_int(C), ctypes.c_int(H), ctypes.c_int(W), ctypes.c_int(oh),
                            ctypes.c_int(ow), ctypes.c_int(2), ctypes.c_int(2),
                            ctypes.c_int(int(guardOutput)))
    return outputState


def poolChangeIndexes(changeIndexes, inSize, outSize):
    """Indexes of the 2x2/stride-2 pool OUTPUT pixels whose window holds a changed input pixel (ascending,
    unique) -- the set cbconv2d_cg_backend.cu:214-226 recomputes, in the order changeIndexesExtr would
    give it.  (The reference itself hands the input list on unchanged, conv2d.py:80-83.)"""
    idx = np.asarray(changeIndexes, dtype=np.int64)
    (iH, iW), (oH, oW) = inSize, outSize
    yo, xo = (idx // iW) // 2, (idx % iW) // 2
    keep = (yo < oH) & (xo < oW)
    return np.unique(yo[keep] * oW + xo[keep]).astype(np.int32)


def changeDetectionFG(inp, prevInput, threshold, cmp=CMP_GT, diffs_init=None):
    """cbconv2d_fg_backend.cu:7-35.  diffs are only written where changed; elsewhere they keep
    diffs_init (default 0 here; uninitialised in the reference)."""
    inp, prevInput = _f32(inp), _f32(prevInput)
    diffs = np.zeros_like(inp) if diffs_init is None else _f32(diffs_init).copy()
    cm = np.zeros(inp.shape, dtype=np.int8)
    lib().orc_changeDetectionFG(_p(inp), _p(prevInput), _p(diffs), _p(cm), ctypes.c_long(inp.size),
                                ctypes.c_float(threshold), ctypes.c_int(cmp))
    return diffs, cm


def updateOutputFG(diffs, weight, output, changeCoords):
    """cbconv2d_fg_backend.cu:37-79; output updated in place."""
    diffs, weight = _f32(diffs), _f32(weight)
    coords = np.ascontiguousarray(changeCoords, dtype=np.int64).reshape(-1)
    K, C, kH, kW = weight.shape
    H, W = output.shape[-2:]
    assert output.dtype == np.float32 and output.flags.c_contiguous
    lib().orc_updateOutputFG(_p(diffs), _p(weight), _p(output), _p(coords), ctypes.c_int(K),
                             ctypes.c_int(C), ctypes.c_int(H), ctypes.c_int(W), ctypes.c_int(kH),
                             ctypes.c_int(kW), ctypes.c_long(coords.size))
    return output


def conv2d_fg_cpu(inp, prevInput, output, weight, threshold):
    """cbconv2d_fg_backend.cu:81-112 (single-threaded); output updated in place."""
    inp, prevInput, weight = _f32(inp), _f32(prevInput), _f32(weight)
    K, C, kH, kW = weight.shape
    H, W = inp.shape[-2:]
    assert output.dtype == np.float32 and output.flags.c_contiguous
    lib().orc_conv2d_fg_cpu(_p(inp), _p(prevInput), _p(output), _p(weight),
                            ctypes.c_float(threshold), ctypes.c_int(K), ctypes.c_int(C),
                            ctypes.c_int(H), ctypes.c_int(W), ctypes.c_int(kH), ctypes.c_int(kW))
    return output


def conv2d_dense(inp, weight, bias, relu=False):
    """Ground truth: conv2d(pad=k//2)+bias accumulated in double."""
    inp, weight = _f32(inp), _f32(weight)
    K, C, kH, kW = weight.shape
    H, W = inp.shape[-2:]
    out = np.empty((1, K, H, W), dtype=np.float32)
    b = _f32(bias) if bias is not None else None
    lib().orc_conv2d_dense(_p(inp), _p(weight), _p(b) if b is not None else None, _p(out),
                           ctypes.c_int(K), ctypes.c_int(C), ctypes.c_int(H), ctypes.c_int(W),
                           ctypes.c_int(kH), ctypes.c_int(kW), ctypes.c_int(int(relu)))
    return out


def maxpool_dense(inp, ceil_mode=False):
    """F.max_pool2d(inp, 2, 2, ceil_mode) on a [1,C,H,W] array (ground truth for CBPoolMax2d)."""
    _, C, H, W = inp.shape
    oh, ow = ((H - 1) // 2 + 1, (W - 1) // 2 + 1) if ceil_mode else (H // 2, W // 2)
    pad = np.full((1, C, 2 * oh, 2 * ow), -np.inf, dtype=inp.dtype)
    hh, ww = min(H, 2 * oh), min(W, 2 * ow)
    pad[:, :, :hh, :ww] = inp[:, :, :hh, :ww]
    return pad.reshape(1, C, oh, 2, ow, 2).max(axis=(3, 5))


# ----------------------------------------------------------------------------------------------
# fp16 change predicate (cbconv2d_cg_half_backend.cu:10-88), restated in numpy
# ----------------------------------------------------------------------------------------------
def changeDetection_half(inp, state, filtSize, threshold, updateInputState=False, cmp=CMP_GT):
    """diff = __hsub(state, in) (one rounding to half), changed = diff > th16 | diff < -th16 with
    th16 = __float2half(th) (cg_half.cu:24-29, :60-65); dilation and state update as in fp32.
    cmp=CMP_GE: the python twin's comparison on half tensors ((input-prevInput).abs().ge(th), conv2d_cg.py:126)
    -- the two only differ on exact ties, which the reference-derived fixtures avoid."""
    assert inp.dtype == np.float16 and state.dtype == np.float16
    _, C, H, W = inp.shape
    th16 = np.float16(np.float32(threshold))
    # exact difference in float64 (11-bit significands, exponent span < 42 bits), rounded once
    diff = (state.astype(np.float64) - inp.astype(np.float64)).astype(np.float16)
    if cmp == CMP_GE:
        changed = ((diff >= th16) | (diff <= -th16)).any(axis=1)[0]
    else:
        changed = ((diff > th16) | (diff < -th16)).any(axis=1)[0]        # [H,W]
    cm = changePropagation(changed.astype(np.int8), filtSize)
    if updateInputState:
        sel = np.broadcast_to(changed[None, None], state.shape)
        state[sel] = inp[sel]
    return cm


# ----------------------------------------------------------------------------------------------
# module level: restatement of conv2d.py's state machines on numpy arrays
# ----------------------------------------------------------------------------------------------
class OracleCBConv2d:
    """CBConv2d.forward_normal (conv2d.py:178-259) and forward_fg (:160-176) on numpy arrays."""

    def __init__(self, weight, bias, threshold, withReLU=False, feedbackLoop=False,
                 propChangeIndexes=False, finegrained=False, copyInput=True, cmp=CMP_GT,
                 accMode=0):
        self.weight = _f32(weight)
        self.bias = _f32(bias)
        self.kernel_size = tuple(self.weight.shape[2:])
        self.threshold = float(threshold)
        self.withReLU = withReLU
        self.feedbackLoop = feedbackLoop
        self.propChangeIndexes = propChangeIndexes
        self.finegrained = finegrained
        self.copyInput = copyInput
        self.cmp = cmp
        self.accMode = accMode
        self.clearMemory()

    def clearMemory(self):
        self.prevInput = np.zeros((0,), np.float32)
        self.prevOutput = np.zeros((0,), np.float32)
        self.changeMap = None
        self.changeIndexes = None

    def forward(self, inp):
        if self.finegrained:
            return self._forward_fg(inp)
        changeIndexes = None
        if isinstance(inp, tuple):
            assert inp[0] == 'changeIndexes'
            x, changeIndexes = _f32(inp[1]), np.asarray(inp[2], dtype=np.int32)
        else:
            x = _f32(inp)
        K = self.weight.shape[0]
        if self.prevInput.shape != x.shape:                       # conv2d.py:192-194
            self.prevInput = np.full(x.shape, np.inf, np.float32)
        oshape = (1, K) + x.shape[2:]
        if self.prevOutput.shape != oshape:                       # :195-199
            self.prevOutput = np.full(oshape, np.inf, np.float32)
        if changeIndexes is None:                                 # :220-232
            self.changeMap = changeDetection(x, self.prevInput, self.kernel_size, self.threshold,
                                             updateInputState=self.feedbackLoop, cmp=self.cmp)
            changeIndexes = changeIndexesExtr(self.changeMap)
        if not self.feedbackLoop:                                 # :234-238
            # copyInput=False ALIASES the caller's tensor (:237-238): if the producer later updates it
            # in place (a preceding CBConv2d returns its own prevOutput), changes go undetected --
            # reproduced here on purpose, the reference's apps only use it behind out-of-place ops
            self.prevInput = x.copy() if self.copyInput else x
        self.changeIndexes = changeIndexes
        if changeIndexes.size:                                    # :240-251
            X = genXMatrix(self.prevInput, changeIndexes, self.kernel_size)
            Y = matrixMult(X, self.weight, self.bias, accMode=self.accMode)
            updateOutput(np.ascontiguousarray(Y.T), changeIndexes, self.prevOutput,
                         withReLU=self.withReLU)
        if self.propChangeIndexes:                                # :256-259
            return ('changeIndexes', self.prevOutput, changeIndexes)
        return self.prevOutput

    def _forward_fg(self, inp):
        x = _f32(inp)
        if self.prevInput.shape != x.shape:                       # conv2d.py:163-167
            self.prevOutput = conv2d_dense(x, self.weight, self.bias)
        else:                                                     # :169-170, conv2d_fg.py:75-85
            po = self.prevOutput.copy()
            diffs, cm = changeDetectionFG(x, self.prevInput, self.threshold, cmp=self.cmp)
            coords = np.nonzero(cm.reshape(-1))[0].astype(np.int64)
            if coords.size:
                updateOutputFG(diffs, self.weight, po, coords)
            self.prevOutput = po
        out = np.maximum(self.prevOutput, 0) if self.withReLU else self.prevOutput  # :172-174
        self.prevInput = x.copy()                                 # :175
        return out


class OracleCBPoolMax2d:
    """CBPoolMax2d.forward (conv2d.py:49-78)."""

    def __init__(self, ceil_mode=False, propChangeIndexes=False):
        self.ceil_mode = ceil_mode
        self.propChangeIndexes = propChangeIndexes
        self.clearMemory()

    def clearMemory(self):
        self.outputState = np.zeros((0,), np.float32)

    def forward(self, inp):
        assert isinstance(inp, tuple) and inp[0] == 'changeIndexes'
        x, idx = _f32(inp[1]), np.asarray(inp[2], dtype=np.int32)
        if idx.size:
            _, C, H, W = x.shape
            oh, ow = ((H - 1) // 2 + 1, (W - 1) // 2 + 1) if self.ceil_mode else (H // 2, W // 2)
            if self.outputState.shape != (1, C, oh, ow):
                self.outputState = np.full((1, C, oh, ow), np.inf, np.float32)
            maxPool2d(x, self.outputState, idx, guardOutput=True)
        out = self.outputState.copy()
        if self.propChangeIndexes:
            return ('changeIndexes', out, idx)
        return out


class OracleReLU:
    def forward(self, x):
        return np.maximum(x, 0)

    def clearMemory(self):
        pass


class OracleMaxPool2d:
    def __init__(self, ceil_mode=False):
        self.ceil_mode = ceil_mode

    def forward(self, x):
        return maxpool_dense(x, self.ceil_mode)

    def clearMemory(self):
        pass


class OracleSequential:
    def __init__(self, layers):
        self.layers = list(layers)

    def forward(self, x):
        for l in self.layers:
            x = l.forward(x)
        return x

    def clearMemory(self):
        for l in self.layers:
            l.clearMemory()


# ----------------------------------------------------------------------------------------------
# a17: operation-count statistics (conv2d.py:201-218), restated in numpy
# ----------------------------------------------------------------------------------------------
def compStats(inp, prevInput, weightShape, threshold):
    """The five operation counts CBConv2d gathers when gatherComputationStats is set.  `proped` is a
    VALID (unpadded) grouped convolution of the per-value change tensor with a ones filter
    (conv2d.py:205-209: F.conv2d without a padding argument), i.e. a sliding-window OR over
    (H-kH+1) x (W-kW+1) positions."""
    x, p = np.asarray(inp, dtype=np.float32), np.asarray(prevInput, dtype=np.float32)
    K, C, kH, kW = weightShape
    with np.errstate(invalid='ignore'):
        changeTensor = np.abs(x - p) > np.float32(threshold)          # [1,C,H,W], strict (torch .gt)
    H, W = changeTensor.shape[-2:]
    win = np.lib.stride_tricks.sliding_window_view(changeTensor[0], (kH, kW), axis=(1, 2))
    proped = win.any(axis=(-1, -2))                                    # [C, H-kH+1, W-kW+1]
    opsPerValue = K * kH * kW * 2
    return dict(
        numInputChangesPerFeatureMap=int(changeTensor.sum()) * opsPerValue,
        numInputChanges=int(changeTensor[0].any(axis=0).sum()) * C * opsPerValue,
        numInputPropedChangesPerFeatureMap=int(proped.sum()) * opsPerValue,
        numInputPropedChanges=int(proped.any(axis=0).sum()) * C * opsPerValue,
        totalInputValues=W * H * C * opsPerValue)


# ----------------------------------------------------------------------------------------------
# cg_half path: module state machine on float16 arrays (cbconv2d_cg_half_backend.cu + conv2d.py)
# ----------------------------------------------------------------------------------------------
def genXMatrix_half(inp, changeIndexes, filtSize):
    """cbconv2d_cg_half_backend.cu:146-169 (pure data movement, zero outside the image)."""
    assert inp.dtype == np.float16
    _, C, H, W = inp.shape
    kH, kW = filtSize
    ph, pw = (kH - 1) // 2, (kW - 1) // 2
    pad = np.zeros((C, H + kH - 1, W + kW - 1), dtype=np.float16)
    pad[:, ph:ph + H, pw:pw + W] = inp[0]
    win = np.lib.stride_tricks.sliding_window_view(pad, (kH, kW), axis=(1, 2))   # [C,H,W,kH,kW]
    idx = np.asarray(changeIndexes, dtype=np.int64)
    X = win[:, idx // W, idx % W]                                     # [C,N,kH,kW]
    return np.ascontiguousarray(X.transpose(1, 0, 2, 3)).reshape(idx.size, C * kH * kW)


def matrixMult_half(X, weight, bias):
    """conv2d_cg.py:342-349 on half tensors: fp16 operands; the products and the sum are formed
    exactly (float64) and rounded to fp16 ONCE, bias included.  (torch's hgemm + add_ would round the
    product sum and the biased sum separately; the accumulation precision of the reference's cuBLAS
    hgemm is unpinned, SURVEY 7 'fp16 parity' -- this is the tightest statement both satisfy within
    the 2-ulp bar of DESIGN.md 6.)"""
    assert X.dtype == np.float16 and weight.dtype == np.float16 and bias.dtype == np.float16
    K = weight.shape[0]
    Y = X.astype(np.float64) @ weight.reshape(K, -1).astype(np.float64).T + bias.astype(np.float64)
    return Y.astype(np.float16)


def updateOutput_half(Yt, changeIndexes, prevOutput, withReLU=False):
    """cbconv2d_cg_half_backend.cu:183-197: v = relu && __hle(v, 0) ? 0 : v."""
    assert Yt.dtype == np.float16 and prevOutput.dtype == np.float16
    idx = np.asarray(changeIndexes, dtype=np.int64)
    K, H, W = prevOutput.shape[-3:]
    v = np.where(Yt <= np.float16(0), np.float16(0), Yt) if withReLU else Yt
    prevOutput.reshape(K, H * W)[:, idx] = v
    return prevOutput


def maxPool2d_half(inp, outputState, changeIndexes):
    """cbconv2d_cg_half_backend.cu:207-237 with the yo<oh / xo<ow guard (see maxPool2d)."""
    assert inp.dtype == np.float16 and outputState.dtype == np.float16
    C, H, W = inp.shape[-3:]
    oh, ow = outputState.shape[-2:]
    idx = np.asarray(changeIndexes, dtype=np.int64)
    yo, xo = (idx // W) // 2, (idx % W) // 2
    keep = (yo < oh) & (xo < ow)
    yo, xo = yo[keep], xo[keep]
    v = np.full((C, yo.size), -np.inf, dtype=np.float16)
    for j in range(2):
        for i in range(2):
            yi, xi = yo * 2 + j, xo * 2 + i
            ok = (yi < H) & (xi < W)
            val = np.where(ok[None, :], inp[0][:, np.minimum(yi, H - 1), np.minimum(xi, W - 1)],
                           np.float16(-np.inf))
            v = np.where(val > v, val, v)                             # __hgt: a NaN is never taken
    outputState[0][:, yo, xo] = v
    return outputState


class OracleCBConv2dHalf(OracleCBConv2d):
    """CBConv2d.forward_normal (conv2d.py:178-259) on float16 arrays through the half backend's ops."""

    def __init__(self, weight, bias, threshold, **kw):
        assert not kw.get('finegrained', False), "the fine-grained path is fp32 only (conv2d_fg.py)"
        super().__init__(np.zeros(weight.shape, np.float32), np.zeros(bias.shape, np.float32),
                         threshold, **kw)
        self.weight = np.ascontiguousarray(weight, dtype=np.float16)
        self.bias = np.ascontiguousarray(bias, dtype=np.float16)
        self.clearMemory()

    def clearMemory(self):
        self.prevInput = np.zeros((0,), np.float16)
        self.prevOutput = np.zeros((0,), np.float16)
        self.changeMap = None
        self.changeIndexes = None

    def forward(self, inp):
        changeIndexes = None
        if isinstance(inp, tuple):
            assert inp[0] == 'changeIndexes'
            x, changeIndexes = inp[1], np.asarray(inp[2], dtype=np.int32)
        else:
            x = inp
        assert x.dtype == np.float16
        x = np.ascontiguousarray(x)
        K = self.weight.shape[0]
        if self.prevInput.shape != x.shape:
            self.prevInput = np.full(x.shape, np.inf, np.float16)
        oshape = (1, K) + x.shape[2:]
        if self.prevOutput.shape != oshape:
            self.prevOutput = np.full(oshape, np.inf, np.float16)
        if changeIndexes is None:
            self.changeMap = changeDetection_half(x, self.prevInput, self.kernel_size, self.threshold,
                                                  updateInputState=self.feedbackLoop, cmp=self.cmp)
            changeIndexes = changeIndexesExtr(self.changeMap)
        if not self.feedbackLoop:
            self.prevInput = x.copy() if self.copyInput else x
        self.changeIndexes = changeIndexes
        if changeIndexes.size:
            X = genXMatrix_half(self.prevInput, changeIndexes, self.kernel_size)
            Y = matrixMult_half(X, self.weight, self.bias)
            updateOutput_half(np.ascontiguousarray(Y.T), changeIndexes, self.prevOutput,
                              withReLU=self.withReLU)
        if self.propChangeIndexes:
            return ('changeIndexes', self.prevOutput, changeIndexes)
        return self.prevOutput


class OracleCBPoolMax2dHalf(OracleCBPoolMax2d):
    def clearMemory(self):
        self.outputState = np.zeros((0,), np.float16)

    def forward(self, inp):
        assert isinstance(inp, tuple) and inp[0] == 'changeIndexes'
        x, idx = np.ascontiguousarray(inp[1]), np.asarray(inp[2], dtype=np.int32)
        assert x.dtype == np.float16
        if idx.size:
            _, C, H, W = x.shape
            oh, ow = ((H - 1) // 2 + 1, (W - 1) // 2 + 1) if self.ceil_mode else (H // 2, W // 2)
            if self.outputState.shape != (1, C, oh, ow):
                self.outputState = np.full((1, C, oh, ow), np.inf, np.float16)
            maxPool2d_half(x, self.outputState, idx)
        out = self.outputState.copy()
        if self.propChangeIndexes:
            return ('changeIndexes', out, idx)
        return out
