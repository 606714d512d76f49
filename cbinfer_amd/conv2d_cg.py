"""Coarse-grained change-based convolution ops on MI355X.

Host-side mirror of the reference's op wrappers (pycbinfer/conv2d_cg.py): same function names,
argument meaning and return values, but every op is a launch of a hand-written HIP kernel through the
C ABI of libcbinfer_hip.so on torch's current stream.  Launch geometry lives in the library.

Additions the reference lacks: `matrixMult` (the contraction on MFMA; the reference calls torch
matmul -> cuBLAS via matrixMult_python), `convChanged` (gather -> MFMA -> bias/ReLU -> scatter in one
launch) and `ChangeIndexes` (a change list whose length stays on the device).
"""
import ctypes

import torch

from ._lib import C, CBinferError, check, dtype_code, ptr, require_device, stream_ptr


# ------------------------------------------------------------------------------------------------
# change lists
# ------------------------------------------------------------------------------------------------
class ChangeIndexes(object):
    """An int32 list of changed pixel indices (y*W+x, ascending) with its length kept on the device.

    It stands where the reference passes an exact-length IntTensor in the
    ('changeIndexes', output, indexes) tuple (conv2d.py:256-259): `.buffer` has capacity H*W,
    `.count` is a device int32 scalar.  Consumers inside this package read the count on the device;
    `.tensor()` synchronises and returns the exact IntTensor the reference would have produced.
    """

    def __init__(self, buffer, count, size=None):
        self.buffer = buffer
        self.count = count
        self.size = tuple(size) if size is not None else None   # (H, W) of the map the indexes address

    # duck-typing for the places the reference touches the third tuple element
    @property
    def data(self):
        return self

    def contiguous(self):
        return self

    def dim(self):
        return 1

    def tensor(self):
        """The exact list as the reference passes it (one host sync for the length).  It is a VIEW of the
        module's index buffer, which the next frame overwrites: `.clone()` (or `self.clone()`) what must
        outlive the frame -- the reference returns a fresh tensor every frame."""
        n = int(self.count.item())
        return self.buffer[:n]

    def numel(self):
        return int(self.count.item())

    def __len__(self):
        return self.numel()

    def clone(self):
        return ChangeIndexes(self.buffer.clone(), self.count.clone(), self.size)


class MaskChangeIndexes(ChangeIndexes):
    """The change list of a frame whose contraction ran mask-driven (cbinfer_conv_changed_rows): the frame
    only left a copy of its change bit mask; the ascending index list and its device-side length are made
    from it by cbinfer_compact_bits the first time somebody asks (one launch on the current stream, no host
    sync) -- a frame none of whose consumers wants the list never pays for it.  Valid until the producing
    layer's next frame."""

    def __init__(self, maskCopy, size, idxBuffer, countBuffer, made=False):
        self._mask, self.size = maskCopy, tuple(size)
        self._idx, self._cnt = idxBuffer, countBuffer
        self._made = made       # (True: the frame's kernel already left list and count in the buffers)

    def _make(self):
        if not self._made:
            H, W = self.size
            check(C.cbinfer_compact_bits(ptr(self._mask), W, H, ptr(self._idx), ptr(self._cnt), None, None,
                                         stream_ptr(self._idx)))
            self._made = True

    @property
    def buffer(self):
        self._make()
        return self._idx

    @property
    def count(self):
        self._make()
        return self._cnt

    def clone(self):
        return ChangeIndexes(self.buffer.clone(), self.count.clone(), self.size)


def _split_indexes(changeIndexes):
    """-> (int32 buffer, device count tensor or None, capacity)."""
    if isinstance(changeIndexes, ChangeIndexes):
        return changeIndexes.buffer, changeIndexes.count, changeIndexes.buffer.numel()
    assert changeIndexes.dim() == 1 and changeIndexes.dtype == torch.int32
    idx = changeIndexes.contiguous()
    return idx, None, idx.numel()


def _check_nchw(t):
    assert t.dim() == 4 and t.size(0) == 1, "expected a [1,C,H,W] tensor"


# ------------------------------------------------------------------------------------------------
# a1 changeDetection  (reference: conv2d_cg.py:100-122)
# ------------------------------------------------------------------------------------------------
def changeDetection(input, prevInput, filtSize, threshold, updateInputState=False, useHalf=False):
    """Per-pixel any-channel |prevInput - input| > threshold, dilated by the filter support.
    Returns an int8 [H,W] map.  With updateInputState the changed pixels of prevInput are refreshed
    in place.  useHalf is implied by the tensor dtype and only checked."""
    require_device(input, prevInput)
    assert input.size() == prevInput.size() and input.dim() == 4
    _check_nchw(input)
    assert input.dtype == prevInput.dtype
    assert prevInput.is_contiguous(), "prevInput is updated in place and must be contiguous"
    if useHalf:
        assert input.dtype == torch.float16
    inp = input.contiguous()
    inC, inH, inW = inp.size(-3), inp.size(-2), inp.size(-1)
    changeMap = torch.empty(inH, inW, dtype=torch.int8, device=inp.device)
    check(C.cbinfer_change_detection(ptr(inp), ptr(prevInput), ptr(changeMap), inW, inH, inC,
                                     (filtSize[0] - 1) // 2, (filtSize[1] - 1) // 2, float(threshold),
                                     int(bool(updateInputState)), dtype_code(inp), stream_ptr(inp)))
    return changeMap


# ------------------------------------------------------------------------------------------------
# a2 changePropagation  (reference: conv2d_cg.py:159-177)
# ------------------------------------------------------------------------------------------------
def changePropagation(changeMap, filtSize):
    assert len(filtSize) == 2
    if filtSize[0] == 1 and filtSize[1] == 1:
        return changeMap  # no propagation for 1x1 filters
    require_device(changeMap)
    assert changeMap.dtype in (torch.int8, torch.uint8, torch.bool)
    cm = changeMap.contiguous()
    h, w = cm.size(-2), cm.size(-1)
    assert cm.numel() == h * w
    out = torch.empty_like(cm)
    check(C.cbinfer_change_propagation(ptr(cm), ptr(out), w, h, (filtSize[0] - 1) // 2,
                                       (filtSize[1] - 1) // 2, stream_ptr(cm)))
    return out


# ------------------------------------------------------------------------------------------------
# a3 changeIndexesExtr  (reference: conv2d_cg.py:200-213 -- torch.nonzero(...).int())
# ------------------------------------------------------------------------------------------------
def changeIndexesExtrAsync(changeMap):
    """Stream compaction of a byte map on the device; returns a ChangeIndexes (no host sync)."""
    require_device(changeMap)
    cm = changeMap.contiguous().view(-1)
    if cm.dtype == torch.bool:
        cm = cm.view(torch.int8)
    assert cm.dtype in (torch.int8, torch.uint8)
    numel = cm.numel()
    dev = cm.device
    scratch = torch.empty((numel + 63) // 64, dtype=torch.int64, device=dev)
    idx = torch.empty(numel, dtype=torch.int32, device=dev)
    count = torch.empty(1, dtype=torch.int32, device=dev)
    check(C.cbinfer_change_indexes_extr(ptr(cm), numel, ptr(scratch), ptr(idx), ptr(count),
                                        stream_ptr(cm)))
    return ChangeIndexes(idx, count)


def changeIndexesExtr(changeMap):
    """Exact-length ascending IntTensor of changed flat indices (synchronises, like torch.nonzero)."""
    return changeIndexesExtrAsync(changeMap).tensor()


# ------------------------------------------------------------------------------------------------
# a5 genXMatrix  (reference: conv2d_cg.py:239-261)
# ------------------------------------------------------------------------------------------------
def genXMatrix(input, changeIndexes, filtSize, useHalf=False):
    require_device(input)
    _check_nchw(input)
    inp = input.contiguous()
    inC, inH, inW = inp.size(-3), inp.size(-2), inp.size(-1)
    kH, kW = filtSize
    idx, count, cap = _split_indexes(changeIndexes)
    XMatrix = inp.new_empty((cap, inC * kH * kW))
    if cap > 0:
        check(C.cbinfer_gen_x_matrix(ptr(XMatrix), ptr(inp), ptr(idx), kW, kH, inC, inW, inH, cap,
                                     ptr(count), dtype_code(inp), stream_ptr(inp)))
    return XMatrix


# ------------------------------------------------------------------------------------------------
# a6/a7 the contraction
# ------------------------------------------------------------------------------------------------
_workspaces = {}


def newConvWorkspace(device):
    """A private, zero-initialised split-K workspace (tickets + slabs) for the persistent contraction
    kernel.  The kernels leave it zero.  Two launches that may run concurrently must not share one, so
    every CBConv2d owns its workspace (launches of ONE module are ordered by its state anyway); the
    pointer is then safe to bake into captured graphs and call plans whatever stream they replay on."""
    return torch.zeros(C.cbinfer_conv_workspace_bytes(), dtype=torch.uint8, device=device)


def convWorkspace(device):
    """Workspace for the OP-LEVEL entry points (convChanged without a module), one per (device, stream):
    launches on one stream are ordered.  Refuses to be created inside a stream capture -- the allocation
    and the fill would become graph nodes, and graphs captured on torch's shared default capture stream
    would end up sharing one workspace while replaying concurrently: pass `workspace=` instead."""
    if device.type == 'cuda':
        key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
    else:
        key = (device.type, device.index, 0)
    ws = _workspaces.get(key)
    if ws is None:
        if device.type == 'cuda' and torch.cuda.is_current_stream_capturing():
            raise CBinferError("convWorkspace: no workspace exists for the capturing stream; create one "
                               "with newConvWorkspace() before the capture and pass it as workspace=")
        ws = newConvWorkspace(device)
        _workspaces[key] = ws
    return ws


def prepWeights(weights, H=1, W=1, arith=None):
    """Pad the [K,C,kH,kW] filter bank to the MFMA tile grid (W[Kpad][CkkPad], k contiguous) and append
    the tap table for an H x W feature map.  arith: the library's arithmetic code when it differs from the
    tensor's dtype (CB_F32S: f32 weights pre-split into three bf16 planes for the bf16x3 contraction)."""
    require_device(weights)
    w = weights.detach().contiguous()
    K = w.size(0)
    if w.dim() == 4:
        Cin, kH, kW = w.size(1), w.size(2), w.size(3)
    else:
        Cin, kH, kW = w.numel() // K, 1, 1
    code = dtype_code(w) if arith is None else arith
    nbytes = C.cbinfer_prepared_weights_bytes(K, Cin, kH, kW, code)
    wp = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    check(C.cbinfer_prep_weights(ptr(w), ptr(wp), K, Cin, kH, kW, H, W, code, stream_ptr(w)))
    return wp


def matrixMult(Xmatrix, weights, bias, transposeOut=False, weightsPrepared=None, count=None):
    """Y = X . W.view(K,-1)^T + bias on MFMA (fp32: exact-f32 MFMA; fp16: f32 accumulation).
    Returns [N,K], or the contiguous transpose [K,N] with transposeOut (conv2d.py:247)."""
    require_device(Xmatrix, weights)
    if Xmatrix.numel() == 0:
        return Xmatrix.clone()
    X = Xmatrix.contiguous()
    K = weights.size(0)
    N, Ckk = X.size(0), X.size(1)
    assert weights.numel() == K * Ckk and X.dtype == weights.dtype
    wp = weightsPrepared if weightsPrepared is not None else prepWeights(weights)
    b = bias.detach().contiguous() if bias is not None else None
    Y = X.new_empty((K, N) if transposeOut else (N, K))
    check(C.cbinfer_matrix_mult(ptr(X), ptr(wp), ptr(b), ptr(Y), N, ptr(count), Ckk, K,
                                int(bool(transposeOut)), dtype_code(X), stream_ptr(X)))
    return Y


def matrixMult_python(Xmatrix, weights, bias, activFun=None):
    """The reference's torch formulation (conv2d_cg.py:342-349); on ROCm torch.matmul runs the vendor
    GEMM.  Kept for API parity and as the library-GEMM comparison point; the modules use matrixMult."""
    if Xmatrix.numel() == 0:
        return Xmatrix.clone()
    Ymatrix = Xmatrix.matmul(weights.view(weights.size(0), -1).transpose(0, 1)).add_(bias)
    if activFun is not None:
        Ymatrix = activFun(Ymatrix, inplace=True)
    return Ymatrix


# ------------------------------------------------------------------------------------------------
# a8 updateOutput  (reference: conv2d_cg.py:292-313)
# ------------------------------------------------------------------------------------------------
def updateOutput(YMatrix, changeIndexes, prevOutput, withReLU=False, useHalf=False):
    """prevOutput[0, :, y, x] <- relu?(YMatrix[:, n]) for every changed pixel n; in place.
    YMatrix is the [K,N] (possibly non-contiguous transposed view) result."""
    require_device(YMatrix, prevOutput)
    assert prevOutput.is_contiguous()
    outC, outH, outW = prevOutput.size(-3), prevOutput.size(-2), prevOutput.size(-1)
    idx, count, cap = _split_indexes(changeIndexes)
    if cap > 0:
        Yt = YMatrix.contiguous()
        assert Yt.size(0) == outC and Yt.size(1) == cap
        check(C.cbinfer_update_output(ptr(Yt), ptr(prevOutput), ptr(idx), outW * outH, cap, ptr(count),
                                      outC, int(bool(withReLU)), dtype_code(prevOutput),
                                      stream_ptr(prevOutput)))
    return prevOutput


# ------------------------------------------------------------------------------------------------
# a5..a8 fused
# ------------------------------------------------------------------------------------------------
def convChanged(input, changeIndexes, weights, bias, prevOutput, withReLU=False, accumulate=False,
                weightsPrepared=None, workspace=None, arith=None):
    """gather -> MFMA -> bias/ReLU -> scatter for the changed pixels in ONE launch; prevOutput is
    updated in place.  Equivalent to genXMatrix + matrixMult + transpose + updateOutput."""
    require_device(input, prevOutput, weights)
    _check_nchw(input)
    inp = input.contiguous()
    assert prevOutput.is_contiguous() and inp.dtype == prevOutput.dtype == weights.dtype
    K, Cin, kH, kW = weights.size()
    H, W = inp.size(-2), inp.size(-1)
    assert inp.size(1) == Cin and prevOutput.size(-3) == K
    idx, count, cap = _split_indexes(changeIndexes)
    if cap == 0:
        return prevOutput
    code = dtype_code(inp) if arith is None else arith
    wp = weightsPrepared if weightsPrepared is not None else prepWeights(weights, H, W, arith=code)
    b = bias.detach().contiguous() if bias is not None else None
    check(C.cbinfer_conv_changed(ptr(inp), ptr(idx), cap, ptr(count), ptr(wp), ptr(b), ptr(prevOutput),
                                 Cin, H, W, K, kH, kW, int(bool(withReLU)), int(bool(accumulate)), None,
                                 0, ptr(workspace if workspace is not None else convWorkspace(inp.device)),
                                 code, stream_ptr(inp)))
    return prevOutput


# ------------------------------------------------------------------------------------------------
# a9 maxPool2d  (reference: conv2d_cg.py:58-82)
# ------------------------------------------------------------------------------------------------
def maxPool2d(input, outputState, changeIndexes, kernelSize, stride, useHalf=False):
    assert len(kernelSize) == 2
    assert tuple(kernelSize) == tuple(stride)
    assert input.size(0) == 1
    require_device(input, outputState)
    sy, sx = stride
    assert sy == 2 and sx == 2
    inp = input.contiguous()
    assert outputState.is_contiguous() and inp.dtype == outputState.dtype
    nc, h, w = inp.size(-3), inp.size(-2), inp.size(-1)
    oh, ow = outputState.size(-2), outputState.size(-1)
    idx, count, cap = _split_indexes(changeIndexes)
    if cap > 0:
        check(C.cbinfer_max_pool2d(ptr(inp), ptr(outputState), ptr(idx), cap, ptr(count), nc, h, w, oh,
                                   ow, dtype_code(inp), stream_ptr(inp)))
    return outputState


def poolChangeIndexes(changeIndexes, inSize, outSize):
    """the change-index list of a 2x2/stride-2 pool's OUTPUT from that of its input --
    ascending, duplicate-free flat indexes yo*oW+xo of every window that holds a changed input pixel --
    as a ChangeIndexes (device-side count, no host sync).  inSize = (iH, iW), outSize = (oH, oW)."""
    idx, count, cap = _split_indexes(changeIndexes)
    require_device(idx)
    (iH, iW), (oH, oW) = inSize, outSize
    dev = idx.device
    words = C.cbinfer_mask_words(oH, oW)
    bits = torch.zeros(words, dtype=torch.int64, device=dev)
    out = torch.empty(oH * oW, dtype=torch.int32, device=dev)
    ocount = torch.zeros(1, dtype=torch.int32, device=dev)
    if cap > 0:
        check(C.cbinfer_pool_change_indexes(ptr(idx), cap, ptr(count), iW, oH, oW, ptr(bits), stream_ptr(idx)))
        check(C.cbinfer_compact_bits(ptr(bits), oW, oH, ptr(out), ptr(ocount), None, None, stream_ptr(idx)))
    return ChangeIndexes(out, ocount, (oH, oW))


def dilateChangeIndexes(changeIndexes, size, filtSize):
    """SURVEY 8f-4, consumers with k > 1: the list of OUTPUT pixels a kH x kW convolution must recompute when the
    listed INPUT pixels of an H x W map changed -- their filter supports, ascending and duplicate-free -- as a
    ChangeIndexes (device-side count, no host sync).  What changeDetection's dilation yields when every listed pixel
    changed; the reference hands the undilated list on (conv2d.py:180-190)."""
    idx, count, cap = _split_indexes(changeIndexes)
    require_device(idx)
    H, W = size
    dev = idx.device
    bits = torch.zeros(C.cbinfer_mask_words(H, W), dtype=torch.int64, device=dev)
    out = torch.empty(H * W, dtype=torch.int32, device=dev)
    ocount = torch.zeros(1, dtype=torch.int32, device=dev)
    if cap > 0:
        check(C.cbinfer_dilate_change_indexes(ptr(idx), min(cap, H * W), ptr(count), H, W, (filtSize[0] - 1) // 2,
                                              (filtSize[1] - 1) // 2, ptr(bits), stream_ptr(idx)))
        check(C.cbinfer_compact_bits(ptr(bits), W, H, ptr(out), ptr(ocount), None, None, stream_ptr(idx)))
    return ChangeIndexes(out, ocount, (H, W))


__all__ = ['ChangeIndexes', 'MaskChangeIndexes', 'convWorkspace', 'newConvWorkspace', 'changeDetection', 'changePropagation', 'changeIndexesExtr',
           'changeIndexesExtrAsync', 'genXMatrix', 'prepWeights', 'matrixMult', 'matrixMult_python',
           'updateOutput', 'convChanged', 'maxPool2d', 'poolChangeIndexes', 'dilateChangeIndexes', 'CBinferError']


class ChannelConcat(object):
    """torch.cat(tensors, dim=1) of batch-1 [1,Ci,H,W] tensors (poseDetection/openPose/PoseModel.py:131) as ONE library launch
    into a buffer this object keeps and reuses: the result has the same address every frame -- what a recorded launch
    program (pycbinfer.FrameProgram) and a consumer's call plan want -- and is overwritten by the next call."""

    def __init__(self):
        self.out, self._key, self._srcs, self._chans = None, None, None, None

    def __call__(self, tensors):
        tensors = [t.detach() for t in tensors]
        t0 = tensors[0]
        require_device(*tensors)
        n = len(tensors)
        if not 1 <= n <= 4 or any(t.dim() != 4 or t.size(0) != 1 or t.shape[2:] != t0.shape[2:] or t.dtype != t0.dtype or
                                  t.device != t0.device or not t.is_contiguous() for t in tensors):
            raise CBinferError("ChannelConcat: 1..4 contiguous [1,Ci,H,W] tensors of one size, dtype and device")
        key = (tuple(tuple(t.shape) for t in tensors), t0.dtype, t0.device)
        if self._key != key:
            C_ = sum(t.size(1) for t in tensors)
            self.out = torch.empty((1, C_) + tuple(t0.shape[2:]), dtype=t0.dtype, device=t0.device)
            self._srcs = (ctypes.c_void_p * n)()
            self._chans = (ctypes.c_int32 * n)(*[t.size(1) for t in tensors])
            self._key = key
        for k, t in enumerate(tensors):
            self._srcs[k] = t.data_ptr()
        check(C.cbinfer_concat_channels(self._srcs, self._chans, n, ptr(self.out), t0.size(2) * t0.size(3),
                                        dtype_code(t0), stream_ptr(t0)))
        return self.out
