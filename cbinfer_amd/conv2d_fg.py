"""Fine-grained (per-value delta) change-based convolution ops on MI355X.

Host-side mirror of pycbinfer/conv2d_fg.py: changeDetectionFG / updateOutputFG / cbconvFG with the
reference's names and argument meaning, launching the HIP kernels of libcbinfer_hip.so.
"""
import torch

from ._lib import C, check, ptr, require_device, stream_ptr
from .conv2d_cg import ChangeIndexes, changeIndexesExtrAsync, convChanged


def changeDetectionFG(input, prevInput, threshold, zeroUnchanged=False):
    """reference: conv2d_fg.py:34-46.  Returns (diffs, changeMap): diffs = input - prevInput where
    |diff| > threshold (elsewhere uninitialised, or 0 with zeroUnchanged), changeMap int8, same shape."""
    require_device(input, prevInput)
    assert input.is_contiguous() and input.dtype == torch.float32
    prev = prevInput.contiguous()
    assert prev.size() == input.size()
    diffs = torch.empty_like(input)
    changeMap = torch.empty(input.size(), dtype=torch.int8, device=input.device)
    check(C.cbinfer_change_detection_fg(ptr(input), ptr(prev), ptr(diffs), ptr(changeMap),
                                        input.numel(), float(threshold), int(bool(zeroUnchanged)),
                                        stream_ptr(input)))
    return diffs, changeMap


def changeCoordsExtrFG(changeMap):
    """a11: the coordinates of the changed VALUES -- ascending flat indexes into [C,H,W] -- as a
    ChangeIndexes (int32 list + device-side count) by the library's ballot/popcount-prefix compaction,
    where the reference blocks on torch.nonzero(changeTensor.view(-1)) (conv2d_fg.py:82).
    `.tensor().long().view(-1, 1)` is the reference's int64 [N,1] tensor (one sync)."""
    return changeIndexesExtrAsync(changeMap)


def updateOutputFG(diffs, weight, output, changeCoords):
    """reference: conv2d_fg.py:48-72.  changeCoords: flat coordinates into [C,H,W] of the changed values,
    either the reference's int64 tensor ([N] or [N,1], torch.nonzero of the flattened change map) or a
    ChangeIndexes from changeCoordsExtrFG (length read on the device: no host sync).  output is updated
    in place with f32 atomic adds (summation order unspecified, as in the reference)."""
    require_device(diffs, weight, output)
    assert output.is_contiguous()
    assert weight.dim() == 4
    numOut, numIn, kH, kW = weight.size()
    inH, inW = output.size(-2), output.size(-1)
    d, w = diffs.contiguous(), weight.detach().contiguous()
    if isinstance(changeCoords, ChangeIndexes):
        check(C.cbinfer_update_output_fg_list(ptr(d), ptr(w), ptr(output), ptr(changeCoords.buffer),
                                              changeCoords.buffer.numel(), ptr(changeCoords.count), numOut,
                                              numIn, inH, inW, kH, kW, stream_ptr(output)))
        return output
    require_device(changeCoords)
    assert changeCoords.dtype == torch.int64
    coords = changeCoords.contiguous()
    check(C.cbinfer_update_output_fg(ptr(d), ptr(w), ptr(output), ptr(coords), numOut, numIn, inH, inW,
                                     kH, kW, coords.size(0), stream_ptr(output)))
    return output


def cbconvFG(input, prevInput, output, weight, threshold):
    """reference: conv2d_fg.py:75-96.  output (= previous output, updated in place and returned)
    receives conv(weight, delta) for every input value whose change exceeds the threshold.  The device
    path is the reference's op sequence -- per-value detection, coordinate extraction, atomic scatter --
    with the coordinate count kept on the device (no torch.nonzero round trip)."""
    if input.is_cuda:
        deltaInput, changeTensor = changeDetectionFG(input, prevInput, threshold)
        return updateOutputFG(deltaInput, weight, output, changeCoordsExtrFG(changeTensor))
    # host tensors: the library's counterpart of the reference's compiled CPU routine
    # (cbconv2d_fg_backend.cu:81-112)
    assert input.dtype == torch.float32 and output.is_contiguous()
    inp, prev, w = input.contiguous(), prevInput.contiguous(), weight.detach().contiguous()
    C.cbinfer_conv2d_fg_cpu(inp.data_ptr(), prev.data_ptr(), output.data_ptr(), w.data_ptr(),
                            float(threshold), w.size(0), w.size(1), inp.size(-2), inp.size(-1),
                            w.size(2), w.size(3))
    return output


def cbconvFG_deterministic(input, prevInput, output, weight, threshold, weightsPrepared=None):
    """Same result as cbconvFG without atomics: the thresholded deltas are kept as a dense tensor
    (zeros where unchanged) and the fused gather->MFMA kernel ACCUMULATES conv(weight, delta) into
    the output pixels reached by a changed value.  Bitwise reproducible; no host sync."""
    from .conv2d_cg import changeDetection  # noqa: F401  (kept local to avoid import cycles)
    require_device(input, prevInput, output)
    deltaInput, changeTensor = changeDetectionFG(input, prevInput, threshold, zeroUnchanged=True)
    K, Cin, kH, kW = weight.size()
    H, W = input.size(-2), input.size(-1)
    # output pixels touched = dilation of the per-pixel any-channel change map by the filter support
    anyc = changeTensor.view(Cin, H * W).amax(dim=0).view(H, W).contiguous()
    from .conv2d_cg import changePropagation, changeIndexesExtrAsync
    touched = changePropagation(anyc, (kH, kW))
    idx = changeIndexesExtrAsync(touched)
    convChanged(deltaInput, idx, weight, None, output, withReLU=False, accumulate=True,
                weightsPrepared=weightsPrepared)
    return output


__all__ = ['changeDetectionFG', 'changeCoordsExtrFG', 'updateOutputFG', 'cbconvFG', 'cbconvFG_deterministic']
