"""Fine-grained (per-value delta) change-based convolution ops on MI355X.

Host-side mirror of pycbinfer/conv2d_fg.py: changeDetectionFG / updateOutputFG / cbconvFG with the
reference's names and argument meaning, launching the HIP kernels of libcbinfer_hip.so.
"""
import torch

from ._lib import C, check, ptr, require_device, stream_ptr
from .conv2d_cg import ChangeIndexes, convChanged


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


def updateOutputFG(diffs, weight, output, changeCoords):
    """reference: conv2d_fg.py:48-72.  changeCoords: int64 flat coordinates into [C,H,W]
    (torch.nonzero of the flattened change map, shape [N] or [N,1]).  output is updated in place
    with f32 atomic adds (summation order unspecified, as in the reference)."""
    require_device(diffs, weight, output, changeCoords)
    assert output.is_contiguous()
    assert weight.dim() == 4 and changeCoords.dtype == torch.int64
    numOut, numIn, kH, kW = weight.size()
    inH, inW = output.size(-2), output.size(-1)
    numChanges = changeCoords.size(0)
    coords = changeCoords.contiguous()
    check(C.cbinfer_update_output_fg(ptr(diffs.contiguous()), ptr(weight.detach().contiguous()),
                                     ptr(output), ptr(coords), numOut, numIn, inH, inW, kH, kW,
                                     numChanges, stream_ptr(output)))
    return output


def cbconvFG(input, prevInput, output, weight, threshold):
    """reference: conv2d_fg.py:75-96.  output (= previous output, updated in place and returned)
    receives conv(weight, delta) for every input value whose change exceeds the threshold."""
    if input.is_cuda:
        deltaInput, changeTensor = changeDetectionFG(input, prevInput, threshold)
        changeIdx = torch.nonzero(changeTensor.view(-1))   # device->host sync, as in the reference
        if changeIdx.numel() != 0:
            output = updateOutputFG(deltaInput, weight, output, changeIdx)
        return output
    # host tensors: the reference's compiled CPU routine (cbconv2d_fg_backend.cu:81-112), here the
    # race-free host function exported by the same library
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


__all__ = ['changeDetectionFG', 'updateOutputFG', 'cbconvFG', 'cbconvFG_deterministic']
