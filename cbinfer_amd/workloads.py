"""Workload definitions for tests and benchmarks: the scene-labeling CNN the reference's experiments
run (shape only -- its trained weights are an external download), the experiment presets that
sceneLabeling/modelLoader.py applies to it, and the synthetic static-camera video of BASELINE.md.

Nothing here is on the hot path; it only builds the modules and inputs that exercise it.
"""
import torch
import torch.nn as nn

from . import CBConv2d, CBPoolMax2d, convert

# (C_in, C_out, k) per conv; pools after conv 1 and 2.  SURVEY.md 8a note 1: inferred from the module
# indexing in sceneLabeling/modelLoader.py:47,65,72-78 and the papers; kept as data so another spec
# can be substituted.
SCENE_LABELING_SPEC = dict(
    convs=[(3, 16, 7), (16, 64, 7), (64, 256, 7), (256, 64, 1), (64, 8, 1)],
    pools_after=(0, 1))


def sceneLabelingBaseline(spec=SCENE_LABELING_SPEC, seed=0, ceil_mode=False):
    """The 11-module baseline nn.Sequential: conv,ReLU,pool, conv,ReLU,pool, conv,ReLU, conv,ReLU, conv
    with default-initialised weights (seeded)."""
    gen_state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    layers = []
    n = len(spec['convs'])
    for i, (ci, co, k) in enumerate(spec['convs']):
        layers.append(nn.Conv2d(ci, co, k, padding=k // 2))
        if i < n - 1:
            layers.append(nn.ReLU())
        if i in spec['pools_after']:
            layers.append(nn.MaxPool2d(2, 2, ceil_mode=ceil_mode))
    torch.random.set_rng_state(gen_state)
    return nn.Sequential(*layers).eval()


def denseOps(spec, H, W):
    """Dense operation count of all convs (2*C*K*kH*kW*H*W each; conv2d.py:216 'totalInputValues')."""
    total, h, w = 0, H, W
    for i, (ci, co, k) in enumerate(spec['convs']):
        total += 2 * ci * co * k * k * h * w
        if i in spec['pools_after']:
            h, w = h // 2, w // 2
    return total


def configureExperiment(modelBaseline, modelConverted, experimentIdx):
    """Apply the module-flag surgery of sceneLabeling/modelLoader.py:41-87 to a converted
    scene-labeling net (children 0,2,3,5,6,8,10 -> positions 0..6).  Returns the test model.

      1 change-index propagation into the two 1x1 layers        (:41-44)
      2 no CBinfer for the 1x1 layers (baseline modules 8..10)  (:45-47)
      3 = 2 + copyInput=False                                   (:48-52)
      4 = 3 + feedbackLoop                                      (:54-61)
      5/6 = 4 + change-based pooling with index propagation     (:62-78)
      7 = 3 + fine-grained convolution                          (:79-87)
    """
    mt = modelConverted
    if experimentIdx == 1:
        mt[4].propChangeIndexes = True
        mt[5].propChangeIndexes = True
        return mt
    mt = nn.Sequential(*[mt[i] for i in range(5)] + [modelBaseline[i] for i in range(8, 11)])
    cbs = [m for m in mt.modules() if type(m) == CBConv2d]
    if experimentIdx == 2:
        return mt
    for m in cbs:
        m.copyInput = False
    if experimentIdx == 3:
        return mt
    if experimentIdx == 7:
        for m in cbs:
            m.finegrained = True
        return mt
    for m in cbs:
        m.feedbackLoop = True
    if experimentIdx == 4:
        return mt
    assert experimentIdx in (5, 6)
    mt[0].propChangeIndexes = True
    pool1 = CBPoolMax2d(mt[1])
    mt[2].copyInput = True
    mt[2].propChangeIndexes = True
    pool3 = CBPoolMax2d(mt[3])
    mt[4].copyInput = True
    return nn.Sequential(*[mt[0], pool1, mt[2], pool3] + list(mt.children())[4:])


def sceneLabelingModels(experimentIdx=6, threshold=0.05, device='cuda', dtype=torch.float32, seed=0):
    """(baseline, change-based test model) as the reference's loadModel() returns them, with random
    weights and one fixed threshold for every layer."""
    base = sceneLabelingBaseline(seed=seed).to(device=device, dtype=dtype)
    test = convert(base, threshold=threshold)
    test = configureExperiment(base, test, experimentIdx)
    return base, test.to(device)


class SyntheticVideo(object):
    """Static-camera video stand-in: frame 0 ~ U[0,1); frame t+1 = frame t
    with `nblocks` non-overlapping block x block regions (all channels) re-drawn from U[0,1).  The
    regions are cells of the block grid, re-chosen every frame from a seeded generator, so the changed
    fraction of pixels is exactly nblocks*block^2/(H*W) every frame."""

    def __init__(self, H=320, W=480, C=3, ratio=0.10, block=32, seed=1234, device='cuda',
                 dtype=torch.float32):
        assert H % block == 0 and W % block == 0
        self.H, self.W, self.C, self.block = H, W, C, block
        self.cells = (H // block) * (W // block)
        self.nblocks = max(0, int(round(ratio * self.cells)))
        self.ratio = self.nblocks / float(self.cells)
        self.device, self.dtype = device, dtype
        self.gen = torch.Generator(device='cpu')
        self.gen.manual_seed(seed)
        self.frame = torch.rand(1, C, H, W, generator=self.gen).to(device=device, dtype=dtype)

    def next(self):
        """Advance by one frame; returns a NEW tensor (the previous frame is left intact)."""
        f = self.frame.clone()
        b = self.block
        gw = self.W // b
        cells = torch.randperm(self.cells, generator=self.gen)[:self.nblocks].tolist()
        for c in cells:
            y0, x0 = (c // gw) * b, (c % gw) * b
            patch = torch.rand(1, self.C, b, b, generator=self.gen)
            f[:, :, y0:y0 + b, x0:x0 + b] = patch.to(device=self.device, dtype=self.dtype)
        self.frame = f
        return f

    def frames(self, T):
        out = [self.frame]
        for _ in range(T - 1):
            out.append(self.next())
        return out
