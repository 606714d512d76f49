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
                 dtype=torch.float32, pattern='blocks'):
        assert H % block == 0 and W % block == 0
        self.H, self.W, self.C, self.block = H, W, C, block
        self.pattern = pattern
        self.cells = (H // block) * (W // block)
        self.nblocks = max(0, int(round(ratio * self.cells)))
        self.ratio = self.nblocks / float(self.cells)
        if pattern == 'region':
            # ONE rectangle with the frame's aspect ratio covering `ratio` of the pixels, moving to a new
            # random position every frame (a single moving object instead of scattered blocks)
            self.rh = max(1, int(round(H * ratio ** 0.5)))
            self.rw = max(1, int(round(ratio * H * W / self.rh)))
            self.ratio = self.rh * self.rw / float(H * W)
        self.device, self.dtype = device, dtype
        self.gen = torch.Generator(device='cpu')
        self.gen.manual_seed(seed)
        self.frame = torch.rand(1, C, H, W, generator=self.gen).to(device=device, dtype=dtype)

    def next(self):
        """Advance by one frame; returns a NEW tensor (the previous frame is left intact)."""
        f = self.frame.clone()
        if self.pattern == 'region':
            y0 = int(torch.randint(0, self.H - self.rh + 1, (1,), generator=self.gen))
            x0 = int(torch.randint(0, self.W - self.rw + 1, (1,), generator=self.gen))
            patch = torch.rand(1, self.C, self.rh, self.rw, generator=self.gen)
            f[:, :, y0:y0 + self.rh, x0:x0 + self.rw] = patch.to(device=self.device, dtype=self.dtype)
            self.frame = f
            return f
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


# ------------------------------------------------------------------------------------------------
# OpenPose (BASELINE.json configs[3]): network SHAPE of poseDetection/openPose/PoseModel.py:34-68 with
# T refinement stages, random weights.  Each entry is (C_in, C_out, k); 'P' is a 2x2/2 max pool.
# ------------------------------------------------------------------------------------------------
OPENPOSE_FRONT = [(3, 64, 3), (64, 64, 3), 'P', (64, 128, 3), (128, 128, 3), 'P',
                  (128, 256, 3), (256, 256, 3), (256, 256, 3), (256, 256, 3), 'P',
                  (256, 512, 3), (512, 512, 3), (512, 256, 3), (256, 128, 3)]


def _openpose_stage(stage, nOut):
    if stage == 1:
        return [(128, 128, 3), (128, 128, 3), (128, 128, 3), (128, 512, 1), (512, nOut, 1)]
    return [(185, 128, 7)] + [(128, 128, 7)] * 4 + [(128, 128, 1), (128, nOut, 1)]


def _seq(spec, relu_after_last):
    layers = []
    convs = [i for i, e in enumerate(spec) if e != 'P']
    for i, e in enumerate(spec):
        if e == 'P':
            layers.append(nn.MaxPool2d(2, 2, 0))
        else:
            ci, co, k = e
            layers.append(nn.Conv2d(ci, co, k, 1, k // 2))
            if relu_after_last or i != convs[-1]:
                layers.append(nn.ReLU(inplace=True))
    return nn.Sequential(*layers)


def _side_stream(device):
    """The side stream of the forked branches: one that verifiably overlaps with the caller's current stream
    (cbinfer_amd/streams.py; kept out of the module: streams do not pickle and belong to a device, not to a model)."""
    from .streams import side_stream
    return side_stream(device)


class OpenPoseModel(nn.Module):
    """Feature extractor `model0` + per stage t two branches `model{t}_1` (38 PAF maps) and
    `model{t}_2` (19 confidence maps); stages t >= 2 see cat(branch1, branch2, features) = 185
    channels (PoseModel.py:122-137)."""

    def __init__(self, T=2, seed=0, concurrentBranches=False, init='default', groupedBranches=False):
        """init='default': nn.Conv2d's own initialisation (kaiming_uniform with a = sqrt(5): activations shrink by ~0.4
        per layer, so on random weights a change dies out behind the fifth conv).  init='kaiming': variance-preserving
        kaiming_normal_(nonlinearity='relu') and zero biases -- activations keep their scale through all 36 layers, so
        a change at the input reaches every layer, as with trained weights (the LIVE network of bench.py / tools/)."""
        super(OpenPoseModel, self).__init__()
        # the two branches of a stage are independent: with concurrentBranches they are enqueued on two
        # HIP streams (fork/join per stage), so that their per-launch fixed costs overlap
        self.concurrentBranches = concurrentBranches
        # ... or (converted networks) walked in lockstep, every pair of layers ONE launch (pycbinfer.BranchGroup, round 6)
        self.groupedBranches = groupedBranches
        self.libraryConcat = False      # (set by hand: the stage inputs concatenated by the library, see forward)
        state = torch.random.get_rng_state()
        torch.manual_seed(seed)
        self.T = T
        self.model0 = _seq(OPENPOSE_FRONT, relu_after_last=True)
        for t in range(1, T + 1):
            setattr(self, 'model%d_1' % t, _seq(_openpose_stage(t, 38), relu_after_last=False))
            setattr(self, 'model%d_2' % t, _seq(_openpose_stage(t, 19), relu_after_last=False))
        if init == 'kaiming':
            for m in self.modules():
                if isinstance(m, nn.Conv2d):
                    nn.init.kaiming_normal_(m.weight, nonlinearity='relu')
                    nn.init.zeros_(m.bias)
        elif init != 'default':
            raise ValueError("OpenPoseModel: init must be 'default' or 'kaiming'")
        torch.random.set_rng_state(state)
        self.eval()

    def forward(self, x):
        feat = self.model0(x)
        cur = feat
        fork = getattr(self, 'concurrentBranches', False) and cur.is_cuda
        side = _side_stream(cur.device) if fork else None
        grouped = getattr(self, 'groupedBranches', False) and cur.is_cuda and not fork
        for t in range(1, self.T + 1):
            if grouped:
                outL, outS = self._branch_group(t)(cur)
            elif fork:
                main = torch.cuda.current_stream(cur.device)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    outS = getattr(self, 'model%d_2' % t)(cur)
                outL = getattr(self, 'model%d_1' % t)(cur)
                main.wait_stream(side)
            else:
                outL = getattr(self, 'model%d_1' % t)(cur)
                outS = getattr(self, 'model%d_2' % t)(cur)
            if t != self.T:
                if getattr(self, 'libraryConcat', False) and cur.is_cuda:
                    # (one library launch into a buffer that keeps its address: pycbinfer.ChannelConcat -- with it a frame of
                    #  the converted network makes library calls only and can be replayed by pycbinfer.FrameProgram)
                    cc = self.__dict__.setdefault('_concats', {})
                    if t not in cc:
                        from . import ChannelConcat
                        cc[t] = ChannelConcat()
                    cur = cc[t]([outL, outS, feat])
                else:
                    cur = torch.cat([outL, outS, feat], 1)
        return outL, outS

    def _branch_group(self, t):
        from .branches import BranchGroup
        pair = (getattr(self, 'model%d_1' % t), getattr(self, 'model%d_2' % t))
        groups = self.__dict__.setdefault('_branchGroups', {})
        g = groups.get(t)
        if g is None or g[0] is not pair[0] or g[1] is not pair[1]:      # (the sub-models may have been converted since)
            g = groups[t] = (pair[0], pair[1], BranchGroup(pair))
        return g[2]

    def submodelNames(self):
        return ['model0'] + ['model%d_%d' % (t, b) for t in range(1, self.T + 1) for b in (1, 2)]


def convertOpenPose(model, threshold=1e-2, feedbackLoop=False):
    """Convert every sub-model with pycbinfer.convert(), as poseDetection/modelConverter.py:20-24 does
    (only nn.Sequential containers are traversed by convert, so the custom module is converted per
    sub-model).  feedbackLoop=True: the reference's experiments 10/11 (modelConverter.py:84-86, "recursive
    mode", eval03.py:31-35): every CBConv2d refreshes its state at the changed pixels only instead of copying
    its whole input every frame.  Returns the same module object with its children replaced."""
    from . import CBConv2d
    for name in model.submodelNames():
        setattr(model, name, convert(getattr(model, name), threshold=threshold))
    if feedbackLoop:
        for m in model.modules():
            if type(m) is CBConv2d:
                m.feedbackLoop = True
    return model


def fuseOpenPoseDetections(model, enabled=True):
    """Execution-level (results unchanged): every chained CBConv2d pair inside the sub-models lets the PRODUCER's
    launch run the consumer's change detection (pycbinfer.fuseDetectionIntoProducer), and the feature extractor's last
    layer does so for the first layers of BOTH branches of stage 1, which consume its output directly
    (PoseModel.py:122-137) -- a link the module tree does not show (pycbinfer.linkConsumers)."""
    from . import CBConv2d, fuseDetectionIntoProducer, linkConsumers
    for name in model.submodelNames():
        fuseDetectionIntoProducer(getattr(model, name), enabled)
    convs = lambda seq: [m for m in seq.children() if type(m) is CBConv2d]
    last = convs(model.model0)[-1]
    heads = [convs(getattr(model, 'model1_%d' % b))[0] for b in (1, 2)]
    linkConsumers(last, heads if enabled else [])
    return model


def calibrateChangeRatio(converted, nextFrame, target=0.10, pairs=4, settle=8, finalSettle=24):
    """Per-layer thresholds that give every CBConv2d of the CONVERTED network a post-dilation change ratio of `target` on
    the video `nextFrame()` yields, found layer by layer in execution order IN the change-based network itself (a
    layer's threshold decides what the layers behind it see): layer L's threshold is bisected on the recorded
    |input - prevInput| maps of `pairs` consecutive frames, with the layers in front of it at their calibrated
    thresholds -- after `settle` frames: a change-based layer's outputs keep drifting for many frames after its
    threshold was raised (the longer a pixel was not recomputed, the larger its jump when it is, so the layer behind
    sees more) -- and itself and the layers behind it at threshold 0 (exact: they recompute every pixel whose input
    differs at all, so nothing is stale when their turn comes).  `finalSettle` more frames bring the whole network to
    its steady state.  A WORKLOAD generator for measurements (BASELINE.md budgets config 4 at 10 % of the dense work,
    SURVEY 8(d): "r = 10 % at every layer"), not the reference's accuracy-driven tuner
    (pycbinfer.tuneThresholdParameters).  Returns the thresholds; the network is left warm: continue the same video."""
    import os
    from . import CBConv2d
    convs = [m for m in converted.modules() if type(m) is CBConv2d]
    for m in convs:
        m.threshold = 0.0
    # (the recorded |input - prevInput| maps need every layer's state as its OWN detection finds it: while a producer's
    #  launch runs its consumer's detection (pycbinfer.fuseDetectionIntoProducer) that state is refreshed before the
    #  consumer's forward -- and this hook -- is reached.  The folding is decided per frame: off while calibrating.)
    saved = os.environ.get('CBINFER_NO_NEXTFOLD')
    os.environ['CBINFER_NO_NEXTFOLD'] = '1'
    try:
        return _calibrate(converted, convs, nextFrame, target, pairs, settle, finalSettle)
    finally:
        if saved is None:
            os.environ.pop('CBINFER_NO_NEXTFOLD', None)
        else:
            os.environ['CBINFER_NO_NEXTFOLD'] = saved


def _calibrate(converted, convs, nextFrame, target, pairs, settle, finalSettle):
    with torch.no_grad():
        converted(nextFrame())
        for m in convs:
            for _ in range(settle):
                converted(nextFrame())
            diffs = []

            def hook(mod, inp, diffs=diffs):
                x = inp[0]
                x = x[1] if isinstance(x, tuple) else x
                prev = mod.prevInput
                if torch.is_tensor(x) and torch.is_tensor(prev) and prev.shape == x.shape:
                    d = torch.nan_to_num((x.detach().float() - prev.float()).abs(), nan=0.0, posinf=3e38)
                    diffs.append(d.amax(dim=1, keepdim=True))
            h = m.register_forward_pre_hook(hook)
            try:
                for _ in range(pairs):
                    converted(nextFrame())
            finally:
                h.remove()
            D = torch.cat(diffs)
            k = m.kernel_size[0]

            def cover(th, D=D, k=k):
                chg = (D > th).float()
                if k > 1:
                    chg = torch.nn.functional.max_pool2d(chg, k, 1, k // 2)
                return float(chg.mean())
            fin = D[D < 1e38]
            lo, hi = 0.0, (float(fin.max()) + 1e-6) if fin.numel() else 1.0
            if cover(0.0) <= target:
                m.threshold = 0.0
            else:
                for _ in range(24):
                    mid = 0.5 * (lo + hi)
                    if cover(mid) > target:
                        lo = mid
                    else:
                        hi = mid
                m.threshold = hi
        for _ in range(finalSettle):
            converted(nextFrame())
    return [float(m.threshold) for m in convs]


def openPoseDenseOps(T, H, W):
    total = 0
    h, w = H, W
    for e in OPENPOSE_FRONT:
        if e == 'P':
            h, w = h // 2, w // 2
        else:
            total += 2 * e[0] * e[1] * e[2] * e[2] * h * w
    for t in range(1, T + 1):
        for nOut in (38, 19):
            for (ci, co, k) in _openpose_stage(t, nOut):
                total += 2 * ci * co * k * k * h * w
    return total
