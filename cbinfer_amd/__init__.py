"""cbinfer_amd -- CBinfer's change-based convolution hot path, written for AMD MI355X (gfx950).

The public surface is the reference's `pycbinfer` package (pycbinfer/__init__.py): convert(),
clearMemory(), getStateTensors(), propChangeIndexesOf1x1(), tuneThresholdParameters() and the
CBConv2d / CBPoolMax2d modules.  `import pycbinfer` (the alias package at the repository root)
resolves to this package, including the `pycbinfer.conv2d.CBConv2d` class paths that pickled
reference models name.
"""
import torch
import torch.nn as nn

from . import _lib                      # loads libcbinfer_hip.so; raises ImportError if not built
from .conv2d import CBConv2d
from .conv2d import CBPoolMax2d
from .conv2d import CBTail1x1
from .conv2d_cg import ChangeIndexes, ChannelConcat
from .pipeline import FramePipeline
from .batch import SequenceBatch
from .branches import BranchGroup
from .program import FrameProgram

__version__ = "0.1.0"

VERBOSE = False   # the reference prints one line per substituted node; opt-in here


def _log(msg):
    if VERBOSE:
        print(msg)


def _rebuild(container, rule):
    """A new nn.Sequential holding the children of `container` under their old names, each passed
    through rule(child, before, after) -> the module to put in its place, or None to leave it out.
    `before` / `after` are the child's neighbours in the ORIGINAL container (None at the ends)."""
    kids = list(container.named_children())
    out = nn.Sequential()
    for pos, (name, child) in enumerate(kids):
        before = kids[pos - 1][1] if pos > 0 else None
        after = kids[pos + 1][1] if pos + 1 < len(kids) else None
        keep = rule(child, before, after)
        if keep is not None:
            out.add_module(name, keep)
    return out


def subsitute(node, threshold=1e-1, finegrained=False):
    """(module, replaced?): exactly torch.nn.Conv2d -- not a subclass -- becomes a CBConv2d sharing its
    parameters (reference: __init__.py:10-17; the spelling of the name is the reference's)."""
    if type(node) is not nn.Conv2d:
        return node, False
    _log('replacing conv2d')
    cb = CBConv2d(node, threshold)
    cb.finegrained = finegrained
    return cb, True


def convertRecur(m, ignoreList=[], threshold=1e-1, finegrained=False):
    """(converted nn.Sequential, anything changed?) for the children of `m` (reference: __init__.py:20-45):
    names preserved; nn.Dropout and the types of ignoreList dropped; nested nn.Sequential containers
    converted recursively -- as in the reference (:28) WITHOUT handing `finegrained` down; every other
    child goes through subsitute().  If anything changed the ReLUs are merged right away: the reference
    calls convert() on its result at this point (:43-44), a pass that finds nothing left to substitute
    and so amounts to mergeReLURecur()."""
    dropped = tuple(ignoreList) + (nn.Dropout,)
    outcomes = []

    def rule(child, before, after):
        if type(child) is nn.Sequential:
            child, hit = convertRecur(child, ignoreList, threshold)
        elif type(child) in dropped:
            _log('removing node %s' % (type(child),))
            child, hit = None, True
        else:
            child, hit = subsitute(child, threshold=threshold, finegrained=finegrained)
        outcomes.append(hit)
        return child

    mout = _rebuild(m, rule)
    changed = any(outcomes)
    if changed:
        mout = mergeReLURecur(mout)
    return mout, changed


def mergeReLURecur(m):
    """An nn.ReLU right behind a CBConv2d is absorbed into it (withReLU=True) and leaves the container,
    nested nn.Sequential containers included (reference: __init__.py:47-66)."""
    def rule(child, before, after):
        if type(child) is nn.Sequential:
            return mergeReLURecur(child)
        if type(child) is CBConv2d and type(after) is nn.ReLU:
            child.withReLU = True
        if type(child) is nn.ReLU and type(before) is CBConv2d:
            _log('merging ReLU layer')
            return None
        return child
    return _rebuild(m, rule)


def _sequentials(rootModule):
    return [m for m in rootModule.modules() if type(m) == nn.Sequential]


def propChangeIndexesOf1x1(rootModule):
    """Let every CBConv2d that feeds a 1x1 CBConv2d pass its change indexes on, so the 1x1 layer skips
    its own change detection (reference: __init__.py:68-77).

    Deliberate deviation: the reference compares the kernel_size TUPLE with the LIST [1,1] (:73), which
    is never equal, so there this function is a no-op and the applications set the flags by hand
    (sceneLabeling/modelLoader.py:43-44).  Here the comparison is done on values, i.e. the function
    does what its name and the hand-written code say."""
    for seq in _sequentials(rootModule):
        kids = list(seq.children())
        for producer, consumer in zip(kids[:-1], kids[1:]):
            if (type(producer) == CBConv2d and type(consumer) == CBConv2d and
                    tuple(consumer.kernel_size) == (1, 1)):
                _log('enabling propagation of change indexes for 1x1')
                producer.propChangeIndexes = True
    return rootModule


def insertCBPooling(rootModule, cloneOutput=True):
    """what sceneLabeling/modelLoader.py:62-78 does by hand for experiments 5/6, for any
    converted network: every 2x2/stride-2 nn.MaxPool2d that directly follows a CBConv2d inside an
    nn.Sequential becomes a CBPoolMax2d fed by that layer's change indexes (propChangeIndexes on the
    conv), so only the windows holding a changed pixel are pooled again.  A CBConv2d consuming the pool
    then needs its own input copy (copyInput) unless it runs in feedback mode.  Returns rootModule."""
    def _pair(v):
        return tuple(v) if isinstance(v, (tuple, list)) else (v, v)
    for seq in _sequentials(rootModule):
        names = list(seq._modules.keys())
        for a, b in zip(names[:-1], names[1:]):
            conv, pool = seq._modules[a], seq._modules[b]
            if (type(conv) == CBConv2d and type(pool) == torch.nn.MaxPool2d and
                    _pair(pool.kernel_size) == (2, 2) and _pair(pool.stride) == (2, 2) and
                    _pair(pool.padding) == (0, 0) and _pair(pool.dilation) == (1, 1)):
                _log('change-based pooling after %s' % a)
                conv.propChangeIndexes = True
                cb = CBPoolMax2d(pool)
                cb.cloneOutput = cloneOutput
                seq._modules[b] = cb
                nxt = names.index(b) + 1
                if nxt < len(names) and type(seq._modules[names[nxt]]) == CBConv2d:
                    consumer = seq._modules[names[nxt]]
                    if not consumer.feedbackLoop:
                        consumer.copyInput = True
    return rootModule


def fusePoolingIntoDetection(rootModule, enabled=True):
    """Execution-level fusion (no change of results): a CBPoolMax2d whose consumer inside the same
    nn.Sequential is a feedback-mode CBConv2d stops pooling and lets that layer compute the pooled values
    inside its change detection (CBPoolMax2d.lazy, cbinfer_cbconv2d_forward_pooled) -- one launch less
    per pool and frame.  The pooled map is then not materialised, so the pool must not hand its indexes
    on (propChangeIndexes) and nothing else may read its outputState.  Returns rootModule."""
    for seq in _sequentials(rootModule):
        kids = list(seq.children())
        for pool, consumer in zip(kids[:-1], kids[1:]):
            if type(pool) == CBPoolMax2d:
                # (a feedback-mode layer, or -- round 4 -- a fine-grained one in its in-place form: its split-state frame
                #  takes the pool's input as it is, cbinfer_split_forward_fg; other fine-grained frames pool first)
                pool.lazy = bool(enabled and type(consumer) == CBConv2d and not pool.propChangeIndexes and
                                 ((consumer.feedbackLoop and not consumer.finegrained) or
                                  (consumer.finegrained and consumer.fgInPlace) or
                                  # (a layer that keeps a copy of its input: the copy-all detection of the split-state
                                  #  kernels takes the pool; where those do not take the layer it pools densely)
                                  (consumer.copyInput and not consumer.feedbackLoop and not consumer.finegrained)))
    return rootModule


def fuseDetectionIntoProducer(rootModule, enabled=True, windowOrder='auto'):
    """Execution-level fusion (no change of results): inside every nn.Sequential a run  CBConv2d (producer) ->
    lazy CBPoolMax2d -> feedback-mode CBConv2d (consumer)  lets the PRODUCER's contraction launch also be the consumer's
    pooled change detection when the producer runs on the row-pair kernel (cb_rowpair.hip: a workgroup owns whole 2x2
    windows) and the consumer on the split-state kernels: one launch less per frame, and the freshly computed outputs
    are compared with the consumer's state while they are still in the workgroup's LDS.  Decided per frame
    (CBConv2d._next_detect): the first frames of a sequence, a frame after the consumer's state was restored or its
    threshold changed run the separate detection.  Call after fusePoolingIntoDetection.  Returns rootModule.

    Round 6: a producer that runs on the split-state kernels itself (<= 64 output channels, a shallow contraction -- the
    16 -> 64 layer of the scene-labeling network) does the same with its contraction in pooling-WINDOW order
    (cbinfer_split_conv_next).  `windowOrder`: True, False (such a producer stays in pixel order, the consumer's detection in
    a launch of its own) or 'auto' (default): the window-order tile holds 16 whole windows -- the unchanged pixels of a
    touched window cost slots -- and its prologue and epilogue are longer, which pays (measured at 160 x 240: 1.7-2.8 us per
    frame from 2 to 35 % recomputed pixels) while its tiles fit ONE round of the grid and costs 2-3 us per round beyond; 'auto'
    decides that when the layer's call plan is made, from the change count of the layer's last frame (CBConv2d._window_fold)."""
    for seq in _sequentials(rootModule):
        kids = list(seq.children())
        for prod, pool, cons in zip(kids[:-2], kids[1:-1], kids[2:]):
            if type(prod) == CBConv2d and type(pool) == CBPoolMax2d and type(cons) == CBConv2d:
                if enabled and getattr(pool, 'lazy', False) and cons.feedbackLoop and prod.feedbackLoop:
                    prod.__dict__['_fusedNext'] = (pool, cons)      # (plain references, not children)
                    prod.__dict__['_winFold'] = windowOrder
                else:
                    prod.__dict__.pop('_fusedNext', None)
        # round 6: a CBConv2d directly behind another one (fp16 layers on the split-state machinery, the consumer in copy
        # mode -- what convert() makes): the producer's contraction launch compares the values it just wrote with the
        # consumer's state, refreshes that state and ORs the consumer's change mask; the consumer runs its contraction
        # alone.  Whether a given frame may do so is decided per frame (CBConv2d._fill_consumers / _detected_upstream).
        for prod, cons in zip(kids[:-1], kids[1:]):
            if type(prod) == CBConv2d and type(cons) == CBConv2d:
                linkConsumers(prod, [cons] if enabled else [])
    return rootModule


def linkConsumers(producer, consumers):
    """Name the CBConv2d layers that consume `producer`'s output tensor directly (same resolution) where the module tree
    does not say so -- e.g. the first layers of the two branches of an OpenPose stage, both fed the feature extractor's
    last output (poseDetection/openPose/PoseModel.py:122-137).  Their change detection then rides in the producer's
    launch whenever that is exact (see fuseDetectionIntoProducer); at most two consumers per producer are folded.  An
    empty list removes the link.  Execution-level only: results do not change."""
    consumers = [c for c in consumers if type(c) == CBConv2d]
    if consumers:
        producer.__dict__['_fusedConsumers'] = list(consumers)      # (plain references, not children)
    else:
        producer.__dict__.pop('_fusedConsumers', None)
    return producer


_STATEFUL = (CBConv2d, CBPoolMax2d, CBTail1x1)


def _stateful(net):
    return [m for m in net.modules() if type(m) in _STATEFUL]


def clearMemory(net):
    """Drop the frame-to-frame state of every change-based module (reference: __init__.py:79-82)."""
    for m in _stateful(net):
        m.clearMemory()


def getStateTensors(net):
    """Flat list of the state tensors of every change-based module (reference: __init__.py:84-89)."""
    return [t for m in _stateful(net) for t in m.getStateTensors()]


def convert(m, ignoreList=[], threshold=1e-1):
    """nn.Sequential in -> nn.Sequential out with every Conv2d replaced by a CBConv2d sharing its
    weights, ReLUs merged, Dropout removed (reference: __init__.py:91-94)."""
    converted, _ = convertRecur(m, ignoreList=ignoreList, threshold=threshold)
    return mergeReLURecur(converted)


def setSyncIndexes(net, enabled):
    """Switch every CBConv2d of `net` between the sync-free frame pipeline (False, default) and the
    reference-structured op sequence that blocks on the change count (True)."""
    for m in net.modules():
        if type(m) == CBConv2d:
            m.syncIndexes = bool(enabled)
    return net


def fuseTail1x1(rootModule, enabled=True):
    """Execution-level fusion (results within the fp32 bar of the unfused network): inside every
    nn.Sequential the run  CBConv2d -> nn.Conv2d 1x1 -> nn.ReLU -> nn.Conv2d 1x1  (the dense tail that
    sceneLabeling/modelLoader.py:45-47 keeps for experiments 2-7) becomes  CBConv2d -> CBTail1x1 : the
    two 1x1 layers are evaluated in ONE launch, only at the pixels of the CBConv2d's change list (its
    propChangeIndexes is switched on; a fine-grained head hands on the output pixels its frame touched) -- a
    1x1 layer's output changes only where its input did.  The new
    module takes the first 1x1 layer's name and shares the parameters of both.  enabled=False is a no-op.
    Returns rootModule."""
    if not enabled:
        return rootModule

    def is1x1(m):
        return type(m) is nn.Conv2d and CBTail1x1.accepts(m)

    for seq in _sequentials(rootModule):
        names = list(seq._modules.keys())
        kids = [seq._modules[n] for n in names]
        for i in range(len(kids) - 3):
            head, a, act, b = kids[i:i + 4]
            if (type(head) is CBConv2d and is1x1(a) and type(act) is nn.ReLU and
                    is1x1(b) and a.out_channels == b.in_channels and
                    CBTail1x1.supported(a.in_channels, a.out_channels, b.out_channels) and
                    a.weight.dtype == torch.float32):
                _log('fusing the 1x1 tail behind %s' % names[i])
                head.propChangeIndexes = True
                tail = CBTail1x1(a, b, relu=True)
                seq._modules[names[i + 1]] = tail
                head.__dict__['_fusedTail'] = tail      # (a plain reference, not a child: see CBConv2d._folded_tail)
                del seq._modules[names[i + 2]]
                del seq._modules[names[i + 3]]
                break
    return rootModule


_THRESHOLD_CEILING = 1e30   # a tolerance that is never exceeded must not loop forever (the reference would)


def tuneThresholdParameters(vidSeqReader, evalSequences, numFramesPerSeq, targetGenerator,
                            preprocessor, modelBaseline, modelTest, evaluator, cbModuleList,
                            lossToleranceList, initThreshold=1e-2, thresholdIncrFactor=1.2):
    """Greedy front-to-back per-layer threshold search with the reference's callback protocol
    (reference: __init__.py:98-149).  Modules are visited in list order; a module's threshold climbs
    from initThreshold by thresholdIncrFactor per step for as long as the summed evaluator loss over
    evalSequences stays within the module's tolerance of the loss measured before the module was
    touched, and the last level that did is kept (the reference's "one step back").  As there,
    initThreshold itself is never evaluated.  The loss the next module is judged against is re-measured
    with the kept level."""
    tolerances = lossToleranceList if type(lossToleranceList) is list \
        else [lossToleranceList] * len(cbModuleList)
    assert len(cbModuleList) == len(tolerances)
    device = next(modelTest.parameters()).device

    def sequenceLoss(seqName):
        frames, target = vidSeqReader.getDataFrames(seqName=seqName, numFrames=numFramesPerSeq)
        if target is None:
            target = targetGenerator(frames[-1])
        with torch.no_grad():
            for frame in frames:
                prediction = modelTest(preprocessor(frame).to(device))
        return evaluator(prediction, target)

    def measure():
        clearMemory(modelTest)
        return sum(sequenceLoss(name) for name in evalSequences)

    anchor = measure()
    for pos, (module, tolerance) in enumerate(zip(cbModuleList, tolerances)):
        _log('adjusting threshold for module %d of %d' % (pos + 1, len(cbModuleList)))
        kept = initThreshold
        while kept * thresholdIncrFactor < _THRESHOLD_CEILING:
            module.threshold = kept * thresholdIncrFactor
            loss = measure()
            _log('. (%f < %f + %f)' % (loss, anchor, tolerance))
            if loss - anchor > tolerance:
                break
            kept = module.threshold
        module.threshold = kept
        anchor = measure()


__all__ = ['CBConv2d', 'CBPoolMax2d', 'CBTail1x1', 'ChangeIndexes', 'ChannelConcat', 'FramePipeline', 'SequenceBatch', 'BranchGroup', 'FrameProgram', 'convert', 'convertRecur', 'subsitute',
           'mergeReLURecur', 'propChangeIndexesOf1x1', 'insertCBPooling', 'fusePoolingIntoDetection',
           'fuseDetectionIntoProducer', 'linkConsumers', 'fuseTail1x1',
           'clearMemory', 'getStateTensors',
           'setSyncIndexes', 'tuneThresholdParameters']
