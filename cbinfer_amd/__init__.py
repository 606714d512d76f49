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
from .conv2d_cg import ChangeIndexes

__version__ = "0.1.0"

VERBOSE = False   # the reference prints one line per substituted node; opt-in here


def _log(msg):
    if VERBOSE:
        print(msg)


def subsitute(node, threshold=1e-1, finegrained=False):
    """Replace exactly torch.nn.Conv2d by CBConv2d (reference: __init__.py:10-17; the spelling of the
    name is the reference's)."""
    if type(node) is torch.nn.modules.conv.Conv2d:
        _log('replacing conv2d')
        m = CBConv2d(node, threshold)
        m.finegrained = finegrained
        return m, True
    return node, False


def convertRecur(m, ignoreList=[], threshold=1e-1, finegrained=False):
    """reference: __init__.py:20-45.  Child names are preserved; nn.Dropout and ignoreList types are
    dropped; nn.Sequential children are converted recursively.  As in the reference `finegrained` is
    not forwarded into nested containers (:28)."""
    changed = False
    mout = nn.Sequential()
    for nodeName, node in m.named_children():
        if type(node) in [nn.Sequential]:
            sub, c = convertRecur(node, ignoreList, threshold)
            mout.add_module(nodeName, sub)
            changed |= c
        elif type(node) in list(ignoreList) + [nn.Dropout]:
            _log('removing node %s' % (type(node),))
            changed = True
            continue
        else:
            nodeOut, newNode = subsitute(node, threshold=threshold, finegrained=finegrained)
            mout.add_module(nodeName, nodeOut)
            changed |= newNode
    if changed:
        # the reference runs convert() once more until nothing changes (:43-44); that second pass
        # finds no Conv2d left and only re-merges ReLUs
        mout = convert(mout, ignoreList)
    return mout, changed


def mergeReLURecur(m):
    """A CBConv2d directly followed by nn.ReLU absorbs it (withReLU=True) (reference: :47-66)."""
    mout = nn.Sequential()
    children = list(m.children())
    for i, (nodeName, node) in enumerate(m.named_children()):
        if type(node) in [nn.Sequential]:
            mout.add_module(nodeName, mergeReLURecur(node))
            continue
        elif type(node) in [CBConv2d]:
            if len(children) > i + 1 and type(children[i + 1]) is torch.nn.modules.activation.ReLU:
                node.withReLU = True
        elif (type(node) is torch.nn.modules.activation.ReLU and i >= 1 and
              type(children[i - 1]) is CBConv2d):
            _log('merging ReLU layer')
            continue
        mout.add_module(nodeName, node)
    return mout


def propChangeIndexesOf1x1(rootModule):
    """Let every CBConv2d that feeds a 1x1 CBConv2d pass its change indexes on, so the 1x1 layer skips
    its own change detection (reference: __init__.py:68-77).

    Deliberate deviation: the reference compares the kernel_size TUPLE with the LIST [1,1] (:73), which
    is never equal, so there this function is a no-op and the applications set the flags by hand
    (sceneLabeling/modelLoader.py:43-44).  Here the comparison is done on values, i.e. the function
    does what its name and the hand-written code say."""
    seqContainers = [m for m in rootModule.modules() if type(m) == torch.nn.Sequential]
    for seqCont in seqContainers:
        mPrev = None
        for m in seqCont:
            if (type(m) == CBConv2d and type(mPrev) == CBConv2d and
                    tuple(m.kernel_size) == (1, 1)):
                _log('enabling propagation of change indexes for 1x1')
                mPrev.propChangeIndexes = True
            mPrev = m
    return rootModule


def insertCBPooling(rootModule, cloneOutput=True):
    """what sceneLabeling/modelLoader.py:62-78 does by hand for experiments 5/6, for any
    converted network: every 2x2/stride-2 nn.MaxPool2d that directly follows a CBConv2d inside an
    nn.Sequential becomes a CBPoolMax2d fed by that layer's change indexes (propChangeIndexes on the
    conv), so only the windows holding a changed pixel are pooled again.  A CBConv2d consuming the pool
    then needs its own input copy (copyInput) unless it runs in feedback mode.  Returns rootModule."""
    def _pair(v):
        return tuple(v) if isinstance(v, (tuple, list)) else (v, v)
    for seq in [m for m in rootModule.modules() if type(m) == torch.nn.Sequential]:
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
    for seq in [m for m in rootModule.modules() if type(m) == torch.nn.Sequential]:
        kids = list(seq.children())
        for pool, consumer in zip(kids[:-1], kids[1:]):
            if type(pool) == CBPoolMax2d:
                pool.lazy = bool(enabled and type(consumer) == CBConv2d and consumer.feedbackLoop and
                                 not consumer.finegrained and not pool.propChangeIndexes)
    return rootModule


def clearMemory(net):
    for m in net.modules():
        if type(m) == CBConv2d or type(m) == CBPoolMax2d:
            m.clearMemory()


def getStateTensors(net):
    state = []
    for m in net.modules():
        if type(m) == CBConv2d or type(m) == CBPoolMax2d:
            state += m.getStateTensors()
    return state


def convert(m, ignoreList=[], threshold=1e-1):
    """nn.Sequential in -> nn.Sequential out with every Conv2d replaced by a CBConv2d sharing its
    weights, ReLUs merged, Dropout removed (reference: __init__.py:91-94)."""
    m1, changed = convertRecur(m, ignoreList=ignoreList, threshold=threshold)
    return mergeReLURecur(m1)


def setSyncIndexes(net, enabled):
    """Switch every CBConv2d of `net` between the sync-free frame pipeline (False, default) and the
    reference-structured op sequence that blocks on the change count (True)."""
    for m in net.modules():
        if type(m) == CBConv2d:
            m.syncIndexes = bool(enabled)
    return net


def tuneThresholdParameters(vidSeqReader, evalSequences, numFramesPerSeq, targetGenerator,
                            preprocessor, modelBaseline, modelTest, evaluator, cbModuleList,
                            lossToleranceList, initThreshold=1e-2, thresholdIncrFactor=1.2):
    """Greedy front-to-back per-layer threshold search with the reference's callback protocol
    (reference: __init__.py:98-149): for each CB module in turn raise its threshold by
    thresholdIncrFactor while the loss increase over the previous level stays within the module's
    tolerance, then step back once."""
    if type(lossToleranceList) is not list:
        lossToleranceList = [lossToleranceList] * len(cbModuleList)
    assert len(cbModuleList) == len(lossToleranceList)
    device = next(modelTest.parameters()).device

    def evaluateModel():
        clearMemory(modelTest)
        totalLoss = 0
        for seqName in evalSequences:
            frames, target = vidSeqReader.getDataFrames(seqName=seqName, numFrames=numFramesPerSeq)
            if target is None:
                target = targetGenerator(frames[-1])
            with torch.no_grad():
                for frame in frames:
                    outTest = modelTest(preprocessor(frame).to(device))
            totalLoss += evaluator(outTest, target)
        return totalLoss

    prevLoss = evaluateModel()
    for i, m in enumerate(cbModuleList):
        _log('adjusting threshold for module %d of %d' % (i + 1, len(cbModuleList)))
        m.threshold = initThreshold
        while True:
            m.threshold *= thresholdIncrFactor
            loss = evaluateModel()
            _log('. (%f < %f + %f)' % (loss, prevLoss, lossToleranceList[i]))
            # (guard absent in the reference: a tolerance that is never exceeded would loop forever)
            if loss - prevLoss > lossToleranceList[i] or not (m.threshold < 1e30):
                m.threshold /= thresholdIncrFactor
                break
        prevLoss = evaluateModel()


__all__ = ['CBConv2d', 'CBPoolMax2d', 'ChangeIndexes', 'convert', 'convertRecur', 'subsitute',
           'mergeReLURecur', 'propChangeIndexesOf1x1', 'insertCBPooling', 'fusePoolingIntoDetection',
           'clearMemory', 'getStateTensors',
           'setSyncIndexes', 'tuneThresholdParameters']
