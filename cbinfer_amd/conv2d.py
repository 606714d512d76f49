"""CBConv2d / CBPoolMax2d: change-based convolution and pooling modules on MI355X.

Same torch.nn.Module surface as the reference (pycbinfer/conv2d.py): constructor arguments, the
attribute flags set from outside (threshold, withReLU, saveChangeMap, propChangeIndexes,
gatherComputationStats, finegrained, copyInput, feedbackLoop), the registered state buffers
(prevInput, prevOutput, outputState), clearMemory()/getStateTensors(), the
('changeIndexes', output, indexes) tuple protocol between modules, and the returned tensor aliasing
the layer state.  What differs is underneath: a frame of a layer is ONE call into libcbinfer_hip.so
that enqueues detection -> compaction -> fused gather/MFMA/scatter on torch's current stream with the
changed-pixel count kept on the device, so a whole network frame runs without a host round trip and
can be captured into a hipGraph (torch.cuda.CUDAGraph).

Two execution modes, chosen per module by `syncIndexes` (default False):
  False: sync-free.  Change lists travel as `ChangeIndexes` (capacity buffer + device count).
  True : like the reference, block on the count after compaction; change lists are exact IntTensors
         and the reference-structured op sequence (changeDetection, changeIndexesExtr, genXMatrix,
         matrixMult, updateOutput) of conv2d_cg.py is what runs.
"""
import ctypes
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from ._lib import C, check, dtype_code, ptr, raw_stream, require_device, stream_ptr
from .conv2d_cg import (ChangeIndexes, MaskChangeIndexes, changeDetection, changeIndexesExtr, dilateChangeIndexes,
                        genXMatrix, matrixMult, maxPool2d, newConvWorkspace, poolChangeIndexes, prepWeights, updateOutput)
from .conv2d_fg import cbconvFG, cbconvFG_deterministic


def _same_shape(t, shape):
    return t is not None and tuple(t.size()) == tuple(shape)


_NO_CHAIN = os.environ.get('CBINFER_NO_CHAIN', '0') == '1'


def _no_tag():
    return None


class _Produced(object):
    """What a CBConv2d leaves on its output buffer for the next layer (CBConv2d._note_upstream): who wrote it, in which
    of its frames, the buffer's version counter then, and where the frame's change count is.  Never travels: a pickled
    or deep-copied tensor carries None instead."""
    __slots__ = ('module', 'serial', 'version', 'count', 'tokens')

    def __init__(self, module, serial, version, count, tokens=()):
        self.module, self.serial, self.version, self.count = module, serial, version, count
        self.tokens = tokens      # (CBConv2d._half_detect_token of every consumer whose detection rode in the launch)

    def __reduce__(self):
        return _no_tag, ()

    def __deepcopy__(self, memo):
        return None


def _flush_side(pend):
    """The feedback refresh of a row-pair layer's state as a launch of its own (pend: frame, state, C, H, W, threshold)."""
    frame, state, Cn, H, W, th = pend
    check(C.cbinfer_refresh_state(ptr(frame), ptr(state), Cn, H, W, float(th), stream_ptr(frame)))


class LazyPool(object):
    """What a CBPoolMax2d with lazy=True hands to the next module instead of a pooled tensor: the pool's
    INPUT and the pooled size.  A feedback-mode CBConv2d folds the pooling into its change detection
    (cbinfer_cbconv2d_forward_pooled) and never needs the pooled map; anything else calls tensor()."""

    def __init__(self, source, outSize, ceil_mode, indexes=None):
        self.source = source
        self.outSize = tuple(outSize)
        self.ceil_mode = ceil_mode
        self.indexes = indexes      # the producer's change indexes of this frame (pre-pool resolution)

    def producerMask(self):
        """The producing layer's change bit mask of this frame, if it ran a mask-driven contraction: the
        consumer's pooled detection then only looks where that layer rewrote something."""
        ix = self.indexes
        if isinstance(ix, MaskChangeIndexes) and ix.size == tuple(self.source.shape[-2:]):
            return ix._mask
        return None

    def tensor(self):
        return F.max_pool2d(self.source, 2, 2, ceil_mode=self.ceil_mode)


class CBPoolMax2d(nn.Module):
    """Change-based 2x2/stride-2 max pooling (reference: conv2d.py:24-84)."""

    def __init__(self, m):
        super(CBPoolMax2d, self).__init__()
        ks = m.kernel_size if isinstance(m.kernel_size, tuple) else (m.kernel_size,) * 2
        st = m.stride if isinstance(m.stride, tuple) else (m.stride,) * 2
        assert ks == (2, 2) and st == (2, 2)
        self.stride = st
        self.kernel_size = ks
        self.ceil_mode = m.ceil_mode
        self.propChangeIndexes = False
        # reference behaviour (conv2d.py:73): hand out a private copy of the state every frame.  With
        # cloneOutput=False the state tensor itself is returned (tagged, so that a consumer about to
        # keep a reference to it -- CBConv2d with copyInput=False -- takes its copy then); this removes
        # one full-tensor copy per frame when the consumer copies or feeds back anyway.
        self.cloneOutput = True
        # reference behaviour (conv2d.py:80-83): with propChangeIndexes the INPUT-resolution list is
        # handed on as it came in.  downsampleIndexes=True hands on the list of changed OUTPUT pixels
        # instead, which is what a consumer working at the pooled resolution needs.
        self.downsampleIndexes = False
        # lazy=True (set by fusePoolingIntoDetection when the consumer is a feedback-mode CBConv2d): do
        # not pool at all, hand the consumer a LazyPool; it computes the pooled values inside its change
        # detection.  One launch less per pool and frame, identical results.
        self.lazy = False
        self.register_buffer('outputState', torch.zeros(0))
        self.clearMemory()

    def clearMemory(self):
        if not hasattr(self, 'outputState') or 'outputState' not in self._buffers:
            self.register_buffer('outputState', torch.zeros(0))
        self.outputState = self.outputState.new_zeros(0)

    def getStateTensors(self):
        state = []
        if hasattr(self, 'outputState'):
            state += [self.outputState]
        return state

    def forward(self, inp):
        assert type(inp) == tuple and inp[0] == 'changeIndexes'
        input = inp[1].detach().contiguous()
        changeIndexes = inp[2]
        require_device(input)
        if getattr(self, 'lazy', False) and not self.propChangeIndexes:
            nc, h, w = input.size(-3), input.size(-2), input.size(-1)
            oh, ow = ((h - 1) // 2 + 1, (w - 1) // 2 + 1) if self.ceil_mode else (h // 2, w // 2)
            return LazyPool(input, (1, nc, oh, ow), self.ceil_mode, changeIndexes)
        exact = isinstance(changeIndexes, torch.Tensor)
        if exact:
            changeIndexes = changeIndexes.detach().contiguous()
            assert changeIndexes.dim() == 1
        if not exact or changeIndexes.numel() != 0:
            nc, h, w = input.size(-3), input.size(-2), input.size(-1)
            if self.ceil_mode:
                oh, ow = (h - 1) // 2 + 1, (w - 1) // 2 + 1
            else:
                oh, ow = h // 2, w // 2
            if (not _same_shape(self.outputState, (1, nc, oh, ow)) or
                    self.outputState.dtype != input.dtype or self.outputState.device != input.device):
                self.outputState = torch.full((1, nc, oh, ow), float('inf'), dtype=input.dtype,
                                              device=input.device)
            maxPool2d(input, self.outputState, changeIndexes, self.kernel_size, self.stride)

        if getattr(self, 'cloneOutput', True):
            # a private copy: a following CBConv2d with copyInput=False keeps a reference to it
            output = self.outputState.clone()
        else:
            output = self.outputState
            output._cbinfer_inplace_state = True
        if self.propChangeIndexes:
            if getattr(self, 'downsampleIndexes', False):
                return 'changeIndexes', output, poolChangeIndexes(
                    changeIndexes, (input.size(-2), input.size(-1)), (output.size(-2), output.size(-1)))
            return 'changeIndexes', output, inp[2]
        return output

    def __repr__(self):
        return ('%s (k=%s, s=%s, ceil_mode=%s, propChgIdxs=%s)' %
                (self.__class__.__name__, self.kernel_size, self.stride, self.ceil_mode,
                 self.propChangeIndexes))


class CBConv2d(nn.Module):
    """Change-based 2-D convolution (reference: conv2d.py:87-304)."""

    def __init__(self, m, threshold):
        super(CBConv2d, self).__init__()
        assert m.groups == 1 and m.transposed == False
        assert m.output_padding == (0, 0) and m.padding == (m.kernel_size[-2] // 2,
                                                             m.kernel_size[-1] // 2)
        assert m.dilation == (1, 1) and m.stride == (1, 1)
        self.groups = m.groups
        self.transposed = m.transposed
        self.output_padding = m.output_padding
        self.padding = m.padding
        self.dilation = m.dilation
        self.stride = m.stride
        self.kernel_size = m.kernel_size
        self.in_channels = m.in_channels
        self.out_channels = m.out_channels

        assert m.weight is not None and m.bias is not None
        self.weight = m.weight   # shared with the source module, as in the reference
        self.bias = m.bias

        self.threshold = threshold
        self.clearMemory()

        self.withReLU = False
        self.saveChangeMap = False
        self.propChangeIndexes = False
        self.gatherComputationStats = False
        self.finegrained = False
        self.copyInput = True
        self.feedbackLoop = False
        # Extension: a layer fed propagated change indexes skips its own change detection whatever its
        # filter size and recomputes exactly the listed pixels (conv2d.py:180-190, :220) -- right for 1x1, short of
        # the filter's reach for k > 1 (the reference's behaviour, kept as the default).  True: the incoming list is
        # dilated by the filter support on the device first (cbinfer_dilate_change_indexes): every output pixel a
        # listed input pixel reaches is recomputed, which is what the layer's own detection would have found.
        self.dilatePropagatedIndexes = False
        self._setDefaultValues()

    # ---------------------------------------------------------------- state
    def clearMemory(self):
        for name in ('prevInput', 'prevOutput'):
            if not hasattr(self, name):
                self.register_buffer(name, self.weight.detach().new_zeros(0))
            elif name not in self._buffers:
                tmp = getattr(self, name)
                delattr(self, name)
                self.register_buffer(name, tmp)
        self.prevInput = self.weight.detach().new_zeros(0)
        self.prevOutput = self.weight.detach().new_zeros(0)
        self.__dict__['_rangeFallback'] = False
        self.__dict__['_upSeen'] = None
        if hasattr(self, 'compStats'):
            self.compStats = None
        # device work buffers (not part of the module state)
        self._work = None
        self._plan = None
        self._lastIndexes = None

    def getStateTensors(self):
        state = []
        if hasattr(self, 'prevInput'):
            state += [self.prevInput]
        if hasattr(self, 'prevOutput'):
            state += [self.prevOutput]
        return state

    def _setDefaultValues(self):
        # back-fill attributes missing in modules pickled by older versions (conv2d.py:292-304)
        for name, val in (('saveChangeMap', False), ('propChangeIndexes', False),
                          ('gatherComputationStats', False), ('finegrained', False),
                          ('copyInput', True), ('feedbackLoop', False), ('syncIndexes', False),
                          ('deterministicFG', False), ('atomicFG', False), ('fgInPlace', False), ('exactF32', False),
                          ('dilatePropagatedIndexes', False), ('_work', None), ('_wprep', None),
                          ('_inputIsLiveState', False), ('_plan', None), ('_wrows', None), ('_lastIndexes', None)):
            if name not in self.__dict__:
                self.__dict__[name] = val

    def __getstate__(self):
        d = dict(self.__dict__)
        d['_work'] = None     # transient device buffers are not serialised
        d['_wprep'] = None
        d['_plan'] = None
        d['_wrows'] = None
        d['_lastIndexes'] = None
        return d

    def lastChangeIndexes(self):
        """The change list of the most recent coarse-grained frame as a ChangeIndexes (None before the first
        frame).  For mask-driven frames it is materialised on this call; valid until the next frame."""
        return self._lastIndexes

    def invalidateWeights(self):
        """Forget the cached re-laid-out copies of the filter bank and the per-frame call plan.  They are
        refreshed on their own when the Parameter object or its version counter changes; a write through
        `weight.data` (the reference's own idiom) bumps neither, so call this after one."""
        self._wprep = self._wrows = self._plan = None

    def _rows_path(self, dtype, H, W):
        """Which mask-driven contraction, if any, runs this layer's sync-free fp32 frame (no int8 mask copy):
          'rows'    cb_rowconv.hip, at most 16 output channels (3->16 7x7: 15 us in the frame against the list
                    kernel's 21 on MI355X at 480x320 @10 %);
          'blocks'  cb_blockconv.hip (bf16x3 arithmetic, so not with exactF32), 17..64 output channels
                    (16->64 7x7: 19 us stand-alone against 27); wider layers stay on the list kernel, whose
                    64-pixel tiles reuse the streamed weights better while few pixels change;
          None      the list kernel (cb_conv.hip).
        CBINFER_NO_ROWCONV=1 / CBINFER_NO_BLOCKCONV=1 switch a kernel off, CBINFER_BLOCKCONV_MAXK moves the bound."""
        K, Cin, kH, kW = self.weight.size()
        if (dtype != torch.float32 or self.syncIndexes or self.saveChangeMap or
                os.environ.get('CBINFER_NO_SELFCOMPACT', '0') == '1'):
            return None
        if (K <= int(os.environ.get('CBINFER_ROWCONV_MAXK', '16')) and
                os.environ.get('CBINFER_NO_ROWCONV', '0') != '1' and C.cbinfer_rowconv_supported(Cin, K, kH, kW)):
            return 'rows'
        if (K <= int(os.environ.get('CBINFER_BLOCKCONV_MAXK', '64')) and self._arith_code(dtype) == _lib.CB_F32S and
                os.environ.get('CBINFER_NO_BLOCKCONV', '0') != '1' and
                C.cbinfer_blockconv_supported(Cin, K, kH, kW)):
            return 'blocks'
        return None

    def _masked_call(self, path):
        """(library entry point, prepared weights) of a mask-driven path."""
        w = self.weight
        key = (path, w.data_ptr(), w._version, w.device)
        if self._wrows is None or self._wrows[0] != key:
            K, Cin, kH, kW = w.size()
            if path == 'rows':
                nbytes, prep = C.cbinfer_rowconv_prepared_bytes(Cin, K, kH, kW), C.cbinfer_rowconv_prep_weights
            else:
                nbytes, prep = C.cbinfer_blockconv_prepared_bytes(Cin, K, kH, kW), C.cbinfer_blockconv_prep_weights
            wp = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
            check(prep(ptr(w.detach().contiguous()), ptr(wp), K, Cin, kH, kW, stream_ptr(w)))
            self._wrows = (key, wp)
        fn = C.cbinfer_cbconv2d_forward_rows if path == 'rows' else C.cbinfer_cbconv2d_forward_blocks
        return fn, self._wrows[1]

    def _pairs_ok(self, H, W):
        """Does the row-segment frame of this layer run on the row-PAIR kernel (cb_rowpair.hip: persistent over the
        non-empty (row pair, mask word) units, and able to do the next layer's pooled detection)?  Feedback mode, at
        most 4 input and 16 output channels, 3x3 / 5x5 / 7x7; CBINFER_NO_ROWPAIRS=1 switches it off."""
        K, Cin, kH, kW = self.weight.size()
        return (self.feedbackLoop and os.environ.get('CBINFER_NO_ROWPAIRS', '0') != '1' and
                bool(C.cbinfer_rowpairs_supported(Cin, K, kH, kW, H, W)))

    def _pair_detect_ok(self, nxt, kH):
        """Does this row-pair layer run its own change detection inside its launch (cbinfer_conv_rowpairs_detect, round 6)?
        Only while the consumer behind the pool takes the detection of ITS input from this launch (nxt) and its last frame
        ran a contraction that can carry this layer's state refresh on its idle workgroups (cbinfer_split_conv_next_refresh
        in window order, cbinfer_split_conv_refresh in pixel order).
        CBINFER_NO_PAIRDET=1 switches the form off."""
        if nxt is None or kH != 7 or os.environ.get('CBINFER_NO_PAIRDET', '0') == '1':
            return False
        link = self.__dict__.get('_fusedNext')
        cons = link[1] if link is not None else None
        cp = cons.__dict__.get('_plan') if cons is not None else None
        return bool(cp is not None and cp.get('split') and cp.get('sideArgs') is not None)

    def _detect_token(self):
        """What a producer that ran this layer's pooled detection inside its own launch must have seen: the identity of
        the state buffers and the threshold.  None while this layer cannot take such a detection (no split-state frame
        yet, a state that was written from outside and must be re-split, a threshold that differs from last frame's)."""
        sp = self._work.get('split') if self._work else None
        if sp is None or sp['stateKey'] is None:
            return None
        prev = self._buffers['prevInput']
        if sp['stateKey'] != (prev.data_ptr(), prev._version):
            return None
        th = float(self.threshold)
        if self.__dict__.get('_pmaskThreshold') != th or self.__dict__.get('_rangeFallback'):
            return None
        return (id(self), prev.data_ptr(), sp['S'].data_ptr(), sp['bits'].data_ptr(), th, sp['arith'])

    def _detect_token_with(self, sp, prev):
        return (id(self), prev.data_ptr(), sp['S'].data_ptr(), sp['bits'].data_ptr(), float(self.threshold),
                sp['arith'])

    def _next_detect(self, H, W):
        """(cbNextDetect, token) if this layer's row-pair launch can also be the pooled change detection of the layer
        behind the following CBPoolMax2d (pycbinfer.fuseDetectionIntoProducer), else (None, None)."""
        link = self.__dict__.get('_fusedNext')
        if (link is None or self.__dict__.get('_noNextFold') or os.environ.get('CBINFER_NO_NEXTFOLD', '0') == '1'):
            return None, None
        pool, cons = link
        # (per-frame fast path: the consumer's token and the flags the full test below looks at are what they were when the
        #  structure was made -- this function runs once per frame and producer inside the call plans)
        nd = self.__dict__.get('_nextStruct')
        if nd is not None and type(cons) is CBConv2d:
            flags = (getattr(pool, 'lazy', False), pool.propChangeIndexes, cons.feedbackLoop, cons.syncIndexes,
                     cons.saveChangeMap, cons.gatherComputationStats, cons.finegrained, H, W)
            if len(nd) > 2 and nd[2] == flags and nd[0] == cons._detect_token():
                return nd[1], nd[0]
        K = self.weight.size(0)
        if (not getattr(pool, 'lazy', False) or pool.propChangeIndexes or type(cons) is not CBConv2d or
                cons.in_channels != K or not cons.feedbackLoop or cons.syncIndexes or cons.saveChangeMap or
                cons.gatherComputationStats or cons.finegrained or cons.weight.dtype != torch.float32):
            return None, None
        H2, W2 = ((H - 1) // 2 + 1, (W - 1) // 2 + 1) if pool.ceil_mode else (H // 2, W // 2)
        prev2 = cons._buffers.get('prevInput')
        if (prev2 is None or tuple(prev2.shape) != (1, K, H2, W2) or prev2.device != self.weight.device or
                not cons._split_ok(torch.float32, H2, W2)):
            return None, None
        tok = cons._detect_token()
        if tok is None:
            return None, None
        nd = self.__dict__.get('_nextStruct')
        if nd is None or nd[0] != tok:
            sp = cons._work['split']
            st = _lib.NextDetect()
            st.state, st.splitState, st.frameMasks = prev2.data_ptr(), sp['S'].data_ptr(), sp['bits'].data_ptr()
            st.rangeFlag, st.H, st.W = sp['flag'].data_ptr(), H2, W2
            st.kH, st.kW, st.threshold = cons.weight.size(2), cons.weight.size(3), float(cons.threshold)
            st.arith = 1 if sp['arith'] == 'x3' else 0
            nd = (tok, st)
        flags = (getattr(pool, 'lazy', False), pool.propChangeIndexes, cons.feedbackLoop, cons.syncIndexes,
                 cons.saveChangeMap, cons.gatherComputationStats, cons.finegrained, H, W)
        nd = self.__dict__['_nextStruct'] = (nd[0], nd[1], flags)
        return nd[1], tok

    def _rows_workspace(self, work, H, W, dev):
        if work['rows'] is None:
            words = C.cbinfer_mask_words(H, W)
            work['rows'] = dict(bits=torch.zeros(words, dtype=torch.int64, device=dev),
                                arrive=torch.zeros(words, dtype=torch.int32, device=dev),
                                copy=torch.zeros(words, dtype=torch.int64, device=dev))
        return work['rows']

    @staticmethod
    def _split_arith():
        """Arithmetic of the split-state kernels for fp32 layers (CBINFER_ARITH): 'x3' (default, round 5) -- bf16
        TRIPLES: every f32 operand exactly, six term products, f32 accumulation: f32-equivalent, what
        conv2d_cg.py:342-349's sgemm multiplies --, or 'f16x2' -- f16 PAIRS, 22-23 significant bits per operand, three
        products (rounds 3-4's default, narrower than f32).  Any other value ('bf16x3') keeps the layer on rounds 1-2's
        list / patch-staged kernels.  Returns 'x3', 'f16x2' or None."""
        a = os.environ.get('CBINFER_ARITH', 'x3')
        return a if a in ('x3', 'f16x2') else None

    def _arith_code(self, dtype):
        if dtype == torch.float16:
            return _lib.CB_F16
        if not self.exactF32 and os.environ.get('CBINFER_EXACT_F32', '0') != '1':
            return _lib.CB_F32S
        return _lib.CB_F32

    def _arith(self, t):
        """Arithmetic code of the fused contraction for the library's list / patch-staged kernels: fp16 as is; fp32 as
        split products on the 16-bit MFMA (CB_F32S: bf16x3 there -- three bf16 terms per operand, the six cross
        products above 2^-24, f32 accumulation; the split-state kernels, which this code also admits (_split_ok), use
        f16 pairs instead: 22-23 significant bits per operand, three products) unless exactF32 asks for the exact f32
        fma chain on the f32 MFMA (CB_F32)."""
        dtype_code(t)     # (rejects anything but fp32 / fp16)
        return self._arith_code(t.dtype)

    def _prepared_weights(self, H=1, W=1, arith=None):
        w = self.weight
        key = (w.data_ptr(), w._version, w.dtype, w.device, H, W, arith)
        if self._wprep is None or self._wprep[0] != key:
            self._wprep = (key, prepWeights(w, H, W, arith=arith))
        return self._wprep[1]

    # ---------------------------------------------------------------- split-state frame (cb_split.hip)
    def _split_ok(self, dtype, H, W):
        """Does this layer's sync-free frame run on the split-state kernels (cbinfer_split_forward)?  fp32,
        feedback mode (the pre-split copy of the state is refreshed at the changed pixels), 16/32/64 input
        channels, default arithmetic (f16-pair products; exactF32 / CBINFER_ARITH=bf16x3 keep the other forms);
        CBINFER_NO_SPLIT=1 switches it off."""
        K, Cin, kH, kW = self.weight.size()
        return (dtype == torch.float32 and (self.feedbackLoop or self.copyInput)
                and not self.syncIndexes and not self.saveChangeMap
                and not self.finegrained and self._arith_code(dtype) == _lib.CB_F32S
                and (not self.__dict__.get('_rangeFallback') or self._split_arith() == 'x3')
                and os.environ.get('CBINFER_NO_SPLIT', '0') != '1'
                and int(os.environ.get('CBINFER_SPLIT_MINK', '0')) <= K <= int(os.environ.get('CBINFER_SPLIT_MAXK', '100000'))
                and self._split_arith() is not None
                and os.environ.get('CBINFER_NO_SELFCOMPACT', '0') != '1'
                and bool(C.cbinfer_split_supported(Cin, K, kH, kW))
                and C.cbinfer_mask_words(H, W) <= C.cbinfer_split_max_mask_words(K)
                and H * W * W < (1 << 32))

    def _hsplit_ok(self, dtype, H, W):
        """Does this fp16 layer's sync-free frame run on the split-state machinery (cbinfer_hsplit_forward: pixel-major
        f16 copy of the state, LDS-DMA contraction)?  Input channels a multiple of 64, feedback mode or a layer that
        keeps a copy of its input; CBINFER_NO_HSPLIT=1 (or CBINFER_NO_SPLIT=1) switches it off."""
        K, Cin, kH, kW = self.weight.size()
        return (dtype == torch.float16 and (self.feedbackLoop or self.copyInput)
                and not self.syncIndexes and not self.saveChangeMap and not self.finegrained
                and os.environ.get('CBINFER_NO_SPLIT', '0') != '1' and os.environ.get('CBINFER_NO_HSPLIT', '0') != '1'
                and os.environ.get('CBINFER_NO_SELFCOMPACT', '0') != '1'
                and bool(C.cbinfer_hsplit_supported(Cin, K, kH, kW))
                # (deep contractions -- 48 k-stages and more: 7x7 on 128 channels, 3x3 on 512 -- are taken since round 5:
                #  on a network whose deep layers do change they are the bulk of the work, and with few tiles the kernel
                #  cuts their depth into 8 or 16 chunks; rounds 4's artefact of random weights -- 31 idle layers, where the
                #  second launch of a deep contraction only cost -- had kept them off: CBINFER_HSPLIT_DEEP=0 does that again.
                #  Channels that are not a multiple of 64 are padded (OpenPose's 185 -> 192); up to a quarter of padding.)
                and (kH * kW * ((Cin + 63) // 64) < 48 or os.environ.get('CBINFER_HSPLIT_DEEP', '1') == '1')
                and 4 * ((Cin + 63) // 64 * 64 - Cin) <= (Cin + 63) // 64 * 64
                and C.cbinfer_mask_words(H, W) <= C.cbinfer_hsplit_max_mask_words(K)
                and C.cbinfer_hsplit_state_bytes(Cin, H, W, kH, kW) < (1 << 31) and H * W * W < (1 << 32))

    def _forward_hsplit(self, input, work, lazy=None):
        """One fp16 frame on the split-state machinery: detection + refresh of prevInput and of its pixel-major copy,
        then the LDS-DMA contraction -- whose launch also runs the change detection of the layers that consume this
        layer's output (pycbinfer.fuseDetectionIntoProducer; round 6), and whose own detection is skipped when the
        PRODUCING layer's launch did it.  `lazy`: the layer sits behind a CBPoolMax2d folded into its detection --
        `input` is the pool's INPUT, the layer runs at the pooled size."""
        K, Cin, kH, kW = self.weight.size()
        H, W = (lazy.outSize[-2], lazy.outSize[-1]) if lazy is not None else (input.size(-2), input.size(-1))
        dev = input.device
        hs = work.get('hsplit')
        if hs is None:
            S = torch.empty(C.cbinfer_hsplit_state_bytes(Cin, H, W, kH, kW), dtype=torch.uint8, device=dev)
            check(C.cbinfer_hsplit_state_init(ptr(S), Cin, H, W, kH, kW, stream_ptr(S)))
            wsBytes = C.cbinfer_hsplit_workspace_bytes(Cin, H, W, K, kH, kW)
            hs = work['hsplit'] = dict(
                S=S, bits=torch.zeros(C.cbinfer_frame_mask_bytes(H, W) // 8, dtype=torch.int64, device=dev),
                copy=torch.zeros(C.cbinfer_mask_words(H, W), dtype=torch.int64, device=dev),
                ws=torch.zeros(wsBytes, dtype=torch.uint8, device=dev) if wsBytes > 0 else None, stateKey=None,
                layer=(_lib.HalfLayer * 1)())
        wp = self._hsplit_weights(H, W)
        prev = self.prevInput
        if not prev.is_contiguous():
            prev = self.prevInput = prev.contiguous()
        stateKey = (prev.data_ptr(), prev._version)
        rebuilt = hs['stateKey'] != stateKey
        if rebuilt:
            # first frame on this path, or prevInput was (re)allocated or written by somebody else: the pixel-major
            # copy is made again from it
            check(C.cbinfer_hsplit_state_rebuild(ptr(prev), ptr(hs['S']), Cin, H, W, kH, kW, stream_ptr(input)))
            hs['stateKey'] = stateKey
        # (the producer-mask shortcut assumes the skipped segments compared below THIS threshold last frame and a
        #  state that has seen every pixel: not on a fresh or restored state, not after a change of the threshold)
        sameTh = self.__dict__.get('_pmaskThreshold') == float(self.threshold)
        self.__dict__['_pmaskThreshold'] = float(self.threshold)
        pmask = None
        if lazy is not None:
            pmask = None if (rebuilt or not sameTh) else lazy.producerMask()
        elif not rebuilt and sameTh:
            pmask = self._chain_mask(H, W)      # (round 5: a chained layer skips the segments its producer left alone)
        pooled = lazy is not None
        L = hs['layer'][0]
        L.upstreamCount, L.input, L.producerMask = None, ptr(input), ptr(pmask)
        L.state, L.pixelState, L.frameMasks = ptr(prev), ptr(hs['S']), ptr(hs['bits'])
        L.output, L.idxOut, L.countOut = ptr(self.prevOutput), ptr(work['idx']), ptr(work['count'])
        L.maskCopy, L.prepared, L.bias = ptr(hs['copy']), ptr(wp), ptr(self.bias.detach())
        L.K, L.threshold, L.relu = K, float(self.threshold), int(bool(self.withReLU))
        # this layer's own detection: done by the producing layer's launch?  (never on a fresh / restored state, after a
        # change of the threshold, or behind a pool)
        mine = None if (rebuilt or not sameTh or pooled) else self._half_token(hs, prev)
        L.detect = 0 if (mine is not None and self._detected_upstream(mine)) else 1
        tokens = self._fill_consumers(L, H, W)
        args = [hs['layer'], 1, int(pooled), input.size(-2) if pooled else 0, input.size(-1) if pooled else 0,
                Cin, H, W, kH, kW, int(bool(self.feedbackLoop)), ptr(hs['ws']), stream_ptr(input)]
        check(C.cbinfer_hsplit_forward_group(*args))
        self.__dict__['_ranSplit'] = True      # (a frame on any OTHER path invalidates hs['stateKey'], see forward)
        self._publish_count(work['count'], tokens)
        if not self._inputIsLiveState:
            self._make_plan(pooled, input, C.cbinfer_hsplit_forward_group, args, None, pmask=ptr(pmask))
            if self._plan is not None:
                self._plan.update(hsplit=True, chain=True, stateVersion=prev._version, checkPmask=pooled,
                                  layer=L, size=(H, W), hs=hs, nextTokens=tokens)
        result = MaskChangeIndexes(hs['copy'], (H, W), work['idx'], work['count'], made=True)
        if self._plan is not None:
            self._plan['indexes'] = result
        return result

    def _hsplit_weights(self, H, W):
        w = self.weight
        K, Cin, kH, kW = w.size()
        key = ('hsplit', w.data_ptr(), w._version, w.device, H, W)
        if self._wrows is None or self._wrows[0] != key:
            wp = torch.empty(C.cbinfer_hsplit_prepared_bytes(Cin, K, kH, kW), dtype=torch.uint8, device=w.device)
            check(C.cbinfer_hsplit_prep_weights(ptr(w.detach().contiguous()), ptr(wp), K, Cin, kH, kW, H, W,
                                                stream_ptr(w)))
            self._wrows = (key, wp)
        return self._wrows[1]

    # ---- the change detection of an fp16 layer inside the launch of the layer that PRODUCES its input (round 6) ----
    # The reference chains layers through their state tensors: a CBConv2d's input is the prevOutput of the one in front
    # (conv2d.py:259), and with propChangeIndexes the producer's change list travels along (:180-186, :256-259).  A
    # consumer in copy mode (feedbackLoop = False, copyInput = True) compares every value of that buffer with its
    # prevInput -- a copy of last frame's buffer -- so only the pixels the producer just recomputed can trigger, and the
    # producer's launch can do the whole detection for them (cb_split.hip: the fp16 epilogue / cbh_reduce_kernel).  What
    # makes that exact is checked on the host every frame, from both sides, with a token that names the consumer's
    # state buffers and threshold: the PRODUCER folds a consumer in only if that consumer's last frame consumed this
    # producer's last frame from this very buffer into the state the token names (its _upSeen) on the fp16 split-state
    # path; the CONSUMER skips its detection launch only if the tag on its input carries its token of this frame.
    def _half_token(self, hs, prev):
        return (id(self), prev.data_ptr(), hs['S'].data_ptr(), hs['bits'].data_ptr(), float(self.threshold))

    def _detected_upstream(self, token):
        d = self.__dict__
        return (d.get('_upNow') is not None and token in (d.get('_upTokens') or ()) and
                os.environ.get('CBINFER_NO_NEXTFOLD', '0') != '1')

    def _half_detect_token(self, prod, prodSerial, pout, H, W):
        """The token under which `prod`'s launch of its NEXT frame may run this layer's change detection, or None."""
        d = self.__dict__
        if (self.feedbackLoop or not self.copyInput or self.syncIndexes or self.saveChangeMap or
                self.gatherComputationStats or self.finegrained or not d.get('_ranSplit')):
            return None
        work = self._work
        hs = work.get('hsplit') if work else None
        prev = self._buffers.get('prevInput')
        if (hs is None or prev is None or prev.dtype != torch.float16 or
                tuple(prev.shape) != (1, prod.out_channels, H, W) or
                hs['stateKey'] != (prev.data_ptr(), prev._version)):
            return None
        seen = d.get('_upSeen')
        if (seen is None or seen[0] is not prod or seen[1] != prodSerial or seen[2] != pout.data_ptr() or
                seen[3] != prev.data_ptr() or seen[4] != prev._version):
            return None
        if d.get('_pmaskThreshold') != float(self.threshold):
            return None
        return self._half_token(hs, prev)

    def _fill_consumers(self, L, H, W):
        """cbHalfLayer.next[] of this frame: the linked consumers (pycbinfer.fuseDetectionIntoProducer) whose detection
        this launch may run.  Returns their tokens."""
        L.nNext = 0
        links = self.__dict__.get('_fusedConsumers')
        if not links or os.environ.get('CBINFER_NO_NEXTFOLD', '0') == '1' or self.weight.size(0) < 64:
            return ()
        pout = self._buffers['prevOutput']
        tag = getattr(pout, '_cbProduced', None)      # (still the PREVIOUS frame's)
        serial = self.__dict__.get('_serial', 0)
        if tag is None or tag.module is not self or tag.serial != serial - 1 or pout._version != tag.version:
            return ()
        tokens = []
        for cons in links:
            if len(tokens) == _lib.HNEXT_MAX:
                break
            tok = cons._half_detect_token(self, serial - 1, pout, H, W) if type(cons) is CBConv2d else None
            if tok is None:
                continue
            n = L.next[len(tokens)]
            n.state, n.pixelState, n.frameMasks = tok[1], tok[2], tok[3]
            n.kH, n.kW, n.threshold = cons.weight.size(2), cons.weight.size(3), tok[4]
            tokens.append(tok)
        L.nNext = len(tokens)
        return tuple(tokens)

    def _split_fg_ok(self, dtype, H, W):
        """Does this layer's fine-grained in-place frame run on the split-state kernels (cbinfer_split_forward_fg)?
        As _split_ok, for a layer in fine-grained mode."""
        K, Cin, kH, kW = self.weight.size()
        return (dtype == torch.float32 and self._arith_code(dtype) == _lib.CB_F32S
                and (not self.__dict__.get('_rangeFallback') or self._split_arith() == 'x3')
                and os.environ.get('CBINFER_NO_SPLIT', '0') != '1' and os.environ.get('CBINFER_NO_SPLIT_FG', '0') != '1'
                and self._split_arith() is not None
                and bool(C.cbinfer_split_supported(Cin, K, kH, kW))
                and C.cbinfer_mask_words(H, W) <= C.cbinfer_split_max_mask_words(K)
                and H * W * W < (1 << 32))

    def _split_weights(self, H, W):
        """(prepared buffer, weight scale).  f16 pairs: a power of two with max |w| * scale in [2^13, 2^14); bf16
        triples ('x3'): the weights as they are -- scale 0.0, which is also what tells the library's frame functions
        which form the buffers hold (include/cbinfer_hip.h)."""
        w = self.weight
        arith = self._split_arith()
        key = ('split', w.data_ptr(), w._version, w.device, H, W, arith)
        if self._wrows is None or self._wrows[0] != key:
            K, Cin, kH, kW = w.size()
            if arith == 'x3':
                scale = 0.0
                wp = torch.empty(C.cbinfer_split3_prepared_bytes(Cin, K, kH, kW), dtype=torch.uint8, device=w.device)
                check(C.cbinfer_split3_prep_weights(ptr(w.detach().contiguous()), ptr(wp), K, Cin, kH, kW, H, W,
                                                    stream_ptr(w)))
            else:
                wmax = float(w.detach().abs().max())          # (one host sync when the weights change)
                scale = 2.0 ** (13 - math.floor(math.log2(wmax))) if wmax > 0 and math.isfinite(wmax) else 1.0
                wp = torch.empty(C.cbinfer_split_prepared_bytes(Cin, K, kH, kW), dtype=torch.uint8, device=w.device)
                check(C.cbinfer_split_prep_weights(ptr(w.detach().contiguous()), ptr(wp), K, Cin, kH, kW, H, W,
                                                   scale, stream_ptr(w)))
            self._wrows = (key, wp, scale)
        return self._wrows[1], self._wrows[2]

    def _split_workspace(self, work, H, W, dev):
        sp = work.get('split')
        arith = self._split_arith()
        if sp is None or sp['arith'] != arith:      # (the records of the two arithmetics differ in size and content)
            K, Cin, kH, kW = self.weight.size()
            words = C.cbinfer_mask_words(H, W)
            if arith == 'x3':
                S = torch.empty(C.cbinfer_split3_state_bytes(Cin, H, W, kH, kW), dtype=torch.uint8, device=dev)
                check(C.cbinfer_split3_state_init(ptr(S), Cin, H, W, kH, kW, stream_ptr(S)))
            else:
                S = torch.empty(C.cbinfer_split_state_bytes(Cin, H, W, kH, kW), dtype=torch.uint8, device=dev)
                check(C.cbinfer_split_state_init(ptr(S), Cin, H, W, kH, kW, stream_ptr(S)))
            # (slabs only for deep contractions -- 48 k-stages and more: 0 bytes otherwise)
            wsBytes = C.cbinfer_split_workspace_bytes(1, Cin, H, W, K, kH, kW)
            ws = torch.zeros(wsBytes, dtype=torch.uint8, device=dev) if wsBytes > 0 else None
            # (a frame mask of its own: the list kernels' protocol -- two masks alternating by a device-side parity --
            #  and the split-state kernels' -- one mask, an arrival counter behind it -- must never meet in one
            #  buffer when a module changes paths in mid-sequence)
            sp = work['split'] = dict(S=S, arith=arith, flag=torch.zeros(1, dtype=torch.int32, device=dev),
                                      bits=torch.zeros(C.cbinfer_frame_mask_bytes(H, W) // 8, dtype=torch.int64,
                                                       device=dev),
                                      copy=torch.zeros(words, dtype=torch.int64, device=dev), ws=ws,
                                      stateKey=None, seq=(_lib.SplitSeq * 1)())
        return sp

    def _folded_tail(self, sp, H, W, dev):
        """The CBTail1x1 behind this layer if its evaluation can ride in this layer's second launch
        (cbinfer_split_forward_tail), with sp['tail'] (cbSplitTail) filled in -- else None: the tail module then
        runs its own launch.  CBINFER_NO_TAILFOLD=1 switches the folding off."""
        t = self.__dict__.get('_fusedTail')
        K, Cin, kH, kW = self.weight.size()
        if (t is None or sp['ws'] is None or not self.propChangeIndexes or t.in_channels != K or
                self.__dict__.get('_noTailFold') or
                t.weight1.dtype != torch.float32 or t.weight1.device != dev or
                # (the second launch reads the layer's bias and the tail's first four values at a time)
                (self.bias.data_ptr() | t.bias1.data_ptr()) & 15 or
                os.environ.get('CBINFER_NO_TAILFOLD', '0') == '1' or
                not C.cbinfer_split_tail_supported(Cin, K, kH, kW, t.hidden_channels, t.out_channels)):
            return None
        out = t._output_for(H, W, dev, torch.float32)
        st = sp.get('tail')
        if st is None:
            st = sp['tail'] = _lib.SplitTail()
        sp['tailKeep'] = (t._prepared(), t.bias1.detach(), t.weight2.detach().contiguous(), t.bias2.detach())
        st.w1Prepared, st.b1, st.w2, st.b2 = [x.data_ptr() for x in sp['tailKeep']]
        st.C1, st.C2, st.relu1, st.relu2 = t.hidden_channels, t.out_channels, int(t.relu), int(bool(t.withReLU))
        st.output[0] = out.data_ptr()
        return t

    def _fusedTailCandidate(self):
        """The fused tail this layer would fold into its second launch right now (cheap form of _folded_tail's test,
        for the call plans), or None."""
        t = self.__dict__.get('_fusedTail')
        if (t is None or not self.propChangeIndexes or self.__dict__.get('_noTailFold') or
                os.environ.get('CBINFER_NO_TAILFOLD', '0') == '1'):
            return None
        return t

    def rangeExceeded(self):
        """True if a state value of this layer ever left the range of the f16-pair arithmetic (|x| >= 2^20, or a
        non-finite input) since the state was cleared (one host sync unless the module has noticed already).  The
        outputs are right either way: from the frame that trips the flag on, the contraction kernel itself computes
        the layer from prevInput with plain f32 arithmetic (cb_split.hip: cbs_exact_tile) -- slowly -- until the module
        notices (`_poll_range`, no sync) and moves the layer to the bf16x3 kernels, which have f32's range."""
        if self.__dict__.get('_rangeFallback'):
            return True
        sp = self._work.get('split') if self._work else None
        return bool(sp is not None and int(sp['flag'].item()) != 0)

    def _poll_range(self, sp):
        """Every 64th split-state frame: an asynchronous copy of the layer's range flag into pinned memory, and a look
        at what the previous copy brought (no sync, no wait).  A tripped flag moves the layer to the bf16x3 kernels for
        good (`_rangeFallback`; clearMemory resets it): the in-kernel exact path that kept the frames since the trip
        right is orders of magnitude slower than either."""
        if sp['arith'] == 'x3':      # (bf16 triples have f32's range: the flag is never set)
            return
        n = sp['poll'] = sp.get('poll', 0) + 1
        if n & 63 or torch.cuda.is_current_stream_capturing():
            return
        host, ev = sp.get('flagHost'), sp.get('flagEvent')
        if host is None:
            host = sp['flagHost'] = torch.zeros(1, dtype=torch.int32).pin_memory()
        elif ev is not None and ev.query() and int(host[0]) != 0:
            self.__dict__['_rangeFallback'] = True
            self._plan = None
            return
        host.copy_(sp['flag'], non_blocking=True)
        ev = sp['flagEvent'] = torch.cuda.Event()
        ev.record()

    def _window_fold(self, sp, tail, H, W):
        """Can this split-state layer's contraction run in pooling-window order and carry the pooled change detection of
        the layer behind the following CBPoolMax2d (cbinfer_split_*_next, round 6)?  -> (eligible at all, cbNextDetect or
        None, the token the consumer will look for or None, _next_detect's token before the library's own test).
        pycbinfer.fuseDetectionIntoProducer(windowOrder=...) says whether: True, False, or 'auto' (default) -- decided when
        the call plan is made (NOT per frame: the one place where the layer's last change count is read back, outside any
        stream capture) from the tiles the window-order form would need: it pays while they fit one round of the grid
        (DESIGN 5.8).  CBINFER_NO_WINFOLD=1 switches the form off."""
        K, Cin, kH, kW = self.weight.size()
        mode = self.__dict__.get('_winFold', 'auto')
        eligible = (tail is None and sp['arith'] == 'x3' and mode is not False and
                    os.environ.get('CBINFER_NO_WINFOLD', '0') != '1')
        nxt, ntok, rawTok = None, None, None
        if eligible:
            nxt, ntok = self._next_detect(H, W)
            rawTok = ntok
            if nxt is not None and not C.cbinfer_split_next_supported(Cin, K, kH, kW, H, W, ctypes.pointer(nxt)):
                nxt, ntok = None, None
            if nxt is not None and mode == 'auto':
                # decided ONCE per module, from the first frame that recomputed some but not all of the layer's pixels (the
                # first frame of a sequence recomputes all of them, a repeated frame none): the tiles of 16 windows the form
                # would need for that many pixels (a touched window's unchanged pixels cost slots: + ~20 %) -- beyond one
                # round of the persistent grid its longer prologue and epilogue are paid once more per round, and its tile
                # count crosses that line before pixel order's does.  Until then (at most four looks, each a read-back of
                # the layer's change count while its call plan is made -- never inside a stream capture): the form is used.
                st = self.__dict__.setdefault('_winAuto', {'fold': None, 'looks': 0})
                if st['fold'] is None and not torch.cuda.is_current_stream_capturing():
                    n = int(self._work['count'].item())
                    st['looks'] += 1
                    if 0 < n < H * W:
                        cus = torch.cuda.get_device_properties(self.weight.device).multi_processor_count
                        st['fold'] = (1.2 * n) / 64.0 <= cus
                    elif st['looks'] >= 4:
                        st['fold'] = True
                if st['fold'] is False:
                    nxt, ntok = None, None
        return eligible, nxt, ntok, rawTok

    def _forward_split(self, src, lazy, work):
        """One frame on the split-state kernels: detection (+ pooling) + refresh of prevInput and of its pre-split
        copy, then the LDS-DMA contraction.  `src` is the layer input, or the pool's input when `lazy`."""
        K, Cin, kH, kW = self.weight.size()
        H, W = self.prevInput.size(-2), self.prevInput.size(-1)
        dev = src.device
        sp = self._split_workspace(work, H, W, dev)
        wp, scale = self._split_weights(H, W)
        prev = self.prevInput
        stateKey = (prev.data_ptr(), prev._version)
        rebuilt = sp['stateKey'] != stateKey
        if rebuilt:
            # first frame, or prevInput was (re)allocated or written by somebody else (restored states,
            # eval03.py:88-95): the pre-split copy is made again from it
            if sp['arith'] == 'x3':
                check(C.cbinfer_split3_state_rebuild(ptr(prev), ptr(sp['S']), Cin, H, W, kH, kW, stream_ptr(src)))
            else:
                check(C.cbinfer_split_state_rebuild(ptr(prev), ptr(sp['S']), Cin, H, W, kH, kW, ptr(sp['flag']),
                                                    stream_ptr(src)))
            sp['stateKey'] = stateKey
        pmask = None
        if lazy is not None:
            sameTh = self.__dict__.get('_pmaskThreshold') == float(self.threshold)
            self.__dict__['_pmaskThreshold'] = float(self.threshold)
            pmask = None if (rebuilt or not sameTh) else lazy.producerMask()
        q = sp['seq'][0]
        q.input, q.state, q.splitState = src.data_ptr(), prev.data_ptr(), sp['S'].data_ptr()
        q.frameMasks, q.producerMask = sp['bits'].data_ptr(), ptr(pmask)
        q.output, q.idxOut, q.countOut = self.prevOutput.data_ptr(), work['idx'].data_ptr(), work['count'].data_ptr()
        q.rangeFlag, q.maskCopy = sp['flag'].data_ptr(), sp['copy'].data_ptr()
        # (mode: bit 0 = behind a folded pool, bit 1 = not in feedback mode -- both states take every value of the frame)
        args = [sp['seq'], 1, int(lazy is not None) | (0 if self.feedbackLoop else 2),
                src.size(-2) if lazy is not None else 0,
                src.size(-1) if lazy is not None else 0, ptr(wp), ptr(self.bias.detach()), Cin, H, W, K, kH, kW,
                float(self.threshold), float(scale), int(bool(self.withReLU)), ptr(sp['ws'])]
        # the fused 1x1 tail behind this layer (pycbinfer.fuseTail1x1) rides in the contraction's second launch
        tail = self._folded_tail(sp, H, W, dev)
        # the contraction alone (cbinfer_split_conv[_tail]): for the frames whose detection the PRODUCING layer's
        # launch has done already (cb_rowpair.hip; its change indexes carry this layer's token)
        cargs = [sp['seq'], 1, ptr(wp), ptr(self.bias.detach()), Cin, H, W, K, kH, kW, float(scale),
                 int(bool(self.withReLU)), ptr(sp['ws']), 0]
        # round 6: with a split-state consumer behind a lazy pool (pycbinfer.fuseDetectionIntoProducer) the contraction runs
        # in pooling-window order and is that consumer's pooled change detection as well (cbinfer_split_*_next), as the
        # row-pair kernel is for its consumer
        nextEligible, nxt, ntok, rawTok = self._window_fold(sp, tail, H, W)
        if tail is not None:
            fn, cfn = C.cbinfer_split_forward_tail, C.cbinfer_split_conv_tail
            args += [0, ctypes.pointer(sp['tail']), stream_ptr(src)]
            cargs += [ctypes.pointer(sp['tail']), stream_ptr(src)]
        elif nxt is not None:
            fn, cfn = C.cbinfer_split_forward_next, C.cbinfer_split_conv_next
            args += [ctypes.pointer(nxt), stream_ptr(src)]
            cargs = cargs[:-1] + [ctypes.pointer(nxt), stream_ptr(src)]      # (no forceSplit argument)
        else:
            fn, cfn = C.cbinfer_split_forward, C.cbinfer_split_conv
            args += [stream_ptr(src)]
            cargs += [stream_ptr(src)]
        myTok = None if rebuilt else self._detect_token_with(sp, prev)
        done = (lazy is not None and myTok is not None and
                getattr(lazy.indexes, 'nextDetect', None) == myTok)
        # (round 6: the row-pair layer in front left its state refresh to this launch's idle workgroups)
        side = sp.get('side')
        if side is None:
            side = sp['side'] = _lib.SideRefresh()
        sfn, sargs = None, None
        if nxt is not None:
            sfn, sargs = C.cbinfer_split_conv_next_refresh, cargs[:-1] + [ctypes.pointer(side), cargs[-1]]
        elif (tail is None and sp['arith'] == 'x3' and
              C.cbinfer_split_refresh_supported(Cin, K, kH, kW, H, W)):      # (pixel order: cbinfer_split_conv's arguments)
            sfn, sargs = C.cbinfer_split_conv_refresh, cargs[:-2] + [ctypes.pointer(side), cargs[-1]]
        pend = self.__dict__.get('_sidePending')
        if pend is not None and done and sargs is not None:
            self.__dict__.pop('_sidePending')
            side.frame, side.state, side.C, side.H, side.W, side.threshold = (ptr(pend[0]), ptr(pend[1]), pend[2], pend[3],
                                                                              pend[4], pend[5])
            check(sfn(*sargs))
        else:
            check(cfn(*cargs) if done else fn(*args))
        self._poll_range(sp)
        self.__dict__['_ranSplit'] = True
        self._inputIsLiveState = False
        self._lastIndexes = MaskChangeIndexes(sp['copy'], (H, W), work['idx'], work['count'], made=True)
        self._lastIndexes.tailDone = tail
        self._lastIndexes.nextDetect = ntok
        w, b = self._parameters.get('weight'), self._parameters.get('bias')
        if (w is not None and b is not None and not self.gatherComputationStats and
                os.environ.get('CBINFER_NO_FASTPATH', '0') != '1'):
            self._plan = dict(
                split=True, arith=sp['arith'], pooled=lazy is not None, shape=tuple(src.shape), dtype=src.dtype,
                device=dev,
                flags=self._flags(), w=(w.data_ptr(), w._version), b=(b.data_ptr(), b._version),
                state=(self._buffers['prevInput'].data_ptr(), self._buffers['prevOutput'].data_ptr()),
                stateVersion=prev._version, stream=args[-1], work=work, args=args, seq=q, pmask=ptr(pmask),
                indexes=self._lastIndexes, fn=fn, tail=tail, tailKey=tail._fold_key() if tail is not None else None,
                convFn=cfn, convArgs=cargs, tailBlocked=bool(self.__dict__.get('_noTailFold')),
                detectToken=(id(self), prev.data_ptr(), sp['S'].data_ptr(), sp['bits'].data_ptr(),
                             float(self.threshold), sp['arith']),
                nextEligible=nextEligible, nextToken=ntok, nextRaw=rawTok, keep=nxt, hw=(H, W), side=side,
                sideFn=sfn, sideArgs=sargs)
        if self.propChangeIndexes:
            return 'changeIndexes', self.prevOutput, self._lastIndexes
        return self.prevOutput

    def _run_split_plan(self, inp):
        plan = self._plan
        if plan['pooled']:
            if type(inp) is not LazyPool:
                return None
            src = inp.source
            pm = inp.producerMask()
            if (pm.data_ptr() if pm is not None else None) != plan['pmask']:
                return None
        else:
            if type(inp) is not torch.Tensor:
                return None
            src = inp
        w, b, bufs = self._parameters['weight'], self._parameters['bias'], self._buffers
        prev = bufs['prevInput']
        if (src.shape != plan['shape'] or src.dtype != plan['dtype'] or src.device != plan['device'] or
                not src.is_contiguous() or self._flags() != plan['flags'] or
                (w.data_ptr(), w._version) != plan['w'] or (b.data_ptr(), b._version) != plan['b'] or
                (prev.data_ptr(), bufs['prevOutput'].data_ptr()) != plan['state'] or
                prev._version != plan['stateVersion'] or
                self._work is not plan['work'] or raw_stream(src.device.index) != plan['stream'] or
                plan['arith'] != os.environ.get('CBINFER_ARITH', 'x3')):
            return None
        if plan['tailBlocked'] != bool(self.__dict__.get('_noTailFold')):
            return None      # (a FramePipeline took the tail's folding away, or gave it back)
        if plan['tail'] is not None and (self.__dict__.get('_fusedTail') is not plan['tail'] or
                                         self.__dict__.get('_noTailFold') or
                                         os.environ.get('CBINFER_NO_TAILFOLD', '0') == '1' or
                                         plan['tail']._fold_key() != plan['tailKey']):
            return None
        if plan['nextEligible'] and self._next_detect(*plan['hw'])[1] != plan['nextRaw']:
            return None      # (the consumer's state or threshold is not the one this plan folds -- or it can fold now)
        if (plan['keep'] is not None and self.__dict__.get('_winFold', 'auto') == 'auto' and
                self.__dict__.get('_winAuto', {}).get('fold') is None and not torch.cuda.is_current_stream_capturing()):
            return None      # ('auto' has not seen a typical frame yet: the plan is made again, with another look)
        plan['seq'].input = src.data_ptr()
        if plan['pooled'] and getattr(inp.indexes, 'nextDetect', None) == plan['detectToken']:
            pend = self.__dict__.get('_sidePending')
            if pend is not None and plan['sideArgs'] is not None:
                # (the row-pair layer in front left its state refresh to this launch's idle workgroups)
                self.__dict__.pop('_sidePending')
                side = plan['side']
                key = (pend[1].data_ptr(),) + pend[2:]
                if plan.get('sideKey') != key:      # (everything but the frame's address stays from frame to frame)
                    side.state, side.C, side.H, side.W, side.threshold = key
                    plan['sideKey'] = key
                side.frame = pend[0].data_ptr()
                status = plan['sideFn'](*plan['sideArgs'])
            else:
                status = plan['convFn'](*plan['convArgs'])      # (the producing layer's launch was this frame's detection)
        else:
            status = plan['fn'](*plan['args'])
        if status != 0:
            check(status)
        self._poll_range(plan['work']['split'])
        self._inputIsLiveState = False
        self._lastIndexes = plan['indexes']
        if self.propChangeIndexes:
            return 'changeIndexes', bufs['prevOutput'], self._lastIndexes
        return bufs['prevOutput']

    def _workspace(self, input, wantMap=None):
        H, W = input.size(-2), input.size(-1)
        # self-compacting frame pipeline (detection + fused kernel, no compaction launch): mask small
        # enough for the kernel's LDS prefix, and no int8 copy of the mask requested
        wantMap = self.saveChangeMap if wantMap is None else wantMap
        selfc = (not wantMap and
                 C.cbinfer_mask_words(H, W) <= C.cbinfer_frame_mask_max_words() and
                 os.environ.get('CBINFER_NO_SELFCOMPACT', '0') != '1')
        key = (H, W, input.device, selfc)
        if self._work is None or self._work['key'] != key:
            dev = input.device
            # split-K workspace of the contraction kernel: owned by the module (kept across a change of
            # the frame size), so graphs and call plans of different modules or sequences never share
            # one, whatever stream they are captured or replayed on
            conv = self._work['conv'] if (self._work is not None and self._work['key'][2] == dev) \
                else newConvWorkspace(dev)
            nbytes = C.cbinfer_frame_mask_bytes(H, W) if selfc else 8 * C.cbinfer_mask_words(H, W)
            self._work = dict(
                key=key, selfc=selfc,
                bits=torch.zeros((nbytes + 7) // 8, dtype=torch.int64, device=dev),
                idx=torch.empty(H * W, dtype=torch.int32, device=dev),
                count=torch.zeros(1, dtype=torch.int32, device=dev),
                conv=conv, map=None,
                # row-segment contraction (cbinfer_conv_changed_rows): single mask, arrival counters, mask copy
                rows=None,
                # split-state frame (cbinfer_split_forward): pre-split state copy, range flag, mask copy, slabs
                split=None)
        if wantMap and self._work['map'] is None:
            self._work['map'] = torch.zeros(H, W, dtype=torch.int8, device=input.device)
        return self._work

    # ---------------------------------------------------------------- fine-grained
    # Execution forms of a fine-grained frame (results agree within the fp32 bar; selected by attributes):
    #   default      cbinfer_cbconv2d_forward_fg: per-value detection + accumulating fused contraction,
    #                two launches, no atomics, no host sync, deterministic
    #   atomicFG     the reference-structured op sequence changeDetectionFG -> compaction ->
    #                updateOutputFG (f32 atomics), also without a host sync
    #   fgInPlace    default form updating the module's own tensors in place: prevOutput (and the
    #                relu'd copy handed out when withReLU) alias the state across frames, as in
    #                coarse-grained mode, and prevInput is the module's own copy -- no clone, no full-tensor
    #                ReLU pass, capturable.  The reference hands out fresh tensors (conv2d.py:169,173) and
    #                keeps the caller's input tensor as state (:175); that is the default here too.
    def _fg_workspace(self, x):
        work = self._workspace(x, wantMap=False)     # (saveChangeMap has no meaning in fine-grained mode)
        assert work['selfc']
        if work.get('delta') is None or work['delta'].shape != x.shape:
            work['delta'] = torch.empty_like(x)
            work['relu'] = None
        return work

    def _forward_fg_split(self, src, lazy, H, W, work, relu):
        """The fine-grained in-place frame on the split-state kernels (round 4): the detection leaves the thresholded
        differences of EVERY value as f16 pairs in the pixel-major records, the LDS-DMA contraction adds W * delta to
        prevOutput at the mask's pixels.  `lazy`: the layer sits behind a CBPoolMax2d folded into its detection -- `src`
        is the pool's INPUT, H x W the pooled size."""
        K, Cin, kH, kW = self.weight.size()
        sp = self._split_workspace(work, H, W, src.device)
        wp, scale = self._split_weights(H, W)
        q = sp['seq'][0]
        q.input, q.state, q.splitState = src.data_ptr(), self.prevInput.data_ptr(), sp['S'].data_ptr()
        q.frameMasks, q.producerMask = sp['bits'].data_ptr(), None
        q.output, q.idxOut, q.countOut = (self.prevOutput.data_ptr(), work['idx'].data_ptr(),
                                          work['count'].data_ptr())
        q.rangeFlag, q.maskCopy = sp['flag'].data_ptr(), sp['copy'].data_ptr()
        q.delta, q.reluOut = work['delta'].data_ptr(), ptr(relu)
        sp['stateKey'] = None      # (the records hold differences now: a coarse-grained frame re-splits the state)
        pooled = lazy is not None
        args = [sp['seq'], 1, int(pooled), src.size(-2) if pooled else 0, src.size(-1) if pooled else 0, ptr(wp),
                Cin, H, W, K, kH, kW, float(self.threshold), float(scale), ptr(sp['ws'])]
        # the fused 1x1 tail behind this layer rides in the contraction's second launch, as in coarse-grained mode
        # (round 5: cbinfer_split_forward_fg_tail); it reads the relu'd copy when the layer has one
        tail = self._folded_tail(sp, H, W, src.device)
        if tail is not None and (not self.withReLU or relu is not None):
            fn = C.cbinfer_split_forward_fg_tail
            args += [ctypes.pointer(sp['tail']), stream_ptr(src)]
        else:
            tail, fn = None, C.cbinfer_split_forward_fg
            args += [stream_ptr(src)]
        check(fn(*args))
        self._poll_range(sp)
        self.__dict__['_ranSplit'] = True
        result = relu if self.withReLU else self.prevOutput
        self._lastIndexes = MaskChangeIndexes(sp['copy'], (H, W), work['idx'], work['count'], made=True)
        self._lastIndexes.tailDone = tail
        if self.propChangeIndexes:
            result = ('changeIndexes', result, self._lastIndexes)
        self._make_plan(pooled, src, fn, args, None, result=result)
        if self._plan is not None:
            self._plan.update(fgSplit=True, arith=sp['arith'], seq=q, wsplit=(wp, scale), relu=relu,
                              tail=tail, tailKey=tail._fold_key() if tail is not None else None,
                              tailCand=self._fusedTailCandidate())
        return result

    def forward_fg(self, inp):
        if isinstance(inp, LazyPool):
            # behind a CBPoolMax2d folded into the detection (pycbinfer.fusePoolingIntoDetection): the in-place frame
            # on the split-state kernels takes the pool's input as it is; anything else pools first
            lazy, size = inp, tuple(inp.outSize)
            src = lazy.source.detach()
            H, W = size[-2], size[-1]
            if (self.fgInPlace and not self.atomicFG and src.is_cuda and src.dtype == torch.float32 and
                    src.is_contiguous() and tuple(self.prevInput.size()) == size and
                    C.cbinfer_mask_words(H, W) <= C.cbinfer_frame_mask_max_words() and
                    os.environ.get('CBINFER_NO_SELFCOMPACT', '0') != '1' and self._split_fg_ok(src.dtype, H, W)):
                if not self.prevInput.is_contiguous():
                    self.prevInput = self.prevInput.contiguous()
                work = self._fg_workspace(self.prevInput)
                relu = None
                if self.withReLU:
                    if work['relu'] is None:
                        work['relu'] = F.relu(self.prevOutput)
                    relu = work['relu']
                return self._forward_fg_split(src, lazy, H, W, work, relu)
            inp = lazy.tensor()
        x = inp.detach()
        K, Cin, kH, kW = self.weight.size()
        if self.prevInput.size() != x.size():
            # first frame / new size: dense convolution incl. bias (conv2d.py:163-167)
            self.prevOutput = F.conv2d(x, self.weight.detach(), bias=self.bias.detach(),
                                       padding=(kH // 2, kW // 2))
            self.prevInput = x.clone() if (self.fgInPlace and x.is_cuda) else x
            if self._work is not None:
                self._work['relu'] = None
            first = F.relu(self.prevOutput) if self.withReLU else self.prevOutput
            if self.propChangeIndexes and x.is_cuda:      # every output pixel is new on the first frame
                H, W = x.size(-2), x.size(-1)
                allpix = torch.arange(H * W, dtype=torch.int32, device=x.device)
                return 'changeIndexes', first, allpix
            return first
        x = x.contiguous()
        H, W = x.size(-2), x.size(-1)
        fused = (x.is_cuda and not self.atomicFG and x.dtype == torch.float32 and
                 C.cbinfer_mask_words(H, W) <= C.cbinfer_frame_mask_max_words() and
                 os.environ.get('CBINFER_NO_SELFCOMPACT', '0') != '1')
        # layers of few output channels: the frame on the mask-driven contractions (row-segment / patch-staged
        # kernel with an accumulating epilogue), as in coarse-grained mode
        path = self._rows_path(x.dtype, H, W) if fused else None
        if fused and self.fgInPlace:
            work = self._fg_workspace(x)
            relu = None
            if self.withReLU:
                if work['relu'] is None:
                    work['relu'] = F.relu(self.prevOutput)
                relu = work['relu']
            if not self.prevInput.is_contiguous():
                self.prevInput = self.prevInput.contiguous()
            if self._split_fg_ok(x.dtype, H, W):
                return self._forward_fg_split(x, None, H, W, work, relu)
            if path:
                rows = self._rows_workspace(work, H, W, x.device)
                args = (int(path == 'blocks'), ptr(x), ptr(self.prevInput), ptr(work['delta']),
                        ptr(self.prevOutput), ptr(relu), ptr(rows['bits']), ptr(rows['arrive']), ptr(rows['copy']),
                        ptr(self._masked_call(path)[1]), Cin, H, W, K, kH, kW, float(self.threshold), 1,
                        stream_ptr(x))
                check(C.cbinfer_cbconv2d_forward_fg_masked(*args))
                result = relu if self.withReLU else self.prevOutput
                self._lastIndexes = MaskChangeIndexes(rows['copy'], (H, W), work['idx'], work['count'])
                if self.propChangeIndexes:
                    result = ('changeIndexes', result, self._lastIndexes)
                self._make_plan(False, x, C.cbinfer_cbconv2d_forward_fg_masked, args, 1, result=result, rows=True)
                return result
            arith = self._arith(x)
            args = (ptr(x), ptr(self.prevInput), ptr(work['delta']), ptr(self.prevOutput), ptr(relu),
                    ptr(work['bits']), ptr(work['idx']), ptr(work['count']),
                    ptr(self._prepared_weights(H, W, arith)), Cin, H, W, K, kH, kW, float(self.threshold), 1,
                    ptr(work['conv']), arith, stream_ptr(x))
            check(C.cbinfer_cbconv2d_forward_fg(*args))
            result = relu if self.withReLU else self.prevOutput
            if self.propChangeIndexes:      # (extension: the reference's forward_fg hands no indexes on)
                result = ('changeIndexes', result, ChangeIndexes(work['idx'], work['count'], (H, W)))
            self._make_plan(False, x, C.cbinfer_cbconv2d_forward_fg, args, 0, result=result)
            return result
        po = self.prevOutput.clone()                                 # conv2d.py:169
        indexes = None
        if fused and path:
            work = self._fg_workspace(x)
            rows = self._rows_workspace(work, H, W, x.device)
            check(C.cbinfer_cbconv2d_forward_fg_masked(
                int(path == 'blocks'), ptr(x), ptr(self.prevInput.contiguous()), ptr(work['delta']), ptr(po), None,
                ptr(rows['bits']), ptr(rows['arrive']), ptr(rows['copy']), ptr(self._masked_call(path)[1]),
                Cin, H, W, K, kH, kW, float(self.threshold), 0, stream_ptr(x)))
            indexes = MaskChangeIndexes(rows['copy'], (H, W), work['idx'], work['count'])
        elif fused:
            work = self._fg_workspace(x)
            indexes = ChangeIndexes(work['idx'], work['count'], (H, W))
            arith = self._arith(x)
            check(C.cbinfer_cbconv2d_forward_fg(
                ptr(x), ptr(self.prevInput.contiguous()), ptr(work['delta']), ptr(po), None,
                ptr(work['bits']), ptr(work['idx']), ptr(work['count']),
                ptr(self._prepared_weights(H, W, arith)), Cin, H, W, K, kH, kW, float(self.threshold), 0,
                ptr(work['conv']), arith, stream_ptr(x)))
        elif x.is_cuda and self.deterministicFG and not self.atomicFG:
            po = cbconvFG_deterministic(x, self.prevInput, po, self.weight.detach(), self.threshold,
                                        weightsPrepared=self._prepared_weights(H, W))
        else:
            po = cbconvFG(x, self.prevInput, po, self.weight.detach(), self.threshold)
        self.prevOutput = po
        self.prevInput = x                                           # conv2d.py:175
        outp = F.relu(po) if self.withReLU else po
        if self.propChangeIndexes and x.is_cuda:
            if indexes is None:
                # the atomic / deterministic / oversized-mask forms keep no list of the output pixels they
                # touched: hand on EVERY pixel, so that a consumer fed by the tuple protocol (CBTail1x1 behind
                # a fine-grained head) stays correct -- it then recomputes the whole map
                indexes = torch.arange(H * W, dtype=torch.int32, device=x.device)
            return 'changeIndexes', outp, indexes
        return outp

    # ---------------------------------------------------------------- coarse-grained
    def forward_normal(self, inp):
        # input parsing and checks (conv2d.py:180-190)
        changeIndexes = None
        pooled = None
        if isinstance(inp, LazyPool):
            if (self.feedbackLoop and not self.syncIndexes and not self.saveChangeMap and
                    not self.gatherComputationStats and inp.source.dtype == self.weight.dtype):
                pooled = inp
                return self._forward_pooled(pooled)
            # (round 4: a layer WITHOUT feedback loop that keeps a copy of its input, on the split-state kernels -- the
            #  detection's copy-all form takes the 2x2 max on the fly and the pooled map lives in prevInput)
            if (self.copyInput and not self.feedbackLoop and not self.syncIndexes and not self.saveChangeMap and
                    not self.gatherComputationStats and inp.source.dtype == self.weight.dtype and
                    inp.source.is_cuda and (self._split_ok(inp.source.dtype, inp.outSize[-2], inp.outSize[-1]) or
                                            self._hsplit_ok(inp.source.dtype, inp.outSize[-2], inp.outSize[-1]))):
                return self._forward_pooled(inp)
            inp = inp.tensor()           # any other configuration: pool densely, then as usual
        src = inp[1] if type(inp) == tuple else inp
        # a producer that hands out its in-place-updated state (CBPoolMax2d.cloneOutput=False) tags it
        self._inputIsLiveState = bool(getattr(src, '_cbinfer_inplace_state', False))
        if type(inp) == tuple:
            assert inp[0] == 'changeIndexes'
            input = inp[1].detach().contiguous()
            changeIndexes = inp[2]
            if isinstance(changeIndexes, torch.Tensor):
                changeIndexes = changeIndexes.detach().contiguous()
            assert changeIndexes.dim() == 1
        else:
            input = inp.detach().contiguous()
        assert input.size(-3) == self.in_channels
        assert input.dim() == 4 and input.size(0) == 1
        require_device(input)
        assert input.dtype == self.weight.dtype, "input and weights must have the same dtype"
        if (changeIndexes is not None and self.dilatePropagatedIndexes and
                (self.weight.size(2) > 1 or self.weight.size(3) > 1)):
            size = (input.size(-2), input.size(-1))
            if isinstance(changeIndexes, ChangeIndexes) and changeIndexes.size not in (None, size):
                raise _lib.CBinferError("CBConv2d: the propagated change indexes address a %dx%d map, this layer "
                                        "runs at %dx%d" % (changeIndexes.size + size))
            dilated = dilateChangeIndexes(changeIndexes, size, (self.weight.size(2), self.weight.size(3)))
            # (the reference-structured mode hands exact tensors on, conv2d_cg.py:207)
            changeIndexes = dilated.tensor().clone() if self.syncIndexes else dilated

        # (re)allocate the state, +inf => the first frame is 100 % change (conv2d.py:192-199)
        if (self.prevInput.size() != input.size() or self.prevInput.dtype != input.dtype or
                self.prevInput.device != input.device):
            self.prevInput = torch.full_like(input, float('inf'))
        outpSize = list(input.size())
        outpSize[-3] = self.out_channels
        if (not _same_shape(self.prevOutput, outpSize) or self.prevOutput.dtype != input.dtype or
                self.prevOutput.device != input.device):
            self.prevOutput = torch.full(outpSize, float('inf'), dtype=input.dtype,
                                         device=input.device)

        if self.gatherComputationStats:
            self._gatherStats(input)

        if self.syncIndexes:
            changeIndexes = self._forward_ops(input, changeIndexes)
            self._lastIndexes = (changeIndexes if isinstance(changeIndexes, ChangeIndexes) else
                                 ChangeIndexes(changeIndexes, torch.tensor([changeIndexes.numel()],
                                                                          dtype=torch.int32, device=input.device),
                                               (input.size(-2), input.size(-1))))
        else:
            changeIndexes = self._forward_fused(input, changeIndexes)

        if self.propChangeIndexes:
            return 'changeIndexes', self.prevOutput, changeIndexes
        return self.prevOutput

    def _forward_pooled(self, lazy):
        """Feedback-mode layer behind a lazy CBPoolMax2d: detection on pooled values computed on the fly
        + self-compacting contraction (cbinfer_cbconv2d_forward_pooled).  Falls back to dense pooling
        when the mask is too large for the self-compacting kernel."""
        src = lazy.source.detach().contiguous()
        size = lazy.outSize
        H, W = size[-2], size[-1]
        assert size[-3] == self.in_channels and src.dim() == 4 and src.size(0) == 1
        require_device(src)
        if C.cbinfer_mask_words(H, W) > C.cbinfer_frame_mask_max_words():
            return self.forward_normal(lazy.tensor())
        fresh = (tuple(self.prevInput.size()) != tuple(size) or self.prevInput.dtype != src.dtype or
                 self.prevInput.device != src.device)
        if fresh:
            self.prevInput = torch.full(size, float('inf'), dtype=src.dtype, device=src.device)
        outpSize = list(size)
        outpSize[-3] = self.out_channels
        if (not _same_shape(self.prevOutput, outpSize) or self.prevOutput.dtype != src.dtype or
                self.prevOutput.device != src.device):
            self.prevOutput = torch.full(outpSize, float('inf'), dtype=src.dtype, device=src.device)
        self._inputIsLiveState = False
        work = self._workspace(self.prevInput)
        if not work['selfc']:
            return self.forward_normal(lazy.tensor())
        K, Cin, kH, kW = self.weight.size()
        if self._split_ok(src.dtype, H, W):
            return self._forward_split(src, lazy, work)
        if self._hsplit_ok(src.dtype, H, W):
            self._lastIndexes = self._forward_hsplit(src, work, lazy)
            if self.propChangeIndexes:
                return 'changeIndexes', self.prevOutput, self._lastIndexes
            return self.prevOutput
        if not self.feedbackLoop:      # (the other pooled frames refresh the state at the changed pixels only)
            return self.forward_normal(lazy.tensor())
        path = self._rows_path(src.dtype, H, W)
        if path:
            rows = self._rows_workspace(work, H, W, src.device)
            fn, wprep = self._masked_call(path)
            # (a state that was just (re)allocated must see every pixel, whatever the producer rewrote; and the
            #  producer-mask shortcut assumes the skipped segments compared below THIS threshold last frame: the
            #  first frame after a change of the threshold looks at every segment again)
            sameTh = self.__dict__.get('_pmaskThreshold') == float(self.threshold)
            self.__dict__['_pmaskThreshold'] = float(self.threshold)
            pmask = None if (fresh or not sameTh) else lazy.producerMask()
            args = (None, ptr(src), src.size(-2), src.size(-1), ptr(pmask), ptr(self.prevInput),
                    ptr(self.prevOutput), ptr(rows['bits']), ptr(rows['arrive']), ptr(rows['copy']), ptr(wprep),
                    ptr(self.bias.detach()), Cin, H, W, K, kH, kW, float(self.threshold), 1, 0,
                    int(bool(self.withReLU)), stream_ptr(src))
            check(fn(*args))
            self._make_plan(True, src, fn, args, 1, rows=True, pmask=ptr(pmask))
            self._lastIndexes = MaskChangeIndexes(rows['copy'], (H, W), work['idx'], work['count'])
            if self.propChangeIndexes:
                return 'changeIndexes', self.prevOutput, self._lastIndexes
            return self.prevOutput
        arith = self._arith(src)
        args = (ptr(src), src.size(-2), src.size(-1), ptr(self.prevInput), ptr(self.prevOutput),
                ptr(work['bits']), ptr(work['idx']), ptr(work['count']),
                ptr(self._prepared_weights(H, W, arith)), ptr(self.bias.detach()), Cin, H, W, K, kH, kW,
                float(self.threshold), int(bool(self.withReLU)), ptr(work['conv']), arith, stream_ptr(src))
        check(C.cbinfer_cbconv2d_forward_pooled(*args))
        self._make_plan(True, src, C.cbinfer_cbconv2d_forward_pooled, args, 0)
        self._lastIndexes = ChangeIndexes(work['idx'], work['count'], (H, W))
        if self.propChangeIndexes:
            return 'changeIndexes', self.prevOutput, self._lastIndexes
        return self.prevOutput

    def _forward_fused(self, input, changeIndexes):
        """One library call per frame: no host sync (see cbinfer_cbconv2d_forward)."""
        work = self._workspace(input)
        K, Cin, kH, kW = self.weight.size()
        H, W = input.size(-2), input.size(-1)
        have = changeIndexes is not None
        if have and self.feedbackLoop:
            # the reference skips detection here and so never refreshes prevInput (conv2d.py:220-238):
            # its gather would read the +inf initial state.  Refuse instead of computing garbage.
            raise _lib.CBinferError("CBConv2d: feedbackLoop=True cannot be combined with propagated "
                                    "change indexes (the layer state would never be updated)")
        if have:
            if isinstance(changeIndexes, ChangeIndexes):
                idx, count, cap = changeIndexes.buffer, changeIndexes.count, changeIndexes.buffer.numel()
                if changeIndexes.size is not None and changeIndexes.size != (H, W):
                    raise _lib.CBinferError(
                        "CBConv2d: the propagated change indexes address a %dx%d map, this layer runs at "
                        "%dx%d (a CBPoolMax2d in between must hand on down-sampled indexes: "
                        "downsampleIndexes=True)" % (changeIndexes.size + (H, W)))
            else:
                idx, cap = changeIndexes, changeIndexes.numel()
                count = None
            if idx.dtype != torch.int32 or not idx.is_contiguous():
                raise _lib.CBinferError("CBConv2d: propagated change indexes must be a contiguous int32 "
                                        "tensor (conv2d_cg.py:207: nonzero(...).int())")
            cap = min(cap, H * W)
            result = changeIndexes
        else:
            idx, count, cap = work['idx'], work['count'], H * W
            result = ChangeIndexes(idx, count, (H, W))
            if work['selfc'] and not self.saveChangeMap:
                # (round 6: the self-compacting contraction leaves a copy of the frame's change mask at a fixed address
                #  inside the frame mask buffer -- a chained consumer's detection skips the segments this layer left alone)
                off = C.cbinfer_frame_mask_copy_offset(H, W) // 8
                result = MaskChangeIndexes(work['bits'][off:off + C.cbinfer_mask_words(H, W)], (H, W), idx, count,
                                           made=True)
        if have and count is None:
            # exact host-side list: write its length where the kernels look for it
            count = torch.full((1,), cap, dtype=torch.int32, device=input.device)
        prev = self.prevInput
        if not prev.is_contiguous():
            prev = self.prevInput = prev.contiguous()
        mapOut = work['map'] if (self.saveChangeMap and not have) else None
        if not have and work['selfc'] and self._split_ok(input.dtype, H, W):
            self._forward_split(input, None, work)
            return self._lastIndexes
        if not have and work['selfc'] and mapOut is None and self._hsplit_ok(input.dtype, H, W):
            self._lastIndexes = self._forward_hsplit(input, work)
            return self._lastIndexes
        path = self._rows_path(input.dtype, H, W) if (not have and work['selfc']) else None
        if path == 'rows' and self._pairs_ok(H, W):
            # the row-pair kernel: persistent over the non-empty units, and -- with a split-state consumer behind a lazy
            # pool (pycbinfer.fuseDetectionIntoProducer) -- that consumer's pooled change detection in the same launch
            rows = self._rows_workspace(work, H, W, input.device)
            _, wprep = self._masked_call(path)
            nxt, tok = self._next_detect(H, W)
            det = self._pair_detect_ok(nxt, kH)
            if det:
                # round 6: this layer's own detection inside the row-pair launch; the state is refreshed by the consumer's
                # contraction (or, failing that, by a launch of its own behind the consumer's: CBConv2d.forward)
                cons = self.__dict__['_fusedNext'][1]
                old = cons.__dict__.pop('_sidePending', None)
                if old is not None:      # (the consumer was not called last frame)
                    _flush_side(old)
                fn = C.cbinfer_conv_rowpairs_detect
                args = (ptr(input), ptr(prev), ptr(self.prevOutput), ptr(rows['copy']), ptr(wprep),
                        ptr(self.bias.detach()), Cin, H, W, K, kH, kW, float(self.threshold), int(bool(self.withReLU)),
                        ctypes.pointer(nxt), stream_ptr(input))
                check(fn(*args))
                cons.__dict__['_sidePending'] = (input, prev, Cin, H, W, float(self.threshold))      # (tensors: kept alive)
            else:
                fn = C.cbinfer_cbconv2d_forward_rowpairs
                args = (ptr(input), ptr(prev), ptr(self.prevOutput), ptr(rows['bits']), ptr(rows['arrive']),
                        ptr(rows['copy']), ptr(wprep), ptr(self.bias.detach()), Cin, H, W, K, kH, kW,
                        float(self.threshold), int(bool(self.withReLU)),
                        ctypes.pointer(nxt) if nxt is not None else None, stream_ptr(input))
                check(fn(*args))
            result = MaskChangeIndexes(rows['copy'], (H, W), work['idx'], work['count'])
            result.nextDetect = tok
            if not self._inputIsLiveState:
                self._make_plan(False, input, fn, args, 0, rows=True)
                if self._plan is not None:
                    self._plan['pairs'], self._plan['nextToken'], self._plan['keep'] = True, tok, nxt
                    self._plan['det'] = det
            cap = 0      # (done)
        elif path:
            rows = self._rows_workspace(work, H, W, input.device)
            fn, wprep = self._masked_call(path)
            args = (ptr(input), None, 0, 0, None, ptr(prev), ptr(self.prevOutput), ptr(rows['bits']),
                    ptr(rows['arrive']), ptr(rows['copy']), ptr(wprep),
                    ptr(self.bias.detach()), Cin, H, W, K, kH, kW, float(self.threshold),
                    int(bool(self.feedbackLoop)), int(bool(self.copyInput)), int(bool(self.withReLU)),
                    stream_ptr(input))
            check(fn(*args))
            if not self._inputIsLiveState:
                self._make_plan(False, input, fn, args, 0, rows=True)
            result = MaskChangeIndexes(rows['copy'], (H, W), work['idx'], work['count'])
            cap = 0      # (done)
        if cap > 0:
            arith = self._arith(input)
            args = (ptr(input), ptr(prev), ptr(self.prevOutput), None if have else ptr(work['bits']),
                    ptr(idx), ptr(count), ptr(mapOut), ptr(self._prepared_weights(H, W, arith)),
                    ptr(self.bias.detach()), Cin, H, W, K, kH, kW, float(self.threshold),
                    int(bool(self.feedbackLoop)), int(bool(self.copyInput)), int(bool(self.withReLU)),
                    int(have), cap, ptr(work['conv']),
                    int(work['selfc'] and not have), arith, stream_ptr(input))
            check(C.cbinfer_cbconv2d_forward(*args))
            if not have and not self._inputIsLiveState:
                if work['selfc'] and mapOut is None:
                    # replayed through the chained entry: a frame in which the layer that produced `input` rewrote
                    # nothing ends both launches at once (_upstream_count decides per frame whether that may be said)
                    cargs = (None,) + args[:6] + args[7:19] + (args[21], args[23], args[24])
                    self._make_plan(False, input, C.cbinfer_cbconv2d_forward_after, cargs, 1)
                    if self._plan is not None:
                        self._plan['chain'] = True
                        self._plan['indexes'] = result
                else:
                    self._make_plan(False, input, C.cbinfer_cbconv2d_forward, args, 0)
            if work['selfc'] and not have:
                self._publish_count(work['count'])
        if mapOut is not None:
            # (a view of the module's work buffer, rewritten by the next frame -- clone it to keep it; the
            #  reference allocates a fresh map per frame)
            self.changeMap = mapOut
        if not self.feedbackLoop and not self.copyInput:
            # alias, conv2d.py:237-238 (a producer's in-place-updated state is copied first)
            self.prevInput = input.clone() if self._inputIsLiveState else input
        self._lastIndexes = result
        return result

    def _forward_ops(self, input, changeIndexes):
        """The reference's op sequence (conv2d.py:220-251), one kernel launch per op, blocking on the
        change count like torch.nonzero does."""
        if isinstance(changeIndexes, ChangeIndexes):
            changeIndexes = changeIndexes.tensor()
        if changeIndexes is None:
            changeMap = changeDetection(input, self.prevInput, self.kernel_size, self.threshold,
                                        updateInputState=self.feedbackLoop)
            if self.saveChangeMap:
                self.changeMap = changeMap
            changeIndexes = changeIndexesExtr(changeMap)
        if not self.feedbackLoop:
            if self.copyInput:
                self.prevInput.copy_(input)
            else:
                self.prevInput = input.clone() if self._inputIsLiveState else input
        if changeIndexes.numel() != 0:
            Xmatrix = genXMatrix(self.prevInput, changeIndexes, self.kernel_size)
            Ymatrix = matrixMult(Xmatrix, self.weight.detach(), self.bias.detach(), transposeOut=True,
                                 weightsPrepared=self._prepared_weights())
            updateOutput(Ymatrix, changeIndexes, self.prevOutput, withReLU=self.withReLU)
        return changeIndexes

    def _gatherStats(self, input):
        """Operation counts behind the 'effective GOp/s' metric (conv2d.py:201-218)."""
        changeTensor = (input - self.prevInput).abs().gt(self.threshold)
        nC = changeTensor.size(-3)
        kH, kW = self.weight.size(2), self.weight.size(3)
        proped = F.conv2d(changeTensor.float(),
                          torch.ones(nC, 1, kH, kW, device=input.device), groups=nC).gt(0)
        opsPerValue = self.weight.size(0) * kH * kW * 2
        self.compStats = dict(
            numInputChangesPerFeatureMap=changeTensor.sum() * opsPerValue,
            numInputChanges=changeTensor.sum(-3).gt(0).sum() * nC * opsPerValue,
            numInputPropedChangesPerFeatureMap=proped.sum() * opsPerValue,
            numInputPropedChanges=proped.sum(-3).gt(0).sum() * nC * opsPerValue,
            totalInputValues=changeTensor.size(-1) * changeTensor.size(-2) * nC * opsPerValue)

    # ---------------------------------------------------------------- per-frame fast path
    # The sync-free forward is one library call; everything around it (shape checks, state allocation,
    # workspace and weight lookups, pointer extraction) is invariant from frame to frame.  After a frame
    # went through the general path, the call is kept as a plan -- a pre-built argument list plus the few
    # facts that must still hold -- and replayed while they hold; anything else falls back.  This is what
    # keeps the host ahead of the GPU without graph capture (~10 instead of ~35 us per layer and frame).
    def _flags(self):
        return (self.threshold, self.feedbackLoop, self.copyInput, self.withReLU, self.propChangeIndexes,
                self.syncIndexes, self.saveChangeMap, self.gatherComputationStats, self.finegrained,
                self.atomicFG, self.fgInPlace, self.exactF32, self.dilatePropagatedIndexes)

    def _make_plan(self, pooled, src, fn, args, srcSlot, result=None, rows=False, pmask=None):
        """Remember a finished sync-free call: fn(*args) with args[srcSlot] = source pointer, args[-1] =
        stream.  Only for configurations whose call does not depend on per-frame host state."""
        if self.syncIndexes or self.saveChangeMap or self.gatherComputationStats:
            return
        if not (self.feedbackLoop or self.copyInput or result is not None):
            return
        w, b = self._parameters.get('weight'), self._parameters.get('bias')
        if w is None or b is None or os.environ.get('CBINFER_NO_FASTPATH', '0') == '1':
            return
        work = self._work
        self._plan = dict(
            pooled=pooled, shape=tuple(src.shape), dtype=src.dtype, device=src.device, flags=self._flags(),
            w=(w.data_ptr(), w._version), b=(b.data_ptr(), b._version),
            state=(self._buffers['prevInput'].data_ptr(), self._buffers['prevOutput'].data_ptr()),
            stream=args[-1], work=work, fn=fn, args=list(args), srcSlot=srcSlot, result=result, rows=rows, pmask=pmask,
            indexes=ChangeIndexes(work['idx'], work['count'], work['key'][:2]))

    def _plan_source(self, inp):
        """The source tensor if the kept call plan (not a split-state fp32 one) applies to `inp` as it stands -- no side
        effects --, else None."""
        plan = self._plan
        if plan['pooled']:
            if type(inp) is not LazyPool:
                return None
            src = inp.source
            if plan['rows'] or plan.get('checkPmask'):      # the producer's mask is baked into the call: it must still be the one offered
                pm = inp.producerMask()
                if (pm.data_ptr() if pm is not None else None) != plan['pmask']:
                    return None
        else:
            if type(inp) is not torch.Tensor:
                return None
            src = inp
        w, b, bufs = self._parameters['weight'], self._parameters['bias'], self._buffers
        if (src.shape != plan['shape'] or src.dtype != plan['dtype'] or src.device != plan['device'] or
                not src.is_contiguous() or self._flags() != plan['flags'] or
                (w.data_ptr(), w._version) != plan['w'] or (b.data_ptr(), b._version) != plan['b'] or
                (bufs['prevInput'].data_ptr(), bufs['prevOutput'].data_ptr()) != plan['state'] or
                self._work is not plan['work'] or raw_stream(src.device.index) != plan['stream']):
            return None
        if plan.get('stateVersion') is not None and bufs['prevInput']._version != plan['stateVersion']:
            return None      # (somebody wrote prevInput through torch: its pixel-major copy must be made again)
        return src

    def _run_plan(self, inp):
        plan = self._plan
        if plan.get('split'):
            return self._run_split_plan(inp)
        src = self._plan_source(inp)
        if src is None:
            return None
        bufs = self._buffers
        if plan.get('hsplit'):
            tokens = self._prepare_hsplit(plan, src, bufs)
            status = plan['fn'](*plan['args'])
            if status != 0:
                check(status)
            return self._finish_hsplit(plan, tokens, bufs)
        if plan.get('pairs'):
            # the next layer's detection rides in this launch: the plan holds only while that layer's state is the one
            # the plan was made for (and starts to fold as soon as it can)
            if self._next_detect(plan['shape'][-2], plan['shape'][-1])[1] != plan['nextToken']:
                return None
            if plan.get('det') != self._pair_detect_ok(plan['keep'], self.weight.size(2)):
                return None
            if plan.get('det'):
                cons = self.__dict__['_fusedNext'][1]
                old = cons.__dict__.pop('_sidePending', None)
                if old is not None:
                    _flush_side(old)
                cons.__dict__['_sidePending'] = (src, bufs['prevInput'], src.size(1), src.size(-2), src.size(-1),
                                                 float(self.threshold))
        args = plan['args']
        if plan.get('fgSplit'):
            if plan['arith'] != os.environ.get('CBINFER_ARITH', 'x3'):
                return None
            t = plan.get('tail')
            if self._fusedTailCandidate() is not plan['tailCand'] or (
                    t is not None and t._fold_key() != plan['tailKey']):
                return None
            plan['seq'].input = src.data_ptr()
        else:
            args[plan['srcSlot']] = src.data_ptr()
        chain = plan.get('chain')
        if chain:
            up = self.__dict__.get('_upNow')
            args[0] = up.data_ptr() if up is not None else None
            cm = plan.get('chainMask')
            if cm is not None:      # (the producer's mask of this frame, under the chain's own conditions)
                pm = self._chain_mask(cm[1], cm[2])
                args[cm[0]] = pm.data_ptr() if pm is not None else None
        status = plan['fn'](*args)
        if status != 0:
            check(status)
        if chain:
            self._publish_count(plan['work']['count'])
        if plan.get('fgSplit'):
            self._poll_range(plan['work']['split'])
        self._inputIsLiveState = False
        if plan['result'] is not None:          # fine-grained in-place frame: prevOutput or its relu'd copy
            res = plan['result']
            if plan['rows']:                    # mask-driven: this frame's list is made from its own mask copy
                work = plan['work']
                self._lastIndexes = MaskChangeIndexes(work['rows']['copy'], work['key'][:2], work['idx'],
                                                      work['count'])
                if isinstance(res, tuple):
                    res = (res[0], res[1], self._lastIndexes)
            return res
        if plan['rows']:                        # mask-driven frame: the list is made when somebody asks
            work = plan['work']
            self._lastIndexes = MaskChangeIndexes(work['rows']['copy'], work['key'][:2], work['idx'],
                                                  work['count'])
            self._lastIndexes.nextDetect = plan.get('nextToken')
        else:
            self._lastIndexes = plan['indexes']
        if self.propChangeIndexes:
            return 'changeIndexes', bufs['prevOutput'], self._lastIndexes
        return bufs['prevOutput']

    def _prepare_hsplit(self, plan, src, bufs):
        """The per-frame part of an fp16 split-state frame in front of the library call (the plan's invariants hold): input
        pointer, the chain's count and mask, whose detection rides where.  Returns the consumers' tokens."""
        L, hs = plan['layer'], plan['hs']
        H, W = plan['size']
        L.input = src.data_ptr()
        up = self.__dict__.get('_upNow')
        L.upstreamCount = up.data_ptr() if up is not None else None
        if not plan['pooled']:
            pm = self._chain_mask(H, W)
            L.producerMask = pm.data_ptr() if pm is not None else None
            L.detect = 0 if self._detected_upstream(self._half_token(hs, bufs['prevInput'])) else 1
        return self._fill_consumers(L, H, W) if '_fusedConsumers' in self.__dict__ else ()

    def _finish_hsplit(self, plan, tokens, bufs):
        self.__dict__['_ranSplit'] = True
        self._publish_count(plan['work']['count'], tokens)
        self._inputIsLiveState = False
        self._lastIndexes = plan['indexes']
        if self.propChangeIndexes:
            return 'changeIndexes', bufs['prevOutput'], self._lastIndexes
        return bufs['prevOutput']

    # ---------------------------------------------------------------- chains of change-based layers
    # A layer whose self-compacting contraction ran leaves its change count on the device (work['count']) and says so
    # on its output buffer: (module, frame serial, buffer version, count).  The next CBConv2d that is handed this
    # very buffer may tell its two launches where that count is (cbinfer_cbconv2d_forward_after): zero there and the
    # frame is over for it after a scalar load each -- the buffer is what it was, nothing can have changed.  What
    # makes that exact is checked here, on the host, every frame: the tag is of the producer's LATEST forward, nobody
    # wrote to the buffer through torch since (version counter), this layer's previous forward consumed the
    # producer's previous frame from the same buffer, and nobody rewrote this layer's state through torch either.
    # (Threshold and mode are pinned by the plan that carries the call.)  CBINFER_NO_CHAIN=1 switches it off.
    def _publish_count(self, count, tokens=()):
        out = self._buffers['prevOutput']
        out._cbProduced = _Produced(self, self.__dict__.get('_serial', 0), out._version, count, tokens)

    def _chain_mask(self, H, W):
        """The change mask the PRODUCING layer of a chain left this frame (its MaskChangeIndexes' mask copy, H x W) while
        the chain's conditions hold (_note_upstream: the input is that layer's output buffer of its latest frame, this
        layer consumed its previous one from the same buffer into the same state) -- else None.  CBINFER_NO_CHAINMASK=1
        switches it off."""
        d = self.__dict__
        if d.get('_upNow') is None or os.environ.get('CBINFER_NO_CHAINMASK', '0') == '1':
            return None
        ix = getattr(d['_upSeen'][0], '_lastIndexes', None)
        if isinstance(ix, MaskChangeIndexes) and tuple(ix.size) == (H, W) and ix._mask is not None:
            return ix._mask
        return None

    def _note_upstream(self, inp):
        d = self.__dict__
        d['_serial'] = d.get('_serial', 0) + 1
        tag = getattr(inp, '_cbProduced', None) if type(inp) is torch.Tensor else None
        seen, now = None, None
        d['_upTokens'] = tag.tokens if tag is not None else None
        if tag is not None:
            prod, serial, version, count = tag.module, tag.serial, tag.version, tag.count
            pin = self._buffers.get('prevInput')
            if (prod is not self and prod.__dict__.get('_serial') == serial and inp._version == version and
                    pin is not None and not _NO_CHAIN):
                seen = (prod, serial, inp.data_ptr(), pin.data_ptr(), pin._version)
                last = d.get('_upSeen')
                if last is not None and last[0] is prod and last[1] == serial - 1 and last[2:] == seen[2:]:
                    now = count
        d['_upSeen'], d['_upNow'] = seen, now

    def forward(self, inp):
        out = self._forward(inp)
        # round 6: the row-pair layer in front detected its changes inside its own launch and left its state alone
        # (cbinfer_conv_rowpairs_detect); this layer's contraction carries that refresh on its idle workgroups when it runs in
        # window order (cbinfer_split_conv_next_refresh) -- and whenever it did not, the refresh is a launch of its own, here
        pend = self.__dict__.pop('_sidePending', None)
        if pend is not None:
            _flush_side(pend)
        return out

    def _forward(self, inp):
        self._note_upstream(inp)
        if self.__dict__.get('_plan') is not None:
            out = self._run_plan(inp)
            if out is not None:
                return out
            self._plan = None
        self._setDefaultValues()
        self.__dict__['_ranSplit'] = False
        if self.finegrained:
            assert self.feedbackLoop == False
            out = self.forward_fg(inp)
        else:
            out = self.forward_normal(inp)
        if not self.__dict__['_ranSplit']:
            # A frame on any other path refreshes prevInput through raw pointers (neither its address nor its
            # version counter moves): the pre-split copy of the state is stale from here on and is made again from
            # prevInput when the module returns to the split-state kernels
            sp = self._work.get('split') if self._work else None
            if sp is not None:
                sp['stateKey'] = None
            hs = self._work.get('hsplit') if self._work else None
            if hs is not None:
                hs['stateKey'] = None
        return out

    def __repr__(self):
        """One line in the reference's format (conv2d.py:271-290): fixed head, then the conv attributes
        that differ from their defaults, then the change-based flags."""
        self._setDefaultValues()
        optional = [('pad', self.padding, (0,) * len(self.padding)),
                    ('dilation', self.dilation, (1,) * len(self.dilation)),
                    ('outpad', self.output_padding, (0,) * len(self.output_padding)),
                    ('grp', self.groups, 1)]
        parts = ['th=%s' % (self.threshold,), '%s->%s' % (self.in_channels, self.out_channels),
                 'k=%s' % (self.kernel_size,), 's=%s' % (self.stride,), 'copyInput=%s' % (self.copyInput,)]
        parts += ['%s=%s' % (label, value) for label, value, default in optional if value != default]
        if self.bias is None:
            parts.append('bias=False')
        if self.withReLU:
            parts.append('withReLU=%s' % (self.withReLU,))
        parts.append('propChgIdxs=%s' % (self.propChangeIndexes,))
        return '%s (%s)' % (self.__class__.__name__, ', '.join(parts))


class CBTail1x1(nn.Module):
    """conv1x1 -> [ReLU] -> conv1x1 evaluated in one launch at the pixels of the change list handed on by
    the CBConv2d in front of it (tuple protocol, conv2d.py:180-186); every other output pixel keeps its
    value.  Stands for two CBConv2d fed by propagated change indexes (sceneLabeling/modelLoader.py:41-44)
    or for the dense 1x1 tail the other experiments keep (:45-47); built by pycbinfer.fuseTail1x1().
    The parameters are shared with the source modules."""

    @staticmethod
    def maxHidden():
        return int(C.cbinfer_tail1x1_max_hidden())

    @staticmethod
    def supported(C0, C1, C2):
        """Channel counts the one-launch kernel takes (hidden width and its LDS budget)."""
        return bool(C.cbinfer_tail1x1_supported(int(C0), int(C1), int(C2)))

    @staticmethod
    def accepts(m):
        def pair(v):
            return tuple(v) if isinstance(v, (tuple, list)) else (v, v)
        return (pair(m.kernel_size) == (1, 1) and pair(m.stride) == (1, 1) and pair(m.padding) == (0, 0) and
                pair(m.dilation) == (1, 1) and m.groups == 1 and m.bias is not None)

    def __init__(self, conv1, conv2, relu=True):
        super(CBTail1x1, self).__init__()
        assert CBTail1x1.accepts(conv1) and CBTail1x1.accepts(conv2)
        assert conv1.out_channels == conv2.in_channels
        if not CBTail1x1.supported(conv1.in_channels, conv1.out_channels, conv2.out_channels):
            raise _lib.CBinferError("CBTail1x1: %d->%d->%d channels exceed the kernel's hidden width or LDS budget"
                                    % (conv1.in_channels, conv1.out_channels, conv2.out_channels))
        self.weight1, self.bias1 = conv1.weight, conv1.bias
        self.weight2, self.bias2 = conv2.weight, conv2.bias
        self.in_channels, self.hidden_channels, self.out_channels = (
            conv1.in_channels, conv1.out_channels, conv2.out_channels)
        self.relu = bool(relu)
        self.withReLU = False
        self.propChangeIndexes = False
        self.register_buffer('prevOutput', torch.zeros(0))
        self._w1prep = None

    def clearMemory(self):
        self.prevOutput = self.weight1.detach().new_zeros(0)

    def getStateTensors(self):
        return [self.prevOutput]

    def __getstate__(self):
        d = dict(self.__dict__)
        d['_w1prep'] = None
        return d

    def _prepared(self):
        w = self.weight1
        key = (w.data_ptr(), w._version, w.device)
        if self._w1prep is None or self._w1prep[0] != key:
            nbytes = C.cbinfer_tail1x1_prepared_bytes(self.hidden_channels, self.in_channels)
            wp = torch.empty(nbytes // 4, dtype=torch.float32, device=w.device)
            check(C.cbinfer_tail1x1_prep(ptr(w.detach().contiguous()), ptr(wp), self.hidden_channels,
                                         self.in_channels, stream_ptr(w)))
            self._w1prep = (key, wp)
        return self._w1prep[1]

    def _output_for(self, H, W, device, dtype):
        size = (1, self.out_channels, H, W)
        if not _same_shape(self.prevOutput, size) or self.prevOutput.device != device:
            self.prevOutput = torch.full(size, float('inf'), dtype=dtype, device=device)
        return self.prevOutput

    def _fold_key(self):
        """What a producing layer's call plan that folds this tail into its own launch depends on."""
        ts = (self.weight1, self.bias1, self.weight2, self.bias2)
        return (tuple((t.data_ptr(), t._version) for t in ts), self._buffers['prevOutput'].data_ptr(), self.relu,
                bool(self.withReLU))

    def forward(self, inp):
        assert type(inp) == tuple and inp[0] == 'changeIndexes', \
            "CBTail1x1 needs the ('changeIndexes', tensor, indexes) tuple of a CBConv2d with propChangeIndexes"
        if getattr(inp[2], 'tailDone', None) is self:
            # the producing layer evaluated this tail in its own second launch (cbinfer_split_forward_tail)
            if self.propChangeIndexes:
                return 'changeIndexes', self.prevOutput, inp[2]
            return self.prevOutput
        x, indexes = inp[1].detach().contiguous(), inp[2]
        require_device(x)
        assert x.dim() == 4 and x.size(0) == 1 and x.size(1) == self.in_channels
        if x.dtype != torch.float32:
            raise _lib.CBinferError("CBTail1x1 is fp32 only")
        H, W = x.size(-2), x.size(-1)
        self._output_for(H, W, x.device, x.dtype)
        if isinstance(indexes, ChangeIndexes):
            idx, count, cap = indexes.buffer, indexes.count, min(indexes.buffer.numel(), H * W)
        else:
            idx = indexes.detach().contiguous()
            assert idx.dim() == 1 and idx.dtype == torch.int32
            count, cap = None, idx.numel()
        if cap > 0:
            check(C.cbinfer_tail1x1(ptr(x), ptr(idx), cap, ptr(count), ptr(self._prepared()),
                                    ptr(self.bias1.detach()), ptr(self.weight2.detach().contiguous()),
                                    ptr(self.bias2.detach()), ptr(self.prevOutput), self.in_channels,
                                    self.hidden_channels, self.out_channels, H, W, int(self.relu),
                                    int(bool(self.withReLU)), stream_ptr(x)))
        if self.propChangeIndexes:
            return 'changeIndexes', self.prevOutput, indexes
        return self.prevOutput

    def __repr__(self):
        return '%s (%s->%s->%s, relu=%s, propChgIdxs=%s)' % (
            self.__class__.__name__, self.in_channels, self.hidden_channels, self.out_channels, self.relu,
            self.propChangeIndexes)
