"""Several independent video sequences through ONE converted network, one launch per step for all of them
("several sequences in flight per GPU (batched GEMMs across sequences)").

The reference is batch-1 by construction (`assert input.size(0) == 1`, conv2d_cg.py:61; one [H,W] change map per
layer): every sequence needs its own prevInput / prevOutput / change masks.  A single sequence at 480x320 is a
chain of eight small dependent launches, each far too small for 256 CUs, so the frame time is a sum of launch
latencies.  `SequenceBatch` keeps S state sets beside one set of (shared, read-only) weights and issues each step
of the frame ONCE for all S sequences -- the kernels take a table of per-sequence tensors and find their work
items across all of them (cb_split.hip; blockIdx.y = sequence in the detection / row-segment / tail kernels) --
so the fixed cost of a launch is paid once per S frames and a short change list no longer leaves CUs idle.
Per sequence the results are bit-identical to running that sequence alone through its own copy of the network
(tests/test_gpu_batch.py).

Supported layer chain (what pycbinfer.fuseTail1x1 + fusePoolingIntoDetection make of the reference's experiment
5/6 networks, sceneLabeling/modelLoader.py:62-78): feedback-mode CBConv2d layers whose frame runs on the
row-segment kernel (plain tensor input) or on the split-state kernels (plain or lazily pooled input), lazy
CBPoolMax2d between them, an optional CBTail1x1 at the end.  Anything else raises: run such networks one
sequence per stream instead.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import C, CBinferError, check, ptr, stream_ptr
from .conv2d import CBConv2d, CBPoolMax2d, CBTail1x1


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() if t is not None else None for t in tensors])


class SequenceBatch(object):
    def __init__(self, net, sequences):
        self.S = int(sequences)
        if not 1 <= self.S <= C.cbinfer_split_max_sequences():
            raise CBinferError("SequenceBatch: 1..%d sequences per launch" % C.cbinfer_split_max_sequences())
        self.net = net
        self.layers = None          # built on the first frame (needs the frame size)
        self._key = None

    # ------------------------------------------------------------------------------------------
    def clearMemory(self):
        self.layers = None
        self._key = None

    def states(self, q):
        """[(prevInput, prevOutput)] of the CBConv2d layers (and (None, prevOutput) of a tail) of sequence q."""
        return [(l.get('prevInput', [None] * self.S)[q], l['prevOutput'][q]) for l in self.layers if 'prevOutput' in l]

    def changeCounts(self, q):
        """Changed pixels per CBConv2d of the last frame of sequence q (host sync; split layers only)."""
        return [int(l['count'][q].item()) for l in self.layers if l['kind'] == 'split']

    # ------------------------------------------------------------------------------------------
    def _build(self, frame):
        dev = frame.device
        S = self.S
        layers = []
        shape = tuple(frame.shape)          # (1, C, H, W) of what the next layer sees
        pooledFrom = None                   # set behind a lazy pool: the producing layer's index
        kids = list(self.net.children())
        if len(_lib.SplitTail().output) != C.cbinfer_split_max_sequences():
            raise CBinferError("SequenceBatch: cbSplitTail.output[] and cbinfer_split_max_sequences() disagree")
        for pos, m in enumerate(kids):
            if type(m) is CBPoolMax2d:
                if not layers:
                    raise CBinferError("SequenceBatch: a CBPoolMax2d needs a producing CBConv2d in front of it")
                if not getattr(m, 'lazy', False) or m.propChangeIndexes:
                    raise CBinferError("SequenceBatch: CBPoolMax2d must be folded into the next detection "
                                       "(pycbinfer.fusePoolingIntoDetection)")
                _, c, h, w = shape
                oh, ow = ((h - 1) // 2 + 1, (w - 1) // 2 + 1) if m.ceil_mode else (h // 2, w // 2)
                pooledFrom = (len(layers) - 1, h, w)
                shape = (1, c, oh, ow)
                continue
            if type(m) is CBTail1x1:
                if not layers or layers[-1]['kind'] != 'split' or pooledFrom is not None:
                    raise CBinferError("SequenceBatch: CBTail1x1 must follow a split-state CBConv2d")
                if pos != len(kids) - 1:
                    raise CBinferError("SequenceBatch: a CBTail1x1 must be the network's last module")
                prod = layers[-1]
                _, c, h, w = shape
                L = dict(kind='tail', m=m, H=h, W=w,
                         prevOutput=[torch.full((1, m.out_channels, h, w), float('inf'), device=dev)
                                     for _ in range(S)])
                seqs = (_lib.TailSeq * S)()
                for q in range(S):
                    seqs[q].input, seqs[q].changeList = prod['prevOutput'][q].data_ptr(), prod['idx'][q].data_ptr()
                    seqs[q].countDev, seqs[q].output = prod['count'][q].data_ptr(), L['prevOutput'][q].data_ptr()
                L['seqs'] = seqs
                # ... or rides in the producing layer's second launch (cbinfer_split_forward_tail)
                pm = prod['m']
                pK, pC, pkH, pkW = pm.weight.size()
                L['folded'] = bool(prod['ws'] is not None and m.in_channels == pK and
                                   os.environ.get('CBINFER_NO_TAILFOLD', '0') != '1' and
                                   C.cbinfer_split_tail_supported(pC, pK, pkH, pkW, m.hidden_channels, m.out_channels))
                if L['folded']:
                    st = _lib.SplitTail()
                    L['keep'] = (m._prepared(), m.bias1.detach(), m.weight2.detach().contiguous(), m.bias2.detach())
                    st.w1Prepared, st.b1, st.w2, st.b2 = [x.data_ptr() for x in L['keep']]
                    st.C1, st.C2, st.relu1, st.relu2 = (m.hidden_channels, m.out_channels, int(m.relu),
                                                        int(bool(m.withReLU)))
                    for q in range(S):
                        st.output[q] = L['prevOutput'][q].data_ptr()
                    prod['tail'] = st
                layers.append(L)
                shape = (1, m.out_channels, h, w)
                continue
            if type(m) is not CBConv2d:
                raise CBinferError("SequenceBatch: unsupported module %s" % type(m).__name__)
            if not m.feedbackLoop or m.finegrained or m.syncIndexes or m.saveChangeMap or m.gatherComputationStats:
                raise CBinferError("SequenceBatch: CBConv2d layers must run the sync-free feedback-mode frame")
            K, Cin, kH, kW = m.weight.size()
            _, c, h, w = shape
            assert c == Cin
            words = C.cbinfer_mask_words(h, w)
            L = dict(m=m, H=h, W=w, K=K, C=Cin, kH=kH, kW=kW,
                     prevInput=[torch.full((1, Cin, h, w), float('inf'), device=dev) for _ in range(S)],
                     prevOutput=[torch.full((1, K, h, w), float('inf'), device=dev) for _ in range(S)],
                     copy=[torch.zeros(words, dtype=torch.int64, device=dev) for _ in range(S)])
            if m._split_ok(torch.float32, h, w) and S * words <= C.cbinfer_split_max_mask_words(K):
                L['kind'] = 'split'
                wp, scale = m._split_weights(h, w)
                L['wp'], L['scale'] = wp, scale
                x3 = L['x3'] = scale == 0.0      # (bf16 triples -- CBConv2d._split_arith, the modules' own choice)
                nbytes = (C.cbinfer_split3_state_bytes if x3 else C.cbinfer_split_state_bytes)(Cin, h, w, kH, kW)
                L['S'] = [torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(S)]
                L['masks'] = [torch.zeros(C.cbinfer_frame_mask_bytes(h, w) // 8, dtype=torch.int64, device=dev)
                              for _ in range(S)]
                L['idx'] = [torch.empty(h * w, dtype=torch.int32, device=dev) for _ in range(S)]
                L['count'] = [torch.zeros(1, dtype=torch.int32, device=dev) for _ in range(S)]
                L['flag'] = torch.zeros(1, dtype=torch.int32, device=dev)
                wsBytes = C.cbinfer_split_workspace_bytes(S, Cin, h, w, K, kH, kW)
                L['ws'] = torch.zeros(wsBytes, dtype=torch.uint8, device=dev) if wsBytes > 0 else None
                for q in range(S):
                    if x3:
                        check(C.cbinfer_split3_state_init(ptr(L['S'][q]), Cin, h, w, kH, kW, stream_ptr(frame)))
                        check(C.cbinfer_split3_state_rebuild(ptr(L['prevInput'][q]), ptr(L['S'][q]), Cin, h, w, kH, kW,
                                                             stream_ptr(frame)))
                        continue
                    check(C.cbinfer_split_state_init(ptr(L['S'][q]), Cin, h, w, kH, kW, stream_ptr(frame)))
                    check(C.cbinfer_split_state_rebuild(ptr(L['prevInput'][q]), ptr(L['S'][q]), Cin, h, w, kH, kW,
                                                        ptr(L['flag']), stream_ptr(frame)))
                L['pooled'] = pooledFrom
                seqs = (_lib.SplitSeq * S)()
                for q in range(S):
                    sq = seqs[q]
                    sq.state, sq.splitState = L['prevInput'][q].data_ptr(), L['S'][q].data_ptr()
                    sq.frameMasks, sq.output = L['masks'][q].data_ptr(), L['prevOutput'][q].data_ptr()
                    sq.idxOut, sq.countOut = L['idx'][q].data_ptr(), L['count'][q].data_ptr()
                    sq.rangeFlag, sq.maskCopy = L['flag'].data_ptr(), L['copy'][q].data_ptr()
                    if pooledFrom is not None:
                        prod = layers[pooledFrom[0]]
                        sq.input = prod['prevOutput'][q].data_ptr()
                        sq.producerMask = None          # (first frame: a fresh state must see every pixel)
                L['seqs'] = seqs
                L['th'] = None
            elif pooledFrom is None and m._rows_path(torch.float32, h, w) == 'rows':
                L['kind'] = 'rows'
                L['wp'] = m._masked_call('rows')[1]
                L['bits'] = [torch.zeros(words, dtype=torch.int64, device=dev) for _ in range(S)]
                L['arrive'] = [torch.zeros(words, dtype=torch.int32, device=dev) for _ in range(S)]
                L['a_state'] = _ptr_array(L['prevInput'])
                L['a_out'] = _ptr_array(L['prevOutput'])
                L['a_bits'] = _ptr_array(L['bits'])
                L['a_arrive'] = _ptr_array(L['arrive'])
                L['a_copy'] = _ptr_array(L['copy'])
                # the row-pair kernel (the modules' own choice for such a layer: the same arithmetic, so a sequence
                # gets the same bits alone and inside a batch), one launch for all sequences
                L['pairs'] = m._pairs_ok(h, w)
                if L['pairs']:
                    ps = (_lib.PairSeq * S)()
                    for q in range(S):
                        ps[q].state, ps[q].output = L['prevInput'][q].data_ptr(), L['prevOutput'][q].data_ptr()
                        ps[q].bits, ps[q].maskCopy = L['bits'][q].data_ptr(), L['copy'][q].data_ptr()
                    L['pseq'] = ps
            else:
                raise CBinferError("SequenceBatch: the %d->%d layer has no batched kernel at %dx%d" % (Cin, K, h, w))
            layers.append(L)
            pooledFrom = None
            shape = (1, K, h, w)
        if pooledFrom is not None:
            raise CBinferError("SequenceBatch: a lazy pool needs a consuming CBConv2d")
        # a row-pair layer whose consumer behind the lazy pool is a split-state layer does that layer's pooled change
        # detection inside its own launch (pycbinfer.fuseDetectionIntoProducer; cb_rowpair.hip)
        for li, L in enumerate(layers):
            if (L['kind'] == 'split' and L.get('pooled') is not None and layers[L['pooled'][0]].get('pairs') and
                    layers[L['pooled'][0]]['K'] == 16 and L['C'] == 16 and L['pooled'][0] == li - 1 and
                    layers[li - 1]['m'].__dict__.get('_fusedNext') is not None and
                    os.environ.get('CBINFER_NO_NEXTFOLD', '0') != '1'):
                P = layers[li - 1]
                nd = _lib.NextDetect()
                nd.H, nd.W, nd.kH, nd.kW = L['H'], L['W'], L['kH'], L['kW']
                nd.arith = 1 if L['x3'] else 0
                for q in range(S):
                    P['pseq'][q].nextState = L['prevInput'][q].data_ptr()
                    P['pseq'][q].nextSplitState = L['S'][q].data_ptr()
                    P['pseq'][q].nextFrameMasks = L['masks'][q].data_ptr()
                    P['pseq'][q].nextRangeFlag = L['flag'].data_ptr()
                P['next'] = (li, nd)
        for L in layers:
            L['wkey'] = self._weight_key(L['m'])
        self.layers = layers
        self._first = True

    @staticmethod
    def _weight_key(m):
        ts = (m.weight1, m.bias1, m.weight2, m.bias2) if type(m) is CBTail1x1 else (m.weight, m.bias)
        return tuple((t.data_ptr(), t._version) for t in ts)

    def _refresh_weights(self, li):
        """The prepared copies of a layer's weights after the module's parameters were written (the states stay)."""
        L = self.layers[li]
        m = L['m']
        if L['kind'] == 'split':
            L['wp'], L['scale'] = m._split_weights(L['H'], L['W'])
            if (L['scale'] == 0.0) != L['x3']:
                raise CBinferError("SequenceBatch: CBINFER_ARITH changed since the batch was built (its split states "
                                   "hold the other arithmetic's records)")
        elif L['kind'] == 'rows':
            L['wp'] = m._masked_call('rows')[1]
        if L['kind'] == 'tail' or L.get('tail') is not None:
            t = m if L['kind'] == 'tail' else self.layers[li + 1]['m']
            owner = L if L['kind'] == 'tail' else self.layers[li + 1]
            owner['keep'] = (t._prepared(), t.bias1.detach(), t.weight2.detach().contiguous(), t.bias2.detach())
            st = (self.layers[li - 1] if L['kind'] == 'tail' else L).get('tail')
            if st is not None:
                st.w1Prepared, st.b1, st.w2, st.b2 = [x.data_ptr() for x in owner['keep']]
        L['wkey'] = self._weight_key(m)

    # ------------------------------------------------------------------------------------------
    def __call__(self, frames):
        """frames: S contiguous fp32 [1,C,H,W] tensors (one per sequence).  Returns the S network outputs (the
        state tensors of the last layer, valid until the next call, as the modules return theirs)."""
        if len(frames) != self.S:
            raise CBinferError("SequenceBatch: %d frames for %d sequences" % (len(frames), self.S))
        f0 = frames[0]
        key = (tuple(f0.shape), f0.dtype, f0.device)
        for f in frames:
            _lib.require_device(f)
            if (tuple(f.shape), f.dtype, f.device) != key or not f.is_contiguous() or f.dtype != torch.float32:
                raise CBinferError("SequenceBatch: frames must be contiguous fp32 tensors of one shape on one device")
        if self.layers is None or self._key != key:
            self._build(f0)
            self._key = key
        st = stream_ptr(f0)
        S = self.S
        first = self._first
        for li, L in enumerate(self.layers):      # (before any launch: a folded tail's weights are used by its producer)
            if self._weight_key(L['m']) != L['wkey']:
                self._refresh_weights(li)
        for li, L in enumerate(self.layers):
            m = L['m']
            if L['kind'] == 'rows':
                a_in = _ptr_array(frames)
                check(C.cbinfer_change_detection_bits_batched(a_in, L['a_state'], L['a_bits'], S, L['W'], L['H'],
                                                              L['C'], (L['kH'] - 1) // 2, (L['kW'] - 1) // 2,
                                                              float(m.threshold), 1, st))
                if L.get('pairs'):
                    nptr = None
                    if L.get('next') is not None:
                        ni, nd = L['next']
                        Ln = self.layers[ni]
                        # (the same condition as the producer-mask shortcut of the consumer's own detection: not on
                        #  the first frame, not on the first frame after a change of its threshold)
                        if (not first) and Ln['th'] == float(Ln['m'].threshold):
                            nd.threshold = float(Ln['m'].threshold)
                            nptr = ctypes.pointer(nd)
                            Ln['detected'] = True
                    check(C.cbinfer_conv_changed_rowpairs_batched(L['pseq'], S, ptr(L['wp']), ptr(m.bias.detach()),
                                                                  L['C'], L['H'], L['W'], L['K'], L['kH'], L['kW'],
                                                                  int(bool(m.withReLU)), nptr, st))
                else:
                    check(C.cbinfer_conv_changed_rows_batched(L['a_state'], L['a_bits'], L['a_arrive'], L['a_copy'],
                                                              L['a_out'], S, ptr(L['wp']), ptr(m.bias.detach()), L['C'],
                                                              L['H'], L['W'], L['K'], L['kH'], L['kW'],
                                                              int(bool(m.withReLU)), st))
            elif L['kind'] == 'split':
                seqs = L['seqs']
                pooled = L['pooled']
                if pooled is None:
                    for q in range(S):
                        seqs[q].input = frames[q].data_ptr() if li == 0 else \
                            self.layers[li - 1]['prevOutput'][q].data_ptr()
                else:
                    # the producer-mask shortcut of the pooled detection assumes that the skipped segments compared
                    # below THIS threshold against THIS state last frame: not on the first frame, and not on the
                    # first frame after a change of the threshold
                    use = (not first) and L['th'] == float(m.threshold)
                    if use != (seqs[0].producerMask is not None):
                        prod = self.layers[pooled[0]]
                        for q in range(S):
                            seqs[q].producerMask = prod['copy'][q].data_ptr() if use else None
                L['th'] = float(m.threshold)
                args = [seqs, S, int(pooled is not None), pooled[1] if pooled else 0, pooled[2] if pooled else 0,
                        ptr(L['wp']), ptr(m.bias.detach()), L['C'], L['H'], L['W'], L['K'], L['kH'], L['kW'],
                        float(m.threshold), float(L['scale']), int(bool(m.withReLU)), ptr(L['ws'])]
                if L.pop('detected', False):
                    # (the producing layer's row-pair launch was this frame's detection: the contraction alone)
                    cargs = [seqs, S, ptr(L['wp']), ptr(m.bias.detach()), L['C'], L['H'], L['W'], L['K'], L['kH'],
                             L['kW'], float(L['scale']), int(bool(m.withReLU)), ptr(L['ws']), 0]
                    if L.get('tail') is not None:
                        check(C.cbinfer_split_conv_tail(*(cargs + [ctypes.pointer(L['tail']), st])))
                    else:
                        check(C.cbinfer_split_conv(*(cargs + [st])))
                elif L.get('tail') is not None:
                    check(C.cbinfer_split_forward_tail(*(args + [0, ctypes.pointer(L['tail']), st])))
                else:
                    check(C.cbinfer_split_forward(*(args + [st])))
            elif L.get('folded'):
                pass          # (evaluated by the producing layer's second launch)
            else:
                check(C.cbinfer_tail1x1_batched(L['seqs'], S, L['H'] * L['W'], ptr(m._prepared()),
                                                ptr(m.bias1.detach()), ptr(m.weight2.detach().contiguous()),
                                                ptr(m.bias2.detach()), m.in_channels, m.hidden_channels,
                                                m.out_channels, L['H'], L['W'], int(m.relu), int(bool(m.withReLU)),
                                                st))
        self._first = False
        return list(self.layers[-1]['prevOutput'])

    def rangeExceeded(self):
        return any(int(L['flag'].item()) != 0 for L in self.layers if L['kind'] == 'split')
