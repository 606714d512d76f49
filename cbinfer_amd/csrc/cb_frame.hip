// One-call-per-layer frame pipeline: the host side of CBConv2d.forward_normal (conv2d.py:178-259)
// expressed as a fixed sequence of launches on one stream, with the changed-pixel count kept on the
// device (the reference blocks on torch.nonzero at this point, conv2d_cg.py:202).  Safe under
// hipGraph capture: no allocation, no synchronisation, no memset node.
#include "cb_common.h"

static int cb_forward(const int32_t* upstreamCount, const void* input, void* prevInput, void* prevOutput,
                      uint64_t* bits, int32_t* idx, int32_t* countDev, int8_t* mapOut, const void* weightsPrepared,
                      const void* bias, int C, int H, int W, int K, int kH, int kW, float threshold,
                      int feedbackLoop, int copyInput, int relu, int haveIndexes, int capN, void* workspace,
                      int selfCompact, int dtype, cbStream_t stream) {
    CB_REQUIRE(input && prevInput && prevOutput && idx && countDev && weightsPrepared);
    CB_REQUIRE(dtype == CB_F32 || dtype == CB_F16 || dtype == CB_F32S);
    CB_REQUIRE(capN >= 0 && capN <= H * W);
    const int edt = dtype == CB_F32S ? CB_F32 : dtype;   // element type (CB_F32S only selects the arithmetic)
    int st;
    if (selfCompact) {
        // detection -> [state copy] -> self-compacting fused kernel: 2 launches per layer and frame.
        // `bits` is a cbinfer_frame_mask_bytes(H,W) buffer (two alternating masks + parity).
        CB_REQUIRE(!haveIndexes && bits && !mapOut);
        // (not in feedback mode, own copy of the input kept -- conv2d.py:234-236: the detection writes every value
        //  it reads into the state itself, updateInputState = 2; rounds 1-3 issued a full-tensor copy behind it: 36
        //  launches and 166 us of a 890 us OpenPose frame)
        const bool copyAll = !feedbackLoop && copyInput && prevInput != input;
        st = cbinfer_change_detection_frame_after(upstreamCount, input, prevInput, bits, W, H, C, (kH - 1) / 2,
                                                  (kW - 1) / 2, threshold, feedbackLoop ? 1 : (copyAll ? 2 : 0), edt,
                                                  stream);
        if (st != CB_OK) return st;
        const void* src = (feedbackLoop || copyInput) ? prevInput : input;
        return cbinfer_conv_changed_from_mask_after(upstreamCount, src, bits, idx, countDev, weightsPrepared, bias,
                                                    prevOutput, C, H, W, K, kH, kW, relu, workspace, dtype, stream);
    }
    CB_REQUIRE(upstreamCount == nullptr);      // (only the two-launch frame knows how to skip itself)
    if (!haveIndexes) {
        CB_REQUIRE(bits != nullptr);
        const bool copyAll = !feedbackLoop && copyInput && prevInput != input;      // (as above: updateInputState = 2)
        st = cbinfer_change_detection_bits(input, prevInput, bits, W, H, C, (kH - 1) / 2, (kW - 1) / 2,
                                           threshold, feedbackLoop ? 1 : (copyAll ? 2 : 0), edt, stream);
        if (st != CB_OK) return st;
        st = cbinfer_compact_bits(bits, W, H, idx, countDev, nullptr, mapOut, stream);
        if (st != CB_OK) return st;
    }
    if (haveIndexes && !feedbackLoop && copyInput && prevInput != input) {      // (propagated indexes: no detection ran)
        const size_t bytes = (size_t)C * H * W * (dtype == CB_F16 ? 2 : 4);
        hipError_t e = hipMemcpyAsync(prevInput, input, bytes, hipMemcpyDeviceToDevice,
                                      (hipStream_t)stream);
        if (e != hipSuccess) return (int)e;
    }
    // The gather reads the layer STATE (conv2d.py:242).  With copyInput=0 (and no feedback loop) the
    // reference re-points the state at the caller's tensor (:237-238), i.e. the gather reads `input`.
    const void* src = (feedbackLoop || copyInput) ? prevInput : input;
    return cbinfer_conv_changed(src, idx, capN, countDev, weightsPrepared, bias, prevOutput, C, H, W, K,
                                kH, kW, relu, 0, bits, bits ? cbinfer_mask_words(H, W) : 0, workspace,
                                dtype, stream);
}

extern "C" int cbinfer_cbconv2d_forward(const void* input, void* prevInput, void* prevOutput,
                                        uint64_t* bits, int32_t* idx, int32_t* countDev,
                                        int8_t* mapOut, const void* weightsPrepared, const void* bias,
                                        int C, int H, int W, int K, int kH, int kW, float threshold,
                                        int feedbackLoop, int copyInput, int relu, int haveIndexes,
                                        int capN, void* workspace, int selfCompact, int dtype,
                                        cbStream_t stream) {
    return cb_forward(nullptr, input, prevInput, prevOutput, bits, idx, countDev, mapOut, weightsPrepared, bias, C, H,
                      W, K, kH, kW, threshold, feedbackLoop, copyInput, relu, haveIndexes, capN, workspace,
                      selfCompact, dtype, stream);
}

// The self-compacting frame of a layer whose input is the OUTPUT BUFFER of another change-based layer, untouched in
// between (conv -> conv chains; the reference has no counterpart -- every one of its layers runs its detection over
// the whole input, conv2d.py:228-233).  upstreamCount: the device word that layer's contraction left its change count
// in this frame (its countDev).  Zero there means not one output pixel was rewritten: the input is bit for bit what
// this layer compared with its state last frame -- and (not in feedback mode) copied into it, or (feedback mode)
// left within the threshold of it -- so the detection would find nothing and the contraction would have nothing to
// do.  Both launches then return after one scalar load, countDev receives 0 (the next layer of the chain is skipped
// the same way) and masks, parity, state and output stay as they are.  The CALLER guarantees the premise: same input
// buffer and same threshold as in this layer's previous frame, which saw the producer's previous frame; no writer to
// the buffer other than that producer.  selfCompact form only (upstreamCount = NULL: cbinfer_cbconv2d_forward).
extern "C" int cbinfer_cbconv2d_forward_after(const int32_t* upstreamCount, const void* input, void* prevInput,
                                              void* prevOutput, uint64_t* bits, int32_t* idx, int32_t* countDev,
                                              const void* weightsPrepared, const void* bias, int C, int H, int W,
                                              int K, int kH, int kW, float threshold, int feedbackLoop,
                                              int copyInput, int relu, void* workspace, int dtype,
                                              cbStream_t stream) {
    return cb_forward(upstreamCount, input, prevInput, prevOutput, bits, idx, countDev, nullptr, weightsPrepared, bias,
                      C, H, W, K, kH, kW, threshold, feedbackLoop, copyInput, relu, 0, H * W, workspace, 1, dtype,
                      stream);
}

// The same for a feedback-mode layer that sits behind a 2x2/stride-2 max pool, with the pool folded into
// the detection (cbinfer_change_detection_frame_pooled): `prePool` [C, pH, pW] is the pool's INPUT, the
// pooled map is never materialised -- the layer gathers from its state, which holds the pooled values
// of every pixel that ever changed.  Saves the pool launch; results are identical to pool + forward.
extern "C" int cbinfer_cbconv2d_forward_pooled(const void* prePool, int pH, int pW, void* prevInput,
                                               void* prevOutput, uint64_t* bits, int32_t* idx,
                                               int32_t* countDev, const void* weightsPrepared,
                                               const void* bias, int C, int H, int W, int K, int kH, int kW,
                                               float threshold, int relu, void* workspace, int dtype,
                                               cbStream_t stream) {
    CB_REQUIRE(prePool && prevInput && prevOutput && bits && idx && countDev && weightsPrepared);
    CB_REQUIRE(dtype == CB_F32 || dtype == CB_F16 || dtype == CB_F32S);
    const int st = cbinfer_change_detection_frame_pooled(prePool, pH, pW, prevInput, bits, W, H, C,
                                                         (kH - 1) / 2, (kW - 1) / 2, threshold,
                                                         dtype == CB_F32S ? CB_F32 : dtype, stream);
    if (st != CB_OK) return st;
    return cbinfer_conv_changed_from_mask(prevInput, bits, idx, countDev, weightsPrepared, bias, prevOutput,
                                          C, H, W, K, kH, kW, relu, workspace, dtype, stream);
}

// CBConv2d.forward_fg (conv2d.py:160-176 -> conv2d_fg.py:75-85) for frames after the first, as two
// launches without a host round trip, atomics or per-value coordinate list (the reference blocks on
// torch.nonzero, conv2d_fg.py:82, and scatters K*kH*kW atomicAdds per changed value, .cu:37-66):
//   1. per-value detection: delta = in - prev where |in - prev| > th, else 0 (dense [C,H,W] workspace);
//      the pixels with a changed value, dilated by the filter support, go to the frame mask;
//      refreshState=1 also performs the reference's `prevInput = input` (conv2d.py:175) in place,
//      otherwise the caller re-points its state at `input` as the reference does;
//   2. the self-compacting contraction ADDS conv(weights, delta) at exactly the masked output pixels --
//      every other output receives only zero contributions in the reference too -- and keeps
//      reluOut = relu(prevOutput) current there when given (conv2d.py:172-174 without a full-tensor pass).
// Deterministic (fixed summation order); idx/countDev receive the list of touched output pixels.
extern "C" int cbinfer_cbconv2d_forward_fg(const float* input, float* prevInput, float* delta,
                                           float* prevOutput, float* reluOut, uint64_t* frameMasks,
                                           int32_t* idx, int32_t* countDev, const void* weightsPrepared,
                                           int C, int H, int W, int K, int kH, int kW, float threshold,
                                           int refreshState, void* workspace, int dtype, cbStream_t stream) {
    CB_REQUIRE(input && prevInput && delta && prevOutput && frameMasks && idx && countDev && weightsPrepared);
    const int st = cbinfer_change_detection_fg_frame(input, prevInput, delta, frameMasks, W, H, C,
                                                     (kH - 1) / 2, (kW - 1) / 2, threshold, refreshState,
                                                     stream);
    if (st != CB_OK) return st;
    return cbinfer_conv_accumulate_from_mask(delta, frameMasks, idx, countDev, weightsPrepared, prevOutput,
                                             reluOut, C, H, W, K, kH, kW, workspace, dtype, stream);
}

// The fine-grained frame on a mask-driven contraction (row-segment kernel, blocks = 0; patch-staged kernel,
// blocks = 1; fp32): per-value detection into the delta tensor and the SINGLE mask, then out += W * delta at
// the mask's pixels (cbinfer_conv_accumulate_rows / _blocks).  maskCopy keeps the touched-pixel mask.
extern "C" int cbinfer_cbconv2d_forward_fg_masked(int blocks, const float* input, float* prevInput, float* delta,
                                                  float* prevOutput, float* reluOut, uint64_t* bits,
                                                  int32_t* arrive, uint64_t* maskCopy, const void* weightsPrepared,
                                                  int C, int H, int W, int K, int kH, int kW, float threshold,
                                                  int refreshState, cbStream_t stream) {
    CB_REQUIRE(input && prevInput && delta && prevOutput && bits && arrive && maskCopy && weightsPrepared);
    const int st = cbinfer_change_detection_fg_bits(input, prevInput, delta, bits, W, H, C, (kH - 1) / 2,
                                                    (kW - 1) / 2, threshold, refreshState, stream);
    if (st != CB_OK) return st;
    return blocks ? cbinfer_conv_accumulate_blocks(delta, bits, arrive, maskCopy, weightsPrepared, prevOutput,
                                                   reluOut, C, H, W, K, kH, kW, stream)
                  : cbinfer_conv_accumulate_rows(delta, bits, arrive, maskCopy, weightsPrepared, prevOutput,
                                                 reluOut, C, H, W, K, kH, kW, stream);
}

// The frame of a layer whose contraction runs on the row-segment kernel (cbinfer_conv_changed_rows, fp32,
// cbinfer_rowconv_supported): detection -> [state copy] -> row-segment contraction, two launches, single
// mask.  prePool != NULL: the layer sits behind a 2x2/stride-2 max pool folded into its detection
// (cbinfer_cbconv2d_forward_pooled's contract; feedback mode); producerMask (optional): the maskCopy the layer
// that PRODUCED prePool left this frame (pre-pool resolution) -- the detection then skips every 64-pixel
// segment none of whose window pixels that layer rewrote (they compare exactly as they did last frame).  maskCopy keeps the frame's mask: the
// change list is made from it on demand (cbinfer_compact_bits), no launch of this frame depends on it.
static int cb_forward_masked(int blocks, const float* input, const float* prePool, int pH, int pW,
                             const uint64_t* producerMask, float* prevInput, float* prevOutput, uint64_t* bits, int32_t* arrive,
                             uint64_t* maskCopy, const void* weights, const float* bias, int C, int H, int W,
                             int K, int kH, int kW, float threshold, int feedbackLoop, int copyInput, int relu,
                             cbStream_t stream) {
    CB_REQUIRE((input || prePool) && prevInput && prevOutput && bits && arrive && maskCopy && weights);
    auto contract = [&](const float* src) {
        return blocks ? cbinfer_conv_changed_blocks(src, bits, arrive, maskCopy, weights, bias, prevOutput, C, H,
                                                    W, K, kH, kW, relu, stream)
                      : cbinfer_conv_changed_rows(src, bits, arrive, maskCopy, weights, bias, prevOutput, C, H, W,
                                                  K, kH, kW, relu, stream);
    };
    int st;
    if (prePool) {
        st = cbinfer_change_detection_bits_pooled(prePool, pH, pW, producerMask, prevInput, bits, W, H, C,
                                                  (kH - 1) / 2, (kW - 1) / 2, threshold, CB_F32, stream);
        if (st != CB_OK) return st;
        return contract(prevInput);
    }
    const bool copyAll = !feedbackLoop && copyInput && prevInput != input;      // (updateInputState = 2: cb_detect.hip)
    st = cbinfer_change_detection_bits(input, prevInput, bits, W, H, C, (kH - 1) / 2, (kW - 1) / 2, threshold,
                                       feedbackLoop ? 1 : (copyAll ? 2 : 0), CB_F32, stream);
    if (st != CB_OK) return st;
    return contract((feedbackLoop || copyInput) ? prevInput : input);
}

extern "C" int cbinfer_cbconv2d_forward_rows(const float* input, const float* prePool, int pH, int pW,
                                             const uint64_t* producerMask,
                                             float* prevInput, float* prevOutput, uint64_t* bits,
                                             int32_t* arrive, uint64_t* maskCopy, const void* rowWeights,
                                             const float* bias, int C, int H, int W, int K, int kH, int kW,
                                             float threshold, int feedbackLoop, int copyInput, int relu,
                                             cbStream_t stream) {
    return cb_forward_masked(0, input, prePool, pH, pW, producerMask, prevInput, prevOutput, bits, arrive, maskCopy, rowWeights,
                             bias, C, H, W, K, kH, kW, threshold, feedbackLoop, copyInput, relu, stream);
}

// The same frame with the patch-staged contraction (cbinfer_conv_changed_blocks; weights from
// cbinfer_blockconv_prep_weights; bf16x3 arithmetic).
extern "C" int cbinfer_cbconv2d_forward_blocks(const float* input, const float* prePool, int pH, int pW,
                                             const uint64_t* producerMask,
                                               float* prevInput, float* prevOutput, uint64_t* bits,
                                               int32_t* arrive, uint64_t* maskCopy, const void* blockWeights,
                                               const float* bias, int C, int H, int W, int K, int kH, int kW,
                                               float threshold, int feedbackLoop, int copyInput, int relu,
                                               cbStream_t stream) {
    return cb_forward_masked(1, input, prePool, pH, pW, producerMask, prevInput, prevOutput, bits, arrive, maskCopy,
                             blockWeights, bias, C, H, W, K, kH, kW, threshold, feedbackLoop, copyInput, relu,
                             stream);
}
