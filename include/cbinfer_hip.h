/*
 * cbinfer_hip.h -- C ABI of libcbinfer_hip.so: hand-written HIP (gfx950 / MI355X) kernels for
 * CBinfer's change-based convolution hot path.
 *
 * This is the drop-in boundary.  Every entry point replaces one native launcher (or one torch op) of
 * the reference; the reference interface it replaces is cited as file:line relative to
 * /root/reference/pycbinfer.  Conventions (the reference has none of these):
 *   - plain pointers and sizes only; all pointers are DEVICE pointers unless stated otherwise;
 *   - the library allocates nothing and keeps no global state: every buffer is caller-owned;
 *   - every launcher takes the HIP stream to enqueue on (cbStream_t = hipStream_t, NULL = default
 *     stream) and never synchronises; it is safe to call while the stream is being graph-captured;
 *   - return value: 0 on success, a positive hipError_t if the launch failed, a negative CB_ERR_* for
 *     rejected arguments.  cbinfer_status_string() turns either into text;
 *   - launch geometry is chosen inside the library (the reference computes it in Python and passes
 *     six ints, conv2d_cg.py:106-111);
 *   - dtype: CB_F32 (cbconv2d_cg_backend.cu) or CB_F16 (cbconv2d_cg_half_backend.cu);
 *   - tensors are NCHW, batch 1, contiguous.  Index lists are int32 flat pixel indices y*W+x in
 *     ascending order.  "count" pointers are device int32 scalars holding the list length N, so that
 *     a whole frame can be enqueued without a host round trip; where a launcher takes both a host
 *     `numChanges` and a device `countDev`, countDev (if non-NULL) wins and numChanges is only the
 *     capacity the grid is sized for.
 *
 * The same library also exports the reference's own symbol names and signatures
 * (cbconv2d_{cg,cg_half,fg}_backend compat shims, cbinfer_amd/csrc/cb_compat.cpp) so the reference's cffi
 * cdef/dlopen (conv2d_cg.py:6-50, conv2d_fg.py:13-32) binds without change.
 */
#ifndef CBINFER_HIP_H
#define CBINFER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* cbStream_t; /* hipStream_t */

#define CB_F32 0
#define CB_F16 1
/* f32 tensors whose contraction runs as bf16x3 split products on the bf16 MFMA (f32 accumulation): every
 * operand x = hi + mid + lo (three bf16 terms, 24 significant bits), the six cross products above 2^-24 of
 * the product are summed -- f32-level accuracy at 16/6 of the f32 MFMA's rate.  Accepted wherever a `dtype`
 * selects the arithmetic of a gather-mode contraction (cbinfer_prep_weights, cbinfer_prepared_weights_bytes,
 * cbinfer_conv_changed[_from_mask], cbinfer_conv_accumulate_from_mask, cbinfer_cbconv2d_forward[_pooled,
 * _fg]); the tensors stay f32 and CB_F32 keeps the exact f32 fma chain. */
#define CB_F32S 2

#define CB_OK 0
#define CB_ERR_BADARG (-1)      /* null pointer, non-positive size, unsupported dtype */
#define CB_ERR_UNSUPPORTED (-2) /* shape outside what the kernels implement (e.g. kWHalf > 63) */

/* 1: round 1.  2: CB_F32S, row-segment / patch-staged contractions, fine-grained frame, fused 1x1 tail,
 * cbinfer_weights_ckkpad(Ckk, dtype).  3: fine-grained frame on the mask-driven contractions
 * (cbinfer_cbconv2d_forward_fg_masked and its parts).  4: split-state frame (cbinfer_split_*, several sequences
 * per launch), cbinfer_tail1x1_supported.  5: cbinfer_split_forward_tail (the 1x1 tail in the contraction's second
 * launch), cbinfer_split_tail_supported.  6: row-pair frame (cbinfer_*rowpairs*), cbinfer_dilate_change_indexes,
 * updateInputState = 2.  7: chained layers (cbinfer_*_after: the producer's change count ends an idle frame).
 * 8: the f32-EQUIVALENT (bf16-triple) form of the split-state frame: cbinfer_split3_*, CBINFER_SPLIT_X3,
 * weightScale == 0, cbNextDetect.arith.  9: cbinfer_hsplit_forward_group (cbHalfLayer / cbHalfNext: two layers of one
 * geometry per launch, the consumers' change detection in the producing launch); cbinfer_hsplit_* take contractions
 * of a single k-stage and up (1x1 layers on >= 64 channels).  10: cbinfer_split_*_next (a split-state layer's contraction
 * in window order carrying the pooled change detection of the layer behind the 2x2 pool). */
#define CBINFER_ABI_VERSION 11

int cbinfer_abi_version(void);
const char* cbinfer_status_string(int status);

/* ---- geometry helpers (host, pure) --------------------------------------------------------- */
/* Row-padded bit mask layout used by the sync-free path: one uint64 word per 64 pixels of a row,
 * rows padded to whole words.  Returns words per row / total words for an H x W map. */
int cbinfer_mask_words_per_row(int W);
long cbinfer_mask_words(int H, int W);
/* Padded sizes of the prepared weight matrix (see cbinfer_prep_weights); the k-depth padding depends on
 * the dtype (fp32: whole 32-deep stages, fp16: stage pairs of 2 x 64).  Size buffers with
 * cbinfer_prepared_weights_bytes. */
int cbinfer_weights_kpad(int K);
int cbinfer_weights_ckkpad(int Ckk, int dtype);

/* ---- a1: change detection + dilation (+ feedback state update) ------------------------------
 * replaces changeDetection, conv2d_cg.py:100-122 -> cbconv2d_cg_backend.cu:83-100 (kernels :6-81),
 * half: cbconv2d_cg_half_backend.cu:10-88.
 * change(p) = OR_c |state[c,p] - in[c,p]| > th (strict; half: compared in half precision after one
 * rounding of the difference).  A changed pixel marks its (2kHHalf+1)x(2kWHalf+1) neighbourhood in
 * changeMap [H,W] int8 and, if updateInputState == 1, gets in[:,p] copied into state[:,p] (the feedback refresh).
 * updateInputState == 2 (round 4): EVERY pixel's in[:,p] is copied into state[:,p] in the same pass -- the
 * reference's prevInput.copy_(input) of a layer that is not in feedback mode (conv2d.py:234-236) without the
 * full-tensor copy launch behind the detection.
 * The map is zeroed by the library (the reference's caller does it, conv2d_cg.py:105). */
int cbinfer_change_detection(const void* input, void* state, int8_t* changeMap, int W, int H, int C,
                             int kHHalf, int kWHalf, float threshold, int updateInputState,
                             int dtype, cbStream_t stream);

/* Same computation, emitting the dilated mask as a row-padded BIT mask (wave ballot -> one word per
 * 64 pixels; bitsOut must be zero on entry, cbinfer_compact_bits re-zeroes a buffer for the next
 * frame).  This is the form the sync-free frame pipeline uses. */
int cbinfer_change_detection_bits(const void* input, void* state, uint64_t* bitsOut, int W, int H,
                                  int C, int kHHalf, int kWHalf, float threshold,
                                  int updateInputState, int dtype, cbStream_t stream);

/* ---- a2: stand-alone mask dilation ------------------------------------------------------------
 * replaces changePropagation, conv2d_cg.py:159-177 -> cbconv2d_cg_backend.cu:126-136 (kernel :101). */
int cbinfer_change_propagation(const int8_t* mapIn, int8_t* mapOut, int W, int H, int kHHalf,
                               int kWHalf, cbStream_t stream);

/* ---- a3: changed-index extraction (stream compaction) ----------------------------------------
 * replaces changeIndexesExtr[_python], conv2d_cg.py:200-213 (torch.nonzero(map.view(-1)).int()).
 * idxOut (capacity numel) receives the ascending flat indices of the non-zero bytes, countDev the
 * number of them.  scratchWords: caller-owned uint64 scratch of ceil(numel/64) words. */
int cbinfer_change_indexes_extr(const int8_t* changeMap, long numel, uint64_t* scratchWords,
                                int32_t* idxOut, int32_t* countDev, cbStream_t stream);

/* Compaction of a row-padded bit mask (from cbinfer_change_detection_bits) into flat indices y*W+x.
 * If clearBits is non-NULL that (other) mask buffer of the same geometry is zeroed for the next
 * frame.  If mapOut is non-NULL the mask is also expanded to an int8 [H,W] map (saveChangeMap). */
int cbinfer_compact_bits(const uint64_t* bits, int W, int H, int32_t* idxOut, int32_t* countDev,
                         uint64_t* clearBits, int8_t* mapOut, cbStream_t stream);

/* ---- a5: gather -> im2col rows of the changed pixels ------------------------------------------
 * replaces genXMatrix, conv2d_cg.py:239-261 -> cbconv2d_cg_backend.cu:163-173 (kernel :138-161).
 * columns [N, C*kH*kW] row-major, column (c*kH+ky)*kW+kx; zero outside the image. */
int cbinfer_gen_x_matrix(void* columns, const void* input, const int32_t* changeList, int kW, int kH,
                         int C, int W, int H, int numChanges, const int32_t* countDev, int dtype,
                         cbStream_t stream);

/* ---- a6/a7: the dense contraction ------------------------------------------------------------
 * replaces matrixMult_python, conv2d_cg.py:342-349 (torch matmul -> cuBLAS) and, with
 * transposeOut=1, also the transpose+contiguous of conv2d.py:247 / conv2d_cg.py:305.
 * Y = X[N,Ckk] . W[K,Ckk]^T + bias; Y is [N,K] (transposeOut=0) or [K,N] (transposeOut=1).
 * fp32: exact-f32 MFMA (v_mfma_f32_32x32x2_f32); fp16: f16 MFMA with f32 accumulation.
 * weightsPrepared is the buffer produced by cbinfer_prep_weights from the [K,C,kH,kW] filter bank:
 * the matrix padded to the MFMA tile grid, W[Kpad][CkkPad] with k contiguous, followed by a k->(c,ky,kx)
 * tap table (byte offset, dy, dx per k, which depends on the H x W of the feature map the layer runs
 * on); cbinfer_prepared_weights_bytes gives its size.  For a plain [K,Ckk] matrix pass C=Ckk,
 * kH=kW=H=W=1. */
long cbinfer_prepared_weights_bytes(int K, int C, int kH, int kW, int dtype);
int cbinfer_prep_weights(const void* weight, void* weightsPrepared, int K, int C, int kH, int kW,
                         int H, int W, int dtype, cbStream_t stream);
int cbinfer_matrix_mult(const void* X, const void* weightsPrepared, const void* bias, void* Y,
                        int N, const int32_t* countDev, int Ckk, int K, int transposeOut, int dtype,
                        cbStream_t stream);

/* ---- a8: scatter-back -------------------------------------------------------------------------
 * replaces updateOutput, conv2d_cg.py:292-313 -> cbconv2d_cg_backend.cu:191-197 (kernel :175-189).
 * output[k*HW + changeList[n]] = relu ? (v <= 0 ? 0 : v) : v with v = Yt[k*N + n]. */
int cbinfer_update_output(const void* Yt, void* output, const int32_t* changeList,
                          int numOutputPixel, int numChanges, const int32_t* countDev,
                          int nOutputPlane, int relu, int dtype, cbStream_t stream);

/* ---- a5+a6+a7+a8 fused: gather -> MFMA -> bias/ReLU -> scatter, no X / Y in HBM --------------
 * One launch replaces conv2d.py:240-251 (genXMatrix, matrixMult_python, transpose, updateOutput).
 * input is the layer state the gather reads from (conv2d.py:242 reads self.prevInput).
 * accumulate=1 adds to output instead of overwriting (no bias/ReLU): used by the deterministic
 * fine-grained variant (a12) where `input` holds the masked deltas.
 * clearBits (optional, clearWords words): a change bit mask this launch zeroes on the way, so the
 * next frame's cbinfer_change_detection_bits finds it clean (keeps the frame free of memset nodes).
 * workspace (optional, cbinfer_conv_workspace_bytes() bytes, ZERO on first use and left zero): lets the
 * persistent kernel split a short change list along k across workgroups (deterministic slab
 * reduction).  One workspace must not be used by two launches that can run concurrently. */
long cbinfer_conv_workspace_bytes(void);
int cbinfer_conv_changed(const void* input, const int32_t* changeList, int numChanges,
                         const int32_t* countDev, const void* weightsPrepared, const void* bias,
                         void* output, int C, int H, int W, int K, int kH, int kW, int relu,
                         int accumulate, uint64_t* clearBits, long clearWords, void* workspace,
                         int dtype, cbStream_t stream);

/* ---- a14 in one call: CBConv2d.forward_normal (conv2d.py:178-259) enqueued without a host sync --
 * detection(+dilation,+feedback) -> compaction -> [state copy] -> fused gather/MFMA/scatter.
 *   bits      : row-padded change mask (cbinfer_mask_words(H,W) words), zero on first use; left zero
 *               (may be NULL when haveIndexes=1)
 *   idx/count : capacity H*W int32 / one int32; on return (stream order) the frame's change list
 *   mapOut    : optional int8 [H,W] copy of the dilated mask (saveChangeMap)
 *   haveIndexes=1: idx/count were produced upstream (propChangeIndexes protocol): skip detection
 *   feedbackLoop=1: prevInput refreshed at changed pixels only; else, if copyInput, prevInput <- input
 *                   (copyInput=0: the gather reads `input` and the caller re-points its state at it,
 *                   conv2d.py:237-238)
 *   capN      : capacity of idx the fused kernel may assume (H*W, or the exact N if the caller
 *               synchronised);  workspace: see cbinfer_conv_changed */
int cbinfer_cbconv2d_forward(const void* input, void* prevInput, void* prevOutput, uint64_t* bits,
                             int32_t* idx, int32_t* countDev, int8_t* mapOut,
                             const void* weightsPrepared, const void* bias, int C, int H, int W,
                             int K, int kH, int kW, float threshold, int feedbackLoop,
                             int copyInput, int relu, int haveIndexes, int capN, void* workspace,
                             int selfCompact, int dtype, cbStream_t stream);

/* selfCompact=1 (fp32 or fp16, no mapOut, no upstream indexes, cbinfer_mask_words(H,W) <=
 * cbinfer_frame_mask_max_words()): `bits` is a zero-initialised buffer of cbinfer_frame_mask_bytes(H,W)
 * bytes holding two alternating masks and a device-side parity; the compaction launch disappears, the
 * fused kernel derives the change list from the mask by itself and still writes idx/countDev.
 * The two launches are also available on their own: */
long cbinfer_frame_mask_bytes(int H, int W);
/* byte offset, inside a frame mask buffer, of the COPY of the current frame's change mask (cbinfer_mask_words(H,W) words)
 * that a self-compacting contraction (selfCompact = 1) leaves behind: the producerMask a chained consumer's detection takes
 * (cbinfer_hsplit_forward[_group]); round 6 */
long cbinfer_frame_mask_copy_offset(int H, int W);
int cbinfer_frame_mask_max_words(void);
int cbinfer_change_detection_frame(const void* input, void* state, uint64_t* frameMasks, int W, int H,
                                   int C, int kHHalf, int kWHalf, float threshold,
                                   int updateInputState, int dtype, cbStream_t stream);
/* A feedback-mode layer behind a 2x2/stride-2 max pool (CBPoolMax2d, conv2d.py:24-84) with the pool
 * folded into its change detection: prePool [C,pH,pW] is the pool's INPUT; H x W is the pooled size
 * (pH/2 or ceil).  The pooled value is computed on the fly, compared with the state and written to the
 * state at the changed pixels; the pooled map itself is never materialised (the layer gathers from its
 * state).  Identical results to cbinfer_max_pool2d + cbinfer_cbconv2d_forward(feedbackLoop=1), one
 * launch less per frame. */
int cbinfer_change_detection_frame_pooled(const void* prePool, int pH, int pW, void* state,
                                          uint64_t* frameMasks, int W, int H, int C, int kHHalf,
                                          int kWHalf, float threshold, int dtype, cbStream_t stream);
int cbinfer_cbconv2d_forward_pooled(const void* prePool, int pH, int pW, void* prevInput,
                                    void* prevOutput, uint64_t* bits, int32_t* idx, int32_t* countDev,
                                    const void* weightsPrepared, const void* bias, int C, int H, int W,
                                    int K, int kH, int kW, float threshold, int relu, void* workspace,
                                    int dtype, cbStream_t stream);
int cbinfer_conv_changed_from_mask(const void* input, uint64_t* frameMasks, int32_t* idxOut,
                                   int32_t* countOut, const void* weightsPrepared, const void* bias,
                                   void* output, int C, int H, int W, int K, int kH, int kW, int relu,
                                   void* workspace, int dtype, cbStream_t stream);
/* Chains of change-based layers (conv -> conv, the producer's output buffer handed on untouched; no counterpart in
 * the reference, whose every layer scans its whole input, conv2d.py:228-233): upstreamCount is the device word the
 * PRODUCING layer's contraction left its change count in this frame (its countDev).  Zero there: the producer
 * rewrote no output pixel, this layer's input is bit for bit what it compared with its state last frame, the
 * detection could find nothing -- both launches return after one scalar load, countDev receives 0 (so that the
 * next layer of the chain is skipped the same way), masks, parity, state and output stay as they are.  Non-zero
 * (or upstreamCount = NULL): cbinfer_cbconv2d_forward(selfCompact=1) exactly.  The CALLER guarantees the premise:
 * same input buffer and threshold as in this layer's previous frame, which followed the producer's previous frame;
 * no other writer to that buffer.  (cbinfer_amd/conv2d.py: CBConv2d._upstream_count checks it per frame.) */
int cbinfer_change_detection_frame_after(const int32_t* upstreamCount, const void* input, void* state,
                                         uint64_t* frameMasks, int W, int H, int C, int kHHalf, int kWHalf,
                                         float threshold, int updateInputState, int dtype, cbStream_t stream);
int cbinfer_conv_changed_from_mask_after(const int32_t* upstreamCount, const void* input, uint64_t* frameMasks,
                                         int32_t* idxOut, int32_t* countOut, const void* weightsPrepared,
                                         const void* bias, void* output, int C, int H, int W, int K, int kH, int kW,
                                         int relu, void* workspace, int dtype, cbStream_t stream);
int cbinfer_cbconv2d_forward_after(const int32_t* upstreamCount, const void* input, void* prevInput,
                                   void* prevOutput, uint64_t* bits, int32_t* idx, int32_t* countDev,
                                   const void* weightsPrepared, const void* bias, int C, int H, int W, int K, int kH,
                                   int kW, float threshold, int feedbackLoop, int copyInput, int relu,
                                   void* workspace, int dtype, cbStream_t stream);

/* ---- a5..a8 fused, row-segment form (fp32, small filter banks) ----------------------------------
 * Same contract as cbinfer_conv_changed (gather -> MFMA -> bias/ReLU -> scatter, replaces conv2d.py:240-251)
 * for layers cbinfer_rowconv_supported() accepts -- the 3->16 and 16->64 7x7 layers of the scene-labeling
 * net -- driven directly by the change BIT MASK: one workgroup per 64-pixel row segment (mask word) stages
 * the kH x (64+kW-1) input rows of every channel once in LDS and derives all taps from there.
 *   bits     : single row-padded mask (cbinfer_mask_words(H,W) words) as cbinfer_change_detection_bits
 *              (or cbinfer_change_detection_bits_pooled) fills it; zero on first use, left zero;
 *   arrive   : cbinfer_mask_words(H,W) int32, zero on first use, left zero;
 *   maskCopy : receives the frame's mask (cbinfer_compact_bits makes the index list from it on demand);
 *   prepared : cbinfer_rowconv_prep_weights' re-layout of the [K,C,kH,kW] filter bank
 *              (cbinfer_rowconv_prepared_bytes bytes).
 * cbinfer_cbconv2d_forward_rows is CBConv2d.forward_normal (conv2d.py:178-259) on it: detection (pooled on
 * the fly when prePool != NULL, see cbinfer_cbconv2d_forward_pooled) -> [state copy] -> contraction. */
int cbinfer_rowconv_supported(int C, int K, int kH, int kW);
long cbinfer_rowconv_prepared_bytes(int C, int K, int kH, int kW);
int cbinfer_rowconv_prep_weights(const float* weight, void* prepared, int K, int C, int kH, int kW,
                                 cbStream_t stream);
int cbinfer_conv_changed_rows(const float* state, uint64_t* bits, int32_t* arrive, uint64_t* maskCopy,
                              const void* prepared, const float* bias, float* output, int C, int H, int W,
                              int K, int kH, int kW, int relu, cbStream_t stream);
/* producerMask (may be NULL): the change mask (cbinfer_mask_words(pH,pW) words, this frame) of the layer that
 * wrote prePool; pooled pixels none of whose window pixels it rewrote are not even read.  ASSUMPTION the
 * caller must uphold: those pixels were compared against the SAME state with the SAME threshold on the previous
 * call (then they compare exactly as they did, i.e. not above it).  After a change of the threshold, or against a
 * state that was (re)allocated or written from outside, pass NULL for one frame (CBConv2d._forward_pooled does). */
int cbinfer_change_detection_bits_pooled(const void* prePool, int pH, int pW, const uint64_t* producerMask,
                                         void* state, uint64_t* bitsOut, int W, int H, int C, int kHHalf,
                                         int kWHalf, float threshold, int dtype, cbStream_t stream);
int cbinfer_cbconv2d_forward_rows(const float* input, const float* prePool, int pH, int pW,
                                  const uint64_t* producerMask, float* prevInput,
                                  float* prevOutput, uint64_t* bits, int32_t* arrive, uint64_t* maskCopy,
                                  const void* rowWeights, const float* bias, int C, int H, int W, int K,
                                  int kH, int kW, float threshold, int feedbackLoop, int copyInput, int relu,
                                  cbStream_t stream);

/* ---- a5..a8 fused, patch-staged form (fp32 tensors, CB_F32S arithmetic, wide layers) -----------
 * Same contract, buffers and mask protocol as cbinfer_conv_changed_rows for the layers with many channels
 * (16->64, 64->256 7x7): one workgroup per unit of R rows x 64 columns and 64 output channels; the input rows
 * under the unit are staged once per 8-channel chunk in LDS, pre-split into three bf16 planes, every tap reads
 * its 16-byte B fragment from there, the pre-split weights stream from L2 in MFMA fragment order
 * (cbinfer_blockconv_prep_weights, cbinfer_blockconv_prepared_bytes bytes). */
int cbinfer_blockconv_supported(int C, int K, int kH, int kW);
long cbinfer_blockconv_prepared_bytes(int C, int K, int kH, int kW);
int cbinfer_blockconv_prep_weights(const float* weight, void* prepared, int K, int C, int kH, int kW,
                                   cbStream_t stream);
int cbinfer_conv_changed_blocks(const float* state, uint64_t* bits, int32_t* arrive, uint64_t* maskCopy,
                                const void* prepared, const float* bias, float* output, int C, int H, int W,
                                int K, int kH, int kW, int relu, cbStream_t stream);
int cbinfer_cbconv2d_forward_blocks(const float* input, const float* prePool, int pH, int pW,
                                    const uint64_t* producerMask, float* prevInput,
                                    float* prevOutput, uint64_t* bits, int32_t* arrive, uint64_t* maskCopy,
                                    const void* blockWeights, const float* bias, int C, int H, int W, int K,
                                    int kH, int kW, float threshold, int feedbackLoop, int copyInput, int relu,
                                    cbStream_t stream);

/* ---- a9: change-based 2x2/stride-2 max pooling -----------------------------------------------
 * replaces maxPool2d, conv2d_cg.py:58-82 -> cbconv2d_cg_backend.cu:229-240 (kernel :199-227).
 * changeIndexes are INPUT-resolution pixel indices.  Unlike the reference, windows with
 * yo >= oH or xo >= oW (odd size, floor mode) are skipped instead of written out of bounds. */
int cbinfer_max_pool2d(const void* input, void* output, const int32_t* changeIndexes, int numChanges,
                       const int32_t* countDev, int C, int iH, int iW, int oH, int oW, int dtype,
                       cbStream_t stream);

/* ---- 8f-4: change indexes through a pool ------------------------------------------------------
 * The reference's CBPoolMax2d passes the INPUT-resolution index list on when propChangeIndexes is set
 * (conv2d.py:80-83), which is not usable at the pooled resolution.  This marks, for every changed input
 * pixel y*iW+x, bit (y/2, x/2) of a row-padded bit mask of the oH x oW pooled map (layout of
 * cbinfer_mask_words; bitsOut must be zero on entry); cbinfer_compact_bits turns it into the ascending
 * duplicate-free list of changed OUTPUT pixels and its device-side count. */
int cbinfer_pool_change_indexes(const int32_t* changeIndexes, int numChanges, const int32_t* countDev,
                                int iW, int oH, int oW, uint64_t* bitsOut, cbStream_t stream);
/* A CBConv2d fed propagated change indexes skips its own change detection whatever its filter size
 * (conv2d.py:180-190, :220; __init__.py:68-77) and recomputes exactly the listed pixels -- right for 1x1, short of
 * the filter's reach for k > 1.  This marks the filter support of every listed pixel (rows y +- kHHalf, columns
 * x +- kWHalf, clipped) in a row-padded bit mask of the H x W map (bitsOut zero on entry; cbinfer_compact_bits makes
 * the list): the output pixels the listed INPUT pixels reach -- what changeDetection's dilation
 * (cbconv2d_cg_backend.cu:62-72) marks when every listed pixel changed.  CBConv2d.dilatePropagatedIndexes. */
int cbinfer_dilate_change_indexes(const int32_t* changeIndexes, int numChanges, const int32_t* countDev, int H, int W,
                                  int kHHalf, int kWHalf, uint64_t* bitsOut, cbStream_t stream);

/* ---- a10-a12: fine-grained path ---------------------------------------------------------------
 * replaces changeDetectionFG, conv2d_fg.py:34-46 -> cbconv2d_fg_backend.cu:25-35 (kernel :7-23):
 * d = in - prev; changeMap = |d| > th; diffs = d where changed.  zeroUnchanged=1 additionally writes
 * diffs = 0 elsewhere (the reference leaves it uninitialised). */
int cbinfer_change_detection_fg(const float* input, const float* prevInput, float* diffs,
                                int8_t* changeMap, long numVals, float threshold, int zeroUnchanged,
                                cbStream_t stream);
/* replaces updateOutputFG, conv2d_fg.py:48-72 -> cbconv2d_fg_backend.cu:68-79 (kernel :37-66):
 * atomicAdd of w[:,ci,ky,kx]*d into the K*kH*kW outputs each changed value touches.
 * changeCoords: int64 flat coordinates into [C,H,W] (torch.nonzero, conv2d_fg.py:82). */
int cbinfer_update_output_fg(const float* diffs, const float* weight, float* output,
                             const int64_t* changeCoords, int K, int C, int H, int W, int kH, int kW,
                             long numChanges, cbStream_t stream);
/* a11 + a12 without the host round trip: the same scatter driven by an int32 coordinate list with a
 * device-side length, as cbinfer_change_indexes_extr produces it from the per-value change map (replaces
 * torch.nonzero, conv2d_fg.py:82, and its int64 list).  capacity: number of entries the grid may assume. */
int cbinfer_update_output_fg_list(const float* diffs, const float* weight, float* output,
                                  const int32_t* changeCoords, long capacity, const int32_t* countDev,
                                  int K, int C, int H, int W, int kH, int kW, cbStream_t stream);

/* ---- a14 (fine-grained) in one call: CBConv2d.forward_fg (conv2d.py:160-176) for the frames after the
 * first, enqueued without a host sync, atomics or coordinate list.
 *   cbinfer_change_detection_fg_frame: delta[c,p] = |in - prev| > th ? in - prev : 0 for EVERY value
 *     (dense fp32 [C,H,W] workspace); pixels with a changed value, dilated by the filter support, are ORed
 *     into the frame mask (layout of cbinfer_change_detection_frame); refreshState=1 also stores
 *     prev <- in wherever they differ (the reference's `prevInput = input`, conv2d.py:175, in place);
 *   cbinfer_conv_accumulate_from_mask: output += conv(weights, delta) at the masked pixels (same fused
 *     gather/MFMA kernel as cbinfer_conv_changed_from_mask, accumulating epilogue, no bias), and
 *     reluOut = relu(output) there if reluOut is non-NULL (conv2d.py:172-174);
 *   cbinfer_cbconv2d_forward_fg: both.  Deterministic.  fp32 only, like the reference's FG path. */
int cbinfer_change_detection_fg_frame(const float* input, float* prevInput, float* delta,
                                      uint64_t* frameMasks, int W, int H, int C, int kHHalf, int kWHalf,
                                      float threshold, int refreshState, cbStream_t stream);
int cbinfer_conv_accumulate_from_mask(const float* delta, uint64_t* frameMasks, int32_t* idxOut,
                                      int32_t* countOut, const void* weightsPrepared, float* output,
                                      float* reluOut, int C, int H, int W, int K, int kH, int kW,
                                      void* workspace, int dtype, cbStream_t stream);
int cbinfer_cbconv2d_forward_fg(const float* input, float* prevInput, float* delta, float* prevOutput,
                                float* reluOut, uint64_t* frameMasks, int32_t* idx, int32_t* countDev,
                                const void* weightsPrepared, int C, int H, int W, int K, int kH, int kW,
                                float threshold, int refreshState, void* workspace, int dtype,
                                cbStream_t stream);

/* The same frame on the mask-driven contractions (single mask + arrival counters + mask copy, as
 * cbinfer_conv_changed_rows / _blocks; weights prepared by cbinfer_rowconv_prep_weights /
 * cbinfer_blockconv_prep_weights): cbinfer_change_detection_fg_bits writes the delta tensor and ORs the dilated
 * touched-pixel mask into `bits`; cbinfer_conv_accumulate_rows / _blocks add conv(weights, delta) to `output`
 * at the mask's pixels (reluOut as above); cbinfer_cbconv2d_forward_fg_masked (blocks = 0 / 1): both. */
int cbinfer_change_detection_fg_bits(const float* input, float* prevInput, float* delta, uint64_t* bits, int W,
                                     int H, int C, int kHHalf, int kWHalf, float threshold, int refreshState,
                                     cbStream_t stream);
int cbinfer_conv_accumulate_rows(const float* delta, uint64_t* bits, int32_t* arrive, uint64_t* maskCopy,
                                 const void* prepared, float* output, float* reluOut, int C, int H, int W,
                                 int K, int kH, int kW, cbStream_t stream);
int cbinfer_conv_accumulate_blocks(const float* delta, uint64_t* bits, int32_t* arrive, uint64_t* maskCopy,
                                   const void* prepared, float* output, float* reluOut, int C, int H, int W,
                                   int K, int kH, int kW, cbStream_t stream);
int cbinfer_cbconv2d_forward_fg_masked(int blocks, const float* input, float* prevInput, float* delta,
                                       float* prevOutput, float* reluOut, uint64_t* bits, int32_t* arrive,
                                       uint64_t* maskCopy, const void* weightsPrepared, int C, int H, int W,
                                       int K, int kH, int kW, float threshold, int refreshState,
                                       cbStream_t stream);

/* ---- change-based 1x1 tail: conv1x1 -> [ReLU] -> conv1x1 at the changed pixels of the producing layer,
 * one launch.  Replaces two CBConv2d fed by propagated change indexes (sceneLabeling/modelLoader.py:41-44,
 * experiment 1) or the dense baseline modules 8..10 the other experiments keep (:45-47): a 1x1 layer's
 * output changes only where its input did.  output [C2,H,W] keeps every other pixel.  fp32.
 *   w1Prepared: w1 [C1,C0] re-laid out by cbinfer_tail1x1_prep (cbinfer_tail1x1_prepared_bytes bytes);
 *   C1 <= cbinfer_tail1x1_max_hidden(); w2 is the plain [C2,C1] matrix. */
int cbinfer_tail1x1_max_hidden(void);
int cbinfer_tail1x1_supported(int C0, int C1, int C2);   /* hidden width and LDS budget of cbinfer_tail1x1 */
long cbinfer_tail1x1_prepared_bytes(int C1, int C0);
int cbinfer_tail1x1_prep(const float* w1, float* w1Prepared, int C1, int C0, cbStream_t stream);
int cbinfer_tail1x1(const float* input, const int32_t* changeList, int numChanges, const int32_t* countDev,
                    const float* w1Prepared, const float* b1, const float* w2, const float* b2,
                    float* output, int C0, int C1, int C2, int H, int W, int relu1, int relu2,
                    cbStream_t stream);

/* ---- a1 + a5..a8 fused, SPLIT-STATE form (round 3; fp32 tensors, feedback mode, 16 / 32 / 64 input channels).
 * Replaces, per frame and layer, the launcher sequence changeDetection (cbconv2d_cg_backend.cu:83-100, with
 * updateInputState) -> torch.nonzero (conv2d_cg.py:200-209) -> genXMatrix (:138-173) -> matmul (conv2d_cg.py:
 * 342-349) -> updateOutput (:175-197) of CBConv2d.forward_normal (conv2d.py:220-251) -- for up to
 * CBINFER_SPLIT_MAX_SEQUENCES independent sequences (own state each, shared weights) in ONE launch per step.
 * The layer state is kept twice: prevInput [C,H,W] f32 (the module's buffer, refreshed at the changed pixels
 * as .cu:74-80 does) and a pixel-major pre-split f16-pair copy with a zero border (cbinfer_split_state_bytes;
 * cbinfer_split_state_init once, then maintained by cbinfer_split_detect), from which the contraction gathers
 * by LDS-DMA.  Arithmetic: x * 2^-4 = hi + lo * 2^-11 with f16 hi, lo (weights likewise, scaled by the power of
 * two weightScale that brings max|w| into [2^13, 2^14)); products hi.hi + (hi.lo + lo.hi) 2^-11 on the f16 MFMA,
 * f32 accumulation: |error| <= 5 * 2^-24 |a||b| per product (two operand roundings of 2^-23 and the dropped lo.lo
 * term; measured beside the exact f32 chain in tests/test_gpu_split.py).  Range of the state values: |x| < 2^20; a
 * refreshed value beyond it sets *rangeFlag, and from that launch on the contraction computes the sequence's tiles from
 * `state` in plain f32 (slow, right) until the caller moves the layer to another arithmetic.
 *   frameMasks : cbinfer_frame_mask_bytes(H,W) bytes, zero on first use: ONE mask (cbinfer_mask_words words) the
 *                detection ORs into and the contraction zeroes again; the second slot holds the (sharded) arrival
 *                counters of the contraction's workgroups
 *   idxOut     : change list of the frame (H*W ints), countOut its length -- by-products, ascending order
 *   workspace  : cbinfer_split_workspace_bytes(nSeq, ...) bytes, zero on first use (0 bytes / NULL for layers of
 *                fewer than 48 k-stages).  A deep contraction is ALWAYS the left-to-right sum of four partial sums
 *                over fixed k-ranges, whether these are computed by four workgroups (few tiles: slabs + a reduce
 *                launch) or one after the other by the same one (many tiles): a sequence gets the same bits alone
 *                and inside a batch.
 * mode bit 0 (CBINFER_SPLIT_POOLED): `input` is the tensor in FRONT of a 2x2/stride-2 max pool [C,pH,pW] (CBPoolMax2d
 * folded into the detection, see cbinfer_cbconv2d_forward_pooled); producerMask as for
 * cbinfer_change_detection_bits_pooled.  mode bit 1 (CBINFER_SPLIT_COPY_ALL, round 4): the layer is NOT in feedback
 * mode and keeps a copy of its input (feedbackLoop=False, copyInput=True: what convert() makes, conv2d.py:234-236,
 * `prevInput.copy_(input)`): the detection writes EVERY value of the frame into prevInput and into the pre-split
 * copy, not only those of the changed pixels -- the gather then reads this frame's input everywhere, as the
 * reference's does.
 *
 * The f32-EQUIVALENT form (round 5, "x3"): the same frame with every f32 operand kept as THREE bf16 terms, exactly
 * (x = b0 + b1 + b2, 8 significant bits each; bf16 has f32's exponent range: no scale, no range flag), and a product
 * made of the six term products above 2^-24 of it on the bf16 MFMA -- b0 w0 in one f32 accumulator, the five small
 * ones in a second, summed at the end: operands of 24 bits as conv2d_cg.py:342-349's sgemm has them, with FEWER
 * accumulation roundings than its fma chain (one per 16 k).  Records are 96 bytes per 16 channels, weights 6 KB per
 * stage and 32-row tile.  Selected by: the cbinfer_split3_* functions for the buffers (state bytes / init / rebuild,
 * prepared bytes / prep_weights), mode bit 3 (CBINFER_SPLIT_X3) of cbinfer_split_detect, and weightScale == 0 wherever
 * a frame function takes a weightScale (the triple form has none; cbinfer_split_forward[_tail,_fg] derive the
 * detection's mode bit from it).  rangeFlag is not used.  Workspace as for the pair form. */
#define CBINFER_SPLIT_MAX_SEQUENCES 8
#define CBINFER_SPLIT_POOLED 1
#define CBINFER_SPLIT_COPY_ALL 2
#define CBINFER_SPLIT_FG 4        /* cbinfer_split_detect only: the fine-grained frame's detection (cbinfer_split_forward_fg) */
#define CBINFER_SPLIT_X3 8        /* the split state holds bf16 triples (cbinfer_split3_state_bytes) */
typedef struct {
    const float* input;           /* this frame's layer input (or the pool's input) */
    float* state;                 /* prevInput [C,H,W] */
    void* splitState;             /* cbinfer_split_state_bytes(C,H,W,kH,kW) */
    uint64_t* frameMasks;
    const uint64_t* producerMask; /* pooled only; may be NULL */
    float* output;                /* prevOutput [K,H,W] */
    int32_t* idxOut;
    int32_t* countOut;
    int32_t* rangeFlag;           /* may be NULL */
    uint64_t* maskCopy;           /* may be NULL: receives this frame's change mask (cbinfer_mask_words words) at a
                                     fixed address -- the producerMask of the next layer's pooled detection */
    float* delta;                 /* fine-grained frame only (cbinfer_split_forward_fg): [C,H,W] f32, the thresholded
                                     differences of this frame (every value written) */
    float* reluOut;               /* fine-grained frame only, may be NULL: [K,H,W] kept at relu(output) */
} cbSplitSeq;
int cbinfer_split_supported(int C, int K, int kH, int kW);
int cbinfer_split_max_sequences(void);
long cbinfer_split_max_mask_words(int K);   /* cbinfer_mask_words(H,W) x sequences a launch of a K-channel layer takes */
long cbinfer_split_state_bytes(int C, int H, int W, int kH, int kW);
long cbinfer_split_prepared_bytes(int C, int K, int kH, int kW);
long cbinfer_split_workspace_bytes(int nSeq, int C, int H, int W, int K, int kH, int kW);
int cbinfer_split_prep_weights(const float* weight, void* prepared, int K, int C, int kH, int kW, int H, int W,
                               float weightScale, cbStream_t stream);
int cbinfer_split_state_init(void* splitState, int C, int H, int W, int kH, int kW, cbStream_t stream);
int cbinfer_split_state_rebuild(const float* state, void* splitState, int C, int H, int W, int kH, int kW,
                                int32_t* rangeFlag, cbStream_t stream);
long cbinfer_split3_state_bytes(int C, int H, int W, int kH, int kW);
long cbinfer_split3_prepared_bytes(int C, int K, int kH, int kW);
int cbinfer_split3_prep_weights(const float* weight, void* prepared, int K, int C, int kH, int kW, int H, int W,
                                cbStream_t stream);
int cbinfer_split3_state_init(void* splitState, int C, int H, int W, int kH, int kW, cbStream_t stream);
int cbinfer_split3_state_rebuild(const float* state, void* splitState, int C, int H, int W, int kH, int kW,
                                 cbStream_t stream);
int cbinfer_split_detect(const cbSplitSeq* seqs, int nSeq, int mode, int pH, int pW, int C, int H, int W,
                         int kH, int kW, float threshold, cbStream_t stream);
int cbinfer_split_conv(const cbSplitSeq* seqs, int nSeq, const void* prepared, const float* bias, int C, int H,
                       int W, int K, int kH, int kW, float weightScale, int relu, void* workspace, int forceSplit,
                       cbStream_t stream);
/* the other launches of a frame for several sequences at once (own tensors per sequence, shared weights):
 * cbinfer_change_detection_bits, cbinfer_conv_changed_rows and cbinfer_tail1x1 with arrays of nSeq pointers */
int cbinfer_change_detection_bits_batched(const float* const* inputs, float* const* states, uint64_t* const* bitsOut,
                                          int nSeq, int W, int H, int C, int kHHalf, int kWHalf, float threshold,
                                          int updateInputState, cbStream_t stream);
int cbinfer_conv_changed_rows_batched(const float* const* states, uint64_t* const* bits, int32_t* const* arrive,
                                      uint64_t* const* maskCopies, float* const* outputs, int nSeq,
                                      const void* prepared, const float* bias, int C, int H, int W, int K, int kH,
                                      int kW, int relu, cbStream_t stream);
typedef struct {
    const float* input;
    const int32_t* changeList;
    const int32_t* countDev; /* may be NULL: numChanges entries */
    float* output;
} cbTailSeq;
int cbinfer_tail1x1_batched(const cbTailSeq* seqs, int nSeq, int numChanges, const float* w1Prepared,
                            const float* b1, const float* w2, const float* b2, int C0, int C1, int C2, int H, int W,
                            int relu1, int relu2, cbStream_t stream);
int cbinfer_split_forward(const cbSplitSeq* seqs, int nSeq, int mode, int pH, int pW, const void* prepared,
                          const float* bias, int C, int H, int W, int K, int kH, int kW, float threshold,
                          float weightScale, int relu, void* workspace, cbStream_t stream);
/* The FINE-GRAINED frame (CBConv2d.forward_fg, conv2d.py:160-176 -> conv2d_fg.py:75-85: changeDetectionFG,
 * torch.nonzero, updateOutputFG's K*kH*kW atomicAdds per changed value, cbconv2d_fg_backend.cu:7-66) on the
 * split-state kernels (round 4), same contract as cbinfer_cbconv2d_forward_fg with refreshState = 1: prevInput
 * (`state`) takes the frame, `delta` the thresholded differences, `output` += conv(weights, delta) at the pixels of
 * the dilated any-channel mask, `reluOut` (optional) = relu(output) there; idxOut / countOut / maskCopy as for
 * cbinfer_split_forward.  No bias; fixed summation order.  pooled != 0: `input` is the tensor in front of a
 * 2x2/stride-2 max pool [C,pH,pW] (CBPoolMax2d folded into the detection; H x W the pooled size). */
int cbinfer_split_forward_fg(const cbSplitSeq* seqs, int nSeq, int pooled, int pH, int pW, const void* prepared, int C,
                             int H, int W, int K, int kH, int kW, float threshold, float weightScale, void* workspace,
                             cbStream_t stream);
/* The layer + the fused 1x1 tail behind it (sceneLabeling/modelLoader.py:45-47: the dense conv1x1 -> ReLU -> conv1x1
 * the experiments keep; cbinfer_tail1x1 evaluates it at the changed pixels) with the tail folded into the second
 * launch of a deep contraction: the launch that finishes the layer's outputs -- summing the partial tiles of a split
 * contraction -- keeps each group of 16 finished pixel columns in LDS and runs the tail on them.  Same arithmetic
 * as cbinfer_split_forward followed by cbinfer_tail1x1 (bit-identical outputs), one launch and the tail's gather
 * pass over prevOutput less.  cbinfer_split_tail_supported: deep contraction (>= 48 k-stages), K a multiple of 16
 * and a tail cbinfer_tail1x1_supported(K, C1, C2) takes with C1 >= 4; `bias` and `b1` 16-byte aligned. */
typedef struct {
    const float* w1Prepared;      /* cbinfer_tail1x1_prep */
    const float* b1;
    const float* w2;              /* [C2, C1] */
    const float* b2;
    int C1, C2, relu1, relu2;
    float* output[CBINFER_SPLIT_MAX_SEQUENCES];   /* [C2,H,W] per sequence */
} cbSplitTail;
int cbinfer_split_tail_supported(int C, int K, int kH, int kW, int C1, int C2);
int cbinfer_split_conv_tail(const cbSplitSeq* seqs, int nSeq, const void* prepared, const float* bias, int C, int H,
                            int W, int K, int kH, int kW, float weightScale, int relu, void* workspace, int forceSplit,
                            const cbSplitTail* tail, cbStream_t stream);
int cbinfer_split_forward_tail(const cbSplitSeq* seqs, int nSeq, int mode, int pH, int pW, const void* prepared,
                               const float* bias, int C, int H, int W, int K, int kH, int kW, float threshold,
                               float weightScale, int relu, void* workspace, int forceSplit, const cbSplitTail* tail,
                               cbStream_t stream);
/* cbinfer_split_forward_fg with the fused 1x1 tail in the contraction's second launch (round 5): that launch adds the
 * partial tiles' sum to `output`, keeps `reluOut`, and evaluates the tail on relu(output) when reluOut is given (what
 * the network hands the next module, conv2d.py:169-173), else on output.  Same bits as cbinfer_split_forward_fg
 * followed by cbinfer_tail1x1. */
int cbinfer_split_forward_fg_tail(const cbSplitSeq* seqs, int nSeq, int pooled, int pH, int pW, const void* prepared,
                                  int C, int H, int W, int K, int kH, int kW, float threshold, float weightScale,
                                  void* workspace, const cbSplitTail* tail, cbStream_t stream);

/* ---- fp16 layers on the split-state machinery (round 4): the frame of a CBConv2d in half precision
 * (cbconv2d_cg_half_backend.cu:10-88 change detection, :146-197 genXMatrix / updateOutput around an fp16 matmul;
 * conv2d.py:178-259) of 64 and more input channels (round 5: a count that is not a multiple of 64 -- OpenPose's 185 --
 * is padded to the next one in the pixel-major copy and the prepared weights: zero records, zero weights; the tensors
 * keep their C).  Beside prevInput [C,H,W] f16 the layer keeps a
 * pixel-major copy [Hp][Wp][C] f16 with a zero border (cbinfer_hsplit_state_bytes; no split -- the values are f16
 * already), so that a pixel's 64 channels of one tap are 128 contiguous bytes and both operands of the contraction go
 * global -> LDS by LDS-DMA; v_mfma_f32_32x32x16_f16, f32 accumulation, outputs rounded to f16 once (the arithmetic
 * of cbinfer_conv_changed with CB_F16, another summation order).  feedbackLoop = 1: both states take the changed
 * pixels' values (.cu:74-80 of the half backend); 0: every value of the frame (the layer keeps a copy of its input,
 * conv2d.py:234-236).  frameMasks: cbinfer_frame_mask_bytes(H,W) bytes, zero once; workspace:
 * cbinfer_hsplit_workspace_bytes (0 for fewer than 48 k-stages), zero once; upstreamCount: optional, as for
 * cbinfer_cbconv2d_forward_after.  idxOut / countOut / maskCopy as for cbinfer_split_forward.  pooled != 0: `input`
 * is the tensor in front of a 2x2/stride-2 max pool [C,pH,pW] (CBPoolMax2d folded into the detection; H x W the
 * pooled size), producerMask as for cbinfer_change_detection_bits_pooled.  pooled == 0 with a producerMask (round
 * 5): `input` is the output buffer of another change-based layer of the SAME resolution and producerMask that layer's
 * change mask of this frame (cbinfer_mask_words(H,W) words): 64-pixel segments it did not rewrite are skipped.  Valid
 * under the conditions of upstreamCount (the buffer is what this layer compared last frame wherever the producer
 * rewrote nothing; state not fresh or restored, threshold unchanged since the last frame). */
int cbinfer_hsplit_supported(int C, int K, int kH, int kW);
long cbinfer_hsplit_max_mask_words(int K);
long cbinfer_hsplit_state_bytes(int C, int H, int W, int kH, int kW);
long cbinfer_hsplit_prepared_bytes(int C, int K, int kH, int kW);
long cbinfer_hsplit_workspace_bytes(int C, int H, int W, int K, int kH, int kW);
int cbinfer_hsplit_prep_weights(const void* weight, void* prepared, int K, int C, int kH, int kW, int H, int W,
                                cbStream_t stream);
int cbinfer_hsplit_state_init(void* pixelState, int C, int H, int W, int kH, int kW, cbStream_t stream);
int cbinfer_hsplit_state_rebuild(const void* state, void* pixelState, int C, int H, int W, int kH, int kW,
                                 cbStream_t stream);
int cbinfer_hsplit_forward(const int32_t* upstreamCount, const void* input, int pooled, int pH, int pW,
                           const uint64_t* producerMask, void* state, void* pixelState, uint64_t* frameMasks,
                           void* output, int32_t* idxOut, int32_t* countOut, uint64_t* maskCopy, const void* prepared,
                           const void* bias, int C, int H, int W, int K, int kH, int kW, float threshold,
                           int feedbackLoop, int relu, void* workspace, cbStream_t stream);

/* ---- a GROUP of fp16 layers in one set of launches, each doing the change detection of its CONSUMERS (round 6,
 * ABI 9).  Two things the reference's structure offers and one launch per op hides:
 *  (1) layers chain through their state tensors -- a CBConv2d's input IS the prevOutput of the CBConv2d in front
 *      (conv2d.py:259 returns the state tensor itself; ('changeIndexes', out, idx) at :180-186, :256-259 hands the
 *      producer's change list on) -- so the producing launch knows every value the consumer's changeDetection
 *      (cbconv2d_cg_half_backend.cu:45-88) could find changed: those of the pixels it just recomputed.  For a consumer
 *      that keeps a COPY of its input (feedbackLoop = False, copyInput = True: conv2d.py:234-236, what convert() makes)
 *      the detection decomposes per VALUE -- compare with the consumer's prevInput in half precision (strict >), write
 *      the value into prevInput and into the consumer's pixel-major copy where it differs bit for bit, OR the dilated
 *      pixel into the consumer's frame mask -- and rides in the epilogue of the producing contraction (in the reduce
 *      launch of a contraction that was split along k).  The consumer then runs with detect = 0: its contraction alone.
 *      Valid exactly where the chained detection's producer-mask shortcut is (cbinfer_hsplit_forward above): the
 *      consumer compared this very buffer last frame into this very state with this very threshold; the caller falls
 *      back to detect = 1 for one frame otherwise.  Results are those of the separate launches, bit for bit.
 *  (2) the two branches of an OpenPose stage (poseDetection/openPose/PoseModel.py:122-137) are independent layers of
 *      ONE geometry: nLayers = 2 issues both as one persistent grid (items of both layers in one launch; own weights,
 *      states, masks, lists each; K may differ while the padded tile height agrees -- 38 and 19 outputs both pad to 64).
 * cbHalfLayer: one layer's tensors (all f16 unless said otherwise); next[]: up to CBINFER_HNEXT_MAX consumers of its
 * output (same H x W; in_channels = K).  workspace: cbinfer_hsplit_group_workspace_bytes(nLayers, ...), zero once. */
#define CBINFER_HGROUP_MAX 2
#define CBINFER_HNEXT_MAX 2
typedef struct {
    void* state;              /* the consumer's prevInput [K,H,W] */
    void* pixelState;         /* its pixel-major copy: cbinfer_hsplit_state_bytes(K, H, W, kH, kW) */
    uint64_t* frameMasks;     /* its frame mask */
    int kH, kW;               /* its filter: dilation, and the border geometry of pixelState */
    float threshold;
    int reserved;
} cbHalfNext;
typedef struct {
    const int32_t* upstreamCount;   /* optional (nLayers == 1 only), as for cbinfer_hsplit_forward */
    const void* input;              /* this frame's input (pooled: the pool's input); unused with detect = 0 */
    const uint64_t* producerMask;   /* optional, as for cbinfer_hsplit_forward */
    void* state;                    /* prevInput [C,H,W] */
    void* pixelState;
    uint64_t* frameMasks;
    void* output;                   /* prevOutput [K,H,W] */
    int32_t* idxOut;
    int32_t* countOut;
    uint64_t* maskCopy;             /* may be NULL */
    const void* prepared;           /* cbinfer_hsplit_prep_weights */
    const void* bias;               /* [K] f16, may be NULL */
    int K;
    float threshold;
    int relu;
    int detect;                     /* 1: this call runs the layer's change detection; 0: a producer's launch did */
    int nNext;
    int reserved;
    cbHalfNext next[CBINFER_HNEXT_MAX];
} cbHalfLayer;
long cbinfer_hsplit_group_workspace_bytes(int nLayers, int C, int H, int W, int K, int kH, int kW);
int cbinfer_hsplit_forward_group(const cbHalfLayer* layers, int nLayers, int pooled, int pH, int pW, int C, int H, int W,
                                 int kH, int kW, int feedbackLoop, void* workspace, cbStream_t stream);

/* ---- a5..a8 fused for a layer of few channels, ROW-PAIR form, with the NEXT layer's pooled change detection folded
 * in (round 4).  Replaces, per frame, the launcher sequence genXMatrix -> matmul -> updateOutput
 * (cbconv2d_cg_backend.cu:138-197, conv2d_cg.py:342-349) of a feedback-mode CBConv2d with at most 4 input and 16
 * output channels AND -- when `next` names the split-state layer (cbinfer_split_*) behind a 2x2/stride-2 max pool --
 * that layer's CBPoolMax2d (conv2d.py:49-78) + changeDetection with updateInputState (cbconv2d_cg_backend.cu:40-81):
 * a workgroup owns a row PAIR of a 64-pixel mask word, i.e. whole pooling windows, so it compares the pooled values
 * of the windows it rewrote with the next layer's state, refreshes that state (and its pre-split copy) and ORs the
 * dilated changes into the next layer's frame mask itself.  The next layer then runs cbinfer_split_conv[_tail] WITHOUT
 * cbinfer_split_detect.  Results are those of the separate launches (same predicate, same refresh, same mask); a
 * window none of whose pixels changed is skipped exactly as the producer-mask shortcut of cbinfer_split_detect skips
 * it, so the caller must fall back to the separate detection for one frame whenever that shortcut is not valid (fresh
 * or restored next-layer state, changed next-layer threshold).
 *   bits / maskCopy : as for cbinfer_conv_changed_rows (every workgroup zeroes the mask words of its own units); ctl:
 *                     unused (kept for the signature: one int32)
 *   prepared        : cbinfer_rowconv_prep_weights
 *   next            : may be NULL (or next->state NULL): no folding */
typedef struct {
    float* state;             /* the next layer's prevInput [K, H, W] */
    void* splitState;         /* its cbinfer_split_state_bytes(K, H, W, kH, kW) buffer (arith 1: cbinfer_split3_state_bytes) */
    uint64_t* frameMasks;     /* its frame mask (cbSplitSeq.frameMasks) */
    int32_t* rangeFlag;       /* may be NULL */
    int H, W, kH, kW;         /* the next layer's map size (behind the pool) and filter */
    float threshold;
    int arith;                /* 0: its split state holds f16 pairs, 1: bf16 triples (cbinfer_split3_*) */
} cbNextDetect;
typedef struct {                /* ONE sequence's tensors for the batched form (own state each, shared weights) */
    const float* state;         /* prevInput [C,H,W] */
    float* output;              /* prevOutput [K,H,W] */
    uint64_t* bits;
    uint64_t* maskCopy;         /* may be NULL */
    float* nextState;           /* the next layer's tensors as in cbNextDetect; NULL: no folding */
    void* nextSplitState;
    uint64_t* nextFrameMasks;
    int32_t* nextRangeFlag;     /* may be NULL */
} cbPairSeq;
int cbinfer_rowpairs_supported(int C, int K, int kH, int kW, int H, int W);
int cbinfer_conv_changed_rowpairs_batched(const cbPairSeq* seqs, int nSeq, const void* prepared, const float* bias,
                                          int C, int H, int W, int K, int kH, int kW, int relu,
                                          const cbNextDetect* next, cbStream_t stream);
/* round 6 (ABI 11): the row-pair launch WITH the layer's own change detection (replaces changeDetection +
 * updateInputState, conv2d_cg.py:100-122 / cbconv2d_cg_backend.cu:40-81, as a launch of its own in front of the contraction).
 * Every workgroup compares the patch of its row pair in `input` and `prevInput`, dilates the changed pixels into its own two
 * mask words (maskCopy takes them: there is no other mask) and multiplies the refreshed values.  prevInput is NOT written --
 * every workgroup must see the old state -- : the caller refreshes it behind this launch (cbinfer_refresh_state).  One
 * sequence, 7x7 filters; `next` as for cbinfer_conv_changed_rowpairs. */
int cbinfer_conv_rowpairs_detect(const float* input, const float* prevInput, float* prevOutput, uint64_t* maskCopy,
                                 const void* prepared, const float* bias, int C, int H, int W, int K, int kH, int kW,
                                 float threshold, int relu, const cbNextDetect* next, cbStream_t stream);
int cbinfer_refresh_state(const float* frame, float* state, int C, int H, int W, float threshold, cbStream_t stream);
int cbinfer_conv_changed_rowpairs(const float* state, uint64_t* bits, int32_t* ctl, uint64_t* maskCopy,
                                  const void* prepared, const float* bias, float* output, int C, int H, int W, int K,
                                  int kH, int kW, int relu, const cbNextDetect* next, cbStream_t stream);
int cbinfer_cbconv2d_forward_rowpairs(const float* input, float* prevInput, float* prevOutput, uint64_t* bits,
                                      int32_t* ctl, uint64_t* maskCopy, const void* prepared, const float* bias, int C,
                                      int H, int W, int K, int kH, int kW, float threshold, int relu,
                                      const cbNextDetect* next, cbStream_t stream);

/* ---- round 6: the same folding for a split-state PRODUCER (the 16 -> 64 layer of the scene-labeling network in front of
 * its second pool; replaces the consumer's cbinfer_split_detect launch, i.e. CBPoolMax2d conv2d.py:49-78 + changeDetection
 * conv2d_cg.py:100-122 / cbconv2d_cg_backend.cu:40-81 of the NEXT layer).  The contraction's work list is ordered by 2x2
 * pooling WINDOW (a tile = 16 windows touched by the change mask; only their changed pixels are computed, the others'
 * stored outputs take part in the maximum), so the workgroup that recomputes a window also owns its pooled pixel: it
 * compares the maximum with the next layer's state (strict >, all channels), refreshes that state and its split copy at the
 * changed pooled pixels and ORs their dilation into the next layer's frame mask.  The change list the layer hands out
 * (cbSplitSeq.idxOut) keeps the reference's row-major order.  Taken: one sequence, bf16-triple arithmetic (weightScale 0),
 * K <= 64 output channels (a multiple of 16), fewer than 48 k-stages, a mask of at most 1280 words; next as for the
 * row-pair kernel.  The caller falls back to the separate detection under the same conditions as there.
 *   cbinfer_split_conv_next    : cbinfer_split_conv + next (this layer's own detection was its producer's business)
 *   cbinfer_split_forward_next : cbinfer_split_forward + next */
/* (ABI 11) `side`: the feedback refresh of ANOTHER layer's state carried by this launch's workgroups without a work item --
 * for the row-pair layer in front when it ran cbinfer_conv_rowpairs_detect (which leaves its state alone). */
typedef struct {
    const float* frame;       /* that layer's input of this frame [C, H, W] */
    float* state;             /* its prevInput */
    int C, H, W;
    float threshold;
} cbSideRefresh;
int cbinfer_split_conv_next_refresh(const cbSplitSeq* seqs, int nSeq, const void* prepared, const float* bias, int C, int H,
                                    int W, int K, int kH, int kW, float weightScale, int relu, void* workspace,
                                    const cbNextDetect* next, const cbSideRefresh* side, cbStream_t stream);
/* ... and the same side job on a PIXEL-order contraction (the consumer does not run in window order): one sequence, the
 * bf16-triple arithmetic, K <= 64, fewer than 48 k-stages, at most 1280 mask words. */
int cbinfer_split_refresh_supported(int C, int K, int kH, int kW, int H, int W);
int cbinfer_split_conv_refresh(const cbSplitSeq* seqs, int nSeq, const void* prepared, const float* bias, int C, int H, int W,
                               int K, int kH, int kW, float weightScale, int relu, void* workspace,
                               const cbSideRefresh* side, cbStream_t stream);
int cbinfer_split_next_supported(int C, int K, int kH, int kW, int H, int W, const cbNextDetect* next);
int cbinfer_split_conv_next(const cbSplitSeq* seqs, int nSeq, const void* prepared, const float* bias, int C, int H,
                            int W, int K, int kH, int kW, float weightScale, int relu, void* workspace,
                            const cbNextDetect* next, cbStream_t stream);
int cbinfer_split_forward_next(const cbSplitSeq* seqs, int nSeq, int mode, int pH, int pW, const void* prepared,
                               const float* bias, int C, int H, int W, int K, int kH, int kW, float threshold,
                               float weightScale, int relu, void* workspace, const cbNextDetect* next,
                               cbStream_t stream);

/* ---- channel concatenation of batch-1 [Ci,H,W] tensors into [sum Ci,H,W] as ONE launch on the caller's stream
 * (round 6).  The reference's pose network does torch.cat(dim=1) between its stages
 * (poseDetection/openPose/PoseModel.py:131); with this entry point a frame of that network consists of library calls only
 * and can be replayed from a recorded launch program (pycbinfer.FrameProgram).  sources / channels: n (<= 4) device
 * pointers and channel counts (host arrays); output: [sum channels, H, W] of the same dtype; HW = H * W. */
#define CBINFER_CONCAT_MAX 4
int cbinfer_concat_channels(const void* const* sources, const int32_t* channels, int n, void* output, long HW, int dtype,
                            cbStream_t stream);

/* replaces conv2d_fg_cpu, cbconv2d_fg_backend.cu:81-112: HOST pointers, host code, race-free. */
void cbinfer_conv2d_fg_cpu(const float* input, const float* prevInput, float* output,
                           const float* weight, float threshold, int no, int ni, int h, int w,
                           int kh, int kw);

#ifdef __cplusplus
}
#endif
#endif /* CBINFER_HIP_H */
