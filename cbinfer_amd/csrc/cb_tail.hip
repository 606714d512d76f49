// Change-based evaluation of a chain of two 1x1 convolutions (conv -> [ReLU] -> conv) at the changed pixels
// of the layer in front of it, in ONE launch: the dense "1x1 tail" of the scene-labeling network
// (sceneLabeling/modelLoader.py:45-47 keeps baseline modules 8..10 dense; experiment 1, :41-44, runs them as
// two CBConv2d fed by propagated change indexes).  A 1x1 layer's output can only change where its input
// did, and the producing CBConv2d rewrites its output exactly at the pixels of its change list, so
// recomputing the chain at those pixels and keeping every other output gives the dense result.
//
// Shape: 16 changed pixels per workgroup, so that even a few thousand pixels spread over all CUs (the
// whole job is ~100 MFLOP: latency, not throughput, is what counts).  Wave w owns hidden channels
// 16w..16w+15: H[16 x 16 px] = W1[16 x C0] . X[C0 x 16 px] on v_mfma_f32_16x16x4_f32 (exact f32 fma
// chain), A fragments straight from the re-laid-out weights (one 16-byte load per four MFMAs and lane),
// the gathered X tile staged once through LDS.  relu(H + b1) goes to LDS, and the C2 x 16 outputs of the
// second (tiny) layer are plain f32 dot products over it.
#include "cb_common.h"
#include "cb_tail_core.h"

namespace {


// W1 [C1, C0] -> fragment order [mt][s4][lane][4]: element (mt, s4, lane, j) = W1[16 mt + lane%16][16 s4 + 4 j + lane/16]
__global__ __launch_bounds__(256) void cb_tail_prep_kernel(const float* __restrict__ w1, float* __restrict__ w1p,
                                                          int C1, int C0, int C0P) {
    const long total = (long)((C1 + 15) / 16) * (C0P / 16) * 64 * 4;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int j = (int)(e & 3), lane = (int)((e >> 2) & 63);
    const long r = e >> 8;
    const int s4 = (int)(r % (C0P / 16)), mt = (int)(r / (C0P / 16));
    const int m = 16 * mt + (lane & 15), k = 16 * s4 + 4 * j + (lane >> 4);
    w1p[e] = (m < C1 && k < C0) ? w1[(long)m * C0 + k] : 0.f;
}

// (several sequences in one launch: blockIdx.y names the sequence, its tensors come from the table)
struct TailSeqs {
    struct {
        const float* x;
        const int32_t* list;
        const int32_t* countDev;
        float* out;
    } seq[CBINFER_SPLIT_MAX_SEQUENCES];
};

__global__ __launch_bounds__(64 * CB_TAIL_MAXW) void cb_tail1x1_kernel(
    TailSeqs tab, int nHost, const float* __restrict__ w1p, const float* __restrict__ b1,
    const float* __restrict__ w2, const float* __restrict__ b2, int C0, int C0P, int C1, int C2,
    int HW, int relu1, int relu2) {
    extern __shared__ float sm[];   // Xs[C0P][16] | Hs[C1P][17] | W2s[C2][C1] | b2s[C2]
    cb_touch_kernarg<sizeof(TailSeqs) + 72>();
    const float* __restrict__ x = tab.seq[blockIdx.y].x;
    const int32_t* __restrict__ list = tab.seq[blockIdx.y].list;
    const int32_t* __restrict__ countDev = tab.seq[blockIdx.y].countDev;
    float* out = tab.seq[blockIdx.y].out;
    const int t = threadIdx.x, NT = blockDim.x;
    const CbTailLds L = cb_tail_lds(sm, C0P, C1, C2, NT);
    float* Xs = L.Xs;
    float* W2s = L.W2s;
    float* b2s = L.b2s;
    __shared__ int s_pix[CB_TAIL_PX];
    CbTailPre P;
    cb_tail_preload(P, w1p, b1, C0P, C1);      // (requested beside the change count: one round trip, not two)
    const int N = countDev ? min(*countDev, nHost) : nHost;
    if ((int)blockIdx.x * CB_TAIL_PX >= N) return;
    // the second layer's (small) matrix and bias live in LDS for the life of the workgroup
    for (int i = t; i < C2 * C1; i += NT) W2s[i] = w2[i];
    for (int i = t; i < C2; i += NT) b2s[i] = b2[i];

    for (int tile = blockIdx.x; tile * CB_TAIL_PX < N; tile += gridDim.x) {
        const int n0 = tile * CB_TAIL_PX;
        __syncthreads();   // previous tile's Hs / s_pix reads are done
        if (t < CB_TAIL_PX) {
            int pix = n0 + t < N ? list[n0 + t] : -1;
            if ((unsigned)pix >= (unsigned)HW) pix = -1;   // foreign-resolution entry: dropped
            s_pix[t] = pix;
        }
        __syncthreads();
        // gather X[c][px]: thread -> (px = t % 16, c = t / 16 + i * NT/16); 16 neighbouring lanes read the
        // 16 pixels of one channel plane (contiguous where the changed pixels are)
        const int px = t & 15;
        const int mypix = s_pix[px], pixLd = max(mypix, 0);
        const int cstep = NT >> 4;
        for (int c0 = t >> 4; c0 < C0P; c0 += 16 * cstep) {   // sixteen loads in flight per lane
            // (never predicated -- a predicated load is a branch of its own, sixteen of them sixteen round trips:
            //  clamped addresses, predicated uses)
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = x[(long)min(c0 + u * cstep, C0 - 1) * HW + pixLd];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int c = c0 + u * cstep;
                if (c < C0P) Xs[c * CB_TAIL_PX + px] = (mypix >= 0 && c < C0) ? v[u] : 0.f;
            }
        }

        cb_tail_tile(L, s_pix, P, w1p, out, C0P, C1, C2, HW, relu1, relu2);
    }
}

}  // namespace

extern "C" {

int cbinfer_tail1x1_max_hidden(void) { return 16 * CB_TAIL_MAXW; }

// 1 if cbinfer_tail1x1 takes these channel counts (hidden width and LDS budget), so that a fusion can be refused
// when it is set up rather than at the first frame
int cbinfer_tail1x1_supported(int C0, int C1, int C2) {
    return C0 > 0 && C1 > 0 && C2 > 0 && C1 <= 16 * CB_TAIL_MAXW && cb_tail_lds_bytes(C0, C1, C2) <= 60 * 1024;
}

long cbinfer_tail1x1_prepared_bytes(int C1, int C0) {
    const long C0P = (C0 + 15) / 16 * 16;
    return (long)((C1 + 15) / 16) * 16 * C0P * 4;
}

int cbinfer_tail1x1_prep(const float* w1, float* w1Prepared, int C1, int C0, cbStream_t stream) {
    CB_REQUIRE(w1 && w1Prepared && C1 > 0 && C0 > 0);
    if (C1 > 16 * CB_TAIL_MAXW) return CB_ERR_UNSUPPORTED;
    const int C0P = (C0 + 15) / 16 * 16;
    const long total = cbinfer_tail1x1_prepared_bytes(C1, C0) / 4;
    hipLaunchKernelGGL(cb_tail_prep_kernel, dim3(cb_div_up(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       w1, w1Prepared, C1, C0, C0P);
    return cb_launch_status();
}

// The same for nSeq sequences in one launch (own input, change list, count and output each; shared weights).
int cbinfer_tail1x1_batched(const cbTailSeq* seqs, int nSeq, int numChanges, const float* w1Prepared,
                            const float* b1, const float* w2, const float* b2, int C0, int C1, int C2, int H, int W,
                            int relu1, int relu2, cbStream_t stream) {
    CB_REQUIRE(seqs && nSeq >= 1 && nSeq <= CBINFER_SPLIT_MAX_SEQUENCES && w1Prepared && b1 && w2 && b2 && C0 > 0 &&
               C1 > 0 && C2 > 0 && H > 0 && W > 0 && numChanges >= 0);
    if (C1 > 16 * CB_TAIL_MAXW) return CB_ERR_UNSUPPORTED;
    if (numChanges == 0) return CB_OK;
    TailSeqs tab;
    for (int q = 0; q < nSeq; ++q) {
        CB_REQUIRE(seqs[q].input && seqs[q].changeList && seqs[q].output);
        tab.seq[q].x = seqs[q].input;
        tab.seq[q].list = seqs[q].changeList;
        tab.seq[q].countDev = seqs[q].countDev;
        tab.seq[q].out = seqs[q].output;
    }
    const int C0P = (C0 + 15) / 16 * 16;
    const int waves = (C1 + 15) / 16;
    const size_t lds = cb_tail_lds_bytes(C0, C1, C2);
    if (lds > 60 * 1024) return CB_ERR_UNSUPPORTED;
    long tiles = ((long)numChanges + CB_TAIL_PX - 1) / CB_TAIL_PX;
    if (tiles > 2048) tiles = 2048;
    hipLaunchKernelGGL(cb_tail1x1_kernel, dim3((unsigned)tiles, nSeq), dim3(64 * waves), lds, (hipStream_t)stream,
                       tab, numChanges, w1Prepared, b1, w2, b2, C0, C0P, C1, C2, H * W, relu1, relu2);
    return cb_launch_status();
}

int cbinfer_tail1x1(const float* input, const int32_t* changeList, int numChanges, const int32_t* countDev,
                    const float* w1Prepared, const float* b1, const float* w2, const float* b2,
                    float* output, int C0, int C1, int C2, int H, int W, int relu1, int relu2,
                    cbStream_t stream) {
    CB_REQUIRE(input && changeList && w1Prepared && b1 && w2 && b2 && output && C0 > 0 && C1 > 0 &&
               C2 > 0 && H > 0 && W > 0 && numChanges >= 0);
    if (C1 > 16 * CB_TAIL_MAXW) return CB_ERR_UNSUPPORTED;
    if (numChanges == 0) return CB_OK;
    TailSeqs tab;
    tab.seq[0].x = input, tab.seq[0].list = changeList, tab.seq[0].countDev = countDev, tab.seq[0].out = output;
    const int C0P = (C0 + 15) / 16 * 16;
    const int waves = (C1 + 15) / 16;
    const size_t lds = cb_tail_lds_bytes(C0, C1, C2);
    if (lds > 60 * 1024) return CB_ERR_UNSUPPORTED;   // stays below the 64 KB that needs no opt-in
    long tiles = ((long)numChanges + CB_TAIL_PX - 1) / CB_TAIL_PX;
    if (tiles > 2048) tiles = 2048;   // grid-stride beyond: the capacity is H*W when the count is on the device
    hipLaunchKernelGGL(cb_tail1x1_kernel, dim3((unsigned)tiles), dim3(64 * waves), lds, (hipStream_t)stream,
                       tab, numChanges, w1Prepared, b1, w2, b2, C0, C0P, C1, C2, H * W, relu1, relu2);
    return cb_launch_status();
}

}  // extern "C"
