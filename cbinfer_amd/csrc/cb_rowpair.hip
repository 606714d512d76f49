// Row-PAIR contraction with the NEXT layer's pooled change detection folded in (round 4): the fused a5..a8 path of a
// feedback-mode CBConv2d with few input channels and at most 16 output channels (conv2d.py:240-251; kernels
// cbconv2d_cg_backend.cu:138-197) -- the 3->16 7x7 layer of the scene-labeling network -- and, in the same launch,
// what the layer BEHIND the 2x2 max pool would otherwise do in a launch of its own: CBPoolMax2d (conv2d.py:49-78) +
// changeDetection with updateInputState (cbconv2d_cg_backend.cu:40-81) of a split-state layer (cb_split.hip).
//
// Why: round 3's frame was seven dependent launches, 35 of its 84 us in launches whose work is a few microseconds of
// latency chains: cb_rowconv_f32_kernel started 2560 workgroups of which 852 had work
// (11.6 us), and cbs_detect_kernel<POOL> then read the freshly written outputs back (6.6 us).
//
//   unit      one workgroup iteration = one ROW PAIR x one 64-pixel mask word: output rows 2yo, 2yo+1, columns
//             64 tx .. 64 tx + 63 -- i.e. exactly the 2x2 windows of the 32 pooled pixels (yo, 32 tx ..): the unit's
//             workgroup owns every window pixel, so the pooled detection needs nobody else's results.
//   schedule  a persistent grid (two workgroups per CU); every workgroup copies the layer's change mask into LDS,
//             makes the ORDERED list of non-empty units with one scan and takes units blockIdx.x, + gridDim.x, ...
//             The last workgroup out zeroes the mask for the next frame (arrival counter; nobody waits).
//   gather    the (kH + 1) x (64 + kW - 1) input rows under the pair are staged once per channel in LDS (coalesced
//             row loads); every B operand of every tap is a ds_read_b32 at `lane base + immediate tap offset`.
//   product   v_mfma_f32_16x16x4_f32 (the exact f32 fma chain), one 16-pixel tile per wave, eight waves: up to
//             four tiles per row; the weights in MFMA fragment order straight from L2 into registers
//             (cb_rowconv.hip's prepared layout).
//   epilogue  bias / ReLU, scatter to prevOutput AND into an LDS tile [2][16][64] of the pair's outputs that was
//             pre-filled with the old outputs; then, per pooled pixel whose window holds a changed pixel: 2x2 max,
//             strict-> comparison with the next layer's state over all channels, feedback refresh of that state and
//             of its pre-split pixel-major copy (f16 pairs), dilation by the next layer's filter support ORed into
//             its frame mask.  A window none of whose pixels changed compares as it did last frame and is skipped --
//             the producer-mask shortcut of cbs_detect_kernel<POOL>, here for free.
#include <stdlib.h>

#include "cb_split_common.h"

namespace cbp {

typedef float floatx4 __attribute__((ext_vector_type(4)));
using cbs::halfx8;

#define CBP_NT 512
#define CBP_NW 8
#define CBP_MAXWORDS 4096           // mask words of the layer (kept in LDS: 32 KB)

struct PairNext {
    float* state;                   // the next layer's prevInput [K, H2, W2]; null: no folding
    char* S;                        // its split state
    unsigned long long* masks;      // its frame mask (the detection ORs into it)
    int* rangeFlag;
    int H2, W2, wpr2, kHH, kWH, Wp, rec, padY, padXL;
    float th;
};
struct PairParams {
    const float* state;             // prevInput [C,H,W]: what the gather reads (conv2d.py:242)
    const float* wq;                // cbinfer_rowconv_prep_weights layout
    const float* bias;
    float* out;                     // prevOutput [K,H,W]
    unsigned long long* bits;       // change mask of this frame (zeroed by the last workgroup out)
    int* ctl;                       // arrival counter (zero between launches)
    unsigned long long* maskCopy;
    int C, H, W, K, relu, wpr, MW, units;
    PairNext next;
};

__host__ __device__ constexpr int cbp_plane_stride(int kH, int kW) {
    int cs = (kH + 1) * (64 + kW - 1);
    while (cs % 32 != 16) ++cs;     // lane quarters q and q+1 hit disjoint bank halves
    return cs;
}

// r-th (0-based) set bit of w, r < popcount(w)
__device__ __forceinline__ int cbp_nth_bit(unsigned long long w, int r) {
    int pos = 0;
#pragma unroll
    for (int width = 32; width >= 1; width >>= 1) {
        const unsigned long long lowmask = ((1ull << width) - 1ull) << pos;
        const int c = __popcll(w & lowmask);
        if (r >= c) {
            r -= c;
            pos += width;
        }
    }
    return pos;
}

template <int KH, int KW>
__global__ __launch_bounds__(CBP_NT, 4) void cbp_rowpair_kernel(PairParams p) {
    cb_touch_kernarg<sizeof(PairParams)>();
    constexpr int RS = 64 + KW - 1, PR = KH + 1, CS = cbp_plane_stride(KH, KW), S = KH * KW, G = (S + 3) / 4;
    constexpr int PH = (KH - 1) / 2, PW = (KW - 1) / 2;
    __shared__ unsigned long long s_mask[CBP_MAXWORDS];
    __shared__ unsigned short s_units[CBP_MAXWORDS];
    __shared__ float s_patch[4 * CS];
    __shared__ float s_out[2 * 16 * 64];          // [row][channel][x]: the pair's outputs after this frame
    __shared__ float s_P[16 * 33];                // pooled values [channel][xo]
    __shared__ unsigned s_chg[CBP_NW];
    __shared__ int s_wsum[CBP_NW];
    __shared__ int s_last;

    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int MW = p.MW, wpr = p.wpr, H = p.H, W = p.W, HW = H * W;
    const bool fold = p.next.state != nullptr;

    // ---- the mask into LDS, a copy of it to its fixed address, the ordered list of non-empty units ----------------
    for (int i = t; i < MW; i += CBP_NT) s_mask[i] = p.bits[i];
    __syncthreads();
    if (p.maskCopy)
        for (int i = blockIdx.x * CBP_NT + t; i < MW; i += gridDim.x * CBP_NT) p.maskCopy[i] = s_mask[i];
    const int units = p.units;
    const int CHU = (units + CBP_NT - 1) / CBP_NT, u0 = t * CHU;
    int cnt = 0;
    for (int u = u0; u < min(u0 + CHU, units); ++u) {
        const int yo = u / wpr, tx = u - yo * wpr, ya = 2 * yo;
        const unsigned long long w = s_mask[ya * wpr + tx] | (ya + 1 < H ? s_mask[(ya + 1) * wpr + tx] : 0ull);
        cnt += w != 0ull;
    }
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    int base = 0, nUnits = 0;
#pragma unroll
    for (int w = 0; w < CBP_NW; ++w) {
        const int v = s_wsum[w];
        base += w < wave ? v : 0;
        nUnits += v;
    }
    nUnits = __builtin_amdgcn_readfirstlane(nUnits);
    {
        int pos = base + incl - cnt;
        for (int u = u0; u < min(u0 + CHU, units); ++u) {
            const int yo = u / wpr, tx = u - yo * wpr, ya = 2 * yo;
            const unsigned long long w = s_mask[ya * wpr + tx] | (ya + 1 < H ? s_mask[(ya + 1) * wpr + tx] : 0ull);
            if (w != 0ull) s_units[pos++] = (unsigned short)u;
        }
    }
    __syncthreads();

    // ---- per-wave constants: the weights (fragment order, every wave the same 16 output channels) and the bias -----
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.wq, 0, G * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t srsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.state, 0, (int)min((long)p.C * HW * 4, (long)0x7fffffff), 0x00020000);
    const __amdgpu_buffer_rsrc_t orsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, (int)min((long)p.K * HW * 4, (long)0x7fffffff), 0x00020000);
    floatx4 a[G];
    float bv[4];
    if (blockIdx.x < (unsigned)nUnits) {
#pragma unroll
        for (int g = 0; g < G; ++g)
            a[g] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lane * 16, g * 1024, 0));
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[r] = p.bias ? p.bias[min(4 * (lane >> 4) + r, p.K - 1)] : 0.f;
    }

    for (int it = blockIdx.x; it < nUnits; it += gridDim.x) {
        const int u = s_units[it];
        const int yo = u / wpr, tx = u - yo * wpr, ya = 2 * yo;
        const bool hasB = ya + 1 < H;
        const unsigned long long wordA = s_mask[ya * wpr + tx], wordB = hasB ? s_mask[(ya + 1) * wpr + tx] : 0ull;
        // ---- requests, all in one burst: the patch, the pair's old outputs, the next layer's state ----------------
        constexpr int PTOT = 4 * PR * RS, PPT = (PTOT + CBP_NT - 1) / CBP_NT;
        float pv[PPT];
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int e = t + CBP_NT * k;
            const int r = e / RS, j = e - r * RS;          // (compile-time divisor)
            const int c = r / PR, pr = r - c * PR;
            const int yy = ya + pr - PH, xx = tx * 64 - PW + j;
            const bool ok = e < PTOT && c < p.C && yy >= 0 && yy < H && xx >= 0 && xx < W;
            // (an invalid element gets an out-of-range offset: the buffer load returns 0 for it)
            pv[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                  srsrc, ok ? ((c * H + yy) * W + xx) * 4 : (1 << 30), 0, 0));
        }
        float ov[4];
        float s2 = 0.f;
        const int c2 = t >> 5, xo = t & 31, gx = tx * 32 + xo;
        const bool valid2 = fold && yo < p.next.H2 && gx < p.next.W2 && c2 < p.K;
        const long H2W2 = fold ? (long)p.next.H2 * p.next.W2 : 0;
        if (fold) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = t + CBP_NT * k;
                const int x = e & 63, m = (e >> 6) & 15, r = e >> 10;
                const int yy = ya + r, xx = tx * 64 + x;
                const bool ok = yy < H && xx < W && m < p.K;
                ov[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                      orsrc, ok ? (m * HW + yy * W + xx) * 4 : (1 << 30), 0, 0));
            }
            // (clamped address, predicated use)
            s2 = p.next.state[(long)min(c2, p.K - 1) * H2W2 + (long)min(yo, p.next.H2 - 1) * p.next.W2 +
                              min(gx, p.next.W2 - 1)];
        }
        __syncthreads();        // (the previous unit's readers of s_patch / s_out / s_P are done)
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int e = t + CBP_NT * k;
            const int r = e / RS, j = e - r * RS;
            const int c = r / PR, pr = r - c * PR;
            if (e < PTOT) s_patch[c * CS + pr * RS + j] = pv[k];
        }
        if (fold) {
#pragma unroll
            for (int k = 0; k < 4; ++k) s_out[t + CBP_NT * k] = ov[k];
        }
        __syncthreads();

        // ---- one 16-pixel tile per wave: waves 0-3 the first row's tiles, 4-7 the second row's -------------------
        const int row = wave >> 2, tile = wave & 3;
        const unsigned long long word = row ? wordB : wordA;
        const int pc = __popcll(word);
        const bool active = tile * 16 < pc;
        const int n = tile * 16 + (lane & 15);
        const int xl = cbp_nth_bit(word, n < pc ? n : 0);
        if (active) {
            const float* pl = s_patch + (lane >> 4) * CS + row * RS + xl;
            floatx4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int ky = s / KW, kx = s - ky * KW;
                const float b = pl[ky * RS + kx];
                if (s & 1)
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s >> 2][s & 3], b, acc1, 0, 0, 0);
                else
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s >> 2][s & 3], b, acc0, 0, 0, 0);
            }
            const floatx4 acc = acc0 + acc1;
            if (n < pc) {
                const int pix = (ya + row) * W + tx * 64 + xl;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = 4 * (lane >> 4) + r;
                    float v = acc[r] + bv[r];
                    if (p.relu) v = v <= 0.f ? 0.f : v;
                    if (m < p.K) {
                        p.out[(long)m * HW + pix] = v;
                        s_out[(row * 16 + m) * 64 + xl] = v;
                    }
                }
            }
        }
        if (!fold) continue;
        __syncthreads();

        // ---- the next layer's detection for the 32 pooled pixels of this unit: thread = (channel c2, pooled xo) ----
        const unsigned long long tw = wordA | wordB;
        const bool touched = ((tw >> (2 * xo)) & 3ull) != 0ull;
        const int cx1 = (tx * 64 + 2 * xo + 1 < W) ? 2 * xo + 1 : 2 * xo, r1 = hasB ? 16 * 64 : 0;
        const float* so = s_out + min(c2, 15) * 64;
        const float pooled = fmaxf(fmaxf(so[2 * xo], so[cx1]), fmaxf(so[r1 + 2 * xo], so[r1 + cx1]));
        const bool chg = valid2 && touched && cb_changed(s2, pooled, p.next.th);
        const unsigned long long bal = __ballot(chg);
        if (lane == 0) s_chg[wave] = (unsigned)(bal | (bal >> 32));
        s_P[min(c2, 15) * 33 + xo] = pooled;
        __syncthreads();
        unsigned m32 = 0;
#pragma unroll
        for (int w = 0; w < CBP_NW; ++w) m32 |= s_chg[w];
        m32 = __builtin_amdgcn_readfirstlane(m32);
        if (m32 == 0u) continue;        // (uniform)
        // feedback: refresh the f32 state at the changed pooled pixels only (.cu:74-80) ...
        if (valid2 && ((m32 >> xo) & 1u)) p.next.state[(long)c2 * H2W2 + (long)yo * p.next.W2 + gx] = pooled;
        // ... and its pre-split pixel-major copy: thread = (pooled pixel, 8-channel half), whole 16-byte pieces
        if (t < 64) {
            const int pl2 = t >> 1, half = t & 1;
            if ((m32 >> pl2) & 1u) {
                halfx8 hi, lo;
                bool over = false;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int c = half * 8 + j;
                    const float v = (c < p.K ? s_P[c * 33 + pl2] : 0.f) * CBS_XSCALE;
                    over |= !(fabsf(v) <= CBS_F16_MAX);
                    _Float16 h, l;
                    cbs::cbs_split(v, h, l);
                    hi[j] = h, lo[j] = l;
                }
                char* rec = p.next.S + CBS_SPAD +
                            ((long)(yo + p.next.padY) * p.next.Wp + (tx * 32 + pl2 + p.next.padXL)) * p.next.rec + half * 16;
                *(halfx8*)rec = hi;
                *(halfx8*)(rec + 32) = lo;
                if (over && p.next.rangeFlag) *p.next.rangeFlag = 1;
            }
        }
        // dilation by the next layer's filter support, ORed into its frame mask: the 32 pooled pixels are one half
        // of word tx / 2 of row yo
        if (wave == 1) {
            const int w2 = tx >> 1, wpr2 = p.next.wpr2;
            const unsigned long long M = (unsigned long long)m32 << ((tx & 1) * 32);
            unsigned long long D = M, SR = 0ull, SL = 0ull;
            for (int d = 1; d <= p.next.kWH; ++d) {
                D |= (M << d) | (M >> d);
                SR |= M >> (64 - d);
                SL |= M << (64 - d);
            }
            auto validMask = [&](int tile) -> unsigned long long {
                const int rem = p.next.W2 - tile * 64;
                return rem >= 64 ? ~0ull : (rem <= 0 ? 0ull : ((1ull << rem) - 1ull));
            };
            D &= validMask(w2);
            SR = (w2 + 1 < wpr2) ? (SR & validMask(w2 + 1)) : 0ull;
            if (w2 == 0) SL = 0ull;
            const int items = 3 * (2 * p.next.kHH + 1);
            for (int i = lane; i < items; i += 64) {
                const int yy = yo + i / 3 - p.next.kHH;
                const int which = i % 3;
                if (yy < 0 || yy >= p.next.H2) continue;
                const unsigned long long v = which == 0 ? D : (which == 1 ? SR : SL);
                const int t2 = which == 0 ? w2 : (which == 1 ? w2 + 1 : w2 - 1);
                if (v) atomicOr(&p.next.masks[(long)yy * wpr2 + t2], v);
            }
        }
    }

    // ---- last workgroup out zeroes the mask (every workgroup copied it into its LDS before anything else) ---------
    __syncthreads();
    if (t == 0) s_last = __hip_atomic_fetch_add(p.ctl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_last == (int)gridDim.x - 1) {
        for (int i = t; i < MW; i += CBP_NT) p.bits[i] = 0ull;
        if (t == 0) __hip_atomic_store(p.ctl, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

static int cbp_num_cus() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
            cus = n;
        else
            cus = 256;
    }
    return cus;
}

}  // namespace cbp
using namespace cbp;

extern "C" {

// Layers the row-pair kernel takes: fp32, at most 4 input and 16 output channels, a square odd filter of 3, 5 or 7,
// a change mask of at most CBP_MAXWORDS words.
int cbinfer_rowpairs_supported(int C, int K, int kH, int kW, int H, int W) {
    if (C < 1 || C > 4 || K < 1 || K > 16 || kH != kW || (kH != 3 && kH != 5 && kH != 7) || H < 1 || W < 1) return 0;
    if (cbinfer_mask_words(H, W) > CBP_MAXWORDS || (long)16 * H * W * 4 >= (1l << 30)) return 0;
    return 1;
}

int cbinfer_conv_changed_rowpairs(const float* state, uint64_t* bits, int32_t* ctl, uint64_t* maskCopy,
                                  const void* prepared, const float* bias, float* output, int C, int H, int W, int K,
                                  int kH, int kW, int relu, const cbNextDetect* next, cbStream_t stream) {
    CB_REQUIRE(state && bits && ctl && prepared && output);
    if (!cbinfer_rowpairs_supported(C, K, kH, kW, H, W)) return CB_ERR_UNSUPPORTED;
    PairParams p;
    p.state = state, p.wq = (const float*)prepared, p.bias = bias, p.out = output;
    p.bits = (unsigned long long*)bits, p.ctl = ctl, p.maskCopy = (unsigned long long*)maskCopy;
    p.C = C, p.H = H, p.W = W, p.K = K, p.relu = relu;
    p.wpr = cbinfer_mask_words_per_row(W), p.MW = (int)cbinfer_mask_words(H, W);
    p.units = ((H + 1) / 2) * p.wpr;
    p.next.state = nullptr;
    if (next && next->state) {
        // the layer behind the 2x2/stride-2 pool: K channels at (H/2 or (H+1)/2) x (W/2 or (W+1)/2), split-state form
        CB_REQUIRE(next->splitState && next->frameMasks);
        CB_REQUIRE((next->H == H / 2 || next->H == (H + 1) / 2) && (next->W == W / 2 || next->W == (W + 1) / 2));
        if (K != 16 || !cbs::cbs_supported(K, 1, next->kH, next->kW)) return CB_ERR_UNSUPPORTED;
        const cbs::CbsGeom g = cbs::cbs_geom(K, next->H, next->W, next->kH, next->kW);
        p.next.state = next->state, p.next.S = (char*)next->splitState;
        p.next.masks = (unsigned long long*)next->frameMasks, p.next.rangeFlag = next->rangeFlag;
        p.next.H2 = next->H, p.next.W2 = next->W, p.next.wpr2 = cbinfer_mask_words_per_row(next->W);
        p.next.kHH = (next->kH - 1) / 2, p.next.kWH = (next->kW - 1) / 2;
        p.next.Wp = g.Wp, p.next.rec = g.rec, p.next.padY = g.padY, p.next.padXL = g.padXL;
        p.next.th = next->threshold;
    }
    int grid = 2 * cbp_num_cus();
    if (grid > p.units) grid = p.units;
    if (kH == 7)
        hipLaunchKernelGGL((cbp_rowpair_kernel<7, 7>), dim3(grid), dim3(CBP_NT), 0, (hipStream_t)stream, p);
    else if (kH == 5)
        hipLaunchKernelGGL((cbp_rowpair_kernel<5, 5>), dim3(grid), dim3(CBP_NT), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL((cbp_rowpair_kernel<3, 3>), dim3(grid), dim3(CBP_NT), 0, (hipStream_t)stream, p);
    return cb_launch_status();
}

// One frame of a feedback-mode CBConv2d (conv2d.py:178-259) on the row-pair kernel: detection with feedback refresh,
// then the contraction -- which, with `next`, also is the next layer's pooled detection.
int cbinfer_cbconv2d_forward_rowpairs(const float* input, float* prevInput, float* prevOutput, uint64_t* bits,
                                      int32_t* ctl, uint64_t* maskCopy, const void* prepared, const float* bias, int C,
                                      int H, int W, int K, int kH, int kW, float threshold, int relu,
                                      const cbNextDetect* next, cbStream_t stream) {
    CB_REQUIRE(input && prevInput && prevOutput && bits);
    const int st = cbinfer_change_detection_bits(input, prevInput, bits, W, H, C, (kH - 1) / 2, (kW - 1) / 2, threshold,
                                                 1, CB_F32, stream);
    if (st != CB_OK) return st;
    return cbinfer_conv_changed_rowpairs(prevInput, bits, ctl, maskCopy, prepared, bias, prevOutput, C, H, W, K, kH, kW,
                                         relu, next, stream);
}

}  // extern "C"
