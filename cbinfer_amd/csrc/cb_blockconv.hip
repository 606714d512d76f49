// Patch-staged contraction for the wide layers (16->64 and 64->256 7x7 of the scene-labeling network):
// gather -> MFMA -> bias/ReLU -> scatter for the changed pixels of a unit of R rows x 64 columns (R words of
// the change bit mask) and 64 output channels per workgroup, fp32 tensors, bf16x3 split arithmetic (CB_F32S).
//
// What bounded the list kernel (cb_conv.hip) after the move to bf16x3 was not the matrix pipe but the feed:
// kH*kW im2col gathers per input value through the texture path and, per 32-k stage, a full LDS write +
// barrier + read round trip of both operands (ablations in DESIGN.md).  Here
//   * the input rows under the unit are staged ONCE per 8-channel chunk -- (R + kH - 1) x (64 + kW) pixels,
//     coalesced row loads, split into three bf16 planes on the way in -- and every tap of every changed
//     pixel reads its B fragment (8 channels = 16 bytes) straight from that patch: ds_read_b128 at
//     `lane base + immediate`; the LDS write traffic per MAC drops by ~kH*kW;
//   * the A operand (weights, pre-split and pre-arranged in MFMA fragment order by cbinfer_blockconv_prep_
//     weights) never touches LDS: three 16-byte loads per lane and k-step, reused for all pixel tiles of the
//     unit, prefetched one step ahead;
//   * v_mfma_f32_16x16x32_bf16: 16 output channels x 16 pixels x (4 horizontally adjacent taps x 8 channels)
//     per instruction, six per step (hi.hi, hi.mid, mid.hi, mid.mid, hi.lo, lo.hi -- f32-level accuracy,
//     see CB_F32S); pixel tiles are runs of up to 16 changed pixels of one row, so a 14-pixel dilated run
//     wastes 12 % where a 32-wide tile would waste 56 %;
//   * waves 0-7 multiply -- one 16-channel tile per pair of waves, which take alternate k-steps, so that
//     one wave's LDS reads hide under the other's MFMAs on the same SIMD; their partial sums meet in LDS at
//     the end in a fixed order -- and waves 8-11 stage the next channel chunk into the other patch buffer
//     meanwhile (loads, splits, LDS writes): one barrier per chunk.
// k order: (channel chunk, ky, kx/4, [kx%4 = lane quarter], 8 channels); kW is padded to a multiple of 4
// with zero weights (7 -> 8: 14 % more MFMA work on a pipe that is no longer the bottleneck).
// Mask protocol: as cb_rowconv.hip (single mask, maskCopy, per-unit arrival counter over blockIdx.z).
#include <stdlib.h>

#include "cb_common.h"

namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct BlkParams {
    const float* state;     // [C,H,W]
    const uint4* wb;        // prepared weights [mtile][chunk][step][plane][lane] x 16 B
    const float* bias;
    float* out;             // [K,H,W]
    unsigned long long* bits;
    int* arrive;
    unsigned long long* maskCopy;
    int C, H, W, K, kH, kW;
    int CH, KXQ, SPC;       // channel chunks of 8, kW/4 rounded up, k-steps per chunk (kH * KXQ)
    int PR, PC, PLANE;      // patch rows (R + kH - 1), columns (64 + 4 KXQ), bytes per bf16 plane (PR*PC*16)
    int MT;                 // 16-row output-channel tiles
    int pcMagic;            // ceil(65536 / PC): q / PC == (q * pcMagic) >> 16 for q < PR*PC
    int relu, wpr;
    int accumulate;         // fine-grained frame: out += W * delta (state = delta tensor), no bias
    float* reluOut;         // ... and optionally relu(out) into a second plane set
};

__device__ __forceinline__ void cb_split3b(float x, unsigned& hi, unsigned& mid, unsigned& lo) {
    const __bf16 h = (__bf16)x;
    const float r1 = x - (float)h;
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    const __bf16 l = (__bf16)r2;
    hi = __builtin_bit_cast(unsigned short, h);
    mid = __builtin_bit_cast(unsigned short, m);
    lo = __builtin_bit_cast(unsigned short, l);
}

__device__ __forceinline__ int cb_nth_bit_b(unsigned long long w, int r) {
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

__device__ __forceinline__ unsigned long long cb_uniform64(unsigned long long v) {
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
           (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}

#ifdef CB_BLK_STAMP
// diagnostic build only (make EXTRA=-DCB_BLK_STAMP): per-workgroup time stamps (100 MHz constant clock);
// slots 0..15: multiplier wave 0 (entry, prologue done, after barrier c = 2+c, loop done, end), 16..31: stager
__device__ unsigned long long cb_blk_stamps[2048 * 32];
#define CB_BSTAMP(who, i)                                                                            \
    do {                                                                                             \
        const unsigned bid = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;         \
        if (threadIdx.x == ((who) ? 512 : 0) && bid < 2048 && (i) < 16)                              \
            cb_blk_stamps[bid * 32 + (who) * 16 + (i)] = __builtin_amdgcn_s_memrealtime();           \
    } while (0)
#else
#define CB_BSTAMP(who, i)
#endif
#define CB_BLK_THREADS 768   // at most: 8 multiplier waves (4 channel tiles x 2 k-halves) + 4 stager waves
#define CB_BLK_SPT 3         // patch pixels per stager thread (PR*PC <= 768)

// MG = 16-channel tiles per workgroup (2 MG multiplier waves + 4 stager waves): 4 for wide layers; 2 for layers
// of at most 64 output channels, whose units are too few to fill the chip with four tiles each
template <int R, int MG>
__global__ __launch_bounds__(64 * (2 * MG + 4)) void cb_blockconv_kernel(BlkParams p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];   // patch [2 buffers][3 planes][PR][PC] x 16 B
    constexpr int NTMAX = 4 * R;
    CB_BSTAMP(0, 0);
    const int tx = blockIdx.x, y0 = blockIdx.y * R, mg = blockIdx.z;
    unsigned long long word[R];
    bool any = false;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        word[r] = y0 + r < p.H ? cb_uniform64(p.bits[(long)(y0 + r) * p.wpr + tx]) : 0ull;
        any |= word[r] != 0ull;
    }
    const int t = threadIdx.x;
    if (!any) {   // nothing changed in this unit
        if (mg == 0 && t < R && y0 + t < p.H) p.maskCopy[(long)(y0 + t) * p.wpr + tx] = 0ull;
        return;
    }
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    int arrived = 0;
    if (t == 0) {
        if (mg == 0) {
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (y0 + r < p.H) p.maskCopy[(long)(y0 + r) * p.wpr + tx] = word[r];
        }
        if (gridDim.z > 1)
            arrived = __hip_atomic_fetch_add(p.arrive + (long)y0 * p.wpr + tx, 1, __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT);
    }
    const int ph = (p.kH - 1) / 2, pw = (p.kW - 1) / 2;
    const int HW = p.H * p.W;
    const int bufBytes = 3 * p.PLANE;

    // pixel tiles of the unit: tile i = 16 consecutive set bits of one row's word
    int nT[R], nTt = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        nT[r] = (__popcll(word[r]) + 15) >> 4;
        nTt += nT[r];
    }

    if (wave >= 2 * MG) {
        // ================= stager waves: patch of chunk c -> buffer c & 1, one chunk ahead of the multipliers
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc((void*)p.state, 0, p.C * HW * 4, 0x00020000);
        const int x0 = tx * 64 - pw;
        // the patch's PR x PC pixels flattened over the 256 stager threads, CB_BLK_SPT per thread; a thread
        // gathers the 8 channels of its pixels (all loads of a chunk in flight at once), splits and writes them
        const int ts = t - 128 * MG;
        int pixoff[CB_BLK_SPT], dstoff[CB_BLK_SPT];
#pragma unroll
        for (int i = 0; i < CB_BLK_SPT; ++i) {
            const int q = ts + 256 * i;
            const int pr = (q * p.pcMagic) >> 16, pc = q - pr * p.PC;    // q / PC, q % PC (checked on the host)
            const int yy = y0 - ph + pr, x = x0 + pc;
            const bool ok = q < p.PR * p.PC && yy >= 0 && yy < p.H && x >= 0 && x < p.W;
            pixoff[i] = ok ? yy * p.W + x : -1;
            dstoff[i] = q < p.PR * p.PC ? q * 16 : -1;
        }
        // barrier c = "chunk c is staged AND the multipliers are done with chunk c-1": chunk c+1 is staged into
        // the buffer chunk c-1 occupied while chunk c is being multiplied.  The loads of chunk c+1 are issued
        // BEFORE chunk c is split and written (two register sets), so their latency hides under that work.
        float va[CB_BLK_SPT][8], vb[CB_BLK_SPT][8];
        auto gather = [&](float (&v)[CB_BLK_SPT][8], int c) {
#pragma unroll
            for (int i = 0; i < CB_BLK_SPT; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int ch = 8 * c + e;
                    // an invalid element gets an out-of-range offset: the buffer load returns 0 for it
                    v[i][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                            rsrc, (pixoff[i] >= 0 && ch < p.C) ? (ch * HW + pixoff[i]) * 4 : (1 << 30),
                                                            0, 0));
                }
        };
        // x = hi + mid + lo: hi and mid by truncation (a mask), lo rounded to nearest -- |x - sum| <= 2^-24 |x|
        auto put = [&](const float (&v)[CB_BLK_SPT][8], int c) {
            char* buf = lds + (c & 1) * bufBytes;
#pragma unroll
            for (int i = 0; i < CB_BLK_SPT; ++i) {
                if (dstoff[i] >= 0) {
                    unsigned h[8], m[8], l[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const unsigned xb = __builtin_bit_cast(unsigned, v[i][e]);
                        const unsigned hb = xb & 0xffff0000u;
                        const float r1 = v[i][e] - __builtin_bit_cast(float, hb);
                        const unsigned mb = __builtin_bit_cast(unsigned, r1) & 0xffff0000u;
                        const float r2 = r1 - __builtin_bit_cast(float, mb);
                        h[e] = hb;
                        m[e] = mb;
                        l[e] = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)r2) << 16;
                    }
                    char* dst = buf + dstoff[i];
#define CB_PK(a, b) (((a) >> 16) | (b))
                    *(uint4*)(dst) = make_uint4(CB_PK(h[0], h[1]), CB_PK(h[2], h[3]), CB_PK(h[4], h[5]), CB_PK(h[6], h[7]));
                    *(uint4*)(dst + p.PLANE) = make_uint4(CB_PK(m[0], m[1]), CB_PK(m[2], m[3]), CB_PK(m[4], m[5]), CB_PK(m[6], m[7]));
                    *(uint4*)(dst + 2 * p.PLANE) = make_uint4(CB_PK(l[0], l[1]), CB_PK(l[2], l[3]), CB_PK(l[4], l[5]), CB_PK(l[6], l[7]));
#undef CB_PK
                }
            }
        };
        CB_BSTAMP(1, 0);
        gather(va, 0);
        for (int c = 0; c < p.CH; c += 2) {
            if (c + 1 < p.CH) gather(vb, c + 1);
            put(va, c);
            CB_BSTAMP(1, 1 + c);
            __syncthreads();   // barrier c (the multipliers run CH of them as well)
            if (c + 1 < p.CH) {
                if (c + 2 < p.CH) gather(va, c + 2);
                put(vb, c + 1);
                CB_BSTAMP(1, 2 + c);
                __syncthreads();   // barrier c + 1
            }
        }
        __syncthreads();   // (the multipliers' two reduction barriers)
        __syncthreads();
    } else {
        // ================= multiplier waves: one 16-channel tile each
        const int mt = MG * mg + wave % MG, kh = wave / MG;   // channel tile, k-half (steps kh, kh+2, ...)
        const bool active = mt < p.MT;
        const int kg = lane >> 4;
        // per tile: LDS byte offset of this lane's pixel (patch row = row in unit, column = x in word) + quarter
        int base[NTMAX], tpix[NTMAX];
        {
            int i = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int pc = __popcll(word[r]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (j < nT[r]) {
                        const int n = 16 * j + (lane & 15);
                        const int xl = cb_nth_bit_b(word[r], n < pc ? n : 0);
                        // tiles are stored compacted: slot i (uniform) <- (r, j)
#pragma unroll
                        for (int s = 0; s < NTMAX; ++s)
                            if (s == i) {
                                base[s] = ((r * p.PC) + xl + kg) * 16;
                                tpix[s] = n < pc ? (y0 + r) * p.W + tx * 64 + xl : -1;
                            }
                        ++i;
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < NTMAX; ++s)
                if (s >= nTt) {
                    base[s] = 0;
                    tpix[s] = -1;
                }
        }
        floatx4 acc[NTMAX];
#pragma unroll
        for (int s = 0; s < NTMAX; ++s) acc[s] = floatx4{0.f, 0.f, 0.f, 0.f};
        float bv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
            bv[r] = p.bias ? p.bias[min(min(mt, p.MT - 1) * 16 + 4 * kg + r, p.K - 1)] : 0.f;

        CB_BSTAMP(0, 1);
        const uint4* Aw = p.wb + (long)min(mt, p.MT - 1) * p.CH * p.SPC * 192 + lane;   // step: 3 planes x 64 lanes
        // Weight fragments: a ring of four register sets, the loads of step s+4 are issued while step s is
        // multiplied (a step is only 12..48 MFMAs = 0.1..0.4 us: one step of look-ahead does not cover an L2
        // hit); they do not depend on the staging, so the ring runs straight across the chunk barriers.
        uint4 a0[3], a1[3], a2[3], a3[3];
        auto loadA = [&](uint4 (&dst)[3], int step) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) dst[pl] = Aw[((long)step * 3 + pl) * 64];
        };
        // this wave's steps: sin = kh, kh + 2, ... of every chunk; (c, sin) walks them, (cA, sA) runs four of
        // them ahead for the weight loads
        int cA = 0, sA = kh;
        auto nextA = [&]() {   // global step index of the look-ahead walker, then advance it (clamped at the end)
            const int g = min(cA, p.CH - 1) * p.SPC + (cA < p.CH ? sA : kh);
            sA += 2;
            if (sA >= p.SPC) {
                sA = kh;
                ++cA;
            }
            return g;
        };
        loadA(a0, nextA());
        loadA(a1, nextA());
        loadA(a2, nextA());
        loadA(a3, nextA());
        auto mul = [&](const uint4 (&a)[3], const char* pat, int off) {
            const bf16x8 ah = __builtin_bit_cast(bf16x8, a[0]), am = __builtin_bit_cast(bf16x8, a[1]),
                         al = __builtin_bit_cast(bf16x8, a[2]);
#pragma unroll
            for (int s = 0; s < NTMAX; s += 2) {
                if (s < nTt) {   // (uniform) two tiles at a time: their MFMA chains interleave
                    const char* q0 = pat + base[s] + off;
                    const char* q1 = pat + base[s + 1 < NTMAX ? s + 1 : s] + off;
                    const bf16x8 bh0 = *(const bf16x8*)q0, bm0 = *(const bf16x8*)(q0 + p.PLANE),
                                 bl0 = *(const bf16x8*)(q0 + 2 * p.PLANE);
                    const bf16x8 bh1 = *(const bf16x8*)q1, bm1 = *(const bf16x8*)(q1 + p.PLANE),
                                 bl1 = *(const bf16x8*)(q1 + 2 * p.PLANE);
                    floatx4 c0 = acc[s], c1 = acc[s + 1 < NTMAX ? s + 1 : s];
                    // smallest terms first
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh0, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh1, c1, 0, 0, 0);
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl0, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl1, c1, 0, 0, 0);
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm0, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm1, c1, 0, 0, 0);
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh0, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh1, c1, 0, 0, 0);
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm0, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm1, c1, 0, 0, 0);
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh0, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh1, c1, 0, 0, 0);
                    acc[s] = c0;
                    if (s + 1 < NTMAX) acc[s + 1] = c1;
                }
            }
        };
        if (!active) {   // a wave without a channel tile only keeps the barrier count
            for (int c = 0; c < p.CH; ++c) __syncthreads();
        } else {
            int c = 0, sin = kh, ky = 0, q = kh;   // sin = ky * KXQ + q; this wave advances by two steps
            while (q >= p.KXQ) {
                q -= p.KXQ;
                ++ky;
            }
            const int own = p.CH * ((p.SPC - kh + 1) >> 1);   // steps of this wave
            const char* pat = lds;
            bool fresh = true;                                // first own step of a chunk: wait for its barrier
#define CB_BLK_STEP(A, J)                                                                           \
    if ((J) < own) {                                                                                \
        if (fresh) {                                                                                \
            CB_BSTAMP(0, 2 + c);                                                                    \
            __syncthreads(); /* barrier c: chunk c staged, every multiplier done with chunk c-1 */  \
            pat = lds + (c & 1) * bufBytes;                                                         \
            fresh = false;                                                                          \
        }                                                                                           \
        mul(A, pat, (ky * p.PC + 4 * q) * 16);                                                      \
        sin += 2;                                                                                   \
        q += 2;                                                                                     \
        while (q >= p.KXQ) {                                                                        \
            q -= p.KXQ;                                                                             \
            ++ky;                                                                                   \
        }                                                                                           \
        if (sin >= p.SPC) {                                                                         \
            sin = kh;                                                                               \
            ky = 0;                                                                                 \
            q = kh;                                                                                 \
            while (q >= p.KXQ) {                                                                    \
                q -= p.KXQ;                                                                         \
                ++ky;                                                                               \
            }                                                                                       \
            ++c;                                                                                    \
            fresh = true;                                                                           \
        }                                                                                           \
    }                                                                                               \
    /* the re-arming loads sit OUTSIDE every branch: with loads on both sides of a join the compiler's */ \
    /* vmcnt bookkeeping falls back to draining the queue (vmcnt(0)) before each use                   */ \
    loadA(A, nextA());
            for (int j = 0; j < own; j += 4) {
                CB_BLK_STEP(a0, j)
                CB_BLK_STEP(a1, j + 1)
                CB_BLK_STEP(a2, j + 2)
                CB_BLK_STEP(a3, j + 3)
            }
#undef CB_BLK_STEP
        }
        // ---- the two k-halves of a channel tile meet in LDS (the patch buffers are free now) ----------------
        __syncthreads();
        if (active && kh == 1) {
#pragma unroll
            for (int s = 0; s < NTMAX; ++s)
                if (s < nTt) *(floatx4*)(lds + (((wave % MG) * NTMAX + s) * 64 + lane) * 16) = acc[s];
        }
        __syncthreads();
        if (active && kh == 0) {
#pragma unroll
            for (int s = 0; s < NTMAX; ++s)
                if (s < nTt) acc[s] += *(const floatx4*)(lds + (((wave % MG) * NTMAX + s) * 64 + lane) * 16);
        }
        CB_BSTAMP(0, 14);
        // ---- bias / ReLU / scatter: D tile col = lane % 16 (pixel), rows 4 (lane/16) + r (channel) ----------
        if (active && kh == 0) {
#pragma unroll
            for (int s = 0; s < NTMAX; ++s) {
                if (s < nTt && tpix[s] >= 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int m = mt * 16 + 4 * kg + r;
                        if (m < p.K) {
                            float v = acc[s][r] + bv[r];
                            if (p.accumulate) {
                                v += p.out[(long)m * HW + tpix[s]];
                                p.out[(long)m * HW + tpix[s]] = v;
                                if (p.reluOut) p.reluOut[(long)m * HW + tpix[s]] = v <= 0.f ? 0.f : v;
                            } else {
                                if (p.relu) v = v <= 0.f ? 0.f : v;
                                p.out[(long)m * HW + tpix[s]] = v;
                            }
                        }
                    }
                }
            }
        }
    }
    CB_BSTAMP(0, 15);
    // the last consumer of the unit zeroes its mask words (and the counter) for the next frame's detection
    if (t == 0) {
        if (gridDim.z == 1 || arrived == (int)gridDim.z - 1) {
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (y0 + r < p.H) p.bits[(long)(y0 + r) * p.wpr + tx] = 0ull;
            if (gridDim.z > 1)
                __hip_atomic_store(p.arrive + (long)y0 * p.wpr + tx, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

struct BlkGeom {
    int CH, KXQ, SPC, R, PR, PC, PLANE, MT, MG, ZM;
    long ldsBytes, wbBytes;
};
BlkGeom blk_geom(int C, int K, int kH, int kW) {
    BlkGeom g;
    g.CH = (C + 7) / 8;
    g.KXQ = (kW + 3) / 4;
    g.SPC = kH * g.KXQ;
    g.MT = (K + 15) / 16;
    g.MG = 2;   // two tiles per workgroup, two workgroups per CU: 65 vs 68 us (64->256, 27 %), 117 vs 131 (100 %)
    {
        static int mgo = -1;
        if (mgo < 0) {
            const char* e = getenv("CBINFER_BLK_MG");   // tuning aid
            mgo = e ? atoi(e) : 0;
        }
        if (mgo == 2 || mgo == 4) g.MG = mgo;
    }
    g.ZM = (g.MT + g.MG - 1) / g.MG;
    g.R = 2;   // measured on 16->64 @160x240, 18 % changed: 18.7 us with two-row units, 22 with one-row units
    {
        static int rr = -1;
        if (rr < 0) {
            const char* e = getenv("CBINFER_BLK_R");   // tuning aid
            rr = e ? atoi(e) : 0;
        }
        if (rr == 1 || rr == 2) g.R = rr;
    }
    g.PR = g.R + kH - 1;
    g.PC = 64 + 4 * g.KXQ;
    g.PLANE = g.PR * g.PC * 16;
    g.ldsBytes = 2l * 3 * g.PLANE;
    // (the same LDS holds the k-halves' partial sums at the end: 4 tiles x 4R pixel tiles x 64 lanes x 16 B)
    if (g.ldsBytes < 4l * 4 * g.R * 1024) g.ldsBytes = 4l * 4 * g.R * 1024;
    g.wbBytes = (long)g.ZM * g.MG * g.CH * g.SPC * 3 * 64 * 16;
    return g;
}

// weights [K,C,kH,kW] f32 -> [mtile][chunk][step = ky*KXQ + q][plane][lane] x (8 bf16): lane = (m % 16, kg),
// element e = channel 8 chunk + e at tap (ky, 4 q + kg); zero outside K / C / kW
__global__ __launch_bounds__(256) void cb_blockconv_prep_kernel(const float* __restrict__ w,
                                                               unsigned short* __restrict__ wb, int K, int C,
                                                               int kH, int kW, int CH, int KXQ, int MTP) {
    const int SPC = kH * KXQ;
    const long total = (long)MTP * CH * SPC * 64 * 8;   // one thread per (mtile, chunk, step, lane, e)
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int e = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
    long r = idx >> 9;
    const int step = (int)(r % SPC);
    r /= SPC;
    const int c = (int)(r % CH), mt = (int)(r / CH);
    const int ky = step / KXQ, q = step - ky * KXQ;
    const int m = 16 * mt + (lane & 15), kx = 4 * q + (lane >> 4), ch = 8 * c + e;
    const float v = (m < K && kx < kW && ch < C) ? w[(((long)m * C + ch) * kH + ky) * kW + kx] : 0.f;
    unsigned hi, mid, lo;
    cb_split3b(v, hi, mid, lo);
    unsigned short* dst = wb + ((((long)mt * CH + c) * SPC + step) * 3 * 64 + lane) * 8 + e;
    dst[0] = (unsigned short)hi;
    dst[64 * 8] = (unsigned short)mid;
    dst[2 * 64 * 8] = (unsigned short)lo;
}

}  // namespace

extern "C" {

// Layers the block kernel takes: fp32 tensors, CB_F32S arithmetic; a double-buffered patch within 64 KB of
// LDS; at least 16 output channels' worth of work per wave.
int cbinfer_blockconv_supported(int C, int K, int kH, int kW) {
    if (C <= 0 || K <= 0 || kH <= 1 || kW <= 1 || kH > 15 || kW > 16) return 0;   // (>= 2 steps per chunk)
    const BlkGeom g = blk_geom(C, K, kH, kW);
    if (g.PR * g.PC > 256 * CB_BLK_SPT) return 0;
    const int magic = (65536 + g.PC - 1) / g.PC;
    for (int q = 0; q < 256 * CB_BLK_SPT; ++q)
        if (((q * magic) >> 16) != q / g.PC) return 0;
    return g.ldsBytes <= 64 * 1024 ? 1 : 0;
}

long cbinfer_blockconv_prepared_bytes(int C, int K, int kH, int kW) { return blk_geom(C, K, kH, kW).wbBytes; }

int cbinfer_blockconv_prep_weights(const float* weight, void* prepared, int K, int C, int kH, int kW,
                                   cbStream_t stream) {
    CB_REQUIRE(weight && prepared);
    if (!cbinfer_blockconv_supported(C, K, kH, kW)) return CB_ERR_UNSUPPORTED;
    const BlkGeom g = blk_geom(C, K, kH, kW);
    const long total = (long)g.ZM * g.MG * g.CH * g.SPC * 64 * 8;
    hipLaunchKernelGGL(cb_blockconv_prep_kernel, dim3(cb_div_up(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       weight, (unsigned short*)prepared, K, C, kH, kW, g.CH, g.KXQ, g.ZM * g.MG);
    return cb_launch_status();
}

// Same buffers and protocol as cbinfer_conv_changed_rows; arrive needs cbinfer_mask_words(H,W) int32.
static int cb_blocks_launch(const float* state, uint64_t* bits, int32_t* arrive, uint64_t* maskCopy,
                            const void* prepared, const float* bias, float* output, int C, int H, int W,
                            int K, int kH, int kW, int relu, int accumulate, float* reluOut, cbStream_t stream) {
    CB_REQUIRE(state && bits && arrive && maskCopy && prepared && (bias || accumulate) && output && H > 0 && W > 0);
    if (!cbinfer_blockconv_supported(C, K, kH, kW)) return CB_ERR_UNSUPPORTED;
    if ((long)C * H * W * 4 >= (1l << 30) || H > 65535) return CB_ERR_UNSUPPORTED;
    const BlkGeom g = blk_geom(C, K, kH, kW);
    BlkParams p;
    p.state = state;
    p.wb = (const uint4*)prepared;
    p.bias = bias;
    p.out = output;
    p.bits = (unsigned long long*)bits;
    p.arrive = arrive;
    p.maskCopy = (unsigned long long*)maskCopy;
    p.C = C;
    p.H = H;
    p.W = W;
    p.K = K;
    p.kH = kH;
    p.kW = kW;
    p.CH = g.CH;
    p.KXQ = g.KXQ;
    p.SPC = g.SPC;
    p.PR = g.PR;
    p.PC = g.PC;
    p.PLANE = g.PLANE;
    p.MT = g.MT;
    p.pcMagic = (65536 + g.PC - 1) / g.PC;
    p.relu = relu;
    p.accumulate = accumulate;
    p.reluOut = reluOut;
    p.wpr = cbinfer_mask_words_per_row(W);
    dim3 grid(p.wpr, (H + g.R - 1) / g.R, g.ZM), block(64 * (2 * g.MG + 4));
    if (g.R == 2 && g.MG == 4)
        hipLaunchKernelGGL((cb_blockconv_kernel<2, 4>), grid, block, (size_t)g.ldsBytes, (hipStream_t)stream, p);
    else if (g.R == 2)
        hipLaunchKernelGGL((cb_blockconv_kernel<2, 2>), grid, block, (size_t)g.ldsBytes, (hipStream_t)stream, p);
    else if (g.MG == 4)
        hipLaunchKernelGGL((cb_blockconv_kernel<1, 4>), grid, block, (size_t)g.ldsBytes, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL((cb_blockconv_kernel<1, 2>), grid, block, (size_t)g.ldsBytes, (hipStream_t)stream, p);
    return cb_launch_status();
}

int cbinfer_conv_changed_blocks(const float* state, uint64_t* bits, int32_t* arrive, uint64_t* maskCopy,
                                const void* prepared, const float* bias, float* output, int C, int H, int W,
                                int K, int kH, int kW, int relu, cbStream_t stream) {
    CB_REQUIRE(bias != nullptr);
    return cb_blocks_launch(state, bits, arrive, maskCopy, prepared, bias, output, C, H, W, K, kH, kW, relu, 0,
                            nullptr, stream);
}

int cbinfer_conv_accumulate_blocks(const float* delta, uint64_t* bits, int32_t* arrive, uint64_t* maskCopy,
                                   const void* prepared, float* output, float* reluOut, int C, int H, int W,
                                   int K, int kH, int kW, cbStream_t stream) {
    return cb_blocks_launch(delta, bits, arrive, maskCopy, prepared, nullptr, output, C, H, W, K, kH, kW, 0, 1,
                            reluOut, stream);
}

}  // extern "C"

#ifdef CB_BLK_STAMP
extern "C" int cbinfer_debug_blk_stamps(void* host, long bytes, int clear) {
    if (clear) {
        void* d = nullptr;
        if (hipGetSymbolAddress(&d, HIP_SYMBOL(cb_blk_stamps)) != hipSuccess) return -1;
        return (int)hipMemset(d, 0, sizeof(cb_blk_stamps));
    }
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(cb_blk_stamps), (size_t)bytes);
}
#endif
