// Pieces of the split-state design (cb_split.hip) shared with the kernels that refresh a split state on behalf of
// the NEXT layer (cb_rowpair.hip): the geometry of the pre-split pixel-major state copy and the operand splits --
// f16 PAIRS (round 3: 22-23 significant bits per operand, three products) and bf16 TRIPLES (round 5, "x3": every
// f32 value exactly, six products: the f32-EQUIVALENT form, conv2d_cg.py:342-349's sgemm operands).
#pragma once
#include "cb_common.h"

namespace cbs {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CBS_MAXSEQ CBINFER_SPLIT_MAX_SEQUENCES
#define CBS_XSCALE 0.0625f            // activations are stored as x * 2^-4 (range up to 2^20, see header)
#define CBS_XSCALE_INV 16.f
#define CBS_LO 2048.f                 // lo terms carry a factor 2^11
#define CBS_F16_MAX 65504.f
#define CBS_SPAD 4096                 // bytes in front of the records of a split state (see cbs_dma16: negative offsets)
#define CBS_PRE_BIG 1536              // mask words of all sequences of a launch: 128-row tile
#define CBS_PRE_X3 1024               // ... 128-row tile, bf16-triple form, mask words kept in LDS beside its 144 KB ring
#define CBS_PRE_SMALL 1280            // ... 64-row tile, two workgroups per CU
#define CBS_PRE_MID2 2600             // ... 64-row tile, two workgroups per CU, mask words not kept in LDS
#define CBS_PRE_MID 5120              // ... 64-row tile, one workgroup per CU
#define CBS_CHUNKS 4                  // canonical k-chunks of a deep contraction (see cbs_conv_kernel)

// ---------------------------------------------------------------------------------------------------
// geometry of the split state / stage list (host and device)
// ---------------------------------------------------------------------------------------------------
struct CbsGeom {
    int C, G, H, W, kH, kW;
    int padY, padXL, padXR, Wp, Hp, rec;
    int pair, kWs, nStages;
    int dummyBase;      // byte offset of the top-left tap record of the all-zero dummy pixel
    int planes;         // 16-bit terms per value: 2 (f16 pairs) or 3 (bf16 triples); a 16-channel group of a record
                        // is [plane][16] = 32 planes bytes, a stage (32 k) 64 planes bytes of a pixel's record
};

__host__ __device__ inline CbsGeom cbs_geom(int C, int H, int W, int kH, int kW, int planes = 2) {
    CbsGeom g;
    g.planes = planes;
    g.C = C, g.G = C / 16, g.H = H, g.W = W, g.kH = kH, g.kW = kW;
    g.padY = kH / 2, g.padXL = kW / 2;
    g.pair = g.G == 1;                      // 16 channels: a stage is two x-adjacent taps
    g.kWs = g.pair ? (kW + 1) / 2 : kW;     // stage columns per filter row
    g.padXR = g.pair ? 2 * g.kWs - 1 - g.padXL : kW / 2;
    g.Wp = W + g.padXL + g.padXR;
    g.Hp = H + 4 * g.padY + 1;              // image + border, then 2 padY + 1 zero rows for the dummy pixel
    g.rec = g.G * 32 * planes;
    g.nStages = g.pair ? kH * g.kWs : kH * kW * (g.G / 2);
    g.dummyBase = ((H + 2 * g.padY) * g.Wp) * g.rec;
    return g;
}

inline bool cbs_supported(int C, int K, int kH, int kW) {
    return (C == 16 || C == 32 || C == 64) && K >= 1 && K <= 1024 && (kH & 1) && (kW & 1) && kH <= 15 && kW <= 15 &&
           cbs_geom(C, 64, 64, kH, kW).nStages >= 4;
}
inline int cbs_bm(int K) { return K <= 64 ? 64 : 128; }
inline int cbs_kp(int K) { const int bm = cbs_bm(K); return (K + bm - 1) / bm * bm; }

// x (already scaled) = hi + lo / 2^11 with f16 hi, lo, both rounded to nearest: |x - hi - lo / 2^11| <= 2^-23 |x|
// (the residual of a round-to-nearest hi is at most half an ulp of hi, and lo keeps 11 bits of it).  f16 SUBNORMAL
// terms are kept: v_mfma_f32_32x32x16_f16 honours subnormal operands on gfx950 (tools/micro/mfma_denorm.hip: every
// product exact), so small values keep a relative bound down to 2^-24 * 2^-11 -- rounds 1-3 dropped such terms to zero
// (not knowing what the matrix unit does with them) and carried an absolute floor of 2^-21 per activation instead.
__device__ __forceinline__ void cbs_split(float x, _Float16& hi, _Float16& lo) {
    const _Float16 h = (_Float16)x;
    const float r = (x - (float)h) * CBS_LO;          // exact difference, exact scaling
    hi = h;
    lo = (_Float16)r;
}

// x = b0 + b1 + b2 EXACTLY, bf16 terms rounded to nearest (8 significant bits each: the residual of a round-to-nearest
// 8-bit head of a 24-bit value has at most 16 bits, that of the second term at most 8).  bf16 has f32's exponent range:
// no scale, no range flag.  A finite |x| above the largest bf16 (3.3895e38 .. FLT_MAX) would round its head to inf: the
// head is taken by truncation there, the residual (< 2^-7 |x|) is finite and splits as usual.
// Non-finite x: b0 carries it, the other terms are zero (inf - inf would make them NaN).  The products of such an operand
// with the residual terms of a weight are inf * 0 = NaN or infinities of both signs: an output that depends on a
// non-finite state value is NaN where the reference's f32 chain yields +-inf or NaN (either way: not a number to use).
__device__ __forceinline__ void cbs_split3(float x, __bf16& b0, __bf16& b1, __bf16& b2) {
    __bf16 h = (__bf16)x;
    const bool fin = fabsf(x) < INFINITY;      // (false for NaN too)
    if (fin && !(fabsf((float)h) < INFINITY))
        h = __builtin_bit_cast(__bf16, (unsigned short)(__builtin_bit_cast(unsigned, x) >> 16));
    const float r1 = x - (float)h;
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    b0 = h;
    b1 = fin ? m : (__bf16)0.f;
    b2 = fin ? (__bf16)r2 : (__bf16)0.f;
}

// The 8-channel part (group grp, half) of a pixel's values v[0..8) into its record: one 16-byte piece per plane.
// planes == 2: f16 pairs of v * 2^-4 (returns true when a value left the pair's range); planes == 3: bf16 triples.
__device__ __forceinline__ bool cbs_store_part(char* recBase, int grp, int half, int planes, const float* v) {
    char* dst = recBase + grp * 32 * planes + half * 16;
    bool over = false;
    if (planes == 2) {
        halfx8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = v[j] * CBS_XSCALE;
            over |= !(fabsf(x) <= CBS_F16_MAX);
            _Float16 h, l;
            cbs_split(x, h, l);
            hi[j] = h, lo[j] = l;
        }
        *(halfx8*)dst = hi;
        *(halfx8*)(dst + 32) = lo;
    } else {
        bf16x8 t0, t1, t2;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            __bf16 a, b, c;
            cbs_split3(v[j], a, b, c);
            t0[j] = a, t1[j] = b, t2[j] = c;
        }
        *(bf16x8*)dst = t0;
        *(bf16x8*)(dst + 32) = t1;
        *(bf16x8*)(dst + 64) = t2;
    }
    return over;
}

// FOUR consecutive channels (quarter `sub` of half `half` of group `grp`) of a pixel's values into its record: the 8-byte
// half of the part's 16-byte piece of every plane -- cbs_store_part's bytes, written by two lanes instead of one.
__device__ __forceinline__ bool cbs_store_quarter(char* recBase, int grp, int half, int sub, int planes, const float* v) {
    typedef _Float16 halfx4 __attribute__((ext_vector_type(4)));
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    char* dst = recBase + grp * 32 * planes + half * 16 + sub * 8;
    bool over = false;
    if (planes == 2) {
        halfx4 hi, lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x = v[j] * CBS_XSCALE;
            over |= !(fabsf(x) <= CBS_F16_MAX);
            _Float16 h, l;
            cbs_split(x, h, l);
            hi[j] = h, lo[j] = l;
        }
        *(halfx4*)dst = hi;
        *(halfx4*)(dst + 32) = lo;
    } else {
        bf16x4 t0, t1, t2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __bf16 a, b, c;
            cbs_split3(v[j], a, b, c);
            t0[j] = a, t1[j] = b, t2[j] = c;
        }
        *(bf16x4*)dst = t0;
        *(bf16x4*)(dst + 32) = t1;
        *(bf16x4*)(dst + 64) = t2;
    }
    return over;
}

}  // namespace cbs
