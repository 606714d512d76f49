// Pieces of the split-state design (cb_split.hip) shared with the kernels that refresh a split state on behalf of
// the NEXT layer (cb_rowpair.hip): the geometry of the pre-split pixel-major state copy and the f16-pair split.
#pragma once
#include "cb_common.h"

namespace cbs {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));

#define CBS_MAXSEQ CBINFER_SPLIT_MAX_SEQUENCES
#define CBS_XSCALE 0.0625f            // activations are stored as x * 2^-4 (range up to 2^20, see header)
#define CBS_XSCALE_INV 16.f
#define CBS_LO 2048.f                 // lo terms carry a factor 2^11
#define CBS_F16_MAX 65504.f
#define CBS_SPAD 4096                 // bytes in front of the records of a split state (see cbs_dma16: negative offsets)
#define CBS_PRE_BIG 1536              // mask words of all sequences of a launch: 128-row tile
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
};

__host__ __device__ inline CbsGeom cbs_geom(int C, int H, int W, int kH, int kW) {
    CbsGeom g;
    g.C = C, g.G = C / 16, g.H = H, g.W = W, g.kH = kH, g.kW = kW;
    g.padY = kH / 2, g.padXL = kW / 2;
    g.pair = g.G == 1;                      // 16 channels: a stage is two x-adjacent taps
    g.kWs = g.pair ? (kW + 1) / 2 : kW;     // stage columns per filter row
    g.padXR = g.pair ? 2 * g.kWs - 1 - g.padXL : kW / 2;
    g.Wp = W + g.padXL + g.padXR;
    g.Hp = H + 4 * g.padY + 1;              // image + border, then 2 padY + 1 zero rows for the dummy pixel
    g.rec = g.G * 64;
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


}  // namespace cbs
