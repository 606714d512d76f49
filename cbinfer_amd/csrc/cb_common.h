// Shared definitions for the gfx950 kernels of libcbinfer_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cbinfer_hip.h"

#define CB_WAVE 64

typedef _Float16 cb_half;

// MFMA tile geometry of the contraction kernels (cb_conv.hip).  The prepared weight matrix is
// padded to these so the k-loop and the m-tiles need no bounds checks.
#define CB_MFMA_M 32      // out-channel rows per MFMA tile
#define CB_BK 16          // k-depth staged in LDS per step (fp32); fp16 uses 32
#define CB_BK_H 32

static inline int cb_div_up(long a, long b) { return (int)((a + b - 1) / b); }

static inline int cb_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? CB_OK : (int)e;
}

#define CB_REQUIRE(cond)                 \
    do {                                 \
        if (!(cond)) return CB_ERR_BADARG; \
    } while (0)

template <typename T>
struct cb_traits;
template <>
struct cb_traits<float> {
    static constexpr int dtype = CB_F32;
};
template <>
struct cb_traits<cb_half> {
    static constexpr int dtype = CB_F16;
};

// change predicate, one channel value
//   fp32: fabs(state - in) > th                                  (cbconv2d_cg_backend.cu:22,56)
//   fp16: d = __hsub(state, in); d > th16 | d < -th16            (cbconv2d_cg_half_backend.cu:27-28)
// bit-for-bit inequality (a value that equals the state is not written again by the copy-all detections)
__device__ __forceinline__ bool cb_differs(float s, float x) {
    return __builtin_bit_cast(unsigned, s) != __builtin_bit_cast(unsigned, x);
}
__device__ __forceinline__ bool cb_differs(cb_half s, cb_half x) {
    return __builtin_bit_cast(unsigned short, s) != __builtin_bit_cast(unsigned short, x);
}
__device__ __forceinline__ bool cb_changed(float s, float x, float th) {
    return fabsf(s - x) > th;
}
__device__ __forceinline__ bool cb_changed(cb_half s, cb_half x, cb_half th) {
    cb_half d = s - x;  // v_sub_f16: one rounding, like __hsub
    return (d > th) | (d < -th);
}
// max pooling primitives (shared by the pool kernel and the pooled change detection)
__device__ __forceinline__ float cb_neg_inf(float*) { return -INFINITY; }
__device__ __forceinline__ cb_half cb_neg_inf(cb_half*) { return (cb_half)(-INFINITY); }
__device__ __forceinline__ float cb_max(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ cb_half cb_max(cb_half a, cb_half b) { return a < b ? b : a; }
__device__ __forceinline__ float cb_threshold(float th, float*) { return th; }
__device__ __forceinline__ cb_half cb_threshold(float th, cb_half*) { return (cb_half)th; }  // RNE

// A kernel whose arguments are a struct of several hundred bytes reads them where it needs them: a handful of
// scalar loads at a time, each group a round trip to the kernel-argument memory (which misses the scalar cache
// the first time a 64-byte line is touched), one after the other along the kernel's critical path.  This touches
// every line of the first BYTES argument bytes in ONE burst at kernel entry, so that all later reads hit.
template <int BYTES>
__device__ __forceinline__ void cb_touch_kernarg() {
    typedef __attribute__((address_space(4))) const int cb_kint;
    cb_kint* ka = (cb_kint*)__builtin_amdgcn_kernarg_segment_ptr();
    int acc = 0;
#pragma unroll
    for (int o = 0; o < BYTES; o += 64) acc |= ka[o / 4];
    asm volatile("" ::"s"(acc));
}
