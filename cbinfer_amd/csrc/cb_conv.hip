// Gather -> im2col, the dense contraction on MFMA, scatter-back, and the fused
// gather->MFMA->bias/ReLU->scatter kernel for gfx950.  Reference behaviour restated from
// cbconv2d_cg_backend.cu:138-197 and conv2d_cg.py:342-349; entry-point contracts in
// include/cbinfer_hip.h.
//
// Contraction layout: D[m][n] = sum_k A[m][k] * B[k][n] with m = output channel, n = position in the
// changed-pixel list, k = (c*kH+ky)*kW+kx.  The pixel index sits on the MFMA *lane* (C/D column), the
// output channel in the accumulator registers, so the scatter epilogue writes 32 (mostly consecutive)
// pixels of one output-channel plane per store instruction.
//   fp32: v_mfma_f32_32x32x2_f32 (exact f32 fma chain)            -- operands A[m][k], B[n][k] in LDS
//         or, CB_F32S, six v_mfma_f32_32x32x16_bf16 per 16 k on three bf16 terms per operand (X3)
//   fp16: v_mfma_f32_32x32x16_f16, f32 accumulation             -- operands A[m][k], B[n][k] in LDS
#include <stdlib.h>

#include "cb_common.h"
#include <type_traits>

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// x = hi + mid + lo with three bf16 terms (24 significant bits): hi = bf16(x), mid = bf16(x - hi),
// lo = bf16(x - hi - mid); both differences are exact in f32.  Returned as the raw 16-bit patterns.
__device__ __forceinline__ void cb_split3(float x, unsigned& hi, unsigned& mid, unsigned& lo) {
    const __bf16 h = (__bf16)x;
    const float r1 = x - (float)h;
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    const __bf16 l = (__bf16)r2;
    hi = __builtin_bit_cast(unsigned short, h);
    mid = __builtin_bit_cast(unsigned short, m);
    lo = __builtin_bit_cast(unsigned short, l);
}
// The same decomposition for the inner loops, cheaper: hi and mid by truncation (a mask instead of a
// conversion and a shift each), lo rounded to nearest; |x - (hi + mid + lo)| <= 2^-24 |x| still.  The terms
// are returned as f32 bit patterns with the bf16 in the TOP half, so two of them pack with one shift-or.
__device__ __forceinline__ void cb_split3t(float x, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = __builtin_bit_cast(unsigned, x) & 0xffff0000u;
    const float r1 = x - __builtin_bit_cast(float, hi);
    mid = __builtin_bit_cast(unsigned, r1) & 0xffff0000u;
    const float r2 = r1 - __builtin_bit_cast(float, mid);
    lo = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)r2) << 16;
}
#define CB_PK2(a, b) (((a) >> 16) | (b))
// Four values at once for the stage stores: the three planes as packed pairs {x0 x1}, {x2 x3}.  The top halves
// of two registers are packed by ONE v_perm_b32 (no mask, shift, or); the subtractions are scalar v_sub_f32 on
// purpose -- the packed f32 form the compiler would choose costs three times as much beside MFMAs.
__device__ __forceinline__ float cb_sub_f32(float a, float b) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void cb_split3t_x4(const float (&x)[4], uint2& hi, uint2& mid, uint2& lo) {
    unsigned u[4], v[4];
    float r1[4], r2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        u[e] = __builtin_bit_cast(unsigned, x[e]);
        r1[e] = cb_sub_f32(x[e], __builtin_bit_cast(float, u[e] & 0xffff0000u));
        v[e] = __builtin_bit_cast(unsigned, r1[e]);
        r2[e] = cb_sub_f32(r1[e], __builtin_bit_cast(float, v[e] & 0xffff0000u));
    }
    hi = make_uint2(__builtin_amdgcn_perm(u[1], u[0], 0x07060302u), __builtin_amdgcn_perm(u[3], u[2], 0x07060302u));
    mid = make_uint2(__builtin_amdgcn_perm(v[1], v[0], 0x07060302u), __builtin_amdgcn_perm(v[3], v[2], 0x07060302u));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 p0 = {r2[0], r2[1]}, p1 = {r2[2], r2[3]};
    lo = make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(p0, bf16x2)),
                    __builtin_bit_cast(unsigned, __builtin_convertvector(p1, bf16x2)));
}
typedef __attribute__((address_space(4))) int cb_const_int;   // constant address space: scalar loads

__device__ __forceinline__ float cb_relu(float v) { return v <= 0.f ? 0.f : v; }
__device__ __forceinline__ cb_half cb_relu(cb_half v) { return v <= (cb_half)0 ? (cb_half)0 : v; }

// ---------------------------------------------------------------------------------------------
// weight preparation: pad to the MFMA tile grid (zeros), W[KP][CkkP] (k contiguous)
// ---------------------------------------------------------------------------------------------
// k -> tap table appended to the prepared weights, two int arrays of CkkP entries each:
//   off[k]  = byte offset of the tap relative to the output pixel in a [C,H,W] tensor of elemSize bytes
//   dydx[k] = (dx << 16) | (dy & 0xffff)   (signed 16-bit each)
// The zero-padded tail k >= Ckk gets offset 2^30 and dy = -32768, i.e. always outside the image.
__device__ __forceinline__ void cb_pack_k(int* tab, int k, int Ckk, int CkkP, int kH, int kW, int H, int W,
                                          int elemSize) {
    if (k >= Ckk) {
        tab[k] = 1 << 30;
        tab[CkkP + k] = 0x8000;
        return;
    }
    const int c = k / (kH * kW), r = k % (kH * kW);
    const int dy = r / kW - (kH - 1) / 2, dx = r % kW - (kW - 1) / 2;
    tab[k] = (c * H * W + dy * W + dx) * elemSize;
    tab[CkkP + k] = (dx << 16) | (dy & 0xffff);
}
__global__ __launch_bounds__(256) void cb_prep_w_f32_kernel(const float* __restrict__ w,
                                                           float* __restrict__ wp, int K, int Ckk,
                                                           int KP, int CkkP, int kH, int kW, int H,
                                                           int W) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < CkkP) cb_pack_k((int*)(wp + (long)KP * CkkP), (int)e, Ckk, CkkP, kH, kW, H, W, 4);
    if (e >= (long)KP * CkkP) return;
    const int k = (int)(e % CkkP), m = (int)(e / CkkP);
    wp[e] = (m < K && k < Ckk) ? w[(long)m * Ckk + k] : 0.f;
}
// fp32 weights for the bf16x3 contraction: W3[KP][CkkP/32][3 planes][32 k] bf16 (a row's 32 k of one stage
// as hi | mid | lo, 192 B, exactly the LDS row image), followed by the same tap table as the f32 layout
__global__ __launch_bounds__(256) void cb_prep_w_f32s_kernel(const float* __restrict__ w,
                                                            unsigned short* __restrict__ wp, int K, int Ckk,
                                                            int KP, int CkkP, int kH, int kW, int H, int W) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < CkkP) cb_pack_k((int*)(wp + (long)KP * CkkP * 3), (int)e, Ckk, CkkP, kH, kW, H, W, 4);
    if (e >= (long)KP * CkkP) return;
    const int k = (int)(e % CkkP), m = (int)(e / CkkP);
    const float v = (m < K && k < Ckk) ? w[(long)m * Ckk + k] : 0.f;
    unsigned hi, mid, lo;
    cb_split3(v, hi, mid, lo);
    unsigned short* row = wp + ((long)m * (CkkP / 32) + k / 32) * 96 + (k & 31);
    row[0] = (unsigned short)hi;
    row[32] = (unsigned short)mid;
    row[64] = (unsigned short)lo;
}
__global__ __launch_bounds__(256) void cb_prep_w_f16_kernel(const cb_half* __restrict__ w,
                                                           cb_half* __restrict__ wp, int K, int Ckk,
                                                           int KP, int CkkP, int kH, int kW, int H,
                                                           int W) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < CkkP) cb_pack_k((int*)(wp + (long)KP * CkkP), (int)e, Ckk, CkkP, kH, kW, H, W, 2);
    if (e >= (long)KP * CkkP) return;
    const int k = (int)(e % CkkP), m = (int)(e / CkkP);
    wp[e] = (m < K && k < Ckk) ? w[(long)m * Ckk + k] : (cb_half)0;
}

// ---------------------------------------------------------------------------------------------
// a5 genXMatrix (materialising form, API parity): one thread per X element, writes coalesced
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void cb_genx_kernel(T* __restrict__ X, const T* __restrict__ in,
                                                     const int32_t* __restrict__ list, int kW, int kH,
                                                     int C, int W, int H, int nHost,
                                                     const int32_t* __restrict__ countDev) {
    const int N = countDev ? min(*countDev, nHost) : nHost;
    const int Ckk = C * kH * kW;
    const long total = (long)N * Ckk;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long)gridDim.x * blockDim.x) {
        const int n = (int)(e / Ckk), j = (int)(e % Ckk);
        const int c = j / (kH * kW), r = j % (kH * kW);
        const int ky = r / kW, kx = r % kW;
        const int pos = list[n];
        const int ix = pos % W + kx - (kW - 1) / 2;
        const int iy = pos / W + ky - (kH - 1) / 2;
        const bool inside = ix >= 0 && ix < W && iy >= 0 && iy < H;
        X[e] = inside ? in[((long)c * H + iy) * W + ix] : (T)0;
    }
}

// ---------------------------------------------------------------------------------------------
// a8 updateOutput (API parity): one thread per Yt element
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void cb_scatter_kernel(const T* __restrict__ Yt, T* out,
                                                        const int32_t* __restrict__ list, int HW,
                                                        int nHost, const int32_t* __restrict__ countDev,
                                                        int K, int relu) {
    // Yt is [K, nHost] (row length = the HOST count the matrix was allocated with)
    const int N = countDev ? min(*countDev, nHost) : nHost;
    const long total = (long)K * N;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long)gridDim.x * blockDim.x) {
        const int k = (int)(e / N), n = (int)(e % N);
        T v = Yt[(long)k * nHost + n];
        if (relu) v = cb_relu(v);
        const int pix = list[n];
        if ((unsigned)pix < (unsigned)HW) out[(long)k * HW + pix] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// the MFMA contraction
// ---------------------------------------------------------------------------------------------
struct ConvParams {
    const void* A;     // prepared weights
    const void* B;     // MODE 0: X [N, Ckk];  MODE 1: layer state [C,H,W]
    const void* bias;  // [K] or null
    void* out;         // EPI 0: Y [N,K]; 1: Yt [K,N]; 2/3: output planes [K,H,W]
    const int32_t* list;
    const int32_t* countDev;
    int nHost;
    int K, KP, Ckk, CkkP;
    int C, H, W, kH, kW;
    int relu;
    unsigned long long* clearBits;  // optional: change bit mask to zero for the next frame
    long clearWords;
    float* slabs;   // optional split-K workspace: gridDim.x partial tiles of 64x64 floats ...
    int* tickets;   // ... and gridDim.x arrival counters (zero between launches)
    // self-compacting mode (SELFC): the kernel derives the change list from the frame's bit mask itself
    unsigned long long* frameMasks;   // [2][maskWords] masks + {parity, done} ints behind them
    int maskWords, wpr;
    int32_t* listOut;                 // the list and its length are written out as a by-product
    int32_t* countOut;
    void* reluOut;                    // EPI_SCATTER_ACC: optional second plane set receiving relu(out)
    int dbg;                          // diagnostic ablations (CBINFER_CONV_DBG): 1 no gather loads, 2 no weight loads
    int xcdMap;                       // XCD-aware item order (CBINFER_XCD_MAP=0 switches it off)
    int seam;                         // split-K slices are summed by a second launch (cb_splitk_reduce_kernel)
    const int32_t* upstream;          // SELFC, optional: the producing layer's change count of this frame (0: skip)
};
// the split-K geometry of a launch, left behind the tickets for cb_splitk_reduce_kernel: {SK, tiles, N}
#define CB_SEAM_INFO 1020

#define CB_SELFC_MAXW 4096
// The layer that produced this layer's input published a change count of zero for this frame: the detection in front
// of this launch returned at once and left the mask empty, the frame does not exist for this layer -- no mask to read
// or zero, no parity to flip (the mask the parity selects is still the clean one).  Its own count goes out as zero,
// so that the layer behind it is skipped the same way.
#define CB_UPSTREAM_IDLE_EXIT                                                          \
    if (SELFC && p.upstream && *p.upstream == 0) {                                     \
        if (blockIdx.x == 0 && threadIdx.x == 0) {                                     \
            p.countOut[0] = 0;                                                         \
            if (p.seam) p.tickets[CB_SEAM_INFO] = 0;                                   \
        }                                                                              \
        return;                                                                        \
    }
// arrival counters of the self-compacting launches, on lines of their own in the free part of the ticket header
#define CB_ARRIVE_SHARDS 640
#define CB_ARRIVE_NSH 8

// "Every workgroup has read the mask": the last workgroup to get here flips the parity for the next frame.  The
// workgroups count themselves on SHARDED counters (a top counter behind them): a returning atomic on ONE word is
// served every 11 ns, so the 512 workgroups of a launch that finds nothing to do -- the common case of a deep layer
// of a network whose change has died out further up -- spent 5.6 us of their 7.3 queueing for their tickets (round 4).
__device__ __forceinline__ void cb_arrive_and_flip(unsigned long long* frameMasks, int maskWords, int* tickets, int par) {
    int* ctl = (int*)(frameMasks + 2 * (long)maskWords);
    bool last;
    if (tickets == nullptr) {
        last = __hip_atomic_fetch_add(ctl + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
    } else {
        const int sh = blockIdx.x % CB_ARRIVE_NSH, expected = ((int)gridDim.x - sh + CB_ARRIVE_NSH - 1) / CB_ARRIVE_NSH;
        int* c = tickets + CB_ARRIVE_SHARDS + sh * 32;
        last = false;
        if (__hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == expected - 1) {
            __hip_atomic_store(c, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int active = min(CB_ARRIVE_NSH, (int)gridDim.x);
            last = __hip_atomic_fetch_add(ctl + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == active - 1;
        }
    }
    if (last) {
        __hip_atomic_store(ctl + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(ctl, par ^ 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// r-th (0-based) set bit of w, r < popcount(w)
__device__ __forceinline__ int cb_select_bit(unsigned long long w, int r) {
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

// The fused kernel is the last consumer of the frame's change mask, so it re-zeroes it (before any
// early exit): the next frame's detection can atomicOr into a clean mask without a memset node.
__device__ __forceinline__ void cb_clear_mask(const ConvParams& p) {
    if (p.clearBits && blockIdx.y == 0)   // (the fp32 kernel's grid is 1-D)
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < p.clearWords;
             i += (long)gridDim.x * blockDim.x)
            p.clearBits[i] = 0ull;
}

#define CB_CONV_GRID_PER_CU 2
#define CB_MODE_MATRIX 0
#define CB_MODE_GATHER 1
#define CB_EPI_Y 0
#define CB_EPI_YT 1
#define CB_EPI_SCATTER 2
#define CB_EPI_SCATTER_ACC 3

// fp32 contraction kernel (persistent, load-balanced).
//   tile      BM x BN = (32*WM) x (32*WN) outputs, one 32x32 MFMA tile per wave; KS wave groups split the
//             k-depth of every LDS stage between them (partials summed through LDS at the end).
//   schedule  a FIXED grid (2 workgroups per CU) walks the work items (tile, k-slice) computed from the
//             device-side list length N: the host never needs N, idle workgroups exit at once, and a
//             short list is split along k (SK slices per tile) so that it still fills the chip.  The SK
//             partial tiles of a tile go to workspace slabs (write-through 16-byte stores).  They are summed
//             in slice order (deterministic) either by the workgroup whose ticket is last -- its sc1 loads
//             behind one L1 invalidate; nobody ever waits -- or, for the 16-wave forms, by a second launch
//             on all CUs (cb_splitk_reduce_kernel).
//   pipeline  BK = 32 per stage, two LDS buffers, FOUR register staging sets: the loads of stage s+4 are
//             issued during stage s, and stage s+1 is written to the idle LDS buffer while the MFMA
//             chain of stage s runs; one barrier per stage.  Every stage issues the same number of loads
//             and the loop body is branch-free (instantiated per fast-path / wave role), so the
//             compiler's counted vmcnt only waits for the set being stored.  Odd wave groups run
//             [MFMA, store, issue loads], even ones [store, issue loads, MFMA].
//   LDS       both operands as rows of the stage's 32 k (A[m][k], B[n][k], 144-byte row stride): a
//             lane's 8 k-slots per operand are contiguous -> 2 + 2 ds_read_b128 per wave and stage for
//             8 MFMAs, one ds_write_b128 per operand and thread.  MFMA step s pairs k-slot s of the
//             wave group's half with slot 16 + s (any pairing is valid as long as A and B agree).
//   gather    a k-row of a stage is wave-uniform: its tap (byte offset, dy, dx) comes from the table
//             built by cbinfer_prep_weights via scalar loads one stage ahead; per lane there remain two
//             adds, the image-bounds test and a select.  The load goes through a raw buffer descriptor
//             over the layer state: an out-of-image tap gets an out-of-range offset, for which the
//             hardware returns 0.
#ifdef CB_STAMP
// diagnostic build only (make EXTRA=-DCB_STAMP): per-workgroup phase time stamps (100 MHz constant clock)
__device__ unsigned long long cb_stamp_buf[1024 * 8];
__device__ unsigned long long cb_stamp_clk[1024 * 2];   // s_memtime (shader clock) at entry / exit
// wide (MS = 2) kernel: shader-clock stamps of waves 0 (store first) and 8 (MFMA first) of workgroup 8 over
// its first 24 stages: [wave][stage][after barrier, after first segment, after second, after third]
__device__ unsigned long long cb_stage_clk[2 * 24 * 4];
#define CB_STAGE_STAMP(pt)                                                                          \
    do {                                                                                            \
        if (blockIdx.x == 8 && (t == 0 || t == 512) && cb_stage_no < 24 && cb_stamp_first)          \
            cb_stage_clk[((t >> 9) * 24 + cb_stage_no) * 4 + (pt)] = __builtin_amdgcn_s_memtime();  \
    } while (0)
#define CB_STAMP_AT(i)                                                                        \
    do {                                                                                      \
        if (threadIdx.x == 0 && blockIdx.x < 1024 && cb_stamp_first)                          \
            cb_stamp_buf[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime();            \
    } while (0)
#else
#define CB_STAMP_AT(i)
#define CB_STAGE_STAMP(pt)
#endif
#define CB_SKMAX 8
#define CB_SKMAX_SEAM 32   // slices per tile when a second launch sums them
#ifndef CB_WIDE_IL
#define CB_WIDE_IL 0
#endif
// diagnostic ablations of the X3 kernel are a build option (make EXTRA=-DCB_CONV_DBG): as run-time
// branches around the stage loads they would cost the loop its counted vmcnt waits
#ifdef CB_CONV_DBG
#define CB_DBG(bit) (p.dbg & (bit))
#else
#define CB_DBG(bit) false
#endif
// X3: the same kernel with every f32 operand split into three bf16 terms (cb_split3) and the six cross
// products that matter -- hi.hi, hi.mid, mid.hi, mid.mid, hi.lo, lo.hi; the dropped ones are below 2^-24
// of the product -- issued on v_mfma_f32_32x32x16_bf16 with f32 accumulation: f32-level accuracy at 16/6
// of the f32 MFMA's rate (the f32-input MFMA runs at 1/16 of the bf16 one on gfx950).  Weights come
// pre-split (cb_prep_w_f32s_kernel); gathered values are split on their way into LDS.  LDS row = the
// stage's 32 k as hi | mid | lo (192 B + 16 B pad: 16-byte fragment reads of any 16 rows are conflict-free).
// MS = 2 (X3 only, output channels in multiples of 128): every wave owns TWO 32-row tiles of output channels
// for its 32 pixels.  Used as <2, 4, 2, ..., MS = 2>: a 1024-thread workgroup, one per CU, on a 128 x 128 tile.
// The vector-memory pipe of a CU (~70 GB/s: MI355X_MICROARCH.md, "Indexed rows") is what bounds this
// contraction -- a 64 x 64 tile moves 20 KB of weights and gathered values per stage and MFMA unit through it,
// the 128 x 128 tile half of that, with the same registers per wave.
// BHALF (with MS = 2, as <4, 2, 2, ..., MS 2, BHALF>: 256 channels x 64 pixels): only the first k-group's
// waves gather, split and store the pixel operand -- 4 values per thread as elsewhere -- so a staged pixel row
// feeds all 256 output channels; the other k-group's waves only move weights.
template <int WM, int WN, int KS, int MODE, int EPI, bool SELFC = false, bool X3 = false, int MS = 1, bool BHALF = false>
__global__ __launch_bounds__(64 * WM * WN * KS)
    __attribute__((amdgpu_waves_per_eu((WM * WN * KS >= 8 ? 4 : WM * WN * KS >= 4 ? 2 : 1)))) void cb_mfma_f32_kernel(
        ConvParams p) {
    constexpr int NT = 64 * WM * WN * KS;
    constexpr int GPC = NT > 512 ? 1 : CB_CONV_GRID_PER_CU;   // workgroups per CU in the persistent grid
    constexpr int BM = 32 * WM * MS;
    constexpr int BN = 32 * WN;
    constexpr int BK = 32;
    constexpr int LDK = X3 ? 52 : BK + 4;   // LDS row in floats: 32 k of one m / one pixel + 16 B pad
    constexpr int A_F4 = X3 ? 12 * BM : BK * BM / 4;   // 16-byte chunks of an A stage
    constexpr int A_PER_T = (A_F4 + NT - 1) / NT;
    constexpr int BT = BHALF ? NT / 2 : NT;   // threads that carry the pixel operand
    constexpr int B_PER_T = BK * BN / BT;
    static_assert(!BHALF || (MS == 2 && KS == 2), "BHALF: the k-group 0 waves of the wide form");
    constexpr int NPP = NT / BK;        // matrix: pixel slots covered by one pass
    constexpr int KSTEP = BK / KS;     // k-depth one wave group handles per stage
    constexpr int S = KSTEP / 2;        // ... = MFMA steps per wave and stage (two k per step)
    constexpr int A_STAGE = BM * LDK, B_STAGE = BN * LDK;
    constexpr int TILE = BM * BN;
    constexpr int RED = KS > 1 ? WM * WN * MS * 64 * 16 : 0;   // floats of the k-group reduction buffer
    static_assert(MS == 1 || (MS == 2 && X3), "two row tiles per wave: X3 only");
    static_assert(BN % 64 == 0, "gather rows must be wave-uniform");
    static_assert(BK * BN % BT == 0 && S % 4 == 0, "bad decomposition");
    static_assert(!X3 || (KS == 2 && MODE == CB_MODE_GATHER && B_PER_T % 4 == 0), "X3: 16 k per wave group");
    static_assert(2 * (A_STAGE + B_STAGE) >= RED, "reduce buffer");

#ifdef CB_STAMP
    bool cb_stamp_first = true;
    if (threadIdx.x == 0 && blockIdx.x < 1024) cb_stamp_clk[blockIdx.x * 2] = __builtin_amdgcn_s_memtime();
#endif
    CB_STAMP_AT(0);
    const int t = threadIdx.x;
    CB_UPSTREAM_IDLE_EXIT
    // ---- SELFC: stream compaction folded into this kernel -------------------------------------------
    // Every workgroup rebuilds the exclusive popcount prefix of the frame's change mask (<= 32 KB, L2
    // resident) in LDS; a tile's pixels are then found by rank (binary search over the prefix + select
    // of the r-th set bit).  No compaction launch, and no hand-off between workgroups.  Two masks
    // alternate by a device-side parity so that this launch can zero the one the NEXT frame's detection
    // will fill while every workgroup still reads the current one; the last workgroup to finish flips
    // the parity.
    __shared__ int s_pre[SELFC ? CB_SELFC_MAXW + 1 : 1];
    __shared__ int s_wsum[SELFC ? NT / 64 : 1];
    __shared__ int s_tilePix[SELFC ? BN : 1];
    const unsigned long long* mask = nullptr;
    int par = 0;
    int N;
    if (SELFC) {
        int* ctl = (int*)(p.frameMasks + 2 * (long)p.maskWords);   // {parity, done}
        par = ctl[0];
        mask = p.frameMasks + (par ? p.maskWords : 0);
        unsigned long long* other = p.frameMasks + (par ? 0 : p.maskWords);
        // (+ round 6: a copy of THIS frame's mask at a fixed address behind the two masks and the control words -- which
        //  of the two masks is the frame's is a device-side matter (the parity); the copy is what a chained consumer's
        //  detection reads as its producer mask: segments this layer did not rewrite are skipped)
        unsigned long long* copy = p.frameMasks + 2 * (long)p.maskWords + 2;
        for (int i = blockIdx.x * NT + t; i < p.maskWords; i += gridDim.x * NT) other[i] = 0ull, copy[i] = mask[i];
        // Exclusive prefix of the per-word popcounts.  Pass 1: coalesced, independent loads (word t,
        // t+NT, ...) -> counts in LDS; pass 2 (LDS only): thread t owns CH consecutive words.
        const int CH = (p.maskWords + NT - 1) / NT;
#pragma unroll 4
        for (int w = t; w < p.maskWords; w += NT) s_pre[w] = __popcll(mask[w]);
        __syncthreads();
        const int wb = t * CH;
        int loc = 0;
        for (int u = 0; u < CH; ++u) {
            const int w = wb + u;
            if (w < p.maskWords) loc += s_pre[w];
        }
        int incl = loc;   // inclusive scan over the wave, then over the waves
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o);
            if ((t & 63) >= o) incl += v;
        }
        if ((t & 63) == 63) s_wsum[t >> 6] = incl;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < (t >> 6); ++w) base += s_wsum[w];
        int run = base + incl - loc;
        for (int u = 0; u < CH; ++u) {
            const int w = wb + u;
            if (w < p.maskWords) {
                const int c = s_pre[w];   // count -> exclusive prefix, in place (this thread's chunk only)
                s_pre[w] = run;
                run += c;
            }
        }
        if (t == NT - 1) s_pre[p.maskWords] = base + incl;   // the last thread's chunk ends the mask
        __syncthreads();
        N = min(s_pre[p.maskWords], p.nHost);
        if (blockIdx.x == 0 && t == 0) p.countOut[0] = N;
    } else {
        if (MODE == CB_MODE_GATHER) cb_clear_mask(p);
        N = p.countDev ? min(*p.countDev, p.nHost) : p.nHost;
    }
    const int MT = p.KP / BM;
    const int T = ((N + BN - 1) / BN) * MT;                  // output tiles
    const int P = (p.CkkP + 4 * BK - 1) / (4 * BK);          // groups of four stages along k (last: partial)
    // Split-K slice count (at most ~sqrt(3.4 P) <= 8: the reducer costs ~4 us of fences + ~1 us per slab)
    int SK = 1;
    const int cus = (int)gridDim.x / GPC;
#ifndef CB_SK_TARGET
#define CB_SK_TARGET 2
#endif
    static_assert(CB_SK_TARGET <= CB_CONV_GRID_PER_CU, "one slab per work item: items <= workgroups");
    // Measured in the frame (not on warm re-launches, which mislead here): a deep contraction (>= 1024 k)
    // with at least half a grid of tiles fills two workgroups per CU; everything else is only split as far
    // as every slice still gets a CU of its own -- below that the slab round trip (~8 us) costs more than
    // the idle CUs, above it slices of co-resident workgroups just slow each other down.
    if (p.slabs && T > 0 && P >= 4) {
        // (summed by a second launch on all CUs the slices cost no reducer time per tile: split as far as the
        //  work items still find a CU each -- a short list then runs few stages per workgroup)
        const int cap = p.seam ? min(CB_SKMAX_SEAM, P) : min(CB_SKMAX, min(P, (int)sqrtf(3.4f * (float)P)));
        if (GPC == 2 && P >= 8 && T * 2 >= cus)
            SK = max(1, min(cap, (CB_SK_TARGET * cus) / T));
        else
            SK = max(1, min(cap, cus / T));
    }
    const int items = T * SK;
    if (p.seam && blockIdx.x == 0 && threadIdx.x == 0) {
        p.tickets[CB_SEAM_INFO] = SK;
        p.tickets[CB_SEAM_INFO + 1] = T;
        p.tickets[CB_SEAM_INFO + 2] = N;
    }
    if (!SELFC && (int)blockIdx.x >= ((items + 7) & ~7)) return;

    __shared__ __attribute__((aligned(16))) float smem[2 * (A_STAGE + B_STAGE)];
    __shared__ int s_last;
    float* const As = smem;                  // [2][BM][LDK]
    float* const Bs = smem + 2 * A_STAGE;    // [2][BN][LDK]

    // the wave index is wave-uniform, but only readfirstlane lets the compiler know: roles derived from
    // it (k-group, tile position, MFMA-first stagger) then stay in scalar registers and real branches
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int ks = wave / (WM * WN), wq = wave % (WM * WN);
    const int wm = wq % WM, wn = wq / WM;
    const int l31 = lane & 31, h = lane >> 5;

    const float* __restrict__ Ag = (const float*)p.A;
    const float* __restrict__ Bg = (const float*)p.B;
    const cb_const_int* koff_c = X3 ? (const cb_const_int*)((const unsigned short*)p.A + (long)p.KP * p.CkkP * 3)
                                    : (const cb_const_int*)(Ag + (long)p.KP * p.CkkP);
    const cb_const_int* kdyx_c = koff_c + p.CkkP;
    const int HW = p.H * p.W;
    const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.B, 0, MODE == CB_MODE_GATHER ? p.C * HW * 4 : 0, 0x00020000);
    // (X3: the pre-split weights through a buffer descriptor -- 32-bit offsets, the stage as scalar offset)
    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.A, 0, X3 ? (int)min((long)p.KP * p.CkkP * 6, (long)0x7fffffff) : 0, 0x00020000);
    float* __restrict__ out = (float*)p.out;
    const float* __restrict__ bias = (const float*)p.bias;

    const int bj = MODE == CB_MODE_GATHER ? t % BN : t / BK;
    const int br = MODE == CB_MODE_GATHER ? __builtin_amdgcn_readfirstlane(t / BN) * B_PER_T : t % BK;

    CB_STAMP_AT(1);
    // XCD-aware item order.  Workgroups are dealt round-robin over the 8 XCDs (blockIdx.x % 8 names the
    // XCD's peers; gridDim.x is a multiple of 8), and each XCD has its own 4 MB L2.  With the natural order
    // every XCD works on every output-channel tile and streams the WHOLE filter bank (3.2 MB f32 / 4.8 MB
    // split: it does not stay in one L2); here the m-tile is a function of item % 8, so an XCD only ever
    // touches the weights of 8/MT... of its own m-tile(s) and finds them in its L2.  Pure re-indexing: item
    // <-> (n-tile, m-tile, slice) stays a bijection, nothing is assumed about placement for correctness.
    const bool xmap = p.xcdMap && (MT == 1 || MT == 2 || MT == 4 || MT == 8) && (gridDim.x & 7) == 0;
    const int itemsP = xmap ? (items + 7) & ~7 : items;
    for (int itemL = blockIdx.x; itemL < itemsP; itemL += gridDim.x) {
        int tile, slice;
        if (xmap) {
            const int q = itemL & 7, u = itemL >> 3;
            const int mtl = q % MT, r = u * (8 / MT) + q / MT;   // r enumerates (n-tile, slice)
            const int ntl = r / SK;
            slice = r - ntl * SK;
            tile = ntl * MT + mtl;
            if (tile >= T) continue;
        } else {
            tile = itemL / SK;
            slice = itemL - tile * SK;
        }
        const int item = tile * SK + slice;   // canonical index (slab slot)
        const int n0 = (tile / MT) * BN, m0 = (tile % MT) * BM;
        // slices are cut at STAGE granularity (the stage loops leave at any stage): 98 stages over 4 slices are
        // 24/25/24/25, not 24/24/24/26 -- the longest slice is what a tile waits for
        const int nStg = p.CkkP / BK;
        const int kBeg = (nStg * slice / SK) * BK, kEnd = (nStg * (slice + 1) / SK) * BK;
        const int kLast = kEnd - BK;

        // per-thread B-load coordinates; a slot past the list is "outside the image"
        int py = -(1 << 20), px = 0, pbase4 = 0;
        if (SELFC) {
            // pixel of slot j of this tile = the (n0+j)-th set bit of the mask
            __syncthreads();   // s_tilePix of the previous item is no longer read
            if (t < BN) {
                const int r = n0 + t;
                int pos = -1;
                if (r < N) {
                    int lo = 0, hi = p.maskWords;   // largest w with s_pre[w] <= r
                    while (hi - lo > 1) {
                        const int mid = (lo + hi) >> 1;
                        if (s_pre[mid] <= r)
                            lo = mid;
                        else
                            hi = mid;
                    }
                    const int bit = cb_select_bit(mask[lo], r - s_pre[lo]);
                    const int row = lo / p.wpr;
                    pos = row * p.W + (lo - row * p.wpr) * 64 + bit;
                    if (m0 == 0 && slice == 0) p.listOut[r] = pos;
                }
                s_tilePix[t] = pos;
            }
            __syncthreads();
            const int pos = s_tilePix[bj];
            if (pos >= 0) {
                py = pos / p.W;
                px = pos - py * p.W;
                pbase4 = pos * 4;
            }
        } else if (MODE == CB_MODE_GATHER) {
            const int n = n0 + bj;
            if (n < N) {
                const int pos = p.list[n];
                py = pos / p.W;
                px = pos - py * p.W;
                pbase4 = pos * 4;
            }
        }
        // wave-uniform: every tap of every pixel of this wave lies inside the image -> no bounds test
        // ... and no padded k in this slice: then the pixel's byte offset (minus the most negative tap's) is
        // the load's vector offset for the whole item and the tap goes in as the SCALAR offset -- no per-tap
        // address arithmetic at all.  (The scalar offset is not range-checked by the hardware, hence the
        // conditions; it is unsigned, hence the bias.)
        bool fast = false;
        int tapBias = 0, pfast = 0;
        if (MODE == CB_MODE_GATHER) {
            const int phh = (p.kH - 1) / 2, pww = (p.kW - 1) / 2;
            fast = __all(py >= phh && py + (p.kH - 1 - phh) < p.H && px >= pww &&
                         px + (p.kW - 1 - pww) < p.W) &&
                   kEnd <= (p.Ckk / BK) * BK;
            tapBias = (phh * p.W + pww) * 4;
            pfast = pbase4 - tapBias;
        }
        // taps of the NEXT stage to load, B_PER_T consecutive table entries per array: ONE scalar load
        // each (s_load_dwordx4), kept in scalar registers across the stage
        typedef int ivec __attribute__((ext_vector_type(B_PER_T)));
        typedef __attribute__((address_space(4))) ivec cb_const_ivec;
        ivec pkOff, pkDyx;
        auto fetch_pk = [&](int k0) {
            if (MODE == CB_MODE_GATHER) {
#pragma unroll
                for (int i = 0; i < 1; ++i) {
                    pkOff = *(const cb_const_ivec*)(koff_c + k0 + br);
                    pkDyx = *(const cb_const_ivec*)(kdyx_c + k0 + br);
                }
            }
        };
        fetch_pk(kBeg);

        // FOUR register staging sets: the loads of stage s+4 are issued during stage s (3-6 % over two
        // sets; the waits stay counted, vmcnt(15..19), never a drain)
        // (vector types, not arrays: arrays handed to the lambdas by reference end up in scratch)
        typedef float avec __attribute__((ext_vector_type(4 * A_PER_T)));
        typedef float bvec __attribute__((ext_vector_type(B_PER_T < 2 ? 2 : B_PER_T)));
        avec a0, a1, a2, a3;
        bvec b0, b1, b2, b3;

        // X3 weight loads: byte offset of the thread's chunk(s) in stage 0 of this item's row tile
        int aoff[A_PER_T ? A_PER_T : 1];
        if (X3) {
#pragma unroll
            for (int i = 0; i < A_PER_T; ++i) {
                const int f = min(t + i * NT, A_F4 - 1);
                aoff[i] = (((m0 + f / 12) * (p.CkkP / 32)) * 12 + f % 12) * 16;
            }
        }
        auto load_stage = [&](auto FASTC, auto DOBC, int k0, avec& areg, bvec& breg) {
            constexpr bool FAST = decltype(FASTC)::value;
            constexpr bool DOB = decltype(DOBC)::value;
#pragma unroll
            for (int i = 0; i < A_PER_T; ++i) {
                const int f = X3 ? min(t + i * NT, A_F4 - 1) : t + i * NT;   // (X3: every thread loads, the
                if (A_F4 % NT == 0 || f < A_F4) {                            //  surplus is not stored)
                    float4 v;
                    if (X3 && CB_DBG(2)) {
                        v = make_float4(1.f, 1.f, 1.f, 1.f);
                    } else if (X3) {   // 12 chunks per row and stage in the pre-split layout: the thread's
                        // byte offset is fixed per item (aoff), the stage goes in as the scalar offset
                        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                        typedef float f32x4 __attribute__((ext_vector_type(4)));
                        const f32x4 w4 = __builtin_bit_cast(
                            f32x4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(arsrc, aoff[i], k0 * 6, 0));
                        v = make_float4(w4.x, w4.y, w4.z, w4.w);
                    } else {
                        const int row = f / (BK / 4), c4 = f % (BK / 4);
                        v = *(const float4*)(Ag + (long)(m0 + row) * p.CkkP + k0 + c4 * 4);
                    }
                    areg[4 * i] = v.x;
                    areg[4 * i + 1] = v.y;
                    areg[4 * i + 2] = v.z;
                    areg[4 * i + 3] = v.w;
                }
            }
            if (!DOB) {
                return;
            } else if (X3 && CB_DBG(1)) {
#pragma unroll
                for (int i = 0; i < B_PER_T; ++i) breg[i] = 1.0f;
            } else if (MODE == CB_MODE_GATHER && FAST) {
#pragma unroll
                for (int i = 0; i < B_PER_T; ++i)
                    breg[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                            brsrc, pfast, pkOff[i] + tapBias, 0));
            } else
#pragma unroll
            for (int i = 0; i < B_PER_T; ++i) {
                float v = 0.f;
                if (MODE == CB_MODE_GATHER) {
                    const int koff4 = pkOff[i];                       // scalar
                    const int dy = (pkDyx[i] << 16) >> 16, dx = pkDyx[i] >> 16;
                    const int iy = py + dy, ix = px + dx;
                    const bool ok = ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
                    const int voff = ok ? pbase4 + koff4 : (1 << 30);
                    v = __builtin_bit_cast(float,
                                           __builtin_amdgcn_raw_buffer_load_b32(brsrc, voff, 0, 0));
                } else {
                    const int n = n0 + bj + i * NPP;
                    const int kg = k0 + br;
                    if (n < N && kg < p.Ckk) v = Bg[(long)n * p.Ckk + kg];
                }
                breg[i] = v;
            }
            fetch_pk(min(k0 + BK, kLast));   // consecutive calls load consecutive stages
        };
        auto store_stage = [&](auto DOBC, int buf, const avec& areg, const bvec& breg) {
            constexpr bool DOB = decltype(DOBC)::value;
            float* as = As + buf * A_STAGE;
            float* bs = Bs + buf * B_STAGE;
#pragma unroll
            for (int i = 0; i < A_PER_T; ++i) {
                // (MS == 2: the surplus threads re-write the last chunk -- same address, same value as its
                //  owner -- so that the stage body stays one basic block for the scheduler)
                const int f = MS == 2 ? min(t + i * NT, A_F4 - 1) : t + i * NT;
                if (X3 && CB_DBG(256)) continue;   // (diagnostic: no weight LDS writes)
                if (MS == 2 || A_F4 % NT == 0 || f < A_F4)
                    *(float4*)(as + (X3 ? (f / 12) * LDK + (f % 12) * 4 : (f / (BK / 4)) * LDK + (f % (BK / 4)) * 4)) =
                        make_float4(areg[4 * i], areg[4 * i + 1], areg[4 * i + 2], areg[4 * i + 3]);
            }
            if (!DOB) {
            } else if (X3) {   // split the thread's k-consecutive values, 8-byte writes into the three planes
#pragma unroll
                for (int q = 0; q < B_PER_T / 4; ++q) {
                    const float x4[4] = {breg[q * 4], breg[q * 4 + 1], breg[q * 4 + 2], breg[q * 4 + 3]};
                    uint2 h2, m2, l2;
                    cb_split3t_x4(x4, h2, m2, l2);
                    char* row = (char*)(bs + bj * LDK) + (br + q * 4) * 2;
                    if (CB_DBG(512)) continue;   // (diagnostic: split, but no pixel-operand LDS writes)
                    *(uint2*)(row) = h2;
                    *(uint2*)(row + 64) = m2;
                    *(uint2*)(row + 128) = l2;
                }
            } else if (MODE == CB_MODE_GATHER && B_PER_T % 4 == 0) {   // the thread's taps are k-consecutive
#pragma unroll
                for (int q = 0; q < B_PER_T / 4; ++q)
                    *(float4*)(bs + bj * LDK + br + q * 4) =
                        make_float4(breg[q * 4], breg[q * 4 + 1], breg[q * 4 + 2], breg[q * 4 + 3]);
            } else {
#pragma unroll
                for (int i = 0; i < B_PER_T; ++i) {
                    if (MODE == CB_MODE_GATHER)
                        bs[bj * LDK + br + i] = breg[i];
                    else
                        bs[(bj + i * NPP) * LDK + br] = breg[i];
                }
            }
        };

        floatx16 acc, acc1;   // (acc1: second row tile, MS == 2)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f, acc1[i] = 0.f;

        // MFMA step s of wave group ks multiplies k-slots {ks*S + s, BK/2 + ks*S + s} of the stage (lane
        // half h takes the second); any pairing works as long as both operands use it, and this one
        // makes a lane's S values per operand contiguous: S/4 ds_read_b128 instead of S ds_read_b32.
        auto compute = [&](int buf) {
            if (MS == 2) {
                const char* ap = (const char*)(As + buf * A_STAGE + (wm * 64 + l31) * LDK) + ks * 32 + h * 16;
                const char* ap1 = ap + 32 * LDK * 4;
                const char* bp = (const char*)(Bs + buf * B_STAGE + (wn * 32 + l31) * LDK) + ks * 32 + h * 16;
                // all nine fragment reads first, in the order of their use (the scheduler otherwise re-uses
                // fragment registers and leaves an exposed LDS round trip between the MFMAs)
                const bf16x8 bh = *(const bf16x8*)bp, al = *(const bf16x8*)(ap + 128);
                const bf16x8 bl = *(const bf16x8*)(bp + 128), ah = *(const bf16x8*)ap;
                const bf16x8 bm = *(const bf16x8*)(bp + 64), am = *(const bf16x8*)(ap + 64);
                const bf16x8 al1 = *(const bf16x8*)(ap1 + 128), ah1 = *(const bf16x8*)ap1, am1 = *(const bf16x8*)(ap1 + 64);
                __builtin_amdgcn_sched_barrier(0);
                if (CB_DBG(8192)) __builtin_amdgcn_s_setprio(2);   // (diagnostic: MFMA chains win issue)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al1, bh, acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bl, acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am1, bm, acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am1, bh, acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bm, acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bh, acc1, 0, 0, 0);
                if (CB_DBG(8192)) __builtin_amdgcn_s_setprio(0);
#if !CB_WIDE_IL
                __builtin_amdgcn_sched_barrier(0);
#endif
                return;
            }
            if (X3) {
                // wave group ks owns k 16 ks .. 16 ks + 15 of the stage; a lane's fragment = 8 consecutive k
                const char* ap = (const char*)(As + buf * A_STAGE + (wm * 32 + l31) * LDK) + ks * 32 + h * 16;
                const char* bp = (const char*)(Bs + buf * B_STAGE + (wn * 32 + l31) * LDK) + ks * 32 + h * 16;
                const bf16x8 ah = *(const bf16x8*)ap, am = *(const bf16x8*)(ap + 64), al = *(const bf16x8*)(ap + 128);
                const bf16x8 bh = *(const bf16x8*)bp, bm = *(const bf16x8*)(bp + 64), bl = *(const bf16x8*)(bp + 128);
                // smallest terms first
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
                return;
            }
            const float* ap = As + buf * A_STAGE + (wm * 32 + l31) * LDK + h * (BK / 2) + ks * S;
            const float* bp = Bs + buf * B_STAGE + (wn * 32 + l31) * LDK + h * (BK / 2) + ks * S;
            float4 av[S / 4], bv[S / 4];
#pragma unroll
            for (int q = 0; q < S / 4; ++q) {
                av[q] = *(const float4*)(ap + q * 4);
                bv[q] = *(const float4*)(bp + q * 4);
            }
#pragma unroll
            for (int q = 0; q < S / 4; ++q) {
                float a[4] = {av[q].x, av[q].y, av[q].z, av[q].w};
                float b[4] = {bv[q].x, bv[q].y, bv[q].z, bv[q].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
                }
            }
        };

        // Every stage issues the same number of loads (past the slice end they re-load its last stage;
        // that copy lands in the idle LDS buffer and is never multiplied).  The loop is instantiated per (interior-fast-path, wave role) so that
        // its body is branch-free: the compiler's vmcnt bookkeeping stays exact (counted waits that
        // leave the three younger stages in flight) only without control flow between the loads.
        CB_STAMP_AT(2);
        auto main_loop = [&](auto FASTC, auto MFC) {
            constexpr bool MF = decltype(MFC)::value;
            const std::integral_constant<bool, !(BHALF && MF)> dob;   // (BHALF: the MFMA-first waves = k-group 1)
#ifdef CB_STAMP
            int cb_stage_no = 0;
#define CB_STAGE_NEXT ++cb_stage_no;
#else
#define CB_STAGE_NEXT
#endif
            if (MS == 2 && MF && CB_DBG(4096)) __builtin_amdgcn_s_setprio(1);   // (diagnostic: younger half wins issue)
            if (MS == 2 && !MF && CB_DBG(16384)) __builtin_amdgcn_s_setprio(1);  // (diagnostic: older half wins)
            if (MS == 2) {
                // the 16-wave form keeps ONE staging set (128 registers per wave: two accumulators and nine
                // fragment registers come first): the loads of stage s+2 are issued during stage s+1... i.e.
                // one stage (~1.5 us) ahead of their use
                load_stage(FASTC, dob, kBeg, a0, b0);
                __builtin_amdgcn_sched_barrier(0);
                store_stage(dob, 0, a0, b0);
                load_stage(FASTC, dob, min(kBeg + BK, kLast), a0, b0);
                __syncthreads();
#if CB_WIDE_IL
                // every wave: fragment reads, then the twelve MFMAs with the next stage's LDS writes, address
                // arithmetic and loads dealt into the gaps between them
#define CB_STAGE2(BUF, KNEXT)                                     \
                compute(BUF);                                         \
                store_stage(dob, (BUF) ^ 1, a0, b0);                       \
                load_stage(FASTC, dob, min(KNEXT, kLast), a0, b0);         \
                _Pragma("unroll") for (int g_ = 0; g_ < 12; ++g_) {   \
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);\
                    __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);\
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);\
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);\
                }                                                     \
                __syncthreads();
#else
#define CB_STAGE2(BUF, KNEXT)                                     \
                CB_STAGE_STAMP(0);                                    \
                if (MF) {   /* (diagnostic bits 1024 / 2048: these waves ONLY multiply / only stage) */ \
                    if (!CB_DBG(8 | 2048)) compute(BUF);              \
                    CB_STAGE_STAMP(1);                                \
                    if (!CB_DBG(4 | 1024)) store_stage(dob, (BUF) ^ 1, a0, b0);   \
                    CB_STAGE_STAMP(2);                                \
                    if (!CB_DBG(16 | 1024)) load_stage(FASTC, dob, min(KNEXT, kLast), a0, b0);     \
                    CB_STAGE_STAMP(3);                                \
                } else {    /* (1024: these waves only stage; 2048: only multiply) */ \
                    if (!CB_DBG(4 | 2048)) store_stage(dob, (BUF) ^ 1, a0, b0);   \
                    CB_STAGE_STAMP(1);                                \
                    if (!CB_DBG(16 | 2048)) load_stage(FASTC, dob, min(KNEXT, kLast), a0, b0);     \
                    CB_STAGE_STAMP(2);                                \
                    if (!CB_DBG(8 | 1024)) compute(BUF);              \
                    CB_STAGE_STAMP(3);                                \
                }                                                     \
                CB_STAGE_NEXT                                         \
                __syncthreads();
#endif
                for (int k0 = kBeg; k0 < kEnd; k0 += 2 * BK) {
                    CB_STAGE2(0, k0 + 2 * BK)
                    if (k0 + BK >= kEnd) break;
                    CB_STAGE2(1, k0 + 3 * BK)
                }
#undef CB_STAGE2
                return;
            }
            // (scheduling fences: the prologue must issue the sets in ring order, otherwise the loop
            // header inherits "set 0 is the youngest" and drains the queue every iteration)
            load_stage(FASTC, dob, kBeg, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            load_stage(FASTC, dob, min(kBeg + BK, kLast), a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            load_stage(FASTC, dob, min(kBeg + 2 * BK, kLast), a2, b2);
            __builtin_amdgcn_sched_barrier(0);
            load_stage(FASTC, dob, min(kBeg + 3 * BK, kLast), a3, b3);
            __builtin_amdgcn_sched_barrier(0);
            // LDS stage 0 <- set 0, which is then re-armed with stage 4
            store_stage(dob, 0, a0, b0);
            load_stage(FASTC, dob, min(kBeg + 4 * BK, kLast), a0, b0);
            __syncthreads();
            // One barrier per stage.  While a wave's MFMA chain on LDS buffer BUF runs it also writes the
            // NEXT stage into the other buffer (free since the barrier just passed: every wave finished
            // reading it) and re-arms that register set four stages ahead -- the LDS write and the load
            // issue hide under the 8 x 64-cycle MFMAs instead of sitting between two compute phases.
#define CB_STAGE(BUF, AREG, BREG, KNEXT)                          \
            if (MF) {                                             \
                compute(BUF);                                     \
                store_stage(dob, (BUF) ^ 1, AREG, BREG);               \
                load_stage(FASTC, dob, min(KNEXT, kLast), AREG, BREG); \
            } else {                                              \
                store_stage(dob, (BUF) ^ 1, AREG, BREG);               \
                load_stage(FASTC, dob, min(KNEXT, kLast), AREG, BREG); \
                compute(BUF);                                     \
            }                                                     \
            __syncthreads();
            // (a slice is a whole number of stages, not of groups of four -- the exits leave the
            // straight-line body, and its counted waits, alone)
            for (int k0 = kBeg; k0 < kEnd; k0 += 4 * BK) {
                CB_STAGE(0, a1, b1, k0 + 5 * BK)
                if (k0 + BK >= kEnd) break;
                CB_STAGE(1, a2, b2, k0 + 6 * BK)
                if (k0 + 2 * BK >= kEnd) break;
                CB_STAGE(0, a3, b3, k0 + 7 * BK)
                if (k0 + 3 * BK >= kEnd) break;
                CB_STAGE(1, a0, b0, k0 + 8 * BK)
            }
#undef CB_STAGE
        };
        typedef std::integral_constant<bool, true> cb_true;
        typedef std::integral_constant<bool, false> cb_false;
        bool mfmaFirst = (KS > 1) && (__builtin_amdgcn_readfirstlane(ks) & 1);
        if (!BHALF) {   // (BHALF ties the roles to the k-group)
            if (CB_DBG(32)) mfmaFirst = true;                      // (diagnostic: no stagger)
            if (CB_DBG(64)) mfmaFirst = (wave >> 2) & 1;           // (diagnostic: stagger by wave quartets)
            if (CB_DBG(128)) mfmaFirst = wave & 1;                 // (diagnostic: stagger by wave parity)
        }
        const bool fastU = __builtin_amdgcn_readfirstlane((int)fast) != 0;
        if (fastU) {
            if (mfmaFirst)
                main_loop(cb_true(), cb_true());
            else
                main_loop(cb_true(), cb_false());
        } else {
            if (mfmaFirst)
                main_loop(cb_false(), cb_true());
            else
                main_loop(cb_false(), cb_false());
        }
        // (the loop ends on a barrier: all LDS stage reads are done and smem may be reused)
        CB_STAMP_AT(3);

        if (KS > 1) {   // sum the wave groups' partial tiles through LDS (fixed order: deterministic)
            float* red = smem + (wq * MS * 16) * 64 + lane;
#pragma unroll
            for (int g = 1; g < KS; ++g) {
                if (ks == g) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[r * 64] = acc[r];
                    if (MS == 2) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) red[(16 + r) * 64] = acc1[r];
                    }
                }
                __syncthreads();
                if (ks == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] += red[r * 64];
                    if (MS == 2) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc1[r] += red[(16 + r) * 64];
                    }
                }
                __syncthreads();
            }
        }

        CB_STAMP_AT(4);
        // one output value: bias / ReLU / the form of the store
        auto emit = [&](int m, int n, int pix, float v) {
            if (EPI != CB_EPI_SCATTER_ACC) {
                if (bias) v += bias[m];
                if (p.relu) v = cb_relu(v);
            }
            if (EPI == CB_EPI_Y)
                out[(long)n * p.K + m] = v;
            else if (EPI == CB_EPI_YT)
                out[(long)m * p.nHost + n] = v;
            else if (EPI == CB_EPI_SCATTER)
                out[(long)m * HW + pix] = v;
            else {
                const float nv = out[(long)m * HW + pix] + v;
                out[(long)m * HW + pix] = nv;
                if (p.reluOut) ((float*)p.reluOut)[(long)m * HW + pix] = cb_relu(nv);
            }
        };
        if (SK > 1) {
            // Publish this slice's partial tile, take a ticket; the last arriver sums the slices and stores.
            // Slab = [BM/4][BN] float4: four consecutive output channels of one pixel -- the registers
            // 4q..4q+3 of a lane's 32x32 accumulator as they stand -- so both sides move 16 bytes per lane in
            // 512-byte runs.  The stores are write-through (sc1) and drained before the workgroup's ticket,
            // which replaces the agent-scope release (an L2 write-back per workgroup); with one workgroup
            // per CU the reducer's sc1 loads also replace the acquire (MI355X_MICROARCH.md, "Inter-workgroup
            // visibility": publish-large / splitk-seam), otherwise it invalidates its L1 first.
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void*)p.slabs, 0, (int)min((long)gridDim.x * TILE * 4, (long)0x7fffffff), 0x00020000);
#define CB_PUBLISH(A, J)                                                                            \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                            \
                const f32x4 f = {A[4 * q], A[4 * q + 1], A[4 * q + 2], A[4 * q + 3]};                   \
                const u32x4 v = __builtin_bit_cast(u32x4, f);                                           \
                const int mq = (wm * MS + (J)) * 8 + 2 * q + h;                                         \
                __builtin_amdgcn_raw_buffer_store_b128(                                                 \
                    v, srsrc, (item * (TILE / 4) + mq * BN + wn * 32 + l31) * 16, 0, 16 /* sc1 */);     \
            }
            if (ks == 0) {
                CB_PUBLISH(acc, 0)
                if (MS == 2) {
                    CB_PUBLISH(acc1, 1)
                }
            }
#undef CB_PUBLISH
            if (p.seam) {   // the slices meet in the next launch: nothing to wait for here
                CB_STAMP_AT(5);
#ifdef CB_STAMP
                cb_stamp_first = false;
#endif
                continue;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t == 0) {
                const int ticket = __hip_atomic_fetch_add(p.tickets + tile, 1, __ATOMIC_RELAXED,
                                                          __HIP_MEMORY_SCOPE_AGENT);
                const int last = ticket == SK - 1;
                if (last) {
                    if (GPC != 1) {
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    __hip_atomic_store(p.tickets + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                s_last = last;
            }
            __syncthreads();
            const bool last = s_last != 0;
            __syncthreads();   // s_last may be rewritten by the next item
            CB_STAMP_AT(5);
#ifdef CB_STAMP
            if (!last) cb_stamp_first = false;
#endif
            if (!last) continue;
            // every thread: TILE / 4 / NT chunks, all slices of a chunk in flight together
#pragma unroll
            for (int c0 = 0; c0 < TILE / 4; c0 += NT) {
                const int c = c0 + t;
                float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 4
                for (int j = 0; j < SK; ++j) {
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(
                        srsrc, ((tile * SK + j) * (TILE / 4) + c) * 16, 0, 16 /* sc1 */);
                    const f32x4 f = __builtin_bit_cast(f32x4, v);   // (whole-vector casts: element-wise
                    s0 += f.x;                                     //  ones on these types came out as splats)
                    s1 += f.y;
                    s2 += f.z;
                    s3 += f.w;
                }
                const int nl = c % BN, mq = c / BN;
                const int n = n0 + nl;
                int pix = 0;
                if (EPI >= CB_EPI_SCATTER && n < N) pix = SELFC ? s_tilePix[nl] : p.list[n];
                if (n < N && (unsigned)pix < (unsigned)HW) {
                    const int m = m0 + 4 * mq;
                    if (m < p.K) emit(m, n, pix, s0);
                    if (m + 1 < p.K) emit(m + 1, n, pix, s1);
                    if (m + 2 < p.K) emit(m + 2, n, pix, s2);
                    if (m + 3 < p.K) emit(m + 3, n, pix, s3);
                }
            }
            CB_STAMP_AT(6);
#ifdef CB_STAMP
            cb_stamp_first = false;
#endif
            continue;
        }

        // epilogue: C/D map of the 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
        const int n = n0 + wn * 32 + l31;
        int pix = 0;
        if (EPI >= CB_EPI_SCATTER && ks == 0 && n < N) pix = SELFC ? s_tilePix[wn * 32 + l31] : p.list[n];
        // (a list entry outside the map -- a propagated list of another resolution -- is dropped, never
        // written through: the reference's scatter would corrupt a neighbouring plane, .cu:187)
        if (ks == 0 && n < N && (unsigned)pix < (unsigned)HW) {
#pragma unroll
            for (int rr = 0; rr < 16 * MS; ++rr) {
                const int r = rr & 15;
                const int m = m0 + (wm * MS + (rr >> 4)) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m >= p.K) continue;
                emit(m, n, pix, rr < 16 ? acc[r] : acc1[r]);
            }
        }
        CB_STAMP_AT(6);
#ifdef CB_STAMP
        cb_stamp_first = false;
#endif
    }
#ifdef CB_STAMP
    cb_stamp_first = true;
    if (threadIdx.x == 0 && blockIdx.x < 1024) cb_stamp_clk[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memtime();
#endif
    CB_STAMP_AT(7);

    if (SELFC && N > 0) {
        // every workgroup has read the mask: the last one to get here flips the parity for the next frame.  (An EMPTY
        // mask -- every workgroup sees that alike -- needs neither: the next detection finds the mask the parity
        // selects as clean as this one left it, and the other one is zeroed by the next launch that has work; the
        // arrival tickets were 1.7 of the 5.2 us of a launch with nothing to do.)
        __syncthreads();
        if (t == 0) cb_arrive_and_flip(p.frameMasks, p.maskWords, p.tickets, par);
    }
}

// Second launch of a split-K contraction ("cut the seam"): every thread sums one float4 of a tile over its SK
// slabs, in slice order, and stores it -- all CUs at once instead of each tile's last workgroup pulling SK x 64 KB
// through one CU behind a ticket (10-11 us at the end of the 64->256 launch; here ~3 us plus the launch
// boundary).  The first launch leaves {SK, tiles, N} behind the tickets; SK <= 1 (every tile was stored by its
// own workgroup): nothing to do.  Slab layout and summation order as in cb_mfma_f32_kernel: same bits.
template <int EPI, bool SELFC>
__global__ __launch_bounds__(256) void cb_splitk_reduce_kernel(ConvParams p, int BM, int BN) {
    const int SK = p.tickets[CB_SEAM_INFO];
    if (SK <= 1) return;
    const int T = p.tickets[CB_SEAM_INFO + 1], N = p.tickets[CB_SEAM_INFO + 2];
    const int MT = p.KP / BM, TILE4 = BM * BN / 4, HW = p.H * p.W;
    const float4* __restrict__ slabs = (const float4*)p.slabs;
    float* __restrict__ out = (float*)p.out;
    const float* __restrict__ bias = (const float*)p.bias;
    const long total = (long)T * TILE4;
    for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long)gridDim.x * 256) {
        const int tile = (int)(g / TILE4), c = (int)(g - (long)tile * TILE4);
        const int nl = c % BN, mq = c / BN;
        const int n = (tile / MT) * BN + nl;
        // the slab loads do not wait for the pixel index (a slot past the list holds zeros or stale sums: never stored)
        const float4* sl = slabs + (long)tile * SK * TILE4 + c;
        const int pix = n < N ? (SELFC ? p.listOut[n] : p.list[n]) : -1;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        for (int j0 = 0; j0 < SK; j0 += CB_SKMAX) {   // eight slices in flight at a time, summed in slice order
            float4 v[CB_SKMAX];
#pragma unroll
            for (int j = 0; j < CB_SKMAX; ++j)
                if (j0 + j < SK) v[j] = sl[(long)(j0 + j) * TILE4];
#pragma unroll
            for (int j = 0; j < CB_SKMAX; ++j)
                if (j0 + j < SK) {
                    s0 += v[j].x;
                    s1 += v[j].y;
                    s2 += v[j].z;
                    s3 += v[j].w;
                }
        }
        if ((unsigned)pix >= (unsigned)HW) continue;
        const int m = (tile % MT) * BM + 4 * mq;
        const float sv[4] = {s0, s1, s2, s3};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (m + e >= p.K) continue;
            float v = sv[e];
            float* o = out + (long)(m + e) * HW + pix;
            if (EPI == CB_EPI_SCATTER) {
                if (bias) v += bias[m + e];
                if (p.relu) v = cb_relu(v);
                *o = v;
            } else {
                const float nv = *o + v;
                *o = nv;
                if (p.reluOut) ((float*)p.reluOut)[(long)(m + e) * HW + pix] = cb_relu(nv);
            }
        }
    }
}

// fp16 contraction kernel (cg_half path): the schedule, pipeline, split-K, in-kernel compaction and
// gather of cb_mfma_f32_kernel with v_mfma_f32_32x32x16_f16 (f32 accumulation).  A lane's MFMA fragment is
// 8 consecutive k, so both LDS operands are k-contiguous rows -- A[m][k], B[n][k], 144-byte row stride so
// that the 16-byte fragment reads of a 16-lane group cover all 64 banks once -- which is exactly what
// the gather produces: a thread owns B_PER_T consecutive k of one pixel and stores them with one (or
// two) 16-byte LDS writes.  BK = 64 halfs per stage.  With only 2-4 MFMAs per wave and stage this
// kernel is bound by the gather/issue stream, not by the matrix pipe.
template <int WM, int WN, int KS, int MODE, int EPI, bool SELFC = false>
__global__ __launch_bounds__(64 * WM * WN * KS) void cb_mfma_f16_kernel(ConvParams p) {
    constexpr int NT = 64 * WM * WN * KS;
    constexpr int BM = 32 * WM;
    constexpr int BN = 32 * WN;
    constexpr int BK = 64;              // halfs per stage
    constexpr int LDH = 72;             // LDS row stride in halfs (144 B)
    constexpr int A_V = BK * BM / 8;    // 16-byte chunks per A stage
    constexpr int A_PER_T = (A_V + NT - 1) / NT;
    constexpr int B_PER_T = BK * BN / NT;
    constexpr int NPP = NT / BK;        // matrix: pixel slots covered by one pass
    constexpr int KSTEP = BK / KS;      // k-depth one wave group handles per stage
    constexpr int A_STAGE = BM * LDH, B_STAGE = BN * LDH;   // in halfs
    constexpr int TILE = BM * BN;
    static_assert(BN % 64 == 0, "gather rows must be wave-uniform");
    static_assert(BK * BN % NT == 0 && KSTEP % 16 == 0 && B_PER_T % 8 == 0, "bad decomposition");
    static_assert(KS == 1 || 2 * (A_STAGE + B_STAGE) * 2 >= WM * WN * 64 * 16 * 4, "reduce buffer");

    const int t = threadIdx.x;
    CB_UPSTREAM_IDLE_EXIT
    // ---- SELFC: stream compaction folded into this kernel -------------------------------------------
    // Every workgroup rebuilds the exclusive popcount prefix of the frame's change mask (<= 32 KB, L2
    // resident) in LDS; a tile's pixels are then found by rank (binary search over the prefix + select
    // of the r-th set bit).  No compaction launch, and no hand-off between workgroups.  Two masks
    // alternate by a device-side parity so that this launch can zero the one the NEXT frame's detection
    // will fill while every workgroup still reads the current one; the last workgroup to finish flips
    // the parity.
    __shared__ int s_pre[SELFC ? CB_SELFC_MAXW + 1 : 1];
    __shared__ int s_wsum[SELFC ? NT / 64 : 1];
    __shared__ int s_tilePix[SELFC ? BN : 1];
    const unsigned long long* mask = nullptr;
    int par = 0;
    int N;
    if (SELFC) {
        int* ctl = (int*)(p.frameMasks + 2 * (long)p.maskWords);   // {parity, done}
        par = ctl[0];
        mask = p.frameMasks + (par ? p.maskWords : 0);
        unsigned long long* other = p.frameMasks + (par ? 0 : p.maskWords);
        // (+ round 6: a copy of THIS frame's mask at a fixed address behind the two masks and the control words -- which
        //  of the two masks is the frame's is a device-side matter (the parity); the copy is what a chained consumer's
        //  detection reads as its producer mask: segments this layer did not rewrite are skipped)
        unsigned long long* copy = p.frameMasks + 2 * (long)p.maskWords + 2;
        for (int i = blockIdx.x * NT + t; i < p.maskWords; i += gridDim.x * NT) other[i] = 0ull, copy[i] = mask[i];
        // Exclusive prefix of the per-word popcounts.  Pass 1: coalesced, independent loads (word t,
        // t+NT, ...) -> counts in LDS; pass 2 (LDS only): thread t owns CH consecutive words.
        const int CH = (p.maskWords + NT - 1) / NT;
#pragma unroll 4
        for (int w = t; w < p.maskWords; w += NT) s_pre[w] = __popcll(mask[w]);
        __syncthreads();
        const int wb = t * CH;
        int loc = 0;
        for (int u = 0; u < CH; ++u) {
            const int w = wb + u;
            if (w < p.maskWords) loc += s_pre[w];
        }
        int incl = loc;   // inclusive scan over the wave, then over the waves
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o);
            if ((t & 63) >= o) incl += v;
        }
        if ((t & 63) == 63) s_wsum[t >> 6] = incl;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < (t >> 6); ++w) base += s_wsum[w];
        int run = base + incl - loc;
        for (int u = 0; u < CH; ++u) {
            const int w = wb + u;
            if (w < p.maskWords) {
                const int c = s_pre[w];   // count -> exclusive prefix, in place (this thread's chunk only)
                s_pre[w] = run;
                run += c;
            }
        }
        if (t == NT - 1) s_pre[p.maskWords] = base + incl;   // the last thread's chunk ends the mask
        __syncthreads();
        N = min(s_pre[p.maskWords], p.nHost);
        if (blockIdx.x == 0 && t == 0) p.countOut[0] = N;
    } else {
        if (MODE == CB_MODE_GATHER) cb_clear_mask(p);
        N = p.countDev ? min(*p.countDev, p.nHost) : p.nHost;
    }
    const int MT = p.KP / BM;
    const int T = ((N + BN - 1) / BN) * MT;                  // output tiles
    const int P = p.CkkP / (2 * BK);                         // stage pairs along k
    // Split along k only while whole CUs would otherwise idle (T below CB_SK_TARGET x the CU count) and
    // the k-depth is long enough to pay for the slab round trip: the reducer costs ~4 us of fences +
    // ~1 us per slab, a stage pair ~1.7 us, so the best slice count is ~sqrt(1.7 P).
    int SK = 1;
    const int cus = (int)gridDim.x / CB_CONV_GRID_PER_CU;
#ifndef CB_SK_TARGET
#define CB_SK_TARGET 2
#endif
    static_assert(CB_SK_TARGET <= CB_CONV_GRID_PER_CU, "one slab per work item: items <= workgroups");
    if (p.slabs && T > 0 && T < CB_SK_TARGET * cus && P >= 8)
        SK = max(1, min(min(CB_SKMAX, (CB_SK_TARGET * cus) / T), (int)sqrtf(1.7f * (float)P)));
    const int items = T * SK;
    if (!SELFC && (int)blockIdx.x >= items) return;

    __shared__ __attribute__((aligned(16))) cb_half smemh[2 * (A_STAGE + B_STAGE)];
    __shared__ int s_last;
    cb_half* const As = smemh;                  // [2][BM][LDH]
    cb_half* const Bs = smemh + 2 * A_STAGE;    // [2][BN][LDH]
    float* const smem = (float*)smemh;          // reduce buffer view

    // the wave index is wave-uniform, but only readfirstlane lets the compiler know: roles derived from
    // it (k-group, tile position, MFMA-first stagger) then stay in scalar registers and real branches
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int ks = wave / (WM * WN), wq = wave % (WM * WN);
    const int wm = wq % WM, wn = wq / WM;
    const int l31 = lane & 31, h = lane >> 5;

    const cb_half* __restrict__ Ag = (const cb_half*)p.A;
    const cb_half* __restrict__ Bg = (const cb_half*)p.B;
    const cb_const_int* koff_c = (const cb_const_int*)(Ag + (long)p.KP * p.CkkP);
    const cb_const_int* kdyx_c = koff_c + p.CkkP;
    const int HW = p.H * p.W;
    const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.B, 0, MODE == CB_MODE_GATHER ? p.C * HW * 2 : 0, 0x00020000);
    // (the weights through a buffer descriptor: per-thread byte offset fixed per item, the stage as scalar offset)
    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.A, 0, (int)min((long)p.KP * p.CkkP * 2, (long)0x7fffffff), 0x00020000);
    cb_half* __restrict__ out = (cb_half*)p.out;
    const cb_half* __restrict__ bias = (const cb_half*)p.bias;

    const int bj = MODE == CB_MODE_GATHER ? t % BN : t / BK;
    const int br = MODE == CB_MODE_GATHER ? __builtin_amdgcn_readfirstlane(t / BN) * B_PER_T : t % BK;

    for (int item = blockIdx.x; item < items; item += gridDim.x) {
        const int tile = item / SK, slice = item - tile * SK;
        const int n0 = (tile / MT) * BN, m0 = (tile % MT) * BM;
        const int kBeg = (P * slice / SK) * 2 * BK, kEnd = (P * (slice + 1) / SK) * 2 * BK;
        const int kLast = kEnd - BK;

        // per-thread B-load coordinates; a slot past the list is "outside the image"
        int py = -(1 << 20), px = 0, pbase4 = 0;
        if (SELFC) {
            // pixel of slot j of this tile = the (n0+j)-th set bit of the mask
            __syncthreads();   // s_tilePix of the previous item is no longer read
            if (t < BN) {
                const int r = n0 + t;
                int pos = -1;
                if (r < N) {
                    int lo = 0, hi = p.maskWords;   // largest w with s_pre[w] <= r
                    while (hi - lo > 1) {
                        const int mid = (lo + hi) >> 1;
                        if (s_pre[mid] <= r)
                            lo = mid;
                        else
                            hi = mid;
                    }
                    const int bit = cb_select_bit(mask[lo], r - s_pre[lo]);
                    const int row = lo / p.wpr;
                    pos = row * p.W + (lo - row * p.wpr) * 64 + bit;
                    if (m0 == 0 && slice == 0) p.listOut[r] = pos;
                }
                s_tilePix[t] = pos;
            }
            __syncthreads();
            const int pos = s_tilePix[bj];
            if (pos >= 0) {
                py = pos / p.W;
                px = pos - py * p.W;
                pbase4 = pos * 2;
            }
        } else if (MODE == CB_MODE_GATHER) {
            const int n = n0 + bj;
            if (n < N) {
                const int pos = p.list[n];
                py = pos / p.W;
                px = pos - py * p.W;
                pbase4 = pos * 2;
            }
        }
        // wave-uniform: every tap of every pixel of this wave lies inside the image -> no bounds test
        // ... and no padded k in this slice: the tap then goes in as the load's (unsigned, unchecked) scalar
        // offset, biased by the most negative tap, and the pixel's byte offset serves the whole item
        bool fast = false;
        int tapBias = 0, pfast = 0;
        if (MODE == CB_MODE_GATHER) {
            const int phh = (p.kH - 1) / 2, pww = (p.kW - 1) / 2;
            fast = __all(py >= phh && py + (p.kH - 1 - phh) < p.H && px >= pww &&
                         px + (p.kW - 1 - pww) < p.W) &&
                   kEnd <= (p.Ckk / BK) * BK;
            tapBias = (phh * p.W + pww) * 2;
            pfast = pbase4 - tapBias;
        }
        int aoff[A_PER_T];
#pragma unroll
        for (int i = 0; i < A_PER_T; ++i) {
            const int f = min(t + i * NT, A_V - 1);
            aoff[i] = ((m0 + f / (BK / 8)) * p.CkkP + (f % (BK / 8)) * 8) * 2;
        }
        // taps of the NEXT stage to load, B_PER_T consecutive table entries per array: ONE scalar load
        // each (s_load_dwordx4), kept in scalar registers across the stage
        typedef int ivec __attribute__((ext_vector_type(B_PER_T)));
        typedef __attribute__((address_space(4))) ivec cb_const_ivec;
        ivec pkOff, pkDyx;
        auto fetch_pk = [&](int k0) {
            if (MODE == CB_MODE_GATHER) {
#pragma unroll
                for (int i = 0; i < 1; ++i) {
                    pkOff = *(const cb_const_ivec*)(koff_c + k0 + br);
                    pkDyx = *(const cb_const_ivec*)(kdyx_c + k0 + br);
                }
            }
        };
        fetch_pk(kBeg);

        // staging registers as vector types (arrays passed by reference may end up in scratch)
        typedef unsigned avec __attribute__((ext_vector_type(4 * A_PER_T)));
        typedef _Float16 bvec __attribute__((ext_vector_type(B_PER_T)));
        avec a0, a1;
        bvec b0, b1;

        auto load_stage = [&](int k0, avec& areg, bvec& breg) {
#pragma unroll
            for (int i = 0; i < A_PER_T; ++i) {
                const int f = t + i * NT;
                if (A_V % NT == 0 || f < A_V) {
                    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                    const u32x4 v4 = __builtin_amdgcn_raw_buffer_load_b128(arsrc, aoff[i], k0 * 2, 0);
                    areg[4 * i + 0] = v4.x;
                    areg[4 * i + 1] = v4.y;
                    areg[4 * i + 2] = v4.z;
                    areg[4 * i + 3] = v4.w;
                }
            }
            if (MODE == CB_MODE_GATHER && fast) {
#pragma unroll
                for (int i = 0; i < B_PER_T; ++i)
                    breg[i] = __builtin_bit_cast(cb_half, __builtin_amdgcn_raw_buffer_load_b16(
                                                              brsrc, pfast, pkOff[i] + tapBias, 0));
            } else
#pragma unroll
            for (int i = 0; i < B_PER_T; ++i) {
                cb_half v = (cb_half)0;
                if (MODE == CB_MODE_GATHER) {
                    const int koff4 = pkOff[i];                       // scalar
                    const int dy = (pkDyx[i] << 16) >> 16, dx = pkDyx[i] >> 16;
                    const int iy = py + dy, ix = px + dx;
                    const bool ok = ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
                    const int voff = ok ? pbase4 + koff4 : (1 << 30);
                    v = __builtin_bit_cast(cb_half, __builtin_amdgcn_raw_buffer_load_b16(brsrc, voff, 0, 0));
                } else {
                    const int n = n0 + bj + i * NPP;
                    const int kg = k0 + br;
                    if (n < N && kg < p.Ckk) v = Bg[(long)n * p.Ckk + kg];
                }
                breg[i] = v;
            }
            fetch_pk(min(k0 + BK, kLast));   // consecutive calls load consecutive stages
        };
        auto store_stage = [&](int buf, const avec& areg, const bvec& breg) {
            cb_half* as = As + buf * A_STAGE;
            cb_half* bs = Bs + buf * B_STAGE;
#pragma unroll
            for (int i = 0; i < A_PER_T; ++i) {
                const int f = t + i * NT;
                if (A_V % NT == 0 || f < A_V) {
                    const int row = f / (BK / 8), c8 = f % (BK / 8);
                    *(uint4*)(as + row * LDH + c8 * 8) =
                        make_uint4(areg[4 * i + 0], areg[4 * i + 1], areg[4 * i + 2], areg[4 * i + 3]);
                }
            }
            if (MODE == CB_MODE_GATHER) {
                // this thread's B_PER_T consecutive k of pixel slot bj: 16-byte LDS writes
#pragma unroll
                for (int q = 0; q < B_PER_T / 8; ++q) {
                    halfx8 v8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v8[e] = breg[8 * q + e];
                    *(halfx8*)(bs + bj * LDH + br + 8 * q) = v8;
                }
            } else {
#pragma unroll
                for (int i = 0; i < B_PER_T; ++i) bs[(bj + i * NPP) * LDH + br] = breg[i];
            }
        };

        floatx16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;

        auto compute = [&](int buf) {
            // lane holds A[row l31][k = kk + 8h + 0..7] and B[k = kk + 8h + 0..7][col l31]
            const cb_half* as = As + buf * A_STAGE + (wm * 32 + l31) * LDH + 8 * h;
            const cb_half* bs = Bs + buf * B_STAGE + (wn * 32 + l31) * LDH + 8 * h;
#pragma unroll
            for (int kk = ks * KSTEP; kk < (ks + 1) * KSTEP; kk += 16) {
                const halfx8 a = *(const halfx8*)(as + kk);
                const halfx8 b = *(const halfx8*)(bs + kk);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
            }
        };

        // Every stage issues the same number of loads (past the slice end they re-load its last stage
        // and are never stored).
        load_stage(kBeg, a0, b0);
        load_stage(kBeg + BK, a1, b1);
        const bool mfmaFirst = (KS > 1) && (ks & 1);
        for (int k0 = kBeg; k0 < kEnd; k0 += 2 * BK) {
            store_stage(0, a0, b0);
            __syncthreads();
            if (mfmaFirst) {
                compute(0);
                load_stage(min(k0 + 2 * BK, kLast), a0, b0);
            } else {
                load_stage(min(k0 + 2 * BK, kLast), a0, b0);
                compute(0);
            }
            store_stage(1, a1, b1);
            __syncthreads();
            if (mfmaFirst) {
                compute(1);
                load_stage(min(k0 + 3 * BK, kLast), a1, b1);
            } else {
                load_stage(min(k0 + 3 * BK, kLast), a1, b1);
                compute(1);
            }
        }
        __syncthreads();   // all LDS stage reads done: smem may be reused

        if (KS > 1) {   // sum the wave groups' partial tiles through LDS (fixed order: deterministic)
            float* red = smem + (wq * 16) * 64 + lane;
#pragma unroll
            for (int g = 1; g < KS; ++g) {
                if (ks == g) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[r * 64] = acc[r];
                }
                __syncthreads();
                if (ks == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] += red[r * 64];
                }
                __syncthreads();
            }
        }

        auto emit = [&](int m, int n, int pix, float v) {
            if (EPI != CB_EPI_SCATTER_ACC) {
                if (bias) v += (float)bias[m];
                if (p.relu) v = cb_relu(v);
            }
            if (EPI == CB_EPI_Y)
                out[(long)n * p.K + m] = (cb_half)v;
            else if (EPI == CB_EPI_YT)
                out[(long)m * p.nHost + n] = (cb_half)v;
            else if (EPI == CB_EPI_SCATTER)
                out[(long)m * HW + pix] = (cb_half)v;
            else
                out[(long)m * HW + pix] = (cb_half)((float)out[(long)m * HW + pix] + v);
        };
        if (SK > 1) {
            // publish this slice's partial tile, take a ticket; the last arriver sums and stores -- the
            // protocol and slab layout of cb_mfma_f32_kernel ([BM/4][BN] float4, write-through stores, no
            // release; two workgroups per CU: the reducer invalidates its L1 before its sc1 loads)
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void*)p.slabs, 0, (int)min((long)gridDim.x * TILE * 4, (long)0x7fffffff), 0x00020000);
            if (ks == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 f = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
                    const u32x4 v = __builtin_bit_cast(u32x4, f);
                    const int mq = wm * 8 + 2 * q + h;
                    __builtin_amdgcn_raw_buffer_store_b128(
                        v, srsrc, (item * (TILE / 4) + mq * BN + wn * 32 + l31) * 16, 0, 16 /* sc1 */);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t == 0) {
                const int ticket = __hip_atomic_fetch_add(p.tickets + tile, 1, __ATOMIC_RELAXED,
                                                          __HIP_MEMORY_SCOPE_AGENT);
                const int last = ticket == SK - 1;
                if (last) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_store(p.tickets + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                s_last = last;
            }
            __syncthreads();
            const bool last = s_last != 0;
            __syncthreads();   // s_last may be rewritten by the next item
            if (!last) continue;
#pragma unroll
            for (int c0 = 0; c0 < TILE / 4; c0 += NT) {
                const int c = c0 + t;
                float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 4
                for (int j = 0; j < SK; ++j) {
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(
                        srsrc, ((tile * SK + j) * (TILE / 4) + c) * 16, 0, 16 /* sc1 */);
                    const f32x4 f = __builtin_bit_cast(f32x4, v);
                    s0 += f.x;
                    s1 += f.y;
                    s2 += f.z;
                    s3 += f.w;
                }
                const int nl = c % BN, mq = c / BN;
                const int n = n0 + nl;
                int pix = 0;
                if (EPI >= CB_EPI_SCATTER && n < N) pix = SELFC ? s_tilePix[nl] : p.list[n];
                if (n < N && (unsigned)pix < (unsigned)HW) {
                    const int m = m0 + 4 * mq;
                    if (m < p.K) emit(m, n, pix, s0);
                    if (m + 1 < p.K) emit(m + 1, n, pix, s1);
                    if (m + 2 < p.K) emit(m + 2, n, pix, s2);
                    if (m + 3 < p.K) emit(m + 3, n, pix, s3);
                }
            }
            continue;
        }

        // epilogue: C/D map of the 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
        const int n = n0 + wn * 32 + l31;
        int pix = 0;
        if (EPI >= CB_EPI_SCATTER && ks == 0 && n < N) pix = SELFC ? s_tilePix[wn * 32 + l31] : p.list[n];
        // (a list entry outside the map -- a propagated list of another resolution -- is dropped, never
        // written through: the reference's scatter would corrupt a neighbouring plane, .cu:187)
        if (ks == 0 && n < N && (unsigned)pix < (unsigned)HW) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m >= p.K) continue;
                emit(m, n, pix, acc[r]);
            }
        }
    }

    if (SELFC && N > 0) {
        // every workgroup has read the mask: the last one to get here flips the parity for the next frame.  (An EMPTY
        // mask -- every workgroup sees that alike -- needs neither: the next detection finds the mask the parity
        // selects as clean as this one left it, and the other one is zeroed by the next launch that has work; the
        // arrival tickets were 1.7 of the 5.2 us of a launch with nothing to do.)
        __syncthreads();
        if (t == 0) cb_arrive_and_flip(p.frameMasks, p.maskWords, p.tickets, par);
    }
}

// Tile configuration.  cfg = 100*WM + 10*WN + KS; the CBINFER_CONV_CFG environment variable overrides
// the heuristic (tuning aid).  KP == 32 (K <= 32): one m-tile, 128 pixels per workgroup; otherwise
// 64 x 64.  KS (in-block split-K): 2 unless the k-depth is a single stage.
int cb_num_cus() {
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

template <int WM, int WN, int KS, int MODE, int EPI, bool SELFC = false, bool X3 = false, int MS = 1, bool BHALF = false>
int launch_f32(const ConvParams& p, hipStream_t s) {
    const long tilesCap = (long)cb_div_up(p.nHost, 32 * WN) * (p.KP / (32 * WM * MS));
    if (tilesCap == 0) return CB_OK;
    // persistent grid: 2 workgroups per CU, 1 of the 1024-thread form (fewer only if the capacity itself is
    // smaller and there is no split-K workspace to spread it with)
    long g = (WM * WN * KS > 8 ? 1 : CB_CONV_GRID_PER_CU) * (long)cb_num_cus();
    if (!p.slabs && tilesCap < g && !SELFC) g = tilesCap;
    dim3 grid((unsigned)g), block(64 * WM * WN * KS);
    // (the second launch pays where a tile's slices are 64 KB each -- the 16-wave forms; with 16 KB slabs the
    //  last workgroup's reduce is cheaper than a launch boundary: OpenPose's 36 small layers lost 5 % to it)
    ConvParams q = p;
    if (MS != 2) q.seam = 0;
    hipLaunchKernelGGL((cb_mfma_f32_kernel<WM, WN, KS, MODE, EPI, SELFC, X3, MS, BHALF>), grid, block, 0, s, q);
    if constexpr (EPI >= CB_EPI_SCATTER && MS == 2) {
        if (p.seam) {
            const int st = cb_launch_status();
            if (st != CB_OK) return st;
            hipLaunchKernelGGL((cb_splitk_reduce_kernel<EPI, SELFC>), dim3(4 * cb_num_cus()), dim3(256), 0, s, p,
                               32 * WM * MS, 32 * WN);
        }
    }
    return cb_launch_status();
}

template <int WM, int WN, int KS, int MODE, int EPI, bool SELFC = false>
int launch_f16(const ConvParams& p, hipStream_t s) {
    const long tilesCap = (long)cb_div_up(p.nHost, 32 * WN) * (p.KP / (32 * WM));
    if (tilesCap == 0) return CB_OK;
    long g = CB_CONV_GRID_PER_CU * (long)cb_num_cus();
    if (!p.slabs && tilesCap < g && !SELFC) g = tilesCap;
    dim3 grid((unsigned)g), block(64 * WM * WN * KS);
    hipLaunchKernelGGL((cb_mfma_f16_kernel<WM, WN, KS, MODE, EPI, SELFC>), grid, block, 0, s, p);
    return cb_launch_status();
}

// k-depth padding of the prepared weights
int cb_ckkpad(int Ckk, int dtype) {
    const int q = dtype == CB_F16 ? 128 : 32;   // fp32 (both layouts): whole 32-deep stages; fp16: stage pairs of 2 x 64
    return (Ckk + q - 1) / q * q;
}

// bf16x3 arithmetic, 16-wave forms: 2 = 256 channels x 64 pixels (output channels in multiples of 256: a
// staged pixel row feeds all of them, and per-value work on the vector ALU -- which does NOT hide under other
// waves' MFMAs here -- is what the stage pays for), 1 = 128 x 128 (multiples of 128), 0 = the 64 x 64 form.
// CBINFER_X3_WIDE caps the choice.
int cb_x3_wide(int KP) {
    static int cap = -1;
    if (cap < 0) {
        const char* e = getenv("CBINFER_X3_WIDE");
        cap = e ? atoi(e) : 2;
    }
    if (cap >= 2 && KP % 256 == 0) return 2;
    if (cap >= 1 && KP % 128 == 0) return 1;
    return 0;
}

int conv_cfg_override() {
    const char* e = getenv("CBINFER_CONV_CFG");
    return e ? atoi(e) : 0;
}
int conv_dbg() {
    static int d = -1;
    if (d < 0) {
        const char* e = getenv("CBINFER_CONV_DBG");
        d = e ? atoi(e) : 0;
    }
    return d;
}

template <int MODE, int EPI>
int launch_mfma(const ConvParams& p0, int dtype, hipStream_t s) {
    ConvParams p = p0;
    p.dbg = conv_dbg();
    {   // split-K slices summed by a second launch (scatter epilogues with a workspace); CBINFER_SPLITK_SEAM=0:
        // by each tile's last workgroup inside the one launch
        static int seam = -1;
        if (seam < 0) {
            const char* e = getenv("CBINFER_SPLITK_SEAM");
            seam = e ? atoi(e) : 1;
        }
        p.seam = seam && p.slabs && EPI >= CB_EPI_SCATTER && (dtype == CB_F32 || dtype == CB_F32S);
    }
    {
        static int xm = -1;
        if (xm < 0) {
            const char* e = getenv("CBINFER_XCD_MAP");
            xm = e ? atoi(e) : 1;
        }
        p.xcdMap = xm;
    }
    const bool narrow = p.KP <= 32;
    if (dtype == CB_F32S) {   // f32 tensors, bf16x3 split products (gather modes only)
        if constexpr (MODE == CB_MODE_GATHER && EPI >= CB_EPI_SCATTER) {
            if (p.frameMasks) {
                if (narrow) return launch_f32<1, 4, 2, CB_MODE_GATHER, EPI, true, true>(p, s);
                if (cb_x3_wide(p.KP) == 2) return launch_f32<4, 2, 2, CB_MODE_GATHER, EPI, true, true, 2, true>(p, s);
                if (cb_x3_wide(p.KP) >= 1) return launch_f32<2, 4, 2, CB_MODE_GATHER, EPI, true, true, 2>(p, s);
                return launch_f32<2, 2, 2, CB_MODE_GATHER, EPI, true, true>(p, s);
            }
            if (narrow) return launch_f32<1, 4, 2, CB_MODE_GATHER, EPI, false, true>(p, s);
            if (cb_x3_wide(p.KP) == 2) return launch_f32<4, 2, 2, CB_MODE_GATHER, EPI, false, true, 2, true>(p, s);
            if (cb_x3_wide(p.KP) >= 1) return launch_f32<2, 4, 2, CB_MODE_GATHER, EPI, false, true, 2>(p, s);
            return launch_f32<2, 2, 2, CB_MODE_GATHER, EPI, false, true>(p, s);
        } else {
            return CB_ERR_UNSUPPORTED;
        }
    }
    if (dtype == CB_F32) {
        int cfg = conv_cfg_override();
        if (cfg == 0) cfg = narrow ? 142 : 222;
        if (narrow && cfg / 100 != 1) cfg = 142;
        if (p.frameMasks) {   // self-compacting frame pipeline: only the default configurations
            if (MODE != CB_MODE_GATHER || EPI < CB_EPI_SCATTER) return CB_ERR_BADARG;
            constexpr int E = EPI < CB_EPI_SCATTER ? CB_EPI_SCATTER : EPI;
            if (narrow) return launch_f32<1, 4, 2, CB_MODE_GATHER, E, true>(p, s);
            return launch_f32<2, 2, 2, CB_MODE_GATHER, E, true>(p, s);
        }
        switch (cfg) {
            case 141: return launch_f32<1, 4, 1, MODE, EPI>(p, s);
            case 142: return launch_f32<1, 4, 2, MODE, EPI>(p, s);
            case 121: return launch_f32<1, 2, 1, MODE, EPI>(p, s);
            case 122: return launch_f32<1, 2, 2, MODE, EPI>(p, s);
            case 124: return launch_f32<1, 2, 4, MODE, EPI>(p, s);
            case 221: return launch_f32<2, 2, 1, MODE, EPI>(p, s);
            case 222: return launch_f32<2, 2, 2, MODE, EPI>(p, s);
            case 224: return launch_f32<2, 2, 4, MODE, EPI>(p, s);
            default: return CB_ERR_BADARG;
        }
    }
    if (p.frameMasks) {
        if (MODE != CB_MODE_GATHER || EPI != CB_EPI_SCATTER) return CB_ERR_BADARG;
        if (narrow) return launch_f16<1, 4, 2, CB_MODE_GATHER, CB_EPI_SCATTER, true>(p, s);
        return launch_f16<2, 2, 2, CB_MODE_GATHER, CB_EPI_SCATTER, true>(p, s);
    }
    if (narrow) return launch_f16<1, 4, 2, MODE, EPI>(p, s);
    return launch_f16<2, 2, 2, MODE, EPI>(p, s);
}

}  // namespace

extern "C" {

// K <= 32 uses the 32-row workgroup tile, anything larger the 64-row one: pad to whole tiles
int cbinfer_weights_kpad(int K) { return K <= CB_MFMA_M ? CB_MFMA_M : (K + 63) / 64 * 64; }
int cbinfer_weights_ckkpad(int Ckk, int dtype) { return cb_ckkpad(Ckk, dtype); }

long cbinfer_conv_workspace_bytes(void) {
    return 4096 + (long)CB_CONV_GRID_PER_CU * cb_num_cus() * 128 * 64 * 4;   // (= one 128 x 128 tile per CU)
}

long cbinfer_prepared_weights_bytes(int K, int C, int kH, int kW, int dtype) {
    const long KP = cbinfer_weights_kpad(K), CkkP = cb_ckkpad(C * kH * kW, dtype);
    return KP * CkkP * (dtype == CB_F16 ? 2 : dtype == CB_F32S ? 6 : 4) + CkkP * 8;
}

int cbinfer_prep_weights(const void* weight, void* weightsPrepared, int K, int C, int kH, int kW,
                         int H, int W, int dtype, cbStream_t stream) {
    CB_REQUIRE(weight && weightsPrepared && K > 0 && C > 0 && kH > 0 && kW > 0 && H > 0 && W > 0);
    if (kH > 255 || kW > 255 || (long)C * H * W * 4 >= (1l << 30)) return CB_ERR_UNSUPPORTED;
    const int Ckk = C * kH * kW;
    const int KP = cbinfer_weights_kpad(K), CkkP = cb_ckkpad(Ckk, dtype);
    const long total = (long)KP * CkkP;
    dim3 grid(cb_div_up(total, 256)), block(256);
    if (dtype == CB_F32)
        hipLaunchKernelGGL(cb_prep_w_f32_kernel, grid, block, 0, (hipStream_t)stream,
                           (const float*)weight, (float*)weightsPrepared, K, Ckk, KP, CkkP, kH, kW, H, W);
    else if (dtype == CB_F16)
        hipLaunchKernelGGL(cb_prep_w_f16_kernel, grid, block, 0, (hipStream_t)stream,
                           (const cb_half*)weight, (cb_half*)weightsPrepared, K, Ckk, KP, CkkP, kH, kW, H, W);
    else if (dtype == CB_F32S)
        hipLaunchKernelGGL(cb_prep_w_f32s_kernel, grid, block, 0, (hipStream_t)stream, (const float*)weight,
                           (unsigned short*)weightsPrepared, K, Ckk, KP, CkkP, kH, kW, H, W);
    else
        return CB_ERR_BADARG;
    return cb_launch_status();
}

int cbinfer_gen_x_matrix(void* columns, const void* input, const int32_t* changeList, int kW, int kH,
                         int C, int W, int H, int numChanges, const int32_t* countDev, int dtype,
                         cbStream_t stream) {
    CB_REQUIRE(columns && input && changeList && kW > 0 && kH > 0 && C > 0 && W > 0 && H > 0 &&
               numChanges >= 0);
    if (numChanges == 0) return CB_OK;
    const long total = (long)numChanges * C * kH * kW;
    long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    dim3 grid((unsigned)blocks), block(256);
    if (dtype == CB_F32)
        hipLaunchKernelGGL((cb_genx_kernel<float>), grid, block, 0, (hipStream_t)stream,
                           (float*)columns, (const float*)input, changeList, kW, kH, C, W, H,
                           numChanges, countDev);
    else if (dtype == CB_F16)
        hipLaunchKernelGGL((cb_genx_kernel<cb_half>), grid, block, 0, (hipStream_t)stream,
                           (cb_half*)columns, (const cb_half*)input, changeList, kW, kH, C, W, H,
                           numChanges, countDev);
    else
        return CB_ERR_BADARG;
    return cb_launch_status();
}

int cbinfer_matrix_mult(const void* X, const void* weightsPrepared, const void* bias, void* Y, int N,
                        const int32_t* countDev, int Ckk, int K, int transposeOut, int dtype,
                        cbStream_t stream) {
    CB_REQUIRE(X && weightsPrepared && Y && N >= 0 && Ckk > 0 && K > 0);
    CB_REQUIRE(dtype == CB_F32 || dtype == CB_F16);
    if (N == 0) return CB_OK;
    ConvParams p = {};
    p.A = weightsPrepared;
    p.B = X;
    p.bias = bias;
    p.out = Y;
    p.countDev = countDev;
    p.nHost = N;
    p.K = K;
    p.KP = cbinfer_weights_kpad(K);
    p.Ckk = Ckk;
    p.CkkP = cb_ckkpad(Ckk, dtype);
    p.H = p.W = p.kH = p.kW = 1;
    if (transposeOut) return launch_mfma<CB_MODE_MATRIX, CB_EPI_YT>(p, dtype, (hipStream_t)stream);
    return launch_mfma<CB_MODE_MATRIX, CB_EPI_Y>(p, dtype, (hipStream_t)stream);
}

int cbinfer_update_output(const void* Yt, void* output, const int32_t* changeList,
                          int numOutputPixel, int numChanges, const int32_t* countDev,
                          int nOutputPlane, int relu, int dtype, cbStream_t stream) {
    CB_REQUIRE(Yt && output && changeList && numOutputPixel > 0 && numChanges >= 0 &&
               nOutputPlane > 0);
    if (numChanges == 0) return CB_OK;
    const long total = (long)numChanges * nOutputPlane;
    long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    dim3 grid((unsigned)blocks), block(256);
    if (dtype == CB_F32)
        hipLaunchKernelGGL((cb_scatter_kernel<float>), grid, block, 0, (hipStream_t)stream,
                           (const float*)Yt, (float*)output, changeList, numOutputPixel, numChanges,
                           countDev, nOutputPlane, relu);
    else if (dtype == CB_F16)
        hipLaunchKernelGGL((cb_scatter_kernel<cb_half>), grid, block, 0, (hipStream_t)stream,
                           (const cb_half*)Yt, (cb_half*)output, changeList, numOutputPixel,
                           numChanges, countDev, nOutputPlane, relu);
    else
        return CB_ERR_BADARG;
    return cb_launch_status();
}

int cbinfer_conv_changed(const void* input, const int32_t* changeList, int numChanges,
                         const int32_t* countDev, const void* weightsPrepared, const void* bias,
                         void* output, int C, int H, int W, int K, int kH, int kW, int relu,
                         int accumulate, uint64_t* clearBits, long clearWords, void* workspace,
                         int dtype, cbStream_t stream) {
    CB_REQUIRE(input && changeList && weightsPrepared && output && C > 0 && H > 0 && W > 0 && K > 0 &&
               kH > 0 && kW > 0 && numChanges >= 0);
    CB_REQUIRE(dtype == CB_F32 || dtype == CB_F16 || dtype == CB_F32S);
    if ((long)C * kH * kW > 65535 || (long)C * H * W * 4 >= (1l << 30)) return CB_ERR_UNSUPPORTED;
    if (numChanges == 0) return CB_OK;
    ConvParams p = {};
    p.A = weightsPrepared;
    p.B = input;
    p.bias = bias;
    p.out = output;
    p.list = changeList;
    p.countDev = countDev;
    p.nHost = numChanges;
    p.K = K;
    p.KP = cbinfer_weights_kpad(K);
    p.Ckk = C * kH * kW;
    p.CkkP = cb_ckkpad(p.Ckk, dtype);
    p.C = C;
    p.H = H;
    p.W = W;
    p.kH = kH;
    p.kW = kW;
    p.relu = relu;
    p.clearBits = (unsigned long long*)clearBits;
    p.clearWords = clearBits ? clearWords : 0;
    if (workspace) {   // [tickets: grid ints, padded to 4 KB][slabs: one partial tile per workgroup of the grid]
        p.tickets = (int*)workspace;
        p.slabs = (float*)((char*)workspace + 4096);
    }
    if (accumulate)
        return launch_mfma<CB_MODE_GATHER, CB_EPI_SCATTER_ACC>(p, dtype, (hipStream_t)stream);
    return launch_mfma<CB_MODE_GATHER, CB_EPI_SCATTER>(p, dtype, (hipStream_t)stream);
}

// Self-compacting form of cbinfer_conv_changed (used by cbinfer_cbconv2d_forward): the change list is
// derived inside the kernel from the frame's bit mask; idxOut/countOut receive it as a by-product.
static int cb_conv_from_mask(const void* input, uint64_t* frameMasks, int32_t* idxOut,
                             int32_t* countOut, const void* weightsPrepared, const void* bias,
                             void* output, int C, int H, int W, int K, int kH, int kW, int relu,
                             void* workspace, int dtype, cbStream_t stream, int accumulate, void* reluOut,
                             const int32_t* upstream = nullptr) {
    CB_REQUIRE(input && frameMasks && idxOut && countOut && weightsPrepared && output && C > 0 && H > 0 &&
               W > 0 && K > 0 && kH > 0 && kW > 0);
    if (dtype != CB_F32 && dtype != CB_F16 && dtype != CB_F32S) return CB_ERR_BADARG;
    const long words = cbinfer_mask_words(H, W);
    if (words > CB_SELFC_MAXW || (long)C * kH * kW > 65535 || (long)C * H * W * 4 >= (1l << 30))
        return CB_ERR_UNSUPPORTED;
    ConvParams p = {};
    p.A = weightsPrepared;
    p.B = input;
    p.bias = bias;
    p.out = output;
    p.nHost = H * W;
    p.K = K;
    p.KP = cbinfer_weights_kpad(K);
    p.Ckk = C * kH * kW;
    p.CkkP = cb_ckkpad(p.Ckk, dtype);
    p.C = C;
    p.H = H;
    p.W = W;
    p.kH = kH;
    p.kW = kW;
    p.relu = relu;
    p.frameMasks = (unsigned long long*)frameMasks;
    p.maskWords = (int)words;
    p.wpr = cbinfer_mask_words_per_row(W);
    p.listOut = idxOut;
    p.countOut = countOut;
    p.upstream = upstream;
    if (workspace) {
        p.tickets = (int*)workspace;
        p.slabs = (float*)((char*)workspace + 4096);
    }
    if (accumulate) {
        if (dtype != CB_F32 && dtype != CB_F32S) return CB_ERR_UNSUPPORTED;
        p.reluOut = reluOut;
        return launch_mfma<CB_MODE_GATHER, CB_EPI_SCATTER_ACC>(p, dtype, (hipStream_t)stream);
    }
    return launch_mfma<CB_MODE_GATHER, CB_EPI_SCATTER>(p, dtype, (hipStream_t)stream);
}

int cbinfer_conv_changed_from_mask(const void* input, uint64_t* frameMasks, int32_t* idxOut,
                                   int32_t* countOut, const void* weightsPrepared, const void* bias,
                                   void* output, int C, int H, int W, int K, int kH, int kW, int relu,
                                   void* workspace, int dtype, cbStream_t stream) {
    return cb_conv_from_mask(input, frameMasks, idxOut, countOut, weightsPrepared, bias, output, C, H, W, K,
                             kH, kW, relu, workspace, dtype, stream, 0, nullptr);
}

// upstreamCount (optional, device): the change count the layer that produced `input`'s source published this frame;
// zero ends the launch at once with countOut = 0 and the masks untouched (cbinfer_cbconv2d_forward_after).
int cbinfer_conv_changed_from_mask_after(const int32_t* upstreamCount, const void* input, uint64_t* frameMasks,
                                         int32_t* idxOut, int32_t* countOut, const void* weightsPrepared,
                                         const void* bias, void* output, int C, int H, int W, int K, int kH, int kW,
                                         int relu, void* workspace, int dtype, cbStream_t stream) {
    return cb_conv_from_mask(input, frameMasks, idxOut, countOut, weightsPrepared, bias, output, C, H, W, K,
                             kH, kW, relu, workspace, dtype, stream, 0, nullptr, upstreamCount);
}

// Fine-grained form: `delta` holds the thresholded per-value differences (0 where unchanged); the kernel
// ADDS conv(weights, delta) to `output` at the pixels of the frame mask and, if reluOut is given, keeps
// relu(output) up to date there.
int cbinfer_conv_accumulate_from_mask(const float* delta, uint64_t* frameMasks, int32_t* idxOut,
                                      int32_t* countOut, const void* weightsPrepared, float* output,
                                      float* reluOut, int C, int H, int W, int K, int kH, int kW,
                                      void* workspace, int dtype, cbStream_t stream) {
    if (dtype != CB_F32 && dtype != CB_F32S) return CB_ERR_BADARG;
    return cb_conv_from_mask(delta, frameMasks, idxOut, countOut, weightsPrepared, nullptr, output, C, H, W,
                             K, kH, kW, 0, workspace, dtype, stream, 1, reluOut);
}

// two alternating masks, 16 bytes of control words, and (round 6) the copy of the current frame's mask the self-compacting
// contractions leave at a fixed address (cbinfer_frame_mask_copy_offset)
long cbinfer_frame_mask_bytes(int H, int W) { return 3 * cbinfer_mask_words(H, W) * 8 + 16; }
long cbinfer_frame_mask_copy_offset(int H, int W) { return 2 * cbinfer_mask_words(H, W) * 8 + 16; }
int cbinfer_frame_mask_max_words(void) { return CB_SELFC_MAXW; }

}  // extern "C"

#ifdef CB_STAMP
extern "C" int cbinfer_debug_stamps(void* host, long bytes) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(cb_stamp_buf), (size_t)bytes);
}
extern "C" int cbinfer_debug_stage_clocks(void* host, long bytes) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(cb_stage_clk), (size_t)bytes);
}
extern "C" int cbinfer_debug_clocks(void* host, long bytes) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(cb_stamp_clk), (size_t)bytes);
}
#endif
