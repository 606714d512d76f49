// Row-segment contraction for layers whose filter bank is small (K*C*kH*kW*4 B <= a few hundred KB: the
// 3->16 and 16->64 7x7 layers of the scene-labeling network): gather -> MFMA -> bias/ReLU -> scatter for the
// changed pixels of ONE 64-pixel row segment (= one word of the change bit mask) per workgroup.
//
// Why a second kernel: for these layers the list-based kernel of cb_conv.hip is bound by everything but the
// matrix pipe -- 49 im2col gathers per input value through the texture path, an in-kernel prefix over the
// whole mask to find a tile's pixels, a serial k-loop of one workgroup per 64-pixel tile (3 % / 16 % of the
// fp32 MFMA peak in round 1).  Changed pixels come in horizontal runs, so here
//   * a workgroup owns the changed pixels of one mask word: no compaction, no prefix, no search -- the word
//     IS the work descriptor (empty words exit at once);
//   * the (kH) x (64 + kW - 1) input rows under the segment are staged ONCE per channel in LDS with
//     coalesced row loads (gather redundancy kH*kW -> ~kH), and every B operand of every tap is a
//     ds_read_b32 at `lane base + uniform tap offset`;
//   * the output channels are cut into 16-row chunks, the pixels into 16-wide MFMA tiles
//     (v_mfma_f32_16x16x4_f32, exact f32 fma chain: K = 16 needs no padding to 32).  A workgroup takes up to
//     four chunks (four waves each, so the patch is staged once for 64 output channels) and up to two pixel
//     tiles of the word -- blockIdx.z = (chunk group, pixel half): a full 64-pixel word x 64 channels would
//     keep one CU busy for 12 us -- and the four waves of a chunk split pixel tiles x k-depth, so short
//     runs still use all four SIMDs; partial sums meet in LDS in a fixed order (deterministic);
//   * the A operand (weights) is pre-arranged in MFMA fragment order and streamed straight from L2 into
//     registers, one 16-byte load per lane and four MFMAs, six loads in flight.
// k is ordered (ky, kx, c) with the channel fastest and padded to a multiple of four, so the four k of an
// MFMA step are four channels of one tap: lane quarter q reads plane 4g+q at the same tap offset.
//
// Mask protocol (single mask, no parity): the detection ORs into `bits`; every workgroup reads its word,
// workgroup z == 0 copies it to `maskCopy` (the frame's mask stays available for an on-demand index list,
// cbinfer_compact_bits), and the last of the consumers of a non-zero word (chunk groups x the pixel halves
// it needs) -- found with a per-word arrival counter, no grid-wide hand-shake -- zeroes it for the next
// frame.
#include <stdlib.h>

#include "cb_common.h"

namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));

struct RowParams {
    const float* state;     // [C,H,W] layer state the gather reads (conv2d.py:242)
    const float* wq;        // prepared weights (fragment order)
    const float* bias;      // [K] (null in accumulate mode)
    float* reluOut;         // accumulate mode: optional second plane set receiving relu(out)
    int accumulate;         // fine-grained frame: out += W * delta at the mask's pixels (no bias, no ReLU on out)
    float* out;             // [K,H,W]
    unsigned long long* bits;
    int* arrive;            // per-word arrival counters (only used when gridDim.z > 1), zero between frames
    unsigned long long* maskCopy;
    int C, CP, H, W, K, kH, kW;
    int RS, CS;             // LDS patch: row stride (64 + kW - 1) and plane stride (kH*RS padded to 16 mod 32)
    int S, G, NB;           // k-steps (kH*kW*CP/4), groups of four steps, full blocks of CB_ROW_BG groups
    int relu, wpr;
    int MCH, MCW;           // 16-row output-channel chunks in all / per workgroup (4 waves each)
    int halves;             // 1: one workgroup per word; 2: words with more than 32 pixels are shared by two
    int dbg;                // diagnostic ablations (CBINFER_ROW_DBG): 1 no staging loads, 2 no k-loop, 4 no stores
    // several sequences in one launch (nSeq > 1): blockIdx.y = sequence * H + row, the per-sequence tensors
    // come from this table instead of the fields above
    int nSeq;
    struct {
        const float* state;
        float* out;
        unsigned long long* bits;
        int* arrive;
        unsigned long long* maskCopy;
    } seq[CBINFER_SPLIT_MAX_SEQUENCES];
};

// r-th (0-based) set bit of w, r < popcount(w)
__device__ __forceinline__ int cb_nth_bit(unsigned long long w, int r) {
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

#ifdef CB_ROW_STAMP
// diagnostic build only (make EXTRA=-DCB_ROW_STAMP): per-workgroup phase time stamps (100 MHz constant clock)
__device__ unsigned long long cb_row_stamps[8192 * 8];
#define CB_RSTAMP(i)                                                                              \
    do {                                                                                          \
        const unsigned bid = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;      \
        if (threadIdx.x == 0 && bid < 8192) cb_row_stamps[bid * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define CB_RSTAMP(i)
#endif
#define CB_ROW_MAXW 16           // waves per workgroup: 4 per output-channel chunk, up to 4 chunks
#define CB_ROW_MAXROWS 256
#define CB_ROW_BG 3             // groups (16-byte weight loads per lane) per block
#define CB_ROW_BS (4 * CB_ROW_BG)   // k-steps per block

__host__ __device__ constexpr int cb_row_plane_stride(int kH, int kW) {
    int cs = kH * (64 + kW - 1);
    while (cs % 32 != 16) ++cs;   // lane quarters q and q+1 hit disjoint bank halves
    return cs;
}

typedef floatx4 cb_ablock[CB_ROW_BG];

// one block of CB_ROW_BS k-steps with a COMPILE-TIME shape: every tap offset is an immediate of its
// ds_read_b32, two accumulators alternate so that consecutive MFMAs do not wait for each other
template <int KH, int KW, int Q, int B>
__device__ __forceinline__ void cb_row_block_ct(const float* __restrict__ pl, const cb_ablock& a, int nsteps,
                                                floatx4& acc0, floatx4& acc1) {
    constexpr int RS = 64 + KW - 1, CS = cb_row_plane_stride(KH, KW), S = KH * KW * Q;
#pragma unroll
    for (int j = 0; j < CB_ROW_BS; ++j) {
        constexpr int dummy = 0;
        (void)dummy;
        const int s = B * CB_ROW_BS + j;
        if (s < S) {   // (compile-time after unrolling)
            const int tap = s / Q, grp = s % Q, ky = tap / KW, kx = tap % KW;
            const float bv = pl[ky * RS + kx + grp * 4 * CS];
            if (j & 1)
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j >> 2][j & 3], bv, acc1, 0, 0, 0);
            else
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j >> 2][j & 3], bv, acc0, 0, 0, 0);
        }
    }
    (void)nsteps;
}
template <int KH, int KW, int Q, int B, int NBLK>
struct cb_row_blocks {
    static __device__ __forceinline__ void run(int b, const float* __restrict__ pl, const cb_ablock& a,
                                               floatx4& acc0, floatx4& acc1) {
        if (b == B)
            cb_row_block_ct<KH, KW, Q, B>(pl, a, CB_ROW_BS, acc0, acc1);
        else
            cb_row_blocks<KH, KW, Q, B + 1, NBLK>::run(b, pl, a, acc0, acc1);
    }
};
template <int KH, int KW, int Q, int NBLK>
struct cb_row_blocks<KH, KW, Q, NBLK, NBLK> {
    static __device__ __forceinline__ void run(int, const float* __restrict__, const cb_ablock&, floatx4&,
                                               floatx4&) {}
};

// KH, KW, Q (= padded channels / 4) > 0: shape known at compile time (the hot layers); 0, 0, 0: any shape
template <int KH, int KW, int Q>
__global__ __launch_bounds__(64 * CB_ROW_MAXW) void cb_rowconv_f32_kernel(RowParams pin) {
    cb_touch_kernarg<sizeof(RowParams)>();
    extern __shared__ float lds[];   // patch [CP][CS] | red [waves][64][4]
    constexpr bool CT = KH > 0;
    CB_RSTAMP(0);
    RowParams p = pin;
    int y = blockIdx.y;
    if (pin.nSeq > 1) {
        const int q = (int)blockIdx.y / pin.H;
        y -= q * pin.H;
        p.state = pin.seq[q].state, p.out = pin.seq[q].out, p.bits = pin.seq[q].bits;
        p.arrive = pin.seq[q].arrive, p.maskCopy = pin.seq[q].maskCopy;
    }
    const int tx = blockIdx.x;
    // pixel half (tiles 2nh, 2nh+1; with halves == 1 one workgroup takes all four tiles), chunk group
    const int nh = p.halves == 2 ? (blockIdx.z & 1) : 0, msc = p.halves == 2 ? (blockIdx.z >> 1) : blockIdx.z;
    const int widx = y * p.wpr + tx;
    const unsigned long long wv = p.bits[widx];
    // (the word is the same for every lane: tell the compiler, so that everything derived from it is scalar)
    const unsigned long long word =
        ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(wv >> 32)) << 32) |
        (unsigned)__builtin_amdgcn_readfirstlane((int)wv);
    if (word == 0ull) {   // nothing changed in this row segment
        if (blockIdx.z == 0 && threadIdx.x == 0) p.maskCopy[widx] = 0ull;
        return;
    }
    CB_RSTAMP(1);
    const int pc = __popcll(word);
    const int nT = (pc + 15) >> 4;          // 16-pixel tiles in the word
    if (2 * nh >= nT) return;               // the second half exists only for words with more than 32 pixels
    const int tilesMax = p.halves == 2 ? 2 : 4;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int NW = blockDim.x >> 6, NTH = blockDim.x;
    const int consumers = p.halves == 2 ? ((int)gridDim.z >> 1) * (nT > 2 ? 2 : 1) : (int)gridDim.z;
    int arrived = 0;
    if (t == 0) {
        if (blockIdx.z == 0) p.maskCopy[widx] = word;
        if (consumers > 1)
            arrived = __hip_atomic_fetch_add(p.arrive + widx, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int kH = CT ? KH : p.kH, kW = CT ? KW : p.kW, CP = CT ? 4 * Q : p.CP;
    const int RS = CT ? 64 + KW - 1 : p.RS, CS = CT ? cb_row_plane_stride(KH ? KH : 1, KW ? KW : 1) : p.CS;
    const int S = CT ? KH * KW * Q : p.S, G = (S + 3) >> 2, NB = S / CB_ROW_BS;

    // ---- roles: wave = (chunk, sub); the four subs of a chunk split this half's 1 or 2 pixel tiles x k-parts
    const int mc = msc * p.MCW + (wave >> 2), sub = wave & 3;
    const int nTr = min(tilesMax, nT - 2 * nh);       // pixel tiles of this workgroup: 1..4
    const int nTw = nTr == 3 ? 4 : nTr;               // ... rounded up to a divisor of the four subs
    const int kparts = 4 / nTw;
    const int nt = 2 * nh + sub % nTw, kp = sub / nTw;
    const bool active = mc < p.MCH && nt < nT;
    const int n = nt * 16 + (lane & 15);
    const int xl = cb_nth_bit(word, n < pc ? n : 0);
    const int base = (lane >> 4) * CS + xl;   // + tap offset (ky*RS + kx + 4 g CS) = LDS float index
    const int bBeg = NB * kp / kparts, bEnd = NB * (kp + 1) / kparts, bLast = max(bEnd - 1, bBeg);

    // ---- the weights: buffer loads with ONE per-lane offset and a scalar offset per 16-byte group, so that no load
    //      needs an address register of its own (with 64-bit per-load addresses the compiler recycled the registers
    //      of a load in flight for the next load's address and waited for it: a round trip to memory before the
    //      patch was even requested; round 3) --------------------------------------------------------------------
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.wq, 0, (int)min((long)p.MCH * G * 1024, (long)0x7fffffff), 0x00020000);
    const int wBase = min(mc, p.MCH - 1) * G * 1024, wLane = lane * 16;      // group g: + g * 1024
    cb_ablock a0, a1, a2, qr;
    auto loadA = [&](cb_ablock& dst, int b) {
#pragma unroll
        for (int i = 0; i < CB_ROW_BG; ++i)
            dst[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(
                                                     wrsrc, wLane, wBase + min(b * CB_ROW_BG + i, G - 1) * 1024, 0));
    };
    float bv[4];                   // (and the bias of this lane's four output channels)

    // ---- stage the input rows under the segment: patch[c][ky][j] = state[c][y+ky-ph][64 tx - pw + j] -----
    const int ph = (kH - 1) / 2, pw = (kW - 1) / 2;
    const int HW = p.H * p.W;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.state, 0, p.C * HW * 4, 0x00020000);
    const int rows = CP * kH;     // <= CB_ROW_MAXROWS (checked by cbinfer_rowconv_supported)
    const int x0 = tx * 64 - pw;
    __shared__ int s_rowsrc[CB_ROW_MAXROWS];   // element offset of the row's first pixel in `state`, < 0: a zero row
    __shared__ int s_rowdst[CB_ROW_MAXROWS];   // float offset of the row in the patch
    if (t < rows) {
        const int c = CT ? t / (KH ? KH : 1) : t / kH, ky = t - c * kH;
        const int yy = y + ky - ph;
        s_rowsrc[t] = (c < p.C && yy >= 0 && yy < p.H) ? (c * p.H + yy) * p.W : -1;
        s_rowdst[t] = c * CS + ky * RS;
    }
    // (an LDS-only barrier: __syncthreads() is also a fence and would wait here for the weight and bias loads just
    //  requested -- a whole round trip to memory in front of the staging loads' round trip; round 3 stamps: 1.2 us of
    //  a 6 us workgroup)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    CB_RSTAMP(2);
    if (!(p.dbg & 1)) {
        // the kW-1 columns beyond the 64: TW (power of two >= kW-1) slots per row, flattened over the threads
        int sh = 0;
        while ((1 << sh) < kW - 1) ++sh;
        const int total = kW > 1 ? rows << sh : 0;
        // columns 0..63 of every row: one row per wave and pass, eight passes' loads in flight
        const int xa = x0 + lane;
        const bool oka = xa >= 0 && xa < p.W;
        auto request = [&](int r0, int e0, float (&v)[8], float (&tv)[2]) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int e = min(e0 + NTH * u, max(total - 1, 0));
                const int r = e >> sh, jj = e & ((1 << sh) - 1);
                const int src = s_rowsrc[r], xb = x0 + 64 + jj;
                // an invalid element gets an out-of-range offset: the buffer load returns 0 for it
                tv[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                      rsrc, (src >= 0 && xb < p.W) ? (src + xb) * 4 : (1 << 30), 0, 0));
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = min(r0 + NW * u, rows - 1);
                const int src = s_rowsrc[r];
                v[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                     rsrc, (src >= 0 && oka) ? (src + xa) * 4 : (1 << 30), 0, 0));
            }
        };
        auto deposit = [&](int r0, int e0, const float (&v)[8], const float (&tv)[2]) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int e = e0 + NTH * u;
                const int r = e >> sh, jj = e & ((1 << sh) - 1);
                if (e < total && jj < kW - 1) lds[s_rowdst[r] + 64 + jj] = tv[u];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = r0 + NW * u;
                if (r < rows) lds[s_rowdst[r] + lane] = v[u];
            }
        };
        // The first pass (the only one for 7x7 over 3 or 4 channels) in straight-line code: the patch is requested
        // FIRST, the weights of this wave's first blocks and the bias right behind it -- all of it one round trip,
        // and the patch, the older request, can be consumed while the weights are still on their way.
        {
            float v[8], tv[2];
            request(wave, t, v, tv);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                bv[r] = p.bias ? p.bias[min(min(mc, p.MCH - 1) * 16 + 4 * (lane >> 4) + r, p.K - 1)] : 0.f;
            loadA(qr, NB);                 // the (< CB_ROW_BS) steps beyond the last full block, taken by k-part 0
            loadA(a0, min(bBeg, bLast));
            loadA(a1, min(bBeg + 1, bLast));
            deposit(wave, t, v, tv);
        }
        for (int r0 = wave + 8 * NW, e0 = t + 2 * NTH; r0 < rows || e0 < total; r0 += 8 * NW, e0 += 2 * NTH) {
            float v[8], tv[2];
            request(r0, e0, v, tv);
            deposit(r0, e0, v, tv);
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            bv[r] = p.bias ? p.bias[min(min(mc, p.MCH - 1) * 16 + 4 * (lane >> 4) + r, p.K - 1)] : 0.f;
        loadA(qr, NB);
        loadA(a0, min(bBeg, bLast));
        loadA(a1, min(bBeg + 1, bLast));
    }
    floatx4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();   // patch complete
    CB_RSTAMP(3);

    if (active && !(p.dbg & 2)) {
        const float* pl = lds + base;
        // run-time shape: the walk over the k-steps in (ky, kx, channel group) order, kept in scalar
        // registers; `off` is the LDS offset of the step's tap relative to a lane's base
        const int Qr = CP >> 2, CS4 = 4 * CS;
        int grp = 0, kx = 0, off = 0;
        auto seek = [&](int s) {
            const int tap = s / Qr;
            grp = s - tap * Qr;
            const int ky = tap / kW;
            kx = tap - ky * kW;
            off = ky * RS + kx + grp * CS4;
        };
        auto next = [&]() {   // offset of the current step, then advance
            const int o = off;
            ++grp;
            off += CS4;
            if (grp == Qr) {
                grp = 0;
                off += 1 - Qr * CS4;
                if (++kx == kW) {
                    kx = 0;
                    off += RS - kW;
                }
            }
            return o;
        };
        auto block_rt = [&](const cb_ablock& a, int nsteps) {
#pragma unroll
            for (int j = 0; j < CB_ROW_BS; ++j) {
                if (j < nsteps) {
                    const float bv = pl[next()];
                    if (j & 1)
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j >> 2][j & 3], bv, acc1, 0, 0, 0);
                    else
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j >> 2][j & 3], bv, acc0, 0, 0, 0);
                }
            }
        };
        auto block = [&](int b, const cb_ablock& a) {
            if (CT)
                cb_row_blocks<KH ? KH : 1, KW ? KW : 1, Q ? Q : 1, 0, CT ? (KH * KW * Q) / CB_ROW_BS : 0>::run(
                    b, pl, a, acc0, acc1);
            else
                block_rt(a, CB_ROW_BS);
        };
        if (!CT && bBeg < bEnd) seek(bBeg * CB_ROW_BS);
        // three register sets: the weights of block b+2 are requested while block b is multiplied
        for (int b = bBeg; b < bEnd; b += 3) {
            loadA(a2, min(b + 2, bLast));
            block(b, a0);
            if (b + 1 >= bEnd) break;
            loadA(a0, min(b + 3, bLast));
            block(b + 1, a1);
            if (b + 2 >= bEnd) break;
            loadA(a1, min(b + 4, bLast));
            block(b + 2, a2);
        }
        if (kp == 0 && NB * CB_ROW_BS < S) {   // the steps beyond the last full block
            if (CT) {
                cb_row_block_ct<KH ? KH : 1, KW ? KW : 1, Q ? Q : 1, CT ? (KH * KW * Q) / CB_ROW_BS : 0>(
                    pl, qr, 0, acc0, acc1);
            } else {
                seek(NB * CB_ROW_BS);
                block_rt(qr, S - NB * CB_ROW_BS);
            }
        }
    }
    floatx4 acc = acc0 + acc1;
    CB_RSTAMP(4);

    // ---- sum the k-parts through LDS (fixed order), then bias / ReLU / scatter -------------------------
    float* red = lds + CP * CS;
    if (active && kp > 0) *(floatx4*)(red + (wave * 64 + lane) * 4) = acc;
    __syncthreads();
    CB_RSTAMP(5);
    if (active && kp == 0) {
        for (int q = 1; q < kparts; ++q) acc += *(const floatx4*)(red + ((wave + q * nTw) * 64 + lane) * 4);
        if (n < pc && !(p.dbg & 4)) {
            const int pix = y * p.W + tx * 64 + xl;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mc * 16 + 4 * (lane >> 4) + r;
                if (m < p.K) {
                    float v = acc[r] + bv[r];
                    if (p.accumulate) {
                        v += p.out[(long)m * HW + pix];
                        p.out[(long)m * HW + pix] = v;
                        if (p.reluOut) p.reluOut[(long)m * HW + pix] = v <= 0.f ? 0.f : v;
                    } else {
                        if (p.relu) v = v <= 0.f ? 0.f : v;
                        p.out[(long)m * HW + pix] = v;
                    }
                }
            }
        }
    }
    CB_RSTAMP(6);
    // the last consumer of the word zeroes it (and its counter) for the next frame's detection
    if (t == 0) {
        if (consumers == 1) {
            p.bits[widx] = 0ull;
        } else if (arrived == consumers - 1) {
            p.bits[widx] = 0ull;
            __hip_atomic_store(p.arrive + widx, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// geometry shared by prep and launch
struct RowGeom {
    int CP, RS, CS, S, G, NB, MCH, MCW;
    long ldsBytes;
};
RowGeom row_geom(int C, int K, int kH, int kW) {
    RowGeom g;
    g.CP = (C + 3) / 4 * 4;
    g.RS = 64 + kW - 1;
    g.CS = cb_row_plane_stride(kH, kW);
    g.S = kH * kW * (g.CP / 4);
    g.G = (g.S + 3) / 4;
    g.NB = g.S / CB_ROW_BS;
    g.MCH = (K + 15) / 16;
    g.MCW = g.MCH < 2 ? g.MCH : 2;   // two chunks (8 waves) per workgroup: measured 24 us vs 28 (4) / 33 (1) on 16->64
    g.ldsBytes = ((long)g.CP * g.CS + 4 * g.MCW * 64 * 4) * 4;   // patch + reduce buffer (one float4 per lane and wave)
    return g;
}

// weights [K,C,kH,kW] -> [mchunk][group][lane][4] fragment order; step s = (ky*kW + kx)*(CP/4) + channel group
__global__ __launch_bounds__(256) void cb_rowconv_prep_kernel(const float* __restrict__ w, float* __restrict__ wq,
                                                             int K, int C, int kH, int kW, int CP, int RS,
                                                             int CS, int S, int G, int MCH) {
    const long nW = (long)MCH * G * 256;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int Q = CP / 4;
    if (e >= nW) return;
    const int j = (int)(e & 3), lane = (int)((e >> 2) & 63);
    const long r = e >> 8;
    const int g = (int)(r % G), mc = (int)(r / G);
    const int s = 4 * g + j;
    float v = 0.f;
    if (s < S) {
        const int tap = s / Q, grp = s - tap * Q;
        const int ky = tap / kW, kx = tap - ky * kW;
        const int m = 16 * mc + (lane & 15), c = 4 * grp + (lane >> 4);
        if (m < K && c < C) v = w[(((long)m * C + c) * kH + ky) * kW + kx];
    }
    wq[e] = v;
}

}  // namespace

extern "C" {

// Layers the row-segment kernel takes: fp32, a filter bank of at most 512 KB (it is re-streamed from L2 by
// every workgroup), a patch that fits 64 KB of LDS, kW - 1 <= 64 extra columns.
int cbinfer_rowconv_supported(int C, int K, int kH, int kW) {
    if (C <= 0 || K <= 0 || kH <= 0 || kW <= 0 || kW > 33 || kH > 33) return 0;
    if (kH * kW == 1) return 0;   // 1x1: nothing to share between taps, the list kernel / the fused tail do better
    const RowGeom g = row_geom(C, K, kH, kW);
    if (g.ldsBytes > 60 * 1024 || g.CP * kH > CB_ROW_MAXROWS) return 0;
    if ((long)g.MCH * g.G * 1024 > 512 * 1024) return 0;
    return 1;
}

long cbinfer_rowconv_prepared_bytes(int C, int K, int kH, int kW) {
    const RowGeom g = row_geom(C, K, kH, kW);
    return (long)g.MCH * g.G * 1024;
}

int cbinfer_rowconv_prep_weights(const float* weight, void* prepared, int K, int C, int kH, int kW,
                                 cbStream_t stream) {
    CB_REQUIRE(weight && prepared);
    if (!cbinfer_rowconv_supported(C, K, kH, kW)) return CB_ERR_UNSUPPORTED;
    const RowGeom g = row_geom(C, K, kH, kW);
    const long total = (long)g.MCH * g.G * 256;
    hipLaunchKernelGGL(cb_rowconv_prep_kernel, dim3(cb_div_up(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       weight, (float*)prepared, K, C, kH, kW, g.CP, g.RS, g.CS, g.S, g.G, g.MCH);
    return cb_launch_status();
}

// bits / arrive / maskCopy: cbinfer_mask_words(H,W) entries each (uint64 / int32 / uint64); bits and arrive
// zero on first use and left zero; maskCopy receives this frame's mask.
struct RowBatch {
    int nSeq;
    struct {
        const float* state;
        float* out;
        unsigned long long* bits;
        int* arrive;
        unsigned long long* maskCopy;
    } seq[CBINFER_SPLIT_MAX_SEQUENCES];
};
static int cb_rows_launch(const float* state, uint64_t* bits, int32_t* arrive, uint64_t* maskCopy,
                          const void* prepared, const float* bias, float* output, int C, int H, int W,
                          int K, int kH, int kW, int relu, int accumulate, float* reluOut, cbStream_t stream,
                          const RowBatch* batch = nullptr) {
    CB_REQUIRE(state && bits && arrive && maskCopy && prepared && (bias || accumulate) && output && H > 0 && W > 0);
    if (!cbinfer_rowconv_supported(C, K, kH, kW)) return CB_ERR_UNSUPPORTED;
    if ((long)C * H * W * 4 >= (1l << 30) || H > 65535) return CB_ERR_UNSUPPORTED;
    const RowGeom g = row_geom(C, K, kH, kW);
    RowParams p;
    p.state = state;
    p.wq = (const float*)prepared;
    p.bias = bias;
    p.out = output;
    p.bits = (unsigned long long*)bits;
    p.arrive = arrive;
    p.maskCopy = (unsigned long long*)maskCopy;
    p.C = C;
    p.CP = g.CP;
    p.H = H;
    p.W = W;
    p.K = K;
    p.kH = kH;
    p.kW = kW;
    p.RS = g.RS;
    p.CS = g.CS;
    p.S = g.S;
    p.G = g.G;
    p.NB = g.NB;
    p.relu = relu;
    p.accumulate = accumulate;
    p.reluOut = reluOut;
    p.nSeq = batch ? batch->nSeq : 1;
    if (batch)
        for (int q = 0; q < batch->nSeq; ++q) {
            p.seq[q].state = batch->seq[q].state, p.seq[q].out = batch->seq[q].out;
            p.seq[q].bits = batch->seq[q].bits, p.seq[q].arrive = batch->seq[q].arrive;
            p.seq[q].maskCopy = batch->seq[q].maskCopy;
        }
    p.wpr = cbinfer_mask_words_per_row(W);
    {
        static int dbg = -1;
        if (dbg < 0) {
            const char* e = getenv("CBINFER_ROW_DBG");
            dbg = e ? atoi(e) : 0;
        }
        p.dbg = dbg;
    }
    p.MCH = g.MCH;
    p.MCW = g.MCW;
    {
        static int mcw = -1;
        if (mcw < 0) {
            const char* e = getenv("CBINFER_ROW_MCW");   // tuning aid: chunks per workgroup
            mcw = e ? atoi(e) : 0;
        }
        if (mcw > 0 && mcw <= 4) p.MCW = mcw < g.MCH ? mcw : g.MCH;
    }
    const size_t ldsBytes = ((size_t)g.CP * g.CS + 4 * p.MCW * 64 * 4) * 4;
    // a word with all 64 pixels changed costs 4 tiles x MCW chunks x S MFMA steps on one CU: share heavy
    // words between two workgroups
    p.halves = (long)p.MCW * g.S >= 256 ? 2 : 1;
    if ((long)H * p.nSeq > 65535) return CB_ERR_UNSUPPORTED;
    dim3 grid(p.wpr, H * p.nSeq, p.halves * ((g.MCH + p.MCW - 1) / p.MCW)), block(256 * p.MCW);
    // the hot shapes of the scene-labeling network get their tap offsets as immediates
    if (kH == 7 && kW == 7 && g.CP == 4)
        hipLaunchKernelGGL((cb_rowconv_f32_kernel<7, 7, 1>), grid, block, ldsBytes, (hipStream_t)stream, p);
    else if (kH == 7 && kW == 7 && g.CP == 16)
        hipLaunchKernelGGL((cb_rowconv_f32_kernel<7, 7, 4>), grid, block, ldsBytes, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL((cb_rowconv_f32_kernel<0, 0, 0>), grid, block, ldsBytes, (hipStream_t)stream, p);
    return cb_launch_status();
}

int cbinfer_conv_changed_rows(const float* state, uint64_t* bits, int32_t* arrive, uint64_t* maskCopy,
                              const void* prepared, const float* bias, float* output, int C, int H, int W,
                              int K, int kH, int kW, int relu, cbStream_t stream) {
    CB_REQUIRE(bias != nullptr);
    return cb_rows_launch(state, bits, arrive, maskCopy, prepared, bias, output, C, H, W, K, kH, kW, relu, 0,
                          nullptr, stream);
}

// nSeq sequences in one launch: states[q] / outputs[q] / bits[q] / arrive[q] / maskCopies[q] as above, shared weights
int cbinfer_conv_changed_rows_batched(const float* const* states, uint64_t* const* bits, int32_t* const* arrive,
                                      uint64_t* const* maskCopies, float* const* outputs, int nSeq,
                                      const void* prepared, const float* bias, int C, int H, int W, int K, int kH,
                                      int kW, int relu, cbStream_t stream) {
    CB_REQUIRE(states && bits && arrive && maskCopies && outputs && bias && nSeq >= 1 &&
               nSeq <= CBINFER_SPLIT_MAX_SEQUENCES);
    RowBatch b;
    b.nSeq = nSeq;
    for (int q = 0; q < nSeq; ++q) {
        CB_REQUIRE(states[q] && bits[q] && arrive[q] && maskCopies[q] && outputs[q]);
        b.seq[q].state = states[q], b.seq[q].out = outputs[q], b.seq[q].bits = (unsigned long long*)bits[q];
        b.seq[q].arrive = arrive[q], b.seq[q].maskCopy = (unsigned long long*)maskCopies[q];
    }
    return cb_rows_launch(states[0], bits[0], arrive[0], maskCopies[0], prepared, bias, outputs[0], C, H, W, K, kH,
                          kW, relu, 0, nullptr, stream, &b);
}

int cbinfer_conv_accumulate_rows(const float* delta, uint64_t* bits, int32_t* arrive, uint64_t* maskCopy,
                                 const void* prepared, float* output, float* reluOut, int C, int H, int W,
                                 int K, int kH, int kW, cbStream_t stream) {
    return cb_rows_launch(delta, bits, arrive, maskCopy, prepared, nullptr, output, C, H, W, K, kH, kW, 0, 1,
                          reluOut, stream);
}

}  // extern "C"

#ifdef CB_ROW_STAMP
// (diagnostic build) resident workgroups per CU of the 7x7 form at `threads` threads and `ldsBytes` of dynamic LDS
extern "C" int cbinfer_debug_row_occupancy(int threads, long ldsBytes) {
    int n = -1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, cb_rowconv_f32_kernel<7, 7, 1>, threads, (size_t)ldsBytes) != hipSuccess)
        return -1;
    return n;
}
extern "C" int cbinfer_debug_row_stamps(void* host, long bytes, int clear) {
    if (clear) {
        void* d = nullptr;
        if (hipGetSymbolAddress(&d, HIP_SYMBOL(cb_row_stamps)) != hipSuccess) return -1;
        return (int)hipMemset(d, 0, sizeof(cb_row_stamps));
    }
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(cb_row_stamps), (size_t)bytes);
}
#endif
