// Split-state change-based convolution for gfx950 (round 3): the fused a1 + a5..a8 path of a feedback-mode
// CBConv2d (conv2d.py:178-259; kernels cbconv2d_cg_backend.cu:40-81, :138-197) for layers of 16/32/64 input
// channels, built around ONE idea: the layer state is kept a second time in exactly the form the matrix
// cores eat.
//
//   state    beside prevInput [C,H,W] f32 (the module's buffer, unchanged), a pixel-major PRE-SPLIT copy
//            S[Hp][Wp][C/16][hi|lo][16] of f16 pairs: x * 2^-4 = hi + lo * 2^-11 (22-23 significant bits; the
//            detection kernel refreshes both at the changed pixels only -- feedback mode).  S has a zero
//            border of the filter's half size, so a tap needs no bounds test, and a block of zero rows that
//            list slots past the end point at.
//   weights  pre-split the same way (power-of-two scale to [2^13, 2^14)), stored in MFMA fragment order per
//            32-k stage, so a stage of an m-tile is one contiguous piece of memory.
//   gather   both operands go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds), 1 KB per wave
//            instruction, no register staging, no vector ALU work and no LDS store traffic per k: a pixel's
//            stage is 128 contiguous bytes of its record (two 16-channel groups of one tap, or -- 16 channels
//            -- two x-adjacent taps), fetched by 8 lanes; an XOR swizzle of the 16-byte chunks makes the
//            ds_read_b128 fragment reads of the [pixel][128 B] image conflict-free.
//   product  v_mfma_f32_32x32x16_f16, three per 16 k: hi.hi into one accumulator, hi.lo + lo.hi into a
//            second one (scaled 2^-11 at the end).  Per product: two operand roundings of <= 2^-23 each and the
//            dropped lo.lo term (<= 2^-24): <= 5 * 2^-24 |a||b| -- NARROWER than f32 operands (24 bits); measured
//            beside the exact f32 chain in tests/test_gpu_split.py::test_split_hostile_data_accuracy_fullsize -- at
//            HALF the matrix work and two thirds of the operand bytes of bf16x3.  A state value beyond the pair's
//            range (|x| >= 2^20) raises the layer's flag and the contraction computes the layer in plain f32.
//   ring     four (eight: the small tile alone on a CU) LDS stage buffers, all but one in flight, one s_barrier per stage, waits counted by
//            hand (s_waitcnt vmcnt(N): the compiler does not track LDS-DMA against LDS reads).
//   schedule persistent grid over (pixel tile, m-tile, k-slice) items of up to CBS_MAXSEQ independent sequences
//            in ONE launch (blockIdx carries no meaning: each workgroup derives every sequence's list length
//            from its change mask); a short list is split along k, slices meet in a second launch.
#include <stdlib.h>

#include <type_traits>

#include "cb_common.h"

#ifdef CBS_STAMP
// diagnostic build only (make EXTRA=-DCBS_STAMP; tools/split_stamps.py): per-workgroup phase stamps, 100 MHz constant
// clock: [64-row tile | 128-row tile | reduce+tail launch | pooled detection] x 2048 workgroups x 16 marks
__device__ unsigned long long cbs_stamp_buf[4 * 2048 * 16];
#define CB_TAIL_STAMP(i)                                                                 \
    do {                                                                                 \
        if (threadIdx.x == 0 && blockIdx.x < 2048)                                       \
            cbs_stamp_buf[(2 * 2048 + blockIdx.x) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#define CBS_DET_STAMP(i)                                                                                            \
    do {                                                                                                            \
        const unsigned wg = blockIdx.y * gridDim.x + blockIdx.x;                                                    \
        if (POOL && threadIdx.x == 0 && blockIdx.z == 0 && wg < 2048)                                               \
            cbs_stamp_buf[(3 * 2048 + wg) * 16 + (i)] = __builtin_amdgcn_s_memrealtime();                            \
    } while (0)
#else
#define CBS_DET_STAMP(i)
#endif
#include "cb_tail_core.h"
#include "cb_split_common.h"

namespace cbs {

typedef __attribute__((address_space(3))) void* cb_lds_ptr;
typedef __attribute__((address_space(4))) const int cbs_const_int;

// ---------------------------------------------------------------------------------------------------
// weights: [stage][row tile of 32][k-step][plane][lane][8 f16] (MFMA A-fragment order) + the stage table
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cbs_prep_kernel(const float* __restrict__ w, halfx8* __restrict__ A,
                                                      int* __restrict__ stageOff, float* __restrict__ wPlain,
                                                      CbsGeom g, int K, int KP, float wscale) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    // (the plain f32 filter bank, as it came, behind the stage table: what the exact path of a layer whose state left
    //  the f16 pair's range multiplies with -- cbs_exact_tile; the bf16-triple form has f32's range and no such path)
    if (wPlain)
        for (long i = idx; i < (long)K * g.C * g.kH * g.kW; i += (long)gridDim.x * blockDim.x) wPlain[i] = w[i];
    const int PL = g.planes;      // 2: f16 pairs, 3: bf16 triples -- blocks [stage][row tile][k-step][plane], 1 KB each
    if (idx < g.nStages) {
        const int s = (int)idx;
        int off;
        if (g.pair) {
            off = ((s / g.kWs) * g.Wp + 2 * (s % g.kWs)) * g.rec;
        } else {
            const int tap = s / (g.G / 2), sub = s % (g.G / 2);
            off = ((tap / g.kW) * g.Wp + tap % g.kW) * g.rec + sub * 64 * PL;
        }
        stageOff[s] = off;
    }
    const int RT = KP / 32;
    const long total = (long)g.nStages * RT * 2 * PL * 64;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    const long blk = idx >> 6;
    const int plane = (int)(blk % PL), ks = (int)((blk / PL) & 1);
    const int rt = (int)((blk / (2 * PL)) % RT), stage = (int)((blk / (2 * PL)) / RT);
    const int m = rt * 32 + (lane & 31);
    halfx8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int kl = 8 * (lane >> 5) + j;
        int ky, kx, c;
        if (g.pair) {
            ky = stage / g.kWs, kx = 2 * (stage % g.kWs) + ks, c = kl;
        } else {
            const int tap = stage / (g.G / 2), sub = stage % (g.G / 2);
            ky = tap / g.kW, kx = tap % g.kW, c = (sub * 2 + ks) * 16 + kl;
        }
        float v = 0.f;
        if (m < K && kx < g.kW) v = w[(((long)m * g.C + c) * g.kH + ky) * g.kW + kx] * wscale;
        if (PL == 3) {      // (wscale is 1: the triple carries the weight as it is)
            __bf16 t[3];
            cbs_split3(v, t[0], t[1], t[2]);
            o[j] = __builtin_bit_cast(_Float16, t[plane]);
            continue;
        }
        _Float16 hi, lo;
        cbs_split(v, hi, lo);
        o[j] = plane ? lo : hi;
    }
    A[idx] = o;
}

// split state: zero everywhere, +inf (hi plane) at the image pixels -- the f32 state starts as +inf
// (conv2d.py:192-199), and a pixel the detection never refreshes must not contribute finite numbers either
__global__ __launch_bounds__(256) void cbs_state_init_kernel(uint4* __restrict__ S, CbsGeom g) {
    const long chunks = (long)g.Hp * g.Wp * (g.rec / 16);
    if (blockIdx.x == 0) S[threadIdx.x] = make_uint4(0u, 0u, 0u, 0u);     // the CBS_SPAD bytes in front (256 x 16)
    S += CBS_SPAD / 16;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < chunks; i += (long)gridDim.x * blockDim.x) {
        const long pix = i / (g.rec / 16);
        const int q = (int)(i % (g.rec / 16));        // chunk within the record: [group][plane (2 or 3)][half]
        const int py = (int)(pix / g.Wp), px = (int)(pix % g.Wp);
        const bool inside = py >= g.padY && py < g.padY + g.H && px >= g.padXL && px < g.padXL + g.W;
        const bool hiPlane = ((q >> 1) % g.planes) == 0;
        const unsigned v = (inside && hiPlane) ? (g.planes == 3 ? 0x7f807f80u : 0x7c007c00u) : 0u;      // f16 / bf16 +inf
        S[i] = make_uint4(v, v, v, v);
    }
}

// the split state made again from the f32 state (after the module's prevInput was written from outside: a restored
// state, eval03.py:88-95): one thread per (pixel, 8-channel part); the border and the dummy rows are left alone
__global__ __launch_bounds__(256) void cbs_state_rebuild_kernel(const float* __restrict__ state, char* __restrict__ S,
                                                               CbsGeom g, int* rangeFlag) {
    const long HW = (long)g.H * g.W;
    const int parts = g.C / 8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < HW * parts; i += (long)gridDim.x * blockDim.x) {
        const int part = (int)(i / HW);
        const long pix = i % HW;                       // (consecutive threads: consecutive pixels of one channel)
        const int y = (int)(pix / g.W), x = (int)(pix % g.W);
        const int grp = part >> 1, half = part & 1;
        if (g.planes == 3) {
            float v8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v8[j] = state[(long)(grp * 16 + half * 8 + j) * HW + pix];
            cbs_store_part(S + CBS_SPAD + ((long)(y + g.padY) * g.Wp + (x + g.padXL)) * g.rec, grp, half, 3, v8);
            continue;
        }
        halfx8 hi, lo;
        bool over = false;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = state[(long)(grp * 16 + half * 8 + j) * HW + pix] * CBS_XSCALE;
            over |= !(fabsf(v) <= CBS_F16_MAX) && v == v && fabsf(v) != INFINITY;
            _Float16 h, l;
            cbs_split(v, h, l);
            if (fabsf(v) == INFINITY) l = (_Float16)0;     // the initial +inf state: as cbs_state_init_kernel leaves it
            hi[j] = h, lo[j] = l;
        }
        char* rec = S + CBS_SPAD + ((long)(y + g.padY) * g.Wp + (x + g.padXL)) * g.rec + grp * 64 + half * 16;
        *(halfx8*)rec = hi;
        *(halfx8*)(rec + 32) = lo;
        if (over && rangeFlag) *rangeFlag = 1;
    }
}

// ---------------------------------------------------------------------------------------------------
// change detection (+ 2x2 max pool folded in) + dilation + feedback refresh of BOTH states
// ---------------------------------------------------------------------------------------------------
struct CbsDetSeq {
    const float* in;                      // [C,H,W], or with POOL the pool's input [C,pH,pW]
    float* state;                         // prevInput [C,H,W]
    char* S;                              // split state
    unsigned long long* masks;            // frame mask: [words], zeroed by the contraction (cbSplitSeq.frameMasks)
    const unsigned long long* prodMask;   // POOL: the producing layer's change mask of this frame, or null
    int* rangeFlag;                       // set to 1 when a refreshed value leaves the f16 pair's range
    float* delta;                         // fine-grained mode: the thresholded differences [C,H,W] (f32, every value)
};
struct CbsDetArgs {
    CbsDetSeq seq[CBS_MAXSEQ];
    int W, H, C, kHH, kWH, wpr, pH, pW, Wp, rec, padY, padXL;
    long words;
    float th;
    int planes;       // 2: the records are f16 pairs, 3: bf16 triples (cb_split_common.h)
    int copyAll;      // the layer is NOT in feedback mode and keeps a copy of its input (conv2d.py:234-236): both
                      // states take EVERY value of the frame, not only those of the changed pixels (round 4)
    int fg;           // fine-grained frame (conv2d.py:160-176, cbconv2d_fg_backend.cu:7-23; implies copyAll): prevInput
                      // takes the frame, the PRE-SPLIT copy takes d = in - prev where |d| > th and 0 elsewhere -- the
                      // operand of out += W * delta -- and so does the f32 delta tensor (the exact path's operand)
};

__device__ __forceinline__ unsigned long long cbs_valid_mask(int W, int tile) {
    const int rem = W - tile * 64;
    return rem >= 64 ? ~0ull : ((1ull << rem) - 1ull);
}

// One workgroup = one 64-pixel row segment x all channels of one sequence (blockIdx.z); wave g owns channels
// g, g+G, g+2G, g+3G (G = C/4 waves).  Decomposition, predicate (strict >, cbconv2d_cg_backend.cu:56), dilation
// and producer-mask shortcut are cb_detect_kernel's (cb_detect.hip); what is new is the second state: the new
// values of the segment go through LDS ([channel][pixel]) and come out as whole pixel records, 16 bytes per
// lane, for the changed pixels only.
template <bool POOL>
__global__ __launch_bounds__(1024) void cbs_detect_kernel(CbsDetArgs a) {
    cb_touch_kernarg<sizeof(CbsDetArgs)>();
    const CbsDetSeq sq = a.seq[blockIdx.z];
    const int W = a.W, H = a.H, C = a.C, pH = a.pH, pW = a.pW;
    CBS_DET_STAMP(0);
    if (POOL && sq.prodMask) {
        const int pwpr = (pW + 63) >> 6;
        unsigned long long any = 0ull;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) {      // (clamped, not predicated: four loads in one round trip)
                const int yy = 2 * (int)blockIdx.y + j, ww = 2 * (int)blockIdx.x + i;
                const unsigned long long v = sq.prodMask[(long)min(yy, pH - 1) * pwpr + min(ww, pwpr - 1)];
                any |= (yy < pH && ww < pwpr) ? v : 0ull;
            }
        if (__builtin_amdgcn_readfirstlane((int)(any != 0ull)) == 0) return;
    }
    CBS_DET_STAMP(1);
    unsigned long long* bits = sq.masks;      // (single mask: the contraction zeroes it once every workgroup has it)
    const int lane = threadIdx.x & 63;
    const int g = threadIdx.x >> 6;
    const int G = blockDim.x >> 6;
    const int tx = blockIdx.x, y = blockIdx.y;
    const int x = tx * 64 + lane;
    const bool valid = x < W;
    const long HW = (long)H * W;
    const long p = (long)y * W + x;
    const float* __restrict__ in = sq.in;
    float* state = sq.state;

    const long pHW = (long)pH * pW;
    const int py0 = 2 * y, px0 = 2 * x;
    // The 2x2 window by four UNCONDITIONAL loads with clamped coordinates (a window cut off by the map's edge reads a
    // pixel twice: max(a, a) = a).  A per-lane `if (inside) ... else ...` around the loads is a branch region of its
    // own for every channel, and the compiler then waits for one channel's loads before it requests the next: four
    // round trips instead of one (2.6 of the 4.5 us of this kernel's per-workgroup chain, round 3 stamps).
    const int px1 = min(px0 + 1, pW - 1) - px0, py1 = (min(py0 + 1, pH - 1) - py0) * pW;
    auto ldin = [&](int c) -> float {
        if (!POOL) return in[(long)c * HW + p];
        const float* q = in + (long)c * pHW + (long)py0 * pW + px0;
        return fmaxf(fmaxf(q[0], q[px1]), fmaxf(q[py1], q[py1 + px1]));
    };

    // C == 4 G: four channels per wave, eight independent loads in flight per lane
    bool chg = false;
    unsigned dbits = 0xfu;      // copy-all form: which of this lane's four values differ from the state bit for bit
    float k0 = 0.f, k1 = 0.f, k2 = 0.f, k3 = 0.f;
    if (valid) {
        const float s0 = state[(long)g * HW + p], x0 = ldin(g);
        const float s1 = state[(long)(g + G) * HW + p], x1 = ldin(g + G);
        const float s2 = state[(long)(g + 2 * G) * HW + p], x2 = ldin(g + 2 * G);
        const float s3 = state[(long)(g + 3 * G) * HW + p], x3 = ldin(g + 3 * G);
        const bool c0 = cb_changed(s0, x0, a.th), c1 = cb_changed(s1, x1, a.th), c2 = cb_changed(s2, x2, a.th),
                   c3 = cb_changed(s3, x3, a.th);
        chg = c0 | c1 | c2 | c3;
        k0 = x0, k1 = x1, k2 = x2, k3 = x3;
        if (a.copyAll && !a.fg)
            dbits = (unsigned)cb_differs(s0, x0) | (cb_differs(s1, x1) << 1) | (cb_differs(s2, x2) << 2) |
                    (cb_differs(s3, x3) << 3);
        if (a.fg) {      // per VALUE: the state takes the input, the records (and the delta tensor) the difference
            float* dl = sq.delta;
            state[(long)g * HW + p] = x0, state[(long)(g + G) * HW + p] = x1;
            state[(long)(g + 2 * G) * HW + p] = x2, state[(long)(g + 3 * G) * HW + p] = x3;
            k0 = c0 ? x0 - s0 : 0.f, k1 = c1 ? x1 - s1 : 0.f, k2 = c2 ? x2 - s2 : 0.f, k3 = c3 ? x3 - s3 : 0.f;
            dl[(long)g * HW + p] = k0, dl[(long)(g + G) * HW + p] = k1;
            dl[(long)(g + 2 * G) * HW + p] = k2, dl[(long)(g + 3 * G) * HW + p] = k3;
        }
    }

    __shared__ unsigned long long sm[16], smd[16];
    __shared__ float T[64][65];
    const unsigned long long b = __ballot(chg), bd = __ballot(valid && dbits != 0u);
    if (lane == 0) sm[g] = b, smd[g] = bd;
    __syncthreads();
    unsigned long long m = 0, md = 0;
    for (int i = 0; i < G; ++i) m |= sm[i], md |= smd[i];
    CBS_DET_STAMP(2);
    if (m == 0 && !a.copyAll) return;   // uniform over the workgroup
    // the pixels whose states take the new values: the changed ones (feedback mode), or -- copy-all -- those of the
    // segment that hold a value that differs from the state bit for bit (behind another change-based layer the frame is
    // bit-identical to the last one wherever that layer recomputed nothing); fine-grained: all (the records hold deltas)
    const unsigned long long upd = a.fg ? cbs_valid_mask(W, tx) : (a.copyAll ? md : m);
    if (upd == 0) return;               // (copy-all, nothing differs: nothing above the threshold either)

    // feedback: refresh the f32 state at the (pre-dilation) changed pixels only (.cu:74-80) ...
    if (!a.fg && ((upd >> lane) & 1ull)) {
        if (dbits & 1u) state[(long)g * HW + p] = k0;
        if (dbits & 2u) state[(long)(g + G) * HW + p] = k1;
        if (dbits & 4u) state[(long)(g + 2 * G) * HW + p] = k2;
        if (dbits & 8u) state[(long)(g + 3 * G) * HW + p] = k3;
    }
    // ... and the split state: [channel][pixel] through LDS, out as records
    T[g][lane] = k0;
    T[g + G][lane] = k1;
    T[g + 2 * G][lane] = k2;
    T[g + 3 * G][lane] = k3;
    __syncthreads();
    CBS_DET_STAMP(3);
    {
        const int parts = C / 8;                 // 8-channel parts per pixel: (group, half)
        const int t = threadIdx.x;
        if (t < 64 * parts) {
            const int pl = t / parts, part = t % parts;
            if (((upd >> pl) & 1ull) != 0ull) {
                const int grp = part >> 1, half = part & 1;
                float v8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v8[j] = T[grp * 16 + half * 8 + j][pl];
                char* rec = sq.S + CBS_SPAD + ((long)(y + a.padY) * a.Wp + (tx * 64 + pl + a.padXL)) * a.rec;
                const bool over = cbs_store_part(rec, grp, half, a.planes, v8);
                if (over && sq.rangeFlag) *sq.rangeFlag = 1;
            }
        }
    }

    CBS_DET_STAMP(4);
    if (m == 0) return;
    // dilation of the 64-pixel word (+ the parts spilling into the neighbour words), ORed into the frame mask
    unsigned long long D = m, SR = 0, SL = 0;
    for (int d = 1; d <= a.kWH; ++d) {
        D |= (m << d) | (m >> d);
        SR |= m >> (64 - d);
        SL |= m << (64 - d);
    }
    D &= cbs_valid_mask(W, tx);
    SR = (tx + 1 < a.wpr) ? (SR & cbs_valid_mask(W, tx + 1)) : 0ull;
    if (tx == 0) SL = 0;
    if (g == 0) {
        const int items = 3 * (2 * a.kHH + 1);
        for (int i = lane; i < items; i += 64) {
            const int yy = y + i / 3 - a.kHH;
            const int which = i % 3;
            if (yy < 0 || yy >= H) continue;
            const unsigned long long v = which == 0 ? D : (which == 1 ? SR : SL);
            const int t2 = which == 0 ? tx : (which == 1 ? tx + 1 : tx - 1);
            if (v) atomicOr(&bits[(long)yy * a.wpr + t2], v);
        }
    }
    CBS_DET_STAMP(5);
}

// ---------------------------------------------------------------------------------------------------
// the contraction
// ---------------------------------------------------------------------------------------------------
struct CbsSeq {
    const char* S;
    const float* state;               // prevInput [C,H,W] f32 (read by the exact path only)
    const int* rangeFlag;             // != 0: the split state is not usable (a value left the f16 pair's range)
    float* out;                       // prevOutput [K,H,W]
    unsigned long long* masks;        // frame masks
    int32_t* listOut;                 // change list (by-product), capacity H*W
    int32_t* countOut;
    unsigned long long* maskCopy;     // optional: this frame's change mask, at a fixed address
    float* reluOut;                   // accumulate form only, optional: second plane set kept at relu(out)
};
struct CbsParams {
    CbsSeq seq[CBS_MAXSEQ];
    int nSeq;
    const char* A;                    // prepared weights
    const int* stageOff;
    const float* wPlain;              // the f32 filter bank [K,C,kH,kW] (exact path)
    const float* bias;
    float* slabs;                     // split-K workspace: slabCap partial tiles of BM x BN floats
    int* info;                        // left for the reduce launch: {SK, MT, nSeq, tilesP[q]..., N[q]...}
    int slabCap;                      // partial tiles the workspace holds (cbs_slab_capacity)
    int K, KP, H, W, Wp, rec, nStages, maskWords, wpr, relu, dummyBase, kH, kW;
    long stateBytes, aBytes;
    float outScale;
    unsigned long long magicMW, magicWpr, magicW, magicMT;   // floor(2^32 / d) + 1: x / d = (x * magic) >> 32 for x d < 2^32
    int arriveShards;                 // arrival counters in the second mask slot (lines of their own): min(8, MW / 16)
    int forceSK;                      // > 0: tuning / test aid
    int accumulate;                   // fine-grained frame: out += W * delta (no bias, no ReLU on `out`; reluOut beside it)
    int splitRounds;                  // a deep contraction is split along k while its work items <= splitRounds x grid
    int halfOut;                      // fp16 layer (cbs_conv_kernel<..., HALF>): bias and outputs are f16
    const int* upstream;              // optional (one sequence): the producing layer's change count of this frame; 0 ends
                                      // the launch at once (cbinfer_cbconv2d_forward_after's contract)
    int dbg;                          // diagnostic ablations (builds with -DCBS_DBG only; CBINFER_SPLIT_DBG)
    int maxChunks;                    // k-chunks of a deep contraction: CBS_CHUNKS; fp16 layers: up to 16 (chosen on the device)
#ifdef CBS_LAST_ARRIVER
    int lastArrive;                   // experiment (builds with -DCBS_LAST_ARRIVER, CBINFER_SPLIT_LAST=1; x3, 128-row tile): the k-slices of a tile meet INSIDE
                                      // the launch -- the last one to arrive sums the others' partial tiles and scatters
    int* arriveTiles;                 // ... one arrival counter per (pixel tile, row tile), zero between launches
#endif
};
// a wave-uniform pointer the compiler may not know to be uniform, into scalar registers (the "s" operand of the saddr
// forms of global_load / global_store in inline asm)
static __device__ __forceinline__ const void* cbs_uniform_ptr(const void* ptr) {
    const unsigned long a = (unsigned long)ptr;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    return (const void*)(((unsigned long)hi << 32) | lo);
}
// fp16 layers only (AR = 1; round 6): what differs between the layers of a GROUP -- the two branches of an OpenPose
// stage issued as one persistent grid: CbsParams.seq[q] holds layer q's tensors, this its weights, bias, channel count
// -- and the CONSUMERS of each layer's output whose change detection (copy mode: feedbackLoop = False, copyInput = True,
// conv2d.py:234-236; predicate cbconv2d_cg_half_backend.cu:24-35) rides in this launch's epilogue.  A second kernel
// argument of the fp16 instances only: the fp32 instances keep their argument block (and their scalar registers).
#define CBH_GROUP CBINFER_HGROUP_MAX
#define CBH_NEXT CBINFER_HNEXT_MAX
struct CbhNext {
    _Float16* state;                  // consumer's prevInput [K,H,W]
    char* S;                          // its pixel-major copy (cbh_geom(K, H, W, kH, kW))
    unsigned long long* masks;        // its frame mask
    int kHH, kWH, Wp, rec, padY, padXL;
    float th;
    int pad_;
};
struct CbhLayer {
    const char* A;                    // prepared weights of this layer (the stage table behind them depends on the
                                      // geometry only: the group shares CbsParams.stageOff)
    const _Float16* bias;
    int K, relu, nNext, pad_;
    CbhNext next[CBH_NEXT];
};
struct CbhExt {
    CbhLayer L[CBH_GROUP];
    int slabCap1;                     // partial tiles the workspace of ONE layer alone would hold (the chunk rule's bound)
    int pad_;
};
struct CbsNoExt {};
// AR = 3 ("x3" arithmetic + WINDOW order, round 6): the feedback-mode layer behind the 2x2 max pool that follows this
// layer -- its pooled change detection (CBPoolMax2d, conv2d.py:49-78, + changeDetection with updateInputState,
// cbconv2d_cg_backend.cu:40-81; what cbs_detect_kernel<POOL> does in a launch of its own) rides in this launch's epilogue,
// as it does in cbp_rowpair_kernel's.  A second kernel argument of that one instance only.
struct CbsWinExt {
    float* nstate;                    // the consumer's prevInput [K, H2, W2]
    char* nS;                         // its split state (pixel-major records)
    unsigned long long* nmasks;       // its frame mask (the detection ORs into it)
    int* nflag;                       // its range flag (f16 pairs) or null
    int H2, W2, wpr2, kHH, kWH, Wp, rec, padY, padXL, planes;
    float th;
    unsigned long long magicW2;
    // a side job for workgroups without a work item (round 6): the feedback refresh of ANOTHER layer's state -- the
    // row-pair layer in front, whose launch detected its changes itself and had to leave its state alone
    // (cbinfer_conv_rowpairs_detect): state[:, p] = frame[:, p] wherever some channel differs by more than the threshold
    const float* sideFrame;           // null: no side job
    float* sideState;
    int sideC, sideHW;
    float sideTh;
};
template <int AR>
struct CbsExtOf {
    typedef CbsNoExt type;
};
template <>
struct CbsExtOf<1> {
    typedef CbhExt type;
};
template <>
struct CbsExtOf<3> {
    typedef CbsWinExt type;
};
// AR = 4: the bf16-triple arithmetic in PIXEL order + the side job alone (the consumer of a self-detecting row-pair layer
// when it does not run in window order: section 5.9)
struct CbsSideExt {
    const float* sideFrame;
    float* sideState;
    int sideC, sideHW;
    float sideTh;
};
template <>
struct CbsExtOf<4> {
    typedef CbsSideExt type;
};

// One pixel of a consumer's dilated change mask: rows y - kHH .. y + kHH, columns x - kWH .. x + kWH, clipped to the map
// (what cbs_detect_kernel's word shifts produce for a single set bit)
__device__ __forceinline__ void cbh_or_dilated(unsigned long long* masks, int y, int x, int H, int W, int wpr, int kHH,
                                               int kWH) {
    const int lo = max(x - kWH, 0), hi = min(x + kWH, W - 1);
    const int w0 = lo >> 6, w1 = hi >> 6;
    const unsigned long long m0 = (~0ull << (lo & 63)) & (w1 == w0 ? (~0ull >> (63 - (hi & 63))) : ~0ull);
    const unsigned long long m1 = ~0ull >> (63 - (hi & 63));
    for (int yy = max(y - kHH, 0); yy <= min(y + kHH, H - 1); ++yy) {
        atomicOr(&masks[(long)yy * wpr + w0], m0);
        if (w1 != w0) atomicOr(&masks[(long)yy * wpr + w1], m1);
    }
}

// diagnostic ablations are a build option (make EXTRA=-DCBS_DBG; tools/split_dbg_run.sh): 1 every pixel-operand
// DMA reads the dummy pixel, 2 no fragment reads / MFMAs, 4 no DMA at all, 8 no pixel-operand fragment reads,
// 16 no weight fragment reads, 32 no epilogue stores, 64 every DMA issued dead (no memory traffic), 128 / 256 the weight /
// pixel DMAs dead, 512 (x3) the second column tile's fragments are not read (a third of the LDS read volume less)
#ifdef CBS_STAMP
// diagnostic build only (make EXTRA=-DCBS_STAMP; tools/split_stamps.py): per-workgroup phase stamps of its FIRST item,
// 100 MHz constant clock: 0 entry, 1 list lengths known, 2 item set up, 3 ring primed, 4 stage loop done, 5 epilogue
// done; 6 / 7: shader clock (s_memtime) at entry / after the stage loop; 8.. finer prologue marks (see tools/split_stamps.py)
#define CBS_STAMP_AT(i)                                                                                   \
    do {                                                                                                  \
        if (threadIdx.x == 0 && blockIdx.x < 2048 && cbs_first)                                           \
            cbs_stamp_buf[(BM >= 128 ? 2048 * 16 : 0) + blockIdx.x * 16 + (i)] =                            \
                ((i) == 6 || (i) == 7) ? __builtin_amdgcn_s_memtime() : __builtin_amdgcn_s_memrealtime();               \
    } while (0)
#else
#define CBS_STAMP_AT(i)
#endif
#ifdef CBS_DBG
#define CBS_DBGBIT(b) (p.dbg & (b))
#else
#define CBS_DBGBIT(b) false
#endif
#define CBS_INFO_SK 0
#define CBS_INFO_MT 1
#define CBS_INFO_TP 4                  // tilesP[q] (pixel tiles of sequence q), then N[q]
#define CBS_INFO_CH 24                 // fp16 group: k-chunks of layer q (behind the 2 x CBS_MAXSEQ ints above)

// One LDS-DMA instruction: every lane moves 16 bytes from (resource base + voff + soff) to lds + 16 * lane.
// A NON-template wrapper on purpose: inside a kernel template the target builtin is a dependent call that the HOST
// pass of the compilation cannot resolve, and clang then drops the kernel's host stub without a diagnostic.
// OFF (immediate, < 4096) is added to the LDS address AND to the memory address.
template <int OFF>
__device__ __forceinline__ void cbs_dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (cb_lds_ptr)lds, 16, voff, soff, OFF, 0);
}

// A fragment read the COMPILER DOES NOT SEE as an LDS access: it would otherwise put a full s_waitcnt vmcnt(0) in
// front of every read that may alias an LDS-DMA in flight -- i.e. drain the ring every stage.  The result
// registers are valid only behind an s_waitcnt lgkmcnt that names them (CBS_STEP's wait does).
template <int OFF>
__device__ __forceinline__ halfx8 cbs_lds_read16(unsigned addr) {
    halfx8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}

// x / d for x d < 2^32 by the precomputed magic = floor(2^32 / d) + 1 (a 64-bit product: d = 1 included)
__device__ __forceinline__ int cbs_div(int x, unsigned long long magic) {
    return (int)(((unsigned long long)(unsigned)x * magic) >> 32);
}

// BM x BN output tile per workgroup, WM x WN waves, each wave TN = BN/WN/32 column tiles of one 32-row tile.
// HALF (round 4): the fp16 layers (cbconv2d_cg_half_backend.cu:146-197): tensors, state and weights are f16 as they
// come -- ONE plane, a stage is 64 k = the same 128 bytes of a pixel's record and the same 4 KB weight block per row
// tile, so DMA, ring, swizzle and fragment addresses are the f16-pair form's; what differs is the matrix work of a
// stage (four k-steps of one product instead of two of three), no scale, and f16 outputs.
//
// AR = 2, bf16 TRIPLES ("x3", round 5): the f32-EQUIVALENT form.  Every f32 operand is three bf16 terms, exactly
// (cbs_split3); a product is the six term products above 2^-24: b0 w0 into the main accumulator, b0 w1 + b1 w0 + b1 w1
// + b0 w2 + b2 w0 into a second one (its roundings are relative to a sum 2^-8 of the main one's), summed at the end.
// What is dropped (b1 w2 + b2 w1 + b2 w2) is below 2^-25 |x w| per product; the arithmetic error left is the f32
// accumulation of one 16-k block sum per MFMA -- fewer roundings than the f32 fma chain of conv2d_cg.py:342-349's
// sgemm.  No scale, no range flag: bf16 has f32's exponent range.  A stage is still 32 k, now 192 bytes of a
// pixel's record ([k-step][plane][k-half] pieces of 16 bytes) and 6 KB of weights per 32-row tile; the pixel image of a
// stage in LDS is [16-pixel block][quad of pieces][pixel][piece ^ swizzle] -- four consecutive DMA lanes fetch 64
// contiguous bytes of one pixel, and the ds_read_b128 lane groups of a fragment read touch all 64 banks once --; the
// stage loop works in UNITS of one k-step (the fragment registers of a whole stage would not fit beside two
// accumulator sets): unit (s, 0) is multiplied while (s, 1) is read, then the barrier of the stage, then (s, 1) is
// multiplied while (s + 1, 0) is read.
template <int BM, int BN, int WM, int WN, int PRE_CAP, bool MASK_LDS, int RING, int AR = 0>
__global__ __launch_bounds__(64 * WM * WN) void cbs_conv_kernel(CbsParams p, typename CbsExtOf<AR>::type ext) {
    constexpr bool HALF = AR == 1, X3 = AR == 2 || AR == 3 || AR == 4, WIN = AR == 3, SIDE = AR == 3 || AR == 4;
    // (before anything else -- the burst over the kilobyte of arguments included: an idle frame is this one load)
    if (HALF && p.upstream && *p.upstream == 0) {      // (the detection in front returned the same way: the mask is empty)
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            p.seq[0].countOut[0] = 0;
            if (p.info) p.info[CBS_INFO_SK] = 1;      // (nothing for the second launch)
        }
        return;
    }
    cb_touch_kernarg<sizeof(CbsParams) + (HALF ? sizeof(CbhExt) : (WIN ? sizeof(CbsWinExt) : (SIDE ? sizeof(CbsSideExt) : 0)))>();
    constexpr int NW = WM * WN, NT = 64 * NW;
    constexpr int TN = BN / WN / 32;
    static_assert(BM == 32 * WM, "one 32-row tile per wave");
    constexpr int PB = X3 ? 192 : 128;                   // bytes of one pixel's (one weight row's) stage
    constexpr int ABLK = BM * PB / 1024, BBLK = BN * PB / 1024;      // 1-KB DMA blocks per stage
    constexpr int APW = ABLK / NW, BPW = BBLK / NW;      // ... per wave
    static_assert(APW * NW == ABLK && BPW * NW == BBLK && APW >= 1 && BPW >= 1, "DMA blocks must deal evenly");
    constexpr int DPW = APW + BPW;
    constexpr int A_BYTES = BM * PB, B_BYTES = BN * PB, STAGE = A_BYTES + B_BYTES;
    constexpr int TILE = BM * BN;
    // static LDS (a workgroup may declare up to 160 KB without any opt-in): RING stages, then the popcount
    // prefix over the mask words of ALL sequences of the launch (one scan) and -- MASK_LDS -- the words themselves.
    // PRE_CAP = capacity in words.  The 64-row tile comes in three sizes: the two smaller ones leave room for two
    // workgroups per CU, the middle one by fetching the one word a pixel lookup needs from memory.
    __shared__ __attribute__((aligned(1024))) char ring[RING * STAGE];
    __shared__ int s_pre[PRE_CAP + 1];
    __shared__ unsigned long long s_mask[MASK_LDS ? PRE_CAP : 1];
    __shared__ const unsigned long long* s_maskPtr[CBS_MAXSEQ];
    __shared__ int s_wsum[NW];
    __shared__ int s_tilePix[BN];
    __shared__ float s_bias[BM];
    __shared__ unsigned s_nchg[HALF ? CBH_NEXT * (BN / 32) : 1];      // fp16: a consumer's changed pixels of this tile
    // WIN: window-count prefix over the (row pair, mask word) units; per tile: every slot's pixel whether it changed or
    // not, the pooled pixel of each of the tile's BN / 4 windows, the windows whose
    // pooled pixel changed
    __shared__ int s_preW[WIN ? PRE_CAP / 2 + 2 : 1];
    __shared__ int s_tileAny[WIN ? BN : 1];
    __shared__ int s_winPos[WIN ? BN / 4 : 1];
    __shared__ unsigned s_wchg;
    __shared__ unsigned long long s_wsum64[WIN ? 64 * WM * WN / 64 : 1];
    static_assert(!WIN || (BN / WN / 32 == 1 && MASK_LDS), "window order: one column tile per wave, the mask words in LDS");

    const int t = threadIdx.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int l31 = lane & 31, h = lane >> 5;
    const int MW = p.maskWords, MT = p.KP / BM;
#ifdef CBS_STAMP
    bool cbs_first = true;
#endif
    CBS_STAMP_AT(0);
    CBS_STAMP_AT(6);

    // ---- every workgroup: change-list lengths of all sequences from their masks, ONE scan ------------
    // (per-sequence facts live in small LDS tables, not in registers: eight sequences' worth of scalars would
    //  crowd out the kernel's own)
    __shared__ int s_seqN[CBS_MAXSEQ], s_seqRank[CBS_MAXSEQ], s_seqTile[CBS_MAXSEQ + 1], s_exact[CBS_MAXSEQ];
    // Mask protocol (single mask per sequence): the detection ORs into it; here every workgroup takes a copy into
    // LDS and never looks at the mask again.  On its way OUT of the kernel each workgroup counts itself on an
    // arrival counter behind the mask, and the one that learns it was the last zeroes the masks for the next
    // frame's detection.  Nobody waits for anybody; the next launch finds them clean.
    for (int q = t; q < p.nSeq; q += NT) s_maskPtr[q] = p.seq[q].masks;
    // the layer's range flag (set by the detection when a state value left the f16 pair's range, sticky): requested
    // here, looked at behind the scan -- such a sequence's tiles are then computed by plain f32 arithmetic from the
    // f32 state (cbs_exact_tile below): slow, right, in this very frame
    int flagv = 0;
    if (t < p.nSeq) {
        const int* rf = p.seq[t].rangeFlag;
        if (rf) flagv = *rf;
    }
    __syncthreads();
    const int E = p.nSeq * MW;
    // Thread t owns the CH consecutive words [t CH, (t+1) CH) of the concatenated masks: their popcounts stay in
    // registers, one wave scan + one exchange of the wave totals gives every thread its base, and the exclusive
    // prefix goes to LDS once -- two barriers, no serial section for a single sequence.
    int totAll = 0, nWin = 0;      // changed pixels of all sequences; WIN: touched 2x2 windows
    if constexpr (WIN) {
        // Window order (one sequence): the scan runs over the UNITS (row pair yo, mask word tx) instead of the words, with
        // three counts per unit -- changed pixels of the upper row (A), of the lower row (B), touched windows (W) -- packed
        // into one 64-bit value (20 bits each).  Exclusive prefixes: PA at s_pre[u], PB at s_pre[EU + 1 + u], PW at
        // s_preW[u].  The row-major rank of a changed pixel -- its place in the change list, which keeps the reference's
        // order -- is  PA[u] + PB[u0] + bits below  in the upper row and  PA[u0 + wpr] + PB[u] + bits below  in the lower
        // one (u0 = yo wpr: the row pair's first unit).
        const int HU = (p.H + 1) >> 1, EU = HU * p.wpr;
        constexpr int CUMAX = (PRE_CAP / 2 + NT) / NT;
        const int CHU = (EU + NT - 1) / NT, ub = t * CHU;
        unsigned long long cntU[CUMAX], wA[CUMAX], wB[CUMAX];
        unsigned long long locU = 0ull;
        const unsigned long long* mk = s_maskPtr[0];
#pragma unroll
        for (int u = 0; u < CUMAX; ++u) {
            const int i = ub + u;
            cntU[u] = 0ull, wA[u] = 0ull, wB[u] = 0ull;
            if (u < CHU && i < EU) {
                const int yo = cbs_div(i, p.magicWpr), tx = i - yo * p.wpr, ya = 2 * yo;
                const unsigned long long a = mk[ya * p.wpr + tx];
                const unsigned long long b0 = mk[min(ya + 1, p.H - 1) * p.wpr + tx];      // (clamped, not predicated)
                const unsigned long long b = ya + 1 < p.H ? b0 : 0ull;
                const unsigned long long v = a | b;
                wA[u] = a, wB[u] = b;
                cntU[u] = (unsigned long long)__popcll(a) | ((unsigned long long)__popcll(b) << 20) |
                          ((unsigned long long)__popcll((v | (v >> 1)) & 0x5555555555555555ull) << 40);
            }
            locU += cntU[u];
        }
        {
            unsigned long long* copy = p.seq[0].maskCopy;
            if (copy)
                for (int i = blockIdx.x * NT + t; i < MW; i += gridDim.x * NT) copy[i] = mk[i];
        }
        unsigned long long incl = locU;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        if (lane == 63) s_wsum64[t >> 6] = incl;
        __syncthreads();
        CBS_STAMP_AT(8);
        unsigned long long base = 0ull, tot = 0ull;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const unsigned long long v = s_wsum64[w];
            base += w < (t >> 6) ? v : 0ull;
            tot += v;
        }
        const int totA = (int)(tot & 0xfffffull), totB = (int)((tot >> 20) & 0xfffffull), totW = (int)(tot >> 40);
        totAll = __builtin_amdgcn_readfirstlane(totA + totB);
        nWin = __builtin_amdgcn_readfirstlane(totW);
        unsigned long long run = base + incl - locU;
#pragma unroll
        for (int u = 0; u < CUMAX; ++u) {
            const int i = ub + u;
            if (u < CHU && i < EU) {
                const int yo = cbs_div(i, p.magicWpr), tx = i - yo * p.wpr, ya = 2 * yo;
                s_pre[i] = (int)(run & 0xfffffull);
                s_pre[EU + 1 + i] = (int)((run >> 20) & 0xfffffull);
                s_preW[i] = (int)(run >> 40);
                s_mask[ya * p.wpr + tx] = wA[u];
                if (ya + 1 < p.H) s_mask[(ya + 1) * p.wpr + tx] = wB[u];
                run += cntU[u];
            }
        }
        if (t == 0) {
            s_pre[EU] = totA, s_pre[2 * EU + 1] = totB, s_preW[EU] = totW;
            s_seqRank[0] = 0, s_seqN[0] = totA + totB, s_seqTile[0] = 0, s_seqTile[1] = (totW + BN / 4 - 1) / (BN / 4);
            if (blockIdx.x == 0) p.seq[0].countOut[0] = totA + totB;
        }
        if (t < CBS_MAXSEQ) s_exact[t] = 0;
        __syncthreads();
    } else {
        constexpr int CHMAX = (PRE_CAP + NT - 1) / NT;
        const int CHW = (E + NT - 1) / NT, wb = t * CHW;
        int cnt[CHMAX];
        unsigned long long wordReg[MASK_LDS ? CHMAX : 1];
        int loc = 0;
    #pragma unroll
        for (int u = 0; u < CHMAX; ++u) {
            const int i = wb + u;
            cnt[u] = 0;
            if (u < CHW && i < E) {
                const int q = cbs_div(i, p.magicMW), w = i - q * MW;
                const unsigned long long word = s_maskPtr[q][w];
                cnt[u] = __popcll(word);
                if (MASK_LDS) wordReg[u] = word;
            }
            loc += cnt[u];
        }
        for (int q = 0; q < p.nSeq; ++q) {      // this launch also leaves a copy of the frame's masks at a fixed address
            unsigned long long* copy = p.seq[q].maskCopy;
            if (copy)
                for (int i = blockIdx.x * NT + t; i < MW; i += gridDim.x * NT) copy[i] = s_maskPtr[q][i];
        }
        {
            int incl = loc;
    #pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int v = __shfl_up(incl, o);
                if (lane >= o) incl += v;
            }
            if (lane == 63) s_wsum[t >> 6] = incl;
            __syncthreads();
            CBS_STAMP_AT(8);
            int base = 0, tot = 0;
    #pragma unroll
            for (int w = 0; w < NW; ++w) {
                const int v = s_wsum[w];
                base += w < (t >> 6) ? v : 0;
                tot += v;
            }
            totAll = __builtin_amdgcn_readfirstlane(tot);
            int run = base + incl - loc;
    #pragma unroll
            for (int u = 0; u < CHMAX; ++u) {
                const int i = wb + u;
                if (u < CHW && i < E) {
                    s_pre[i] = run;
                    if (MASK_LDS) s_mask[i] = wordReg[u];
                    run += cnt[u];
                }
            }
            if (t < CBS_MAXSEQ) s_exact[t] = flagv;
            if (t == 0) {
                s_pre[E] = tot;
                if (p.nSeq == 1) {      // (one sequence: its tables need nothing from LDS)
                    s_seqRank[0] = 0, s_seqN[0] = tot, s_seqTile[0] = 0, s_seqTile[1] = (tot + BN - 1) / BN;
                    if (blockIdx.x == 0) p.seq[0].countOut[0] = tot;
                }
            }
            __syncthreads();
        }
    }
    if (p.nSeq > 1) {
        if (t == 0) {
            int tiles = 0;
            for (int q = 0; q < p.nSeq; ++q) {
                const int rb = s_pre[q * MW], n = s_pre[(q + 1) * MW] - rb;
                s_seqRank[q] = rb, s_seqN[q] = n, s_seqTile[q] = tiles;
                tiles += (n + BN - 1) / BN;
                if (blockIdx.x == 0) p.seq[q].countOut[0] = n;
            }
            s_seqTile[p.nSeq] = tiles;
        }
        __syncthreads();
    }
    CBS_STAMP_AT(1);
    const int TP = WIN ? (nWin + BN / 4 - 1) / (BN / 4)                       // (tiles of BN / 4 whole windows)
                       : (p.nSeq == 1 ? (totAll + BN - 1) / BN                // pixel tiles of all sequences
                                      : __builtin_amdgcn_readfirstlane(s_seqTile[p.nSeq]));
    // Split along k while whole CUs would idle and the k-depth pays for the slab round trip -- WITHOUT letting the
    // partitioning into the arithmetic: a deep contraction (>= 48 stages) is always the sum, left to right, of
    // CBS_CHUNKS partial sums over fixed stage ranges, each accumulated from zero.  Split, every chunk is a work
    // item of its own and the reduce launch adds the slabs in chunk order; unsplit, one workgroup walks the whole
    // depth and folds its accumulators into a running sum at the chunk boundaries: the same additions in the
    // same order, so a sequence gets the same bits whether it runs alone (split) or beside others (unsplit).
    // fp16 layers (HALF): with few tiles the depth is cut into 8 or 16 chunks instead -- OpenPose's 7x7 layers on 128
    // channels at 46x81 recompute some 400 pixels, i.e. 14 tiles of 98 stages: 56 work items where there are 256 CUs
    // (round 5).  p.maxChunks: 4, or up to 16 for HALF.  The chunk count is chosen PER LAYER of a group from that layer's
    // own tile count, by the rule a launch of the layer alone applies (round 6): a layer gets the same bits alone and
    // beside the other branch of its stage.
    int CH = (p.nStages >= 48 && p.slabs) ? CBS_CHUNKS : 1;      // (the host refuses a deep layer without slabs)
    int chQ[HALF ? CBH_GROUP : 1], itemBase[HALF ? CBH_GROUP + 1 : 1];
    int itemsSplit = 0, itemsWhole = 0;
    if constexpr (HALF) {
#pragma unroll
        for (int u = 0; u < CBH_GROUP; ++u) {
            int tu = 0, c = CH;
            if (u < p.nSeq) {
                tu = p.nSeq == 1 ? TP : __builtin_amdgcn_readfirstlane(s_seqTile[u + 1] - s_seqTile[u]);
                if (CH > 1)
                    while (2 * c <= min(p.maxChunks, 16) && tu * MT * 2 * c <= (int)gridDim.x &&
                           tu * MT * 2 * c <= ext.slabCap1 && p.nStages >= 8 * c)      // (at least four stages per chunk)
                        c *= 2;
            }
            chQ[u] = c;
            itemBase[u] = tu;      // (tiles for now)
            itemsSplit += tu * MT * c, itemsWhole += tu * MT;
        }
    }
    const int chShift = CH >= 16 ? 4 : (CH >= 8 ? 3 : 2);      // (CH is 1, 4, 8 or 16: the host clamps maxChunks to <= 16)
    int anyExact = 0;
    for (int q = 0; q < p.nSeq; ++q) anyExact |= s_exact[q];
    anyExact = __builtin_amdgcn_readfirstlane(anyExact);
    int SK = 1;
    // (round 4: up to splitRounds = 2 work items per workgroup -- 65..128 tiles used to run unsplit, a quarter to a half of
    //  the CUs walking the whole depth while the others idled: 67 us where two rounds of quarter-depth items take 34)
    if constexpr (HALF) {
        if (TP > 0 && CH > 1 && itemsSplit <= p.splitRounds * (int)gridDim.x && itemsSplit <= p.slabCap)
            SK = max(chQ[0], chQ[CBH_GROUP - 1]);      // (> 1: split; each layer's own count is chQ[])
    } else {
        if (TP > 0 && CH > 1 && TP * MT * CH <= p.splitRounds * (int)gridDim.x && TP * MT * CH <= p.slabCap && !anyExact)
            SK = CH;
        // (tests: 1 = unsplit, >= 4 = split -- but never more partial tiles than the workspace holds)
        if (p.forceSK > 0 && CH > 1) SK = (p.forceSK >= CH && TP * MT * CH <= p.slabCap && !anyExact) ? CH : 1;
    }
    const int CMB = MT * SK;
    int items = TP * CMB;
    if constexpr (HALF) {      // item ranges of the group's layers: [itemBase[u], itemBase[u + 1])
        items = SK > 1 ? itemsSplit : itemsWhole;
        int run = 0;
#pragma unroll
        for (int u = 0; u < CBH_GROUP; ++u) {
            const int n = itemBase[u] * MT * (SK > 1 ? chQ[u] : 1);
            itemBase[u] = run;
            run += n;
        }
        itemBase[CBH_GROUP] = run;
    }
    // (builds with -DCBS_LAST_ARRIVER only: compiled in but switched off, the form costs the 128-row kernel 0.85 us per
    //  launch -- profiles/r06_ab_last_arriver.txt)
#ifdef CBS_LAST_ARRIVER
    const bool lastArr = X3 && BM == 128 && p.lastArrive && p.arriveTiles && SK > 1;
#else
    constexpr bool lastArr = false;
#endif
    if (blockIdx.x == 0 && t == 0 && p.info) {
        p.info[CBS_INFO_SK] = lastArr ? 1 : SK;      // (last-arriver form: the outputs are final when the launch ends)
        p.info[CBS_INFO_MT] = MT;
        p.info[2] = p.nSeq;
        for (int q = 0; q < CBS_MAXSEQ; ++q) {
            p.info[CBS_INFO_TP + q] = q < p.nSeq ? s_seqTile[q + 1] - s_seqTile[q] : 0;
            p.info[CBS_INFO_TP + CBS_MAXSEQ + q] = q < p.nSeq ? s_seqN[q] : 0;
        }
        if constexpr (HALF) {
#pragma unroll
            for (int u = 0; u < CBH_GROUP; ++u) p.info[CBS_INFO_CH + u] = chQ[u];
        }
    }

    const int aRecords = (int)min(p.aBytes, (long)0x7fffffff);
    cbs_const_int* stageOff = (cbs_const_int*)p.stageOff;
    const int HW = p.H * p.W;
    const float lo2 = 1.0f / CBS_LO;

    // Stage buffer layout: every wave owns DPW = 4 consecutive KB -- its two weight blocks, then its two pixel
    // blocks -- so that ONE value of M0 (the LDS address register of LDS-DMA) serves all four of its DMA
    // instructions of a stage, the block being chosen by the instruction's immediate offset: one scalar write per
    // stage instead of four.  (tools/micro/ldsdma_m0.hip: both forms -- M0 rewritten between back-to-back DMA
    // instructions, or one M0 and immediate offsets -- place every byte correctly, also under load.)
    //   weight block b = (row tile, k-step, plane) -> (b / APW) * 4096 + (b % APW) * 1024
    //   pixel n: block c = n / 8                  -> (c / BPW) * 4096 + (APW + c % BPW) * 1024 + (n % 8) * 128,
    //            chunk (k-step ks, plane pl, k-half h) at slot (ks * 4 + pl * 2 + h) ^ ((n >> 1) & 7)
    // x3: every wave owns 6 KB -- its three weight blocks, then ONE block of 16 pixels x 192 B laid out as
    //   [quad cq = piece / 4][pixel i][piece % 4 ^ (i >> 2)] x 16 B,  piece = (k-step * 3 + plane) * 2 + k-half
    // (two M0 values per stage: the instruction's immediate offset ends at 4095)
    //   weight block a = (row tile) * 6 + (k-step * 3 + plane) -> (a / 3) * 6144 + (a % 3) * 1024
    static_assert(APW == (X3 ? 3 : 2) && BPW == APW, "four (x3: six) DMA instructions per wave and stage");
    int aRead[4];      // this lane's A fragment of (row tile wm; e = ks * 2 + plane); x3: [0] + immediates
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int b = wm * 4 + e;
        aRead[e] = X3 ? (2 * wm) * 6144 + lane * 16 : (b / APW) * 4096 + (b % APW) * 1024 + lane * 16;
    }
    int bRead[TN];     // x3: the pixel's slot base (piece parity 0; the other parity = ^ 32), + (piece / 4) KB immediates
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = (wn * TN + j) * 32 + l31, c = n >> 3;
        if (X3)
            bRead[j] = (n >> 4) * 6144 + 3072 + (n & 15) * 64 + ((h ^ ((n >> 2) & 3)) << 4);
        else
            bRead[j] = (c / BPW) * 4096 + (APW + c % BPW) * 1024 + (n & 7) * 128 + ((h ^ ((n >> 1) & 7)) << 4);
    }

    for (int it = blockIdx.x; it < items; it += gridDim.x) {
        // it = (ptg * SK + slice) * MT + mt -- MT by its magic, SK (1 or CBS_CHUNKS, powers of two) by shifts: no
        // integer division by a run-time value on the way to the first DMA
        int d = cbs_div(it, p.magicMT), mt = it - d * MT;
        int slice = d & (SK - 1), ptg = SK == 1 ? d : d >> chShift;
        int q = 0, N = totAll, rb = 0, tile0 = 0;
        int CHi = CH, chShiftI = chShift;      // (fp16 group: the item's layer's own chunk count)
        if constexpr (HALF) {
            // the group's layers own item ranges of their own (their chunk counts differ); inside one:
            // local = (tile * chunks + slice) * MT + mt
            static_assert(CBH_GROUP == 2, "two item ranges");
            q = (p.nSeq > 1 && it >= itemBase[1]) ? 1 : 0;
            const int local = it - (q ? itemBase[1] : 0);
            CHi = q ? chQ[CBH_GROUP - 1] : chQ[0];
            chShiftI = CHi >= 16 ? 4 : (CHi >= 8 ? 3 : 2);
            d = cbs_div(local, p.magicMT), mt = local - d * MT;
            slice = SK == 1 ? 0 : d & (CHi - 1);
            ptg = SK == 1 ? d : d >> chShiftI;      // (the tile inside its layer)
            if (p.nSeq > 1) {
                N = __builtin_amdgcn_readfirstlane(s_seqN[q]), rb = __builtin_amdgcn_readfirstlane(s_seqRank[q]);
            }
        } else if (p.nSeq > 1) {      // (one sequence: all of it known without LDS)
            for (int u = 1; u < p.nSeq; ++u)
                if (ptg >= s_seqTile[u]) q = u;
            q = __builtin_amdgcn_readfirstlane(q);
            N = __builtin_amdgcn_readfirstlane(s_seqN[q]), rb = __builtin_amdgcn_readfirstlane(s_seqRank[q]);
            tile0 = __builtin_amdgcn_readfirstlane(s_seqTile[q]);
        }
        const int n0 = (ptg - tile0) * BN, m0 = mt * BM;
        // (fp16 group: layer q's own weights, bias, channel count and consumers)
        const char* Aq = p.A;
        const float* biasq = p.bias;
        int Kq = p.K, reluq = p.relu, nNext = 0;
        if constexpr (HALF) {
            Aq = ext.L[q].A, biasq = (const float*)ext.L[q].bias, Kq = ext.L[q].K, reluq = ext.L[q].relu;
            nNext = __builtin_amdgcn_readfirstlane(ext.L[q].nNext);
        }
        // Stages of this item: one chunk (split) or all of them (then the accumulators are folded into the running
        // sum at the chunk boundaries).  Chunk c = stage pairs [P c / CH, P (c+1) / CH) of the P = nStages / 2 pairs
        // (+ the odd last stage in the last chunk): every boundary lies an even number of stages behind the start,
        // which keeps the two fragment register sets in step with the loop below.
        const int P2 = p.nStages >> 1;
        static_assert(CBS_CHUNKS == 4, "chunk boundaries by shifts");
        // (bf16 triples: a step is a whole stage -- both register sets per stage --, so the boundaries need not be even:
        //  98 stages are 24 + 25 + 24 + 25 instead of 24 + 24 + 24 + 26, and the longest slice sets the launch's time)
        auto chunkBeg = [&](int c) {
            return c >= CHi ? p.nStages : (CHi == 1 ? 0 : (X3 ? (p.nStages * c) >> chShiftI : 2 * ((P2 * c) >> chShiftI)));
        };
        const int c0 = SK == 1 ? 0 : slice, c1 = SK == 1 ? CHi : slice + 1;
        const int sBeg = chunkBeg(c0), sEnd = chunkBeg(c1);
        // (the first stage's tap offset: a scalar load whose round trip runs beside the pixel lookup below)
        int bNext = stageOff[sBeg];

        // Pixel of slot j of this tile = the (n0+j)-th set bit of sequence q's mask.  Two cooperative steps instead
        // of a binary search and a bit select per pixel: every thread tests a few words for holding the tile's first
        // / last rank; then the waves expand the words in between, one word per wave and step, one bit per lane.
        __shared__ int s_wRange[2];
        __syncthreads();   // (s_tilePix and the ring of the previous item are no longer read)
        if constexpr (WIN) {
            // Window order: the tile is the BN / 4 touched windows of ranks [R0, R1]; slot 4 r + 2 dy + dx is pixel (dy, dx)
            // of its r-th window.  s_tilePix holds the CHANGED pixels (the contraction computes those and nothing else),
            // s_tileAny every pixel of the window inside the map, s_winPos the window's pooled pixel (-1: none -- the odd
            // last row or column of a floor-mode pool).  One unit per wave and step, one column per lane.
            const int EU = ((p.H + 1) >> 1) * p.wpr;
            const int R0 = ptg * (BN / 4), R1 = min(R0 + BN / 4, nWin) - 1;
            for (int i = t; i < EU; i += NT) {
                const int a = s_preW[i], b = s_preW[i + 1];
                if (a <= R0 && R0 < b) s_wRange[0] = i;
                if (a <= R1 && R1 < b) s_wRange[1] = i;
            }
            if (t < BN) s_tilePix[t] = -1, s_tileAny[t] = -1;
            if (t < BN / 4) s_winPos[t] = -1;
            if (t == 0) s_wchg = 0u;
            __syncthreads();
            const int w0 = __builtin_amdgcn_readfirstlane(s_wRange[0]), w1 = __builtin_amdgcn_readfirstlane(s_wRange[1]);
            int32_t* listOut = p.seq[0].listOut;
            for (int i = w0 + wave; i <= w1; i += NW) {
                const int yo = cbs_div(i, p.magicWpr), tx = i - yo * p.wpr, ya = 2 * yo, u0 = yo * p.wpr;
                const bool hasB = ya + 1 < p.H;
                const unsigned long long A = s_mask[ya * p.wpr + tx];
                const unsigned long long B = hasB ? s_mask[(ya + 1) * p.wpr + tx] : 0ull;
                const unsigned long long V = A | B, Wm = (V | (V >> 1)) & 0x5555555555555555ull;
                if (Wm == 0ull) continue;
                const int k2 = lane & ~1;      // bit of this lane's window in Wm
                if ((Wm >> k2) & 1ull) {
                    const int r = s_preW[i] + __popcll(Wm & ((1ull << k2) - 1ull)) - R0;
                    if (r >= 0 && r < BN / 4) {
                        const int x = tx * 64 + lane;
                        const unsigned long long below = (1ull << lane) - 1ull;
                        const int posA = ya * p.W + x, posB = posA + p.W, slotA = 4 * r + (lane & 1);
                        if (x < p.W) {
                            s_tileAny[slotA] = posA;
                            if ((A >> lane) & 1ull) {
                                s_tilePix[slotA] = posA;
                                listOut[s_pre[i] + s_pre[EU + 1 + u0] + __popcll(A & below)] = posA;
                            }
                            if (hasB) {
                                s_tileAny[slotA + 2] = posB;
                                if ((B >> lane) & 1ull) {
                                    s_tilePix[slotA + 2] = posB;
                                    listOut[s_pre[u0 + p.wpr] + s_pre[EU + 1 + i] + __popcll(B & below)] = posB;
                                }
                            }
                        }
                        if ((lane & 1) == 0) {
                            const int gx = tx * 32 + (lane >> 1);
                            s_winPos[r] = (yo < ext.H2 && gx < ext.W2) ? yo * ext.W2 + gx : -1;
                        }
                    }
                }
            }
        } else {
            const int R0 = rb + n0, R1 = rb + min(n0 + BN, N) - 1;
            for (int i = q * MW + t; i < (q + 1) * MW; i += NT) {
                const int a = s_pre[i], b = s_pre[i + 1];
                if (a <= R0 && R0 < b) s_wRange[0] = i;
                if (a <= R1 && R1 < b) s_wRange[1] = i;
            }
            if (t < BN) s_tilePix[t] = -1;
            if (HALF && t < CBH_NEXT * (BN / 32)) s_nchg[t] = 0u;
            __syncthreads();
            const int w0 = __builtin_amdgcn_readfirstlane(s_wRange[0]), w1 = __builtin_amdgcn_readfirstlane(s_wRange[1]);
            const bool first = mt == 0 && slice == 0;
            int32_t* listOut = p.seq[q].listOut;
            for (int i = w0 + wave; i <= w1; i += NW) {
                const unsigned long long word = MASK_LDS ? s_mask[i] : s_maskPtr[q][i - q * MW];
                if (word == 0ull) continue;
                if ((word >> lane) & 1ull) {
                    const int r = s_pre[i] + __popcll(word & ((1ull << lane) - 1ull)) - R0;
                    if (r >= 0 && r < BN) {
                        const int w = i - q * MW;
                        const int row = cbs_div(w, p.magicWpr);
                        const int pos = row * p.W + (w - row * p.wpr) * 64 + lane;
                        s_tilePix[r] = pos;
                        if (first) listOut[n0 + r] = pos;
                    }
                }
            }
        }
        CBS_STAMP_AT(9);
        // (the m-tile's bias for the epilogue: requested now, put into LDS once the ring is primed)
        const float biasv = (t < BM && biasq && m0 + t < Kq)
                                ? (HALF ? (float)((const _Float16*)biasq)[m0 + t] : biasq[m0 + t]) : 0.f;
        __syncthreads();
        CBS_STAMP_AT(10);
        if (AR == 0 && __builtin_expect(__builtin_amdgcn_readfirstlane(s_exact[q]) != 0, 0)) {
            // The sequence's state left the range of the f16 pairs (|x| >= 2^20, cbs_detect_kernel): its split state is
            // not usable.  The tile is computed from prevInput and the plain filter bank instead, one f32 fma chain per
            // output in (channel, ky, kx) order like conv2d_cg.py:342-349's sgemm -- orders of magnitude slower, and
            // right: the layer never hands out numbers made from an overflowed operand.  (SK is 1 in such a launch.)
            // (the arguments only this path needs are read through an opaque copy of the kernel-argument pointer:
            //  the compiler would otherwise load them at kernel entry and carry them through the stage loop in
            //  scalar registers it does not have)
            const char* ka = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(ka));
            const CbsParams* pp = (const CbsParams*)ka;
            const float* __restrict__ st = pp->seq[q].state;
            const float* __restrict__ wf = pp->wPlain;
            float* __restrict__ outx = pp->seq[q].out;
            const float* biasx = pp->bias;
            const int Cn = pp->rec >> 2, kH = pp->kH, kW = pp->kW, Wd = pp->W, Hd = pp->H, Kd = pp->K, relux = pp->relu;
            const unsigned long long magicWx = pp->magicW;
            const int HWx = Hd * Wd;
            for (int o = t; o < BM * BN; o += NT) {
                const int nl = o % BN, ml = o / BN, m = m0 + ml;
                const int pix = s_tilePix[nl];
                if (pix < 0 || m >= Kd) continue;
                const int y = cbs_div(pix, magicWx), x = pix - y * Wd;
                const float* wrow = wf + (long)m * Cn * kH * kW;
                float acc = 0.f;
                for (int c = 0; c < Cn; ++c)
                    for (int ky = 0; ky < kH; ++ky) {
                        const int yy = y + ky - kH / 2;
                        for (int kx = 0; kx < kW; ++kx) {
                            const int xx = x + kx - kW / 2;
                            const bool in = yy >= 0 && yy < Hd && xx >= 0 && xx < Wd;
                            const float v = in ? st[(long)c * HWx + (long)yy * Wd + xx] : 0.f;
                            acc = fmaf(v, wrow[(c * kH + ky) * kW + kx], acc);
                        }
                    }
                float v = acc + (biasx ? biasx[m] : 0.f);
                if (relux) v = v <= 0.f ? 0.f : v;
                if (pp->accumulate) {      // (fine-grained: `st` is the delta tensor)
                    v += outx[(long)m * HWx + pix];
                    float* rp = pp->seq[q].reluOut;
                    if (rp) rp[(long)m * HWx + pix] = v <= 0.f ? 0.f : v;
                }
                outx[(long)m * HWx + pix] = v;
            }
            continue;
        }

        const char* Sq = p.seq[q].S;
        const int bRecords = (int)min(p.stateBytes + CBS_SPAD, (long)0x7fffffff);
        // DMA source offsets of this wave (the instruction's immediate offset -- i KB for its i-th DMA of a stage
        // -- counts for the memory address too: taken off here; the pixel operand's may go negative by up to 3 KB,
        // hence the CBS_SPAD bytes in front of the records): its APW weight blocks are contiguous in memory as in
        // LDS, so one offset serves both; BPW pixel blocks of 8 pixels x 128 B
        const int aVoff = wave * APW * 1024 + lane * 16;
        int bVoff[BPW];
#pragma unroll
        for (int i = 0; i < BPW; ++i) {
            // x3: ONE pixel per lane for its three DMAs -- pixel lane / 4 of the wave's block of 16, quad i of its 12
            // pieces, piece (lane & 3) ^ swizzle of the quad (the B DMAs' LDS base is mine + 3 KB: immediates 0, 1, 2 KB)
            const int nl = X3 ? wave * 16 + (lane >> 2) : (wave * BPW + i) * 8 + (lane >> 3);
            const int pos = s_tilePix[nl];
            const int py = cbs_div(max(pos, 0), p.magicW);
            const int base = (pos < 0 || CBS_DBGBIT(1)) ? p.dummyBase : (py * p.Wp + (pos - py * p.W)) * p.rec;
            if (X3)
                bVoff[i] = CBS_SPAD + base + ((i * 4 + ((lane & 3) ^ ((lane >> 4) & 3))) << 4) - i * 1024;
            else
                bVoff[i] = CBS_SPAD + base + (((lane & 7) ^ ((nl >> 1) & 7)) << 4) - (APW + i) * 1024;
        }
        const int aItem = (m0 / 32) * (PB * 32);            // row tiles of this m-tile inside a stage
        const int aStageBytes = (p.KP / 32) * (PB * 32);

        // (the pixel operand's stage offset comes from the table by a scalar load: requested one stage early)
        CBS_STAMP_AT(11);
        // All DMA of stage s of this wave, into ring slot s % RING.  Past the item's last stage the same four
        // instructions are issued DEAD -- through a resource of zero records, so that every access is out of range:
        // no memory traffic, zeros into a ring slot nobody reads any more (tools/micro/ldsdma_oob.hip) -- which keeps
        // the number of DMA instructions in flight, and with it the s_waitcnt vmcnt(N) of a step, the same in every
        // step: one step body, no peeled tail.
        // In the stage loop the instructions of a stage are not issued in one burst: issue(s, true) only sets the stage up
        // (descriptors, LDS base, offsets) and issuePart<i>() issues its i-th DMA -- one per gap between the first matrix
        // instructions behind the barrier (round 5, measured on one box against the burst behind the fourth matrix
        // instruction: 64->256 35.8 -> 33.8 us, 16->64 19.7 -> 18.2 us, eight sequences 244 -> 224 us: a CU takes a DMA
        // instruction every ~30 cycles, eight waves handing it six each at the same moment queue up behind each other
        // with their matrix instructions waiting behind the DMAs in program order).
        __amdgpu_buffer_rsrc_t isA, isB;
        char* isMine = ring;
        int isAS = 0, isBS = 0;
        auto issue = [&](int s, bool setUpOnly = false) {
            char* dst = ring + (s % RING) * STAGE;
            const bool live = s < sEnd;
            const int aS = live ? s * aStageBytes + aItem : 0;
            const int bS = bNext;
            bNext = stageOff[min(s + 1, sEnd - 1)];
            if (CBS_DBGBIT(4)) return;
            const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc(
                (void*)Aq, 0, (live && !CBS_DBGBIT(64 | 128)) ? aRecords : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc(
                (void*)Sq, 0, (live && !CBS_DBGBIT(64 | 256)) ? bRecords : 0, 0x00020000);
            char* mine = dst + wave * (DPW * 1024);      // ONE LDS base (M0) for the four
            if (setUpOnly) {
                isA = ar, isB = br, isMine = mine, isAS = aS, isBS = bS;
                return;
            }
            if constexpr (X3) {
                cbs_dma16<0>(ar, mine, aVoff, aS);
                cbs_dma16<1024>(ar, mine, aVoff, aS);
                cbs_dma16<2048>(ar, mine, aVoff, aS);
                cbs_dma16<0>(br, mine + 3072, bVoff[0], bS);
                cbs_dma16<1024>(br, mine + 3072, bVoff[BPW > 1 ? 1 : 0], bS);
                cbs_dma16<2048>(br, mine + 3072, bVoff[BPW > 2 ? 2 : 0], bS);
            } else {
                cbs_dma16<0>(ar, mine, aVoff, aS);
                cbs_dma16<1024>(ar, mine, aVoff, aS);
                cbs_dma16<2048>(br, mine, bVoff[0], bS);
                cbs_dma16<3072>(br, mine, bVoff[1], bS);
            }
        };

        auto issuePart = [&](auto partTag) {
            constexpr int I = decltype(partTag)::value;
            if (CBS_DBGBIT(4)) return;
            if constexpr (X3) {
                if constexpr (I == 0) cbs_dma16<0>(isA, isMine, aVoff, isAS);
                if constexpr (I == 1) cbs_dma16<1024>(isA, isMine, aVoff, isAS);
                if constexpr (I == 2) cbs_dma16<2048>(isA, isMine, aVoff, isAS);
                if constexpr (I == 3) cbs_dma16<0>(isB, isMine + 3072, bVoff[0], isBS);
                if constexpr (I == 4) cbs_dma16<1024>(isB, isMine + 3072, bVoff[BPW > 1 ? 1 : 0], isBS);
                if constexpr (I == 5) cbs_dma16<2048>(isB, isMine + 3072, bVoff[BPW > 2 ? 2 : 0], isBS);
            } else {
                if constexpr (I == 0) cbs_dma16<0>(isA, isMine, aVoff, isAS);
                if constexpr (I == 1) cbs_dma16<1024>(isA, isMine, aVoff, isAS);
                if constexpr (I == 2) cbs_dma16<2048>(isB, isMine, bVoff[0], isBS);
                if constexpr (I == 3) cbs_dma16<3072>(isB, isMine, bVoff[1], isBS);
            }
        };
        typedef std::integral_constant<int, 0> P0;
        typedef std::integral_constant<int, 1> P1;
        typedef std::integral_constant<int, 2> P2t;
        typedef std::integral_constant<int, 3> P3;
        typedef std::integral_constant<int, 4> P4;
        typedef std::integral_constant<int, 5> P5;

        // x3, one column tile per wave: the main products of the two k-steps of a stage go to accumulators of their own
        // (acc1 / acc3) -- an accumulation chain rounds once per matrix instruction, at the magnitude of what it holds;
        // two chains of half the length hold the large terms half as long (the registers are there: 16 per lane)
        constexpr bool DUALM = X3 && TN == 1;
        floatx16 acc1[TN], acc2[TN], acc3[DUALM ? TN : 1];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc1[j][i] = 0.f, acc2[j][i] = 0.f;
                if (DUALM) acc3[j][i] = 0.f;
            }
        // what the accumulators hold, as one number
        auto comb = [&](int j, int i) -> float {
            if (HALF) return acc1[j][i];
            if (DUALM) return (acc1[j][i] + acc3[j][i]) + acc2[j][i];
            if (X3) return acc1[j][i] + acc2[j][i];
            return acc1[j][i] + acc2[j][i] * lo2;
        };
        // Chunk boundary of an unsplit deep item: running sum = (first ? 0 : running sum) + (acc1 + acc2 / 2^11),
        // accumulators from zero again.  The running sum is kept in REGISTERS (16 TN per lane): with one step body the
        // stage loop has them to spare (round 3's first form, with its peeled loop tails, did not -- it kept the sum
        // in the item's slab, 6 x 64 KB of memory traffic per item).
        int folds = 0;
        floatx16 run[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) run[j][i] = 0.f;
        auto fold = [&]() {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    // (the first chunk's sum is taken as it is: 0 + x would turn a -0 into +0 where the split form,
                    //  which adds the second slab to the first, keeps it)
                    const float c = comb(j, i);
                    run[j][i] = folds > 0 ? run[j][i] + c : c;
                    acc1[j][i] = 0.f, acc2[j][i] = 0.f;
                    if (DUALM) acc3[j][i] = 0.f;
                }
            ++folds;
        };

        // the fragments of one stage in registers: two sets, the reads of stage s+1 run under the MFMAs of stage s
        constexpr int FN = X3 ? 3 : 4;      // fragments per operand tile and register set (x3: one k-step's planes)
        struct Frags {
            halfx8 a[FN];          // [ks * 2 + plane]; x3: [plane] of the unit's k-step (bf16 bits)
            halfx8 b[TN * FN];     // [j * 4 + ks * 2 + plane]; x3: [j * 3 + plane]
        };
        const unsigned ringBase = (unsigned)(size_t)(cb_lds_ptr)ring;
        unsigned bAddr[TN * 4];   // LDS byte address (slot 0) of this lane's B chunk [j][ks * 2 + plane]
#pragma unroll                    // (x3: [j * 4 + parity of the piece pair k-step * 3 + plane], the quad by immediates)
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                bAddr[j * 4 + e] = X3 ? ringBase + (bRead[j] ^ ((e & 1) << 5))
                                      : ringBase + (bRead[j] ^ (((e >> 1) * 4 + (e & 1) * 2) << 4));
        unsigned aAddr[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) aAddr[e] = ringBase + aRead[e];
        auto readA = [&](int s, Frags& f) {
            const unsigned slot = (s % RING) * STAGE;
#pragma unroll
            for (int e = 0; e < FN; ++e) {
                if (CBS_DBGBIT(16))
                    f.a[e] = halfx8{1, 1, 1, 1, 1, 1, 1, 1};
                else
                    f.a[e] = cbs_lds_read16<0>(aAddr[e] + slot);
            }
        };
        auto readB = [&](int s, Frags& f, int j) {
            const unsigned slot = (s % RING) * STAGE;
#pragma unroll
            for (int e = FN * j; e < FN * j + FN; ++e) {
                if (CBS_DBGBIT(8))
                    f.b[e] = halfx8{1, 1, 1, 1, 1, 1, 1, 1};
                else
                    f.b[e] = cbs_lds_read16<0>(bAddr[e] + slot);
            }
        };
        auto readFrags = [&](int s, Frags& f) {
            readA(s, f);
#pragma unroll
            for (int j = 0; j < TN; ++j) readB(s, f, j);
        };
        // x3: the fragments of UNIT (stage s, k-step KS): plane pl = piece pair e = 3 KS + pl -- weight block e of the
        // row tile (immediate), pixel quad e / 2 (immediate), parity e & 1 (address register)
        auto readA3 = [&](auto ksTag, int s, Frags& f) {
            constexpr int KS = decltype(ksTag)::value;
            const unsigned a0 = aAddr[0] + (s % RING) * STAGE;
            constexpr int E0 = 3 * KS, E1 = 3 * KS + 1, E2 = 3 * KS + 2;
            f.a[0] = cbs_lds_read16<(E0 / 3) * 6144 + (E0 % 3) * 1024>(a0);
            f.a[1] = cbs_lds_read16<(E1 / 3) * 6144 + (E1 % 3) * 1024>(a0);
            f.a[2] = cbs_lds_read16<(E2 / 3) * 6144 + (E2 % 3) * 1024>(a0);
        };
        auto readB3 = [&](auto ksTag, int s, Frags& f, int j) {
            constexpr int KS = decltype(ksTag)::value;
            const unsigned slot = (s % RING) * STAGE;
            constexpr int E0 = 3 * KS, E1 = 3 * KS + 1, E2 = 3 * KS + 2;
            if (CBS_DBGBIT(512) && j > 0) {      // (ablation: a third fewer fragment reads -- what 64x64 wave tiles would read)
                f.b[j * 3 + 0] = f.b[0], f.b[j * 3 + 1] = f.b[1], f.b[j * 3 + 2] = f.b[2];
                return;
            }
            f.b[j * 3 + 0] = cbs_lds_read16<(E0 / 2) * 1024>(bAddr[j * 4 + (E0 & 1)] + slot);
            f.b[j * 3 + 1] = cbs_lds_read16<(E1 / 2) * 1024>(bAddr[j * 4 + (E1 & 1)] + slot);
            f.b[j * 3 + 2] = cbs_lds_read16<(E2 / 2) * 1024>(bAddr[j * 4 + (E2 & 1)] + slot);
        };
        // (the same reads one by one: R = 0..2 the weight planes, 3..5 / 6..8 the pixel planes of column tile 0 / 1)
        auto read1 = [&](auto ksTag, auto rTag, int s, Frags& f) {
            constexpr int KS = decltype(ksTag)::value, R = decltype(rTag)::value;
            constexpr int E = 3 * KS + (R % 3);
            if constexpr (R < 3) {
                const unsigned a0 = aAddr[0] + (s % RING) * STAGE;
                f.a[R] = cbs_lds_read16<(E / 3) * 6144 + (E % 3) * 1024>(a0);
            } else if constexpr (R < 3 + 3 * TN) {
                constexpr int j = (R - 3) / 3;
                const unsigned slot = (s % RING) * STAGE;
                f.b[j * 3 + R % 3] = cbs_lds_read16<(E / 2) * 1024>(bAddr[j * 4 + (E & 1)] + slot);
            }
        };
        // x3: the 6 TN matrix instructions of a unit, i = j * 6 + term: the five small products into acc2, b0 w0 into acc1
        constexpr int NM3 = 6 * TN;
        auto mma3 = [&](auto ksTag, const Frags& f, int i0, int i1) {
            constexpr int KS = decltype(ksTag)::value;
            if (CBS_DBGBIT(2)) return;
#pragma unroll
            for (int i = i0; i < (i1 < NM3 ? i1 : NM3); ++i) {
                const int j = i / 6, term = i % 6;
                const int wa = term == 0 ? 2 : (term == 1 ? 0 : (term == 2 ? 1 : (term == 3 ? 1 : 0)));
                const int xb = term == 0 ? 0 : (term == 1 ? 2 : (term == 2 ? 1 : (term == 3 ? 0 : (term == 4 ? 1 : 0))));
                const bf16x8 av = __builtin_bit_cast(bf16x8, f.a[wa]), bv = __builtin_bit_cast(bf16x8, f.b[j * 3 + xb]);
                if (term == 5 && DUALM && KS == 1)
                    acc3[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc3[j], 0, 0, 0);
                else if (term == 5)
                    acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc1[j], 0, 0, 0);
                else
                    acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc2[j], 0, 0, 0);
            }
        };
        // the 6 TN matrix instructions of a stage, numbered i = (ks * TN + j) * 3 + term
        // (HALF: 4 TN, i = ks * TN + j, one product each, fragment e = k-step ks of the stage's four)
        constexpr int NM = HALF ? 4 * TN : 6 * TN;
        auto mmaRange = [&](const Frags& f, int i0, int i1) {
            if (CBS_DBGBIT(2)) return;
            if constexpr (HALF) {
#pragma unroll
                for (int i = i0; i < (i1 < NM ? i1 : NM); ++i) {
                    const int ks = i / TN, j = i % TN;
                    acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[ks], f.b[j * 4 + ks], acc1[j], 0, 0, 0);
                }
                return;
            }
            if constexpr (!X3)
#pragma unroll
            for (int i = i0; i < i1; ++i) {
                const int ks = i / (3 * TN), j = (i / 3) % TN, term = i % 3;
                if (term == 0)
                    acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[ks * 2 + 1], f.b[j * 4 + ks * 2], acc2[j], 0, 0, 0);
                else if (term == 1)
                    acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[ks * 2], f.b[j * 4 + ks * 2 + 1], acc2[j], 0, 0, 0);
                else
                    acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[ks * 2], f.b[j * 4 + ks * 2], acc1[j], 0, 0, 0);
            }
        };
        // s_waitcnt lgkmcnt(0) that NAMES the fragment registers it makes valid (in/out operands): nothing that
        // consumes them can be scheduled in front of it
        auto waitFrags = [&](Frags& f) {
            if constexpr (X3 && TN == 2)
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.a[2]), "+v"(f.b[0]), "+v"(f.b[1]), "+v"(f.b[2]),
                               "+v"(f.b[3]), "+v"(f.b[4]), "+v"(f.b[5])
                             :
                             : "memory");
            else if constexpr (X3)
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.a[2]), "+v"(f.b[0]), "+v"(f.b[1]), "+v"(f.b[2])
                             :
                             : "memory");
            else if constexpr (TN == 2)
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.a[2]), "+v"(f.a[3]), "+v"(f.b[0]), "+v"(f.b[1]),
                               "+v"(f.b[2]), "+v"(f.b[3]), "+v"(f.b[4]), "+v"(f.b[5]), "+v"(f.b[6]), "+v"(f.b[7])
                             :
                             : "memory");
            else
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.a[2]), "+v"(f.a[3]), "+v"(f.b[0]), "+v"(f.b[1]),
                               "+v"(f.b[2]), "+v"(f.b[3])
                             :
                             : "memory");
        };
        // One step = stage s multiplied from registers while stage s+1 is read into the other set.  Before that,
        // everything but the RING-2 youngest stages of this wave's DMA must have landed (vmcnt counts its DMA
        // instructions, DPW per stage, issued in stage order) and -- barrier -- everybody else's: stage s+RING is then
        // aimed at the ring slot of stage s, which nobody reads any more.  The step ENDS with the s_waitcnt that makes
        // the freshly read set valid: the compiler takes the asm reads' results for valid at once, and any copy or
        // spill of such a register it places before the data has arrived -- at a loop edge, a branch join -- would
        // carry garbage (it did: wrong tiles, but only while other kernels kept the CU's LDS busy).  Inside a step
        // there is no control flow, and the reads have the whole MFMA chain to return.
        // The instruction order inside a step is fixed by hand (sched_barrier between the groups): the matrix
        // pipe is fed from the first cycle behind the barrier, and the DMA issue, the address arithmetic and the
        // fragment reads of the next stage sit in the shadows of matrix instructions that are already queued.  Left
        // to itself the compiler puts most reads behind the first matrix instructions and the wait for them right
        // behind the reads -- with most of the stage's matrix work still to be issued.
#define CBS_SB() __builtin_amdgcn_sched_barrier(0)
#define CBS_STEP(WAITN, ISSUE, S, FCUR, FNEXT)                                            \
        do {                                                                               \
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN) : "memory");                  \
            __builtin_amdgcn_s_barrier();                                                  \
            CBS_SB();                                                                      \
            if (ISSUE) issue((S) + RING, true);                                            \
            CBS_SB();                                                                      \
            mmaRange(FCUR, 0, 1); CBS_SB(); if (ISSUE) issuePart(P0()); CBS_SB();          \
            mmaRange(FCUR, 1, 2); CBS_SB(); if (ISSUE) issuePart(P1()); CBS_SB();          \
            mmaRange(FCUR, 2, 3); CBS_SB();                                                \
            readA((S) + 1, FNEXT); CBS_SB(); if (ISSUE) issuePart(P2t()); CBS_SB();        \
            mmaRange(FCUR, 3, 4); CBS_SB();                                                \
            readB((S) + 1, FNEXT, 0); CBS_SB(); if (ISSUE) issuePart(P3()); CBS_SB();      \
            mmaRange(FCUR, 4, 5);                                                          \
            CBS_SB();                                                                      \
            if (TN > 1) readB((S) + 1, FNEXT, TN - 1);                                     \
            CBS_SB();                                                                      \
            mmaRange(FCUR, 5, NM);                                                         \
            waitFrags(FNEXT);                                                              \
        } while (0)

        // x3: a stage is two UNITS (k-steps); F0 holds k-step 0, F1 k-step 1.  Half-step A multiplies (s, 0) while
        // (s, 1) is read -- stage s is in LDS, nothing is aimed at its slot yet.  Half-step B starts with the stage's
        // wait and barrier (stage s + 1 has landed for everybody, everybody's reads of stage s are behind their
        // waits), issues stage s + RING into the slot of stage s, and multiplies (s, 1) while (s + 1, 0) is read.
        // The stage's six DMA instructions go out one per gap behind the barrier, beside the fragment reads (see issue()).
        // (An earlier form let the second half of the waves issue theirs half a stage later: superseded.)
        typedef std::integral_constant<int, 0> KS0;
        typedef std::integral_constant<int, 1> KS1;
        typedef std::integral_constant<int, 6> R6;
        typedef std::integral_constant<int, 7> R7;
        typedef std::integral_constant<int, 8> R8;
#define CBS_STEP3A(S, FCUR, FNEXT)                                                         \
        do {                                                                               \
            CBS_SB();                                                                      \
            mma3(KS0(), FCUR, 0, 1); CBS_SB(); read1(KS1(), P0(), (S), FNEXT); read1(KS1(), P3(), (S), FNEXT); CBS_SB(); \
            mma3(KS0(), FCUR, 1, 2); CBS_SB(); read1(KS1(), P1(), (S), FNEXT); read1(KS1(), P4(), (S), FNEXT); CBS_SB(); \
            mma3(KS0(), FCUR, 2, 3); CBS_SB(); read1(KS1(), P2t(), (S), FNEXT); read1(KS1(), P5(), (S), FNEXT); CBS_SB(); \
            mma3(KS0(), FCUR, 3, 4); CBS_SB(); read1(KS1(), R6(), (S), FNEXT); CBS_SB();   \
            mma3(KS0(), FCUR, 4, 5); CBS_SB(); read1(KS1(), R7(), (S), FNEXT); CBS_SB();   \
            mma3(KS0(), FCUR, 5, 6); CBS_SB(); read1(KS1(), R8(), (S), FNEXT); CBS_SB();   \
            mma3(KS0(), FCUR, 6, NM3);                                                     \
            waitFrags(FNEXT);                                                              \
        } while (0)
#define CBS_STEP3B(WAITN, S, FCUR, FNEXT)                                                  \
        do {                                                                               \
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN) : "memory");                  \
            __builtin_amdgcn_s_barrier();                                                  \
            CBS_SB();                                                                      \
            issue((S) + RING, true);                                                       \
            CBS_SB();                                                                      \
            mma3(KS1(), FCUR, 0, 1); CBS_SB();                                             \
            read1(KS0(), P0(), (S) + 1, FNEXT); read1(KS0(), P3(), (S) + 1, FNEXT); CBS_SB(); issuePart(P0()); CBS_SB(); \
            mma3(KS1(), FCUR, 1, 2); CBS_SB();                                             \
            read1(KS0(), P1(), (S) + 1, FNEXT); read1(KS0(), P4(), (S) + 1, FNEXT); CBS_SB(); issuePart(P1()); CBS_SB(); \
            mma3(KS1(), FCUR, 2, 3); CBS_SB();                                             \
            read1(KS0(), P2t(), (S) + 1, FNEXT); read1(KS0(), P5(), (S) + 1, FNEXT); CBS_SB(); issuePart(P2t()); CBS_SB(); \
            mma3(KS1(), FCUR, 3, 4); CBS_SB(); read1(KS0(), R6(), (S) + 1, FNEXT); CBS_SB(); issuePart(P3()); CBS_SB(); \
            mma3(KS1(), FCUR, 4, 5); CBS_SB(); read1(KS0(), R7(), (S) + 1, FNEXT); CBS_SB(); issuePart(P4()); CBS_SB(); \
            mma3(KS1(), FCUR, 5, 6); CBS_SB(); read1(KS0(), R8(), (S) + 1, FNEXT); CBS_SB(); issuePart(P5()); CBS_SB(); \
            mma3(KS1(), FCUR, 6, NM3);                                                     \
            waitFrags(FNEXT);                                                              \
        } while (0)

        // fp16, one column tile per wave (the 64 x 64 tile: every shallow layer of a pose network), consumers folded in: their
        // state values at this item's pixels are requested HERE -- in front of the ring priming, so they are the wave's
        // OLDEST vector loads (they retire in front of every DMA the hand-counted waits count) and their round trip runs
        // under the whole stage loop -- and the outputs stay in registers for the comparison: the detection adds no round
        // trip to the launch.  (Two column tiles per wave: the registers are not there -- the values are read back in the
        // epilogue; a split item's detection is the reduce launch's.)
        constexpr bool NEXT_REGS = HALF && TN == 1;
        _Float16 hv16[NEXT_REGS ? 16 : 1], sv16[NEXT_REGS ? CBH_NEXT : 1][NEXT_REGS ? 16 : 1];
        if constexpr (NEXT_REGS) {
            if (nNext > 0 && SK == 1) {      // (uniform)
                const int pixLd = max(s_tilePix[wn * 32 + l31], 0);
                const int nx2 = nNext > 1 ? 1 : 0;
                const int mb0 = m0 + wm * 32 + 4 * h;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long o = (long)min(mb0 + 8 * (r >> 2) + (r & 3), Kq - 1) * HW + pixLd;
                    sv16[0][r] = ext.L[q].next[0].state[o];
                    if (CBH_NEXT > 1) sv16[CBH_NEXT > 1 ? 1 : 0][r] = ext.L[q].next[nx2].state[o];
                }
            }
        }
        // window order: what the pooled detection in the epilogue compares and falls back on is requested HERE, like the
        // fp16 consumers' state above (the wave's oldest vector loads; their round trip runs under the stage loop): the
        // consumer's state at the lane's window, and the OLD output of the lane's pixel -- a pixel of a touched window
        // that did not change itself is not computed (its output would come out of other operand bits than the dense first
        // frame's), its stored output takes part in the maximum.
        float oldv[WIN ? 16 : 1], nsv[WIN ? 4 : 1];
        int anyPos = -1, wPos = -1;
        if constexpr (WIN) {
            const int nl = wn * 32 + l31;
            anyPos = s_tileAny[nl], wPos = s_winPos[nl >> 2];
            const int aLd = max(anyPos, 0), wLd = max(wPos, 0);      // (clamped addresses, predicated uses)
            const long H2W2 = (long)ext.H2 * ext.W2;
            const float* oq = p.seq[0].out;
            // (a wave none of whose windows holds an unchanged pixel -- the interior of a changed region -- skips the old
            //  outputs' sixteen load instructions)
            if (__builtin_amdgcn_ballot_w64(anyPos >= 0 && s_tilePix[nl] < 0) != 0ull) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = min(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, Kq - 1);
                    oldv[r] = oq[(long)m * HW + aLd];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) oldv[r] = 0.f;
            }
            // the consumer's state: the four lanes of a window share its channels -- lane k of the quad takes the four
            // consecutive channels wm 32 + 8 k + 4 h + (0..3)
            const int c0 = wm * 32 + 8 * (nl & 3) + 4 * h;
#pragma unroll
            for (int e = 0; e < 4; ++e) nsv[e] = ext.nstate[(long)min(c0 + e, Kq - 1) * H2W2 + wLd];
        }
        Frags F0, F1;
        CBS_STAMP_AT(2);
#pragma unroll
        for (int i = 0; i < RING; ++i) issue(sBeg + i);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 1) * DPW) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (X3) {
            readA3(KS0(), sBeg, F0);
#pragma unroll
            for (int j = 0; j < TN; ++j) readB3(KS0(), sBeg, F0, j);
        } else {
            readFrags(sBeg, F0);
        }
        waitFrags(F0);
        if (t < BM) s_bias[t] = biasv;      // (read in the epilogue, many barriers later)
        CBS_STAMP_AT(3);
        int s = sBeg;
        if constexpr (X3) {
            for (int c = c0; c < c1; ++c) {
                const int cEnd = chunkBeg(c + 1);
                for (; s < cEnd; ++s) {
                    CBS_STEP3A(s, F0, F1);
                    CBS_STEP3B((RING - 2) * DPW, s, F1, F0);
                }
                if (c + 1 < c1) fold();
            }
        }
        // Chunks before the last one (an unsplit deep item) are whole pairs of steps and end in a fold; the last
        // chunk may end in a single step.  The last steps of the item issue dead DMAs and read a stage that does
        // not exist (a ring slot with old bytes) into the fragment set nobody multiplies.
        if constexpr (!X3) {
            for (int c = c0; c < c1; ++c) {
                const int cEnd = chunkBeg(c + 1);
                for (; s + 1 < cEnd; s += 2) {
                    CBS_STEP((RING - 2) * DPW, true, s, F0, F1);
                    CBS_STEP((RING - 2) * DPW, true, s + 1, F1, F0);
                }
                if (c + 1 < c1) fold();
            }
            if (s < sEnd) CBS_STEP((RING - 2) * DPW, true, s, F0, F1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (no dead DMA lands in the ring of the next item)
#undef CBS_STEP
#undef CBS_STEP3A
#undef CBS_STEP3B
#undef CBS_SB
        __builtin_amdgcn_sched_barrier(0);      // (nothing of the epilogue -- its loads, its addresses -- up into the steps)
        CBS_STAMP_AT(4);
        CBS_STAMP_AT(7);

        // ---- epilogue: C/D map of the 32x32 tile: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
        float* out = p.seq[q].out;
        // (the accumulate form's arguments through an opaque copy of the argument pointer, like the exact path's: read
        //  here, not at kernel entry)
        int accum;
        float* reluPlane;
        {
            const char* ka = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(ka));
            const CbsParams* pp = (const CbsParams*)ka;
            accum = pp->accumulate;
            reluPlane = accum ? pp->seq[q].reluOut : nullptr;
        }
        if (CBS_DBGBIT(32)) continue;
        bool finishHere = false;      // (last-arriver experiment: this workgroup holds the tile's final sums in acc1)
        if (SK > 1) {
            // partial tile -> slab [BM/4][BN] float4 (four consecutive output channels of a pixel), plain stores:
            // the slices meet behind the launch boundary (cbs_reduce_kernel)
            float4* slab = (float4*)p.slabs + (long)it * (TILE / 4);
#ifdef CBS_LAST_ARRIVER
            if constexpr (X3 && BM == 128) {
                if (lastArr) {
                    // Experiment (VERDICT round 5, #2b): the slices meet inside the launch.  Partial tiles go out with
                    // device-scope (sc1) stores; a counter per (pixel tile, row tile) tells the last slice to arrive, which
                    // reads the other three with sc1 loads, sums all four in slice order -- its own from registers -- and
                    // runs the ordinary epilogue; the second launch then finds final outputs (its unsplit path).
                    typedef float f32x4 __attribute__((ext_vector_type(4)));
                    const float4* slabU = (const float4*)cbs_uniform_ptr(slab);
                    asm volatile("s_nop 4" ::"s"(slabU));
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int nl = (wn * TN + j) * 32 + l31;
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4) {
                            const int mq = wm * 8 + 2 * r4 + h;
                            const f32x4 v = {comb(j, 4 * r4), comb(j, 4 * r4 + 1), comb(j, 4 * r4 + 2), comb(j, 4 * r4 + 3)};
                            // (inline asm is opaque to the compiler's hazard recogniser: the wait states of a > 64-bit store before its
                            //  data registers are written again, and of a VALU-written SGPR before VMEM reads it, are spelled out)
                            asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" ::"v"((mq * BN + nl) * 16), "v"(v), "s"(slabU) : "memory");
                        }
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __shared__ int s_arrived;
                    __syncthreads();
                    int* ctr = p.arriveTiles + (ptg * MT + mt);      // (one per (pixel tile, row tile))
                    if (t == 0) {
                        const int old = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (old == SK - 1) __hip_atomic_store(ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        s_arrived = old;
                    }
                    __syncthreads();
                    if (s_arrived != SK - 1) continue;      // (uniform: not the last one)
                    const float4* s0 = (const float4*)p.slabs + (long)(it - slice * MT) * (TILE / 4);      // slice 0 of the tile
                    f32x4 o[3][TN * 4];
                    const float4* sk[3];      // (the other three slices, in order; wave-uniform addresses in SGPRs)
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        sk[k] = (const float4*)cbs_uniform_ptr(s0 + (long)(k < slice ? k : k + 1) * MT * (TILE / 4));
                    }
                    asm volatile("s_nop 4" ::"s"(sk[0]), "s"(sk[1]), "s"(sk[2]));      // (v_readfirstlane -> VMEM address: 5 wait states)
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
#pragma unroll
                        for (int j = 0; j < TN; ++j)
#pragma unroll
                            for (int r4 = 0; r4 < 4; ++r4) {
                                const int nl = (wn * TN + j) * 32 + l31, mq = wm * 8 + 2 * r4 + h;
                                asm volatile("global_load_dwordx4 %0, %1, %2 sc1"
                                             : "=v"(o[k][j * 4 + r4])
                                             : "v"((mq * BN + nl) * 16), "s"(sk[k])
                                             : "memory");
                            }
                    }
#pragma unroll
                    for (int k = 0; k < 3; ++k)
#pragma unroll
                        for (int i = 0; i < TN * 4; ++i) asm volatile("s_waitcnt vmcnt(0)" : "+v"(o[k][i])::"memory");
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const float own = comb(j, i);
                            float sum = 0.f;      // (cbs_reduce_tail_kernel's order: 0 + slice 0 + slice 1 + slice 2 + slice 3)
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                const int k = c < slice ? c : c - 1;
                                const float ov = k == 0 ? o[0][j * 4 + (i >> 2)][i & 3]
                                                        : (k == 1 ? o[1][j * 4 + (i >> 2)][i & 3] : o[2][j * 4 + (i >> 2)][i & 3]);
                                sum += c == slice ? own : ov;
                            }
                            acc1[j][i] = sum, acc2[j][i] = 0.f;
                        }
                    finishHere = true;
                }
            }
#endif
            if (!finishHere) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int nl = (wn * TN + j) * 32 + l31;
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const int mq = wm * 8 + 2 * r4 + h;
                        slab[mq * BN + nl] = make_float4(comb(j, 4 * r4), comb(j, 4 * r4 + 1), comb(j, 4 * r4 + 2),
                                                         comb(j, 4 * r4 + 3));
                    }
                }
                CBS_STAMP_AT(5);
#ifdef CBS_STAMP
                cbs_first = false;
#endif
                continue;
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nl = (wn * TN + j) * 32 + l31;
            const int pix = s_tilePix[nl];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, m = m0 + ml;
                float v = comb(j, r);
                if (folds > 0) v = run[j][r] + v;        // (the last chunk joins the running sum)
                v = fmaf(v, p.outScale, s_bias[ml]);      // (an explicit fma here, in the reduce launch and in the
                if (reluq) v = v <= 0.f ? 0.f : v;         //  fused tail: one rounding at all three sites)
                if (HALF) {
                    if (NEXT_REGS) hv16[NEXT_REGS ? r : 0] = m < Kq ? (_Float16)v : (_Float16)0;
                    if (pix >= 0 && m < Kq) ((_Float16*)out)[(long)m * HW + pix] = (_Float16)v;
                    continue;
                }
                if constexpr (WIN)      // (this pixel's value in the window's maximum: new, old, or none -- outside the map)
                    oldv[WIN ? r : 0] = pix >= 0 ? v : (anyPos >= 0 ? oldv[WIN ? r : 0] : -__builtin_inff());
                if (pix >= 0 && m < Kq) {
                    if (accum) {      // fine-grained: the sum of the products joins what the output holds
                        v += out[(long)m * HW + pix];
                        if (reluPlane) reluPlane[(long)m * HW + pix] = v <= 0.f ? 0.f : v;
                    }
                    out[(long)m * HW + pix] = v;
                }
            }
        }
        if constexpr (WIN) {
            // The pooled change detection of the layer behind the 2x2 pool (cbs_detect_kernel<POOL>'s work, per window of
            // this tile): the four pixels of a window sit in four consecutive lanes -- two DPP quad permutes give every lane
            // the maximum (max(max(p00, p01), max(p10, p11)), the detection kernel's association) of its 16 channels --;
            // strict > against the consumer's state over all channels (cbconv2d_cg_backend.cu:56), the window's verdict
            // through LDS (channels live in both lane halves and both row-tile waves); feedback refresh of the f32 state
            // and of its pixel-major split copy at the changed windows (.cu:74-80), their dilation into the frame mask.
            CBS_STAMP_AT(12);
            const int nl = wn * 32 + l31, win = nl >> 2, kq = nl & 3;
            const int c0 = wm * 32 + 8 * kq + 4 * h;      // this lane's four channels of the window (a whole quarter of a part)
            float pv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float x = oldv[WIN ? r : 0];
                x = fmaxf(x, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true)));
                x = fmaxf(x, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true)));
                // accumulator r holds channel wm 32 + (r & 3) + 8 (r >> 2) + 4 h: the lane keeps r = 4 kq + e
                if ((r >> 2) == kq) pv[r & 3] = x;
            }
            const bool mine = wPos >= 0 && c0 < Kq;      // (K is a multiple of 16: a lane's four channels exist or do not)
            bool chg = false;
#pragma unroll
            for (int e = 0; e < 4; ++e) chg |= cb_changed(nsv[WIN ? e : 0], pv[e], ext.th);
            {
                // (the wave's eight windows' verdicts as one uniform value: one LDS atomic by one lane -- a per-lane atomicOr
                //  with differing operands is compiled into a loop over the active lanes)
                const unsigned long long bal = __ballot(chg && mine);
                unsigned q4 = (unsigned)(bal | (bal >> 32));
                q4 |= q4 >> 1;
                q4 |= q4 >> 2;
                unsigned m8 = 0u;
#pragma unroll
                for (int w = 0; w < 8; ++w) m8 |= ((q4 >> (4 * w)) & 1u) << w;
                if (lane == 0 && m8) atomicOr(&s_wchg, m8 << (wn * 8));
            }
            __syncthreads();
            CBS_STAMP_AT(13);
            const unsigned mW = __builtin_amdgcn_readfirstlane(s_wchg);
            if (mW != 0u) {      // (uniform)
                if (mine && ((mW >> win) & 1u)) {
                    const long H2W2 = (long)ext.H2 * ext.W2;
#pragma unroll
                    for (int e = 0; e < 4; ++e) ext.nstate[(long)(c0 + e) * H2W2 + wPos] = pv[e];
                    const int y2 = cbs_div(wPos, ext.magicW2), x2 = wPos - y2 * ext.W2;
                    char* rec = ext.nS + CBS_SPAD + ((long)(y2 + ext.padY) * ext.Wp + (x2 + ext.padXL)) * ext.rec;
                    const bool over = cbs_store_quarter(rec, c0 >> 4, (c0 >> 3) & 1, (c0 >> 2) & 1, ext.planes, pv);
                    if (over && ext.nflag) *ext.nflag = 1;
                }
                CBS_STAMP_AT(14);
                // Dilation into the consumer's frame mask: the tile's changed windows are first united per mask word
                // (lanes of the last wave: one window each, the lowest lane of every (row, word) key collects the bits),
                // then each united word goes out dilated like cbs_detect_kernel's -- a few dozen atomics per tile.  (One
                // atomic chain per WINDOW, the first form, was 22 000 atomics per frame on the ten cache lines of an 80 x 120
                // mask: 25 us of this launch, served one after the other at the memory side.)
                if (wave == NW - 1) {
                    const int w = lane & (BN / 4 - 1);
                    const bool on = lane < BN / 4 && ((mW >> w) & 1u);
                    int key = -1 - lane, y2 = 0, wi = 0;
                    unsigned long long bit = 0ull;
                    if (on) {
                        const int wp = s_winPos[w];
                        y2 = cbs_div(wp, ext.magicW2);
                        const int x2 = wp - y2 * ext.W2;
                        wi = x2 >> 6, key = y2 * ext.wpr2 + wi, bit = 1ull << (x2 & 63);
                    }
                    unsigned long long acc = 0ull;
                    bool leader = on;
#pragma unroll
                    for (int j = 0; j < BN / 4; ++j) {
                        const int kj = __shfl(key, j);
                        const unsigned long long bj = __shfl(bit, j);
                        if (kj == key) {
                            acc |= bj;
                            if (j < lane) leader = false;
                        }
                    }
                    if (leader) {
                        unsigned long long D = acc, SR = 0ull, SL = 0ull;
                        for (int d = 1; d <= ext.kWH; ++d) {
                            D |= (acc << d) | (acc >> d);
                            SR |= acc >> (64 - d);
                            SL |= acc << (64 - d);
                        }
                        D &= cbs_valid_mask(ext.W2, wi);
                        SR = (wi + 1 < ext.wpr2) ? (SR & cbs_valid_mask(ext.W2, wi + 1)) : 0ull;
                        if (wi == 0) SL = 0ull;
                        for (int yy = max(y2 - ext.kHH, 0); yy <= min(y2 + ext.kHH, ext.H2 - 1); ++yy) {
                            unsigned long long* row = ext.nmasks + (long)yy * ext.wpr2 + wi;
                            atomicOr(row, D);
                            if (SR) atomicOr(row + 1, SR);
                            if (SL) atomicOr(row - 1, SL);
                        }
                    }
                }
            }
        }
        if constexpr (HALF) {
            // The change detection of the layers that consume this output (copy mode), on the values just written: a lane
            // holds 4 x 4 consecutive output channels of its pixels = input channels of the consumer, and reads them back
            // from prevOutput (its own stores; the accumulators are dead by now -- rolled loops, a few registers).  Per
            // value: strict > in half precision against the consumer's prevInput (cbconv2d_cg_half_backend.cu:24-35); a
            // value that differs bit for bit goes into prevInput and -- four channels, 8 bytes -- into the consumer's
            // pixel-major record.  Pixels this layer did not recompute hold last frame's bits: nothing to do.
            if (nNext > 0) {      // (uniform)
                const _Float16* outh = (const _Float16*)out;
                // (ONE round trip per column tile: the 16 outputs and the 16 state values of every consumer are requested
                //  together -- clamped addresses, predicated uses: a predicated load is a branch, and a branch per value is
                //  a round trip per value)
                const int nx2 = nNext > 1 ? 1 : 0;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int nl = (wn * TN + j) * 32 + l31;
                    const int pix = s_tilePix[nl];
                    const int pixLd = max(pix, 0);
                    const int py = cbs_div(pixLd, p.magicW), px = pixLd - py * p.W;
                    const int mb0 = m0 + wm * 32 + 4 * h;
                    _Float16 h16[16], s16[CBH_NEXT][16];
                    if constexpr (NEXT_REGS) {      // (requested in front of the output stores, outputs from registers)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            h16[r] = hv16[NEXT_REGS ? r : 0];
                            s16[0][r] = sv16[0][NEXT_REGS ? r : 0];
                            if (CBH_NEXT > 1) s16[CBH_NEXT > 1 ? 1 : 0][r] = sv16[NEXT_REGS && CBH_NEXT > 1 ? 1 : 0][NEXT_REGS ? r : 0];
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const long o = (long)min(mb0 + 8 * (r >> 2) + (r & 3), Kq - 1) * HW + pixLd;
                            h16[r] = outh[o];
                            s16[0][r] = ext.L[q].next[0].state[o];
                            if (CBH_NEXT > 1) s16[CBH_NEXT > 1 ? 1 : 0][r] = ext.L[q].next[nx2].state[o];
                        }
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (mb0 + 8 * (r >> 2) + (r & 3) >= Kq) h16[r] = (_Float16)0;   // (the record's padding channels stay zero)
                    }
#pragma unroll
                    for (int c = 0; c < CBH_NEXT; ++c) {
                        if (c >= nNext || pix < 0) continue;
                        const CbhNext& nx = ext.L[q].next[c];
                        _Float16* nst = nx.state;
                        char* rec = nx.S + CBS_SPAD + ((long)(py + nx.padY) * nx.Wp + (px + nx.padXL)) * nx.rec;
                        const _Float16 thc = (_Float16)nx.th;
                        const int cpad = nx.rec >> 1;      // the consumer's padded channel count
                        bool chg = false;
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4) {
                            const int mb = mb0 + 8 * r4;
                            unsigned dm = 0u;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                if (mb + e >= Kq) continue;
                                chg |= cb_changed(s16[c][4 * r4 + e], h16[4 * r4 + e], thc);
                                dm |= (unsigned)cb_differs(s16[c][4 * r4 + e], h16[4 * r4 + e]) << e;
                            }
                            if (dm) {
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    if ((dm >> e) & 1u) nst[(long)(mb + e) * HW + pix] = h16[4 * r4 + e];
                                if (mb < cpad) {
                                    typedef _Float16 halfx4 __attribute__((ext_vector_type(4)));
                                    *(halfx4*)(rec + mb * 2) =
                                        halfx4{h16[4 * r4], h16[4 * r4 + 1], h16[4 * r4 + 2], h16[4 * r4 + 3]};
                                }
                            }
                        }
                        if (chg) atomicOr(&s_nchg[c * (BN / 32) + (nl >> 5)], 1u << (nl & 31));
                    }
                }
                // the consumers' masks: every changed pixel of the tile, dilated by their filters
                __syncthreads();
#pragma unroll 1
                for (int c = 0; c < nNext; ++c) {
                    const CbhNext& nx = ext.L[q].next[c];
                    if (t < BN && ((s_nchg[c * (BN / 32) + (t >> 5)] >> (t & 31)) & 1u)) {
                        const int pix = s_tilePix[t];
                        const int py = cbs_div(pix, p.magicW), px = pix - py * p.W;
                        cbh_or_dilated(nx.masks, py, px, p.H, p.W, p.wpr, nx.kHH, nx.kWH);
                    }
                }
            }
        }
        CBS_STAMP_AT(5);
#ifdef CBS_STAMP
        cbs_first = false;
#endif
    }

    if constexpr (SIDE) {
        if (ext.sideFrame) {      // (uniform)
            // the side job: by the workgroups that had no item, if there are enough of them -- else by everybody, a slice each
            const int busy = min(items, (int)gridDim.x), nIdle = (int)gridDim.x - busy;
            const bool idleDo = nIdle >= 32;
            const int parts = idleDo ? nIdle : (int)gridDim.x, part = idleDo ? (int)blockIdx.x - busy : (int)blockIdx.x;
            if (part >= 0) {
                const int per = (ext.sideHW + parts - 1) / parts;
                const int i1 = min(ext.sideHW, (part + 1) * per);
                for (int i = part * per + t; i < i1; i += NT) {
                    bool chg = false;
                    for (int c = 0; c < ext.sideC; ++c)
                        chg |= cb_changed(ext.sideState[(long)c * ext.sideHW + i], ext.sideFrame[(long)c * ext.sideHW + i],
                                          ext.sideTh);
                    if (chg)
                        for (int c = 0; c < ext.sideC; ++c)
                            ext.sideState[(long)c * ext.sideHW + i] = ext.sideFrame[(long)c * ext.sideHW + i];
                }
            }
        }
    }

    // Last workgroup out zeroes the masks (every workgroup copied them into its LDS before its first barrier).  The
    // workgroups count themselves on SHARDED counters -- lines of their own in the unused second mask slot, a top
    // counter behind them: a returning atomic on ONE word is served every 11 ns, so a grid of 256-512 workgroups that
    // finish together spent 3-6 us of the launch queueing for their tickets (round 4; the row-pair kernel's first
    // form showed it: 9 us for 768 workgroups).
    // (all masks empty -- every workgroup sees that alike: there is nothing to zero and nobody to wait for)
    if (totAll == 0) return;
    __syncthreads();
    if (t == 0) {
        int* ctl = (int*)(p.seq[0].masks + 2 * (long)MW);
        const int nSh = p.arriveShards;
        int last = 0;
        if (nSh <= 1) {
            last = __hip_atomic_fetch_add(ctl + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
        } else {
            const int sh = blockIdx.x % nSh, expected = ((int)gridDim.x - sh + nSh - 1) / nSh;
            int* c = (int*)(p.seq[0].masks + (long)MW) + sh * 32;
            if (__hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == expected - 1) {
                __hip_atomic_store(c, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int active = min(nSh, (int)gridDim.x);
                last = __hip_atomic_fetch_add(ctl + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == active - 1;
            }
        }
        s_wsum[0] = last;
    }
    __syncthreads();
    if (s_wsum[0]) {
        for (int i = t; i < E; i += NT) {
            const int q = cbs_div(i, p.magicMW), w = i - q * MW;
            ((unsigned long long*)s_maskPtr[q])[w] = 0ull;
        }
        if (t == 0) {
            int* ctl = (int*)(p.seq[0].masks + 2 * (long)MW);
            __hip_atomic_store(ctl + 2, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Second launch of a split contraction: sums the SK slabs of every tile in slice order, scales, adds the bias,
// applies the ReLU and scatters (all CUs, one float4 of four output channels per thread and step).
__global__ __launch_bounds__(256) void cbs_reduce_kernel(CbsParams p, int BM, int BN) {
    cb_touch_kernarg<sizeof(CbsParams) + 8>();
    const int SK = p.info[CBS_INFO_SK];
    if (SK <= 1) return;
    const int MT = p.info[CBS_INFO_MT], CMB = MT * SK, TILE4 = BM * BN / 4, HW = p.H * p.W;
    int tilesBefore[CBS_MAXSEQ + 1];
    tilesBefore[0] = 0;
#pragma unroll
    for (int q = 0; q < CBS_MAXSEQ; ++q) tilesBefore[q + 1] = tilesBefore[q] + p.info[CBS_INFO_TP + q];
    const int TP = tilesBefore[CBS_MAXSEQ];
    const float4* __restrict__ slabs = (const float4*)p.slabs;
    const long total = (long)TP * MT * TILE4;
    for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long)gridDim.x * 256) {
        const int tile = (int)(g / TILE4), c = (int)(g - (long)tile * TILE4);
        const int ptg = tile / MT, mt = tile % MT;
        const int nl = c % BN, mq = c / BN;
        int q = 0;
#pragma unroll
        for (int u = 1; u < CBS_MAXSEQ; ++u)
            if (u < p.nSeq && ptg >= tilesBefore[u]) q = u;
        const int n = (ptg - tilesBefore[q]) * BN + nl;
        const int N = p.info[CBS_INFO_TP + CBS_MAXSEQ + q];
        const float4* sl = slabs + ((long)ptg * CMB + mt) * TILE4 + c;      // slice j: + j * MT * TILE4
        const int pix = n < N ? p.seq[q].listOut[n] : -1;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        for (int j0 = 0; j0 < SK; j0 += 8) {
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j0 + j < SK) v[j] = sl[(long)(j0 + j) * MT * TILE4];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j0 + j < SK) s0 += v[j].x, s1 += v[j].y, s2 += v[j].z, s3 += v[j].w;
        }
        if ((unsigned)pix >= (unsigned)HW) continue;
        const int m = mt * BM + 4 * mq;
        const float sv[4] = {s0, s1, s2, s3};
        float* out = p.seq[q].out;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (m + e >= p.K) continue;
            const float bv = !p.bias ? 0.f : (p.halfOut ? (float)((const _Float16*)p.bias)[m + e] : p.bias[m + e]);
            float v = fmaf(sv[e], p.outScale, bv);
            if (p.relu) v = v <= 0.f ? 0.f : v;
            if (p.halfOut) {      // (fp16 layer: bias and outputs are f16)
                ((_Float16*)out)[(long)(m + e) * HW + pix] = (_Float16)v;
                continue;
            }
            if (p.accumulate) {
                v += out[(long)(m + e) * HW + pix];
                float* rp = p.seq[q].reluOut;
                if (rp) rp[(long)(m + e) * HW + pix] = v <= 0.f ? 0.f : v;
            }
            out[(long)(m + e) * HW + pix] = v;
        }
    }
}

// Second launch of a split fp16 contraction (round 6; cbs_reduce_kernel's work for the fp16 layers): one workgroup per
// group of 16 changed pixels x ALL output channels -- thread (px = t & 15, channel quad t >> 4 + 16 i) sums the SK slabs
// of its float4 in slice order, adds the bias, applies the ReLU, rounds to f16 once and scatters -- and then, with every
// channel of its pixels in one workgroup, the change detection of the layer's CONSUMERS on the values just written (see
// cbs_conv_kernel's fp16 epilogue: the same per-value steps), their dilated masks ORed by (pixel, row) threads.
template <int BM, int BN>
__global__ __launch_bounds__(256) void cbh_reduce_kernel(CbsParams p, CbhExt ext) {
    cb_touch_kernarg<sizeof(CbsParams) + sizeof(CbhExt)>();
    if (p.info[CBS_INFO_SK] <= 1) return;
    const int MT = p.info[CBS_INFO_MT], TILE4 = BM * BN / 4, HW = p.H * p.W;
    int tilesBefore[CBH_GROUP + 1], slabBase[CBH_GROUP], chQ[CBH_GROUP];
    tilesBefore[0] = 0;
    {
        int run = 0;      // (slab = work item index: the layers' item ranges follow each other, cbs_conv_kernel)
#pragma unroll
        for (int q = 0; q < CBH_GROUP; ++q) {
            const int tq = p.info[CBS_INFO_TP + q];
            chQ[q] = p.info[CBS_INFO_CH + q];
            tilesBefore[q + 1] = tilesBefore[q] + tq;
            slabBase[q] = run;
            run += tq * MT * chQ[q];
        }
    }
    constexpr int PXG = 16, GP = BN / PXG;
    const int groups = tilesBefore[CBH_GROUP] * GP;
    const int t = threadIdx.x, px = t & 15;
    __shared__ unsigned s_chg[CBH_NEXT];
    const float4* __restrict__ slabs = (const float4*)p.slabs;
    // work unit = (group of 16 changed pixels, chunk of 32 channel quads = 128 output channels): a layer of 512 output
    // channels has four units per pixel group -- 1000 changed pixels of such a layer are 63 groups, too few workgroups for
    // a launch whose every step is a round trip.  The any-channel OR of the consumers' predicate needs no meeting of the
    // chunks: every unit ORs the pixels ITS channels found changed into the mask (atomic OR is idempotent).
    const int QT = MT * (BM / 4);      // channel quads of a pixel
    const int NCC = (QT + 31) / 32;
    for (int unit = blockIdx.x; unit < groups * NCC; unit += gridDim.x) {
        const int g = unit / NCC, cc = unit - g * NCC;
        const int ptg = g / GP, gi = g - ptg * GP;
        int q = 0;
#pragma unroll
        for (int u = 1; u < CBH_GROUP; ++u)
            if (u < p.nSeq && ptg >= tilesBefore[u]) q = u;
        const int n0 = (ptg - tilesBefore[q]) * BN + gi * PXG;
        const int N = p.info[CBS_INFO_TP + CBS_MAXSEQ + q];
        if (n0 >= N) continue;      // (uniform: a tile's last groups may be empty)
        static_assert(CBH_GROUP == 2, "two layers");
        const int SK = q ? chQ[1] : chQ[0], CMB = MT * SK;
        const long tileSlab = (q ? slabBase[1] : slabBase[0]) + (long)(ptg - tilesBefore[q]) * CMB;
        const CbhLayer& L = ext.L[q];
        const int Kq = L.K, nNext = L.nNext;
        int pix = n0 + px < N ? p.seq[q].listOut[n0 + px] : -1;
        if ((unsigned)pix >= (unsigned)HW) pix = -1;
        const int pixLd = max(pix, 0);
        const int py = cbs_div(pixLd, p.magicW), pxx = pixLd - py * p.W;
        __syncthreads();      // (the previous group's flags have been read)
        if (t < CBH_NEXT) s_chg[t] = 0u;
        __syncthreads();
        _Float16* out = (_Float16*)p.seq[q].out;
        bool chg[CBH_NEXT];
#pragma unroll
        for (int c = 0; c < CBH_NEXT; ++c) chg[c] = false;
        const float4* sl0 = slabs + tileSlab * TILE4 + gi * PXG + px;
        const int nx2 = nNext > 1 ? 1 : 0;
        {
            const int cq0 = cc * 32 + (t >> 4);      // this thread's two quads: cq0 and cq0 + 16
            // (one round trip per round: up to 2 x 8 slab float4 and the consumers' state values of both quads are
            //  requested together -- their addresses need the pixel index only; clamped addresses, predicated uses)
            float4 v[2][8];
            _Float16 s4[2][CBH_NEXT][4];
            float acc[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int cq = min(cq0 + 16 * u, QT - 1);
                const int mt = cq / (BM / 4), mq = cq - mt * (BM / 4), m = mt * BM + 4 * mq;
                const float4* sl = sl0 + (long)mt * TILE4 + mq * BN;      // slice j: + j * MT * TILE4
#pragma unroll
                for (int j = 0; j < 8; ++j) v[u][j] = sl[(long)min(j, SK - 1) * MT * TILE4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const long o = (long)min(m + e, Kq - 1) * HW + pixLd;
                    s4[u][0][e] = nNext > 0 ? L.next[0].state[o] : (_Float16)0;
                    if (CBH_NEXT > 1) s4[u][CBH_NEXT > 1 ? 1 : 0][e] = nNext > 0 ? L.next[nx2].state[o] : (_Float16)0;
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (j < SK) s0 += v[u][j].x, s1 += v[u][j].y, s2 += v[u][j].z, s3 += v[u][j].w;
                acc[u][0] = s0, acc[u][1] = s1, acc[u][2] = s2, acc[u][3] = s3;
            }
            if (SK > 8) {      // (uniform; 16 chunks: the second eight, in slice order behind the first)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int cq = min(cq0 + 16 * u, QT - 1);
                    const int mt = cq / (BM / 4), mq = cq - mt * (BM / 4);
                    const float4* sl = sl0 + (long)mt * TILE4 + mq * BN;
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[u][j] = sl[(long)min(8 + j, SK - 1) * MT * TILE4];
                }
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (8 + j < SK)
                            acc[u][0] += v[u][j].x, acc[u][1] += v[u][j].y, acc[u][2] += v[u][j].z, acc[u][3] += v[u][j].w;
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int cq = cq0 + 16 * u;
                if (cq >= QT) continue;
                const int mt = cq / (BM / 4), mq = cq - mt * (BM / 4), m = mt * BM + 4 * mq;
                if (pix < 0 || m >= Kq) continue;
                _Float16 hv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float bv = (L.bias && m + e < Kq) ? (float)L.bias[m + e] : 0.f;
                    float r = fmaf(acc[u][e], p.outScale, bv);
                    if (L.relu) r = r <= 0.f ? 0.f : r;
                    hv[e] = m + e < Kq ? (_Float16)r : (_Float16)0;
                    if (m + e < Kq) out[(long)(m + e) * HW + pix] = hv[e];
                }
#pragma unroll
                for (int c = 0; c < CBH_NEXT; ++c) {
                    if (c >= nNext) continue;
                    const CbhNext& nx = L.next[c];
                    _Float16* nst = nx.state;
                    const _Float16 thc = (_Float16)nx.th;
                    unsigned dm = 0u;
                    bool ch = false;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (m + e >= Kq) continue;
                        ch |= cb_changed(s4[u][c][e], hv[e], thc);
                        dm |= (unsigned)cb_differs(s4[u][c][e], hv[e]) << e;
                    }
                    chg[c] |= ch;
                    if (dm) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if ((dm >> e) & 1u) nst[(long)(m + e) * HW + pix] = hv[e];
                        if (m < (nx.rec >> 1)) {
                            typedef _Float16 halfx4 __attribute__((ext_vector_type(4)));
                            char* rec = nx.S + CBS_SPAD + ((long)(py + nx.padY) * nx.Wp + (pxx + nx.padXL)) * nx.rec;
                            *(halfx4*)(rec + m * 2) = halfx4{hv[0], hv[1], hv[2], hv[3]};
                        }
                    }
                }
            }
        }
        if (nNext > 0) {      // (uniform)
#pragma unroll
            for (int c = 0; c < CBH_NEXT; ++c)
                if (c < nNext && chg[c]) atomicOr(&s_chg[c], 1u << px);
            __syncthreads();
            // thread (px, row) = (t & 15, t >> 4): one row of one changed pixel's dilated square per thread
            for (int c = 0; c < nNext; ++c) {
                const CbhNext& nx = L.next[c];
                const int dy = (t >> 4) - nx.kHH, yy = py + dy;
                if (pix >= 0 && ((s_chg[c] >> px) & 1u) && dy <= nx.kHH && yy >= 0 && yy < p.H)
                    cbh_or_dilated(nx.masks, yy, pxx, p.H, p.W, p.wpr, 0, nx.kWH);
            }
        }
    }
}

// The second launch of a deep contraction when a fused 1x1 tail (cb_tail.hip) follows the layer: per group of
// CB_TAIL_PX changed pixels it finishes the layer's outputs -- split contraction: the sum of the tile's slabs in
// chunk order, scale, bias, ReLU, scattered to prevOutput; unsplit: already there, gathered -- into an LDS tile
// and evaluates conv1x1 -> [ReLU] -> conv1x1 on it (cb_tail_tile: the tail kernel's own arithmetic).  The tail's
// gather pass over prevOutput and a launch boundary go away.
struct CbsTailArgs {
    const float* w1p;
    const float* b1;
    const float* w2;
    const float* b2;
    float* out[CBS_MAXSEQ];
    int C1, C2, relu1, relu2;
};
template <int BM, int BN>
__global__ __launch_bounds__(64 * CB_TAIL_MAXW) void cbs_reduce_tail_kernel(CbsParams p, CbsTailArgs ta) {
    extern __shared__ float cbs_tail_sm[];
    cb_touch_kernarg<sizeof(CbsParams) + sizeof(CbsTailArgs)>();
    const int C0 = p.K, C0P = (C0 + 15) / 16 * 16, C1 = ta.C1, C2 = ta.C2, HW = p.H * p.W;
    const int t = threadIdx.x, NT = blockDim.x;
    // Round trips to memory are what this launch consists of.  First: the launch info of the contraction, the
    // layer's biases, the second layer's matrix.  Second (per group): the slabs of the group's 16 pixel columns --
    // their address needs no pixel index -- and the 16 pixel indices.  Third: this wave's rows of W1.
    CB_TAIL_STAMP(0);
    const int SK = p.info[CBS_INFO_SK], MT = p.info[CBS_INFO_MT], CMB = MT * SK, TILE4 = BM * BN / 4;
    int tilesBefore[CBS_MAXSEQ + 1];
    tilesBefore[0] = 0;
#pragma unroll
    for (int q = 0; q < CBS_MAXSEQ; ++q) tilesBefore[q + 1] = tilesBefore[q] + p.info[CBS_INFO_TP + q];
    const int GP = BN / CB_TAIL_PX, groups = tilesBefore[CBS_MAXSEQ] * GP;
    if ((int)blockIdx.x >= groups) return;
    CB_TAIL_STAMP(1);
    const CbTailLds L = cb_tail_lds(cbs_tail_sm, C0P, C1, C2, NT);
    __shared__ int s_pix[CB_TAIL_PX];
    for (int i = t; i < C2 * C1; i += NT) L.W2s[i] = ta.w2[i];
    for (int i = t; i < C2; i += NT) L.b2s[i] = ta.b2[i];
    const float4* __restrict__ slabs = (const float4*)p.slabs;
    const bool hasBias = p.bias != nullptr;
    const float* biasp = hasBias ? p.bias : ta.b1;      // (no bias: any readable address, the values are not used)
    const int px = t & 15, cstep = NT >> 4;
    for (int g = blockIdx.x; g < groups; g += gridDim.x) {
        const int ptg = g / GP, gi = g - ptg * GP;
        int q = 0;
#pragma unroll
        for (int u = 1; u < CBS_MAXSEQ; ++u)
            if (u < p.nSeq && ptg >= tilesBefore[u]) q = u;
        const int nl0 = gi * CB_TAIL_PX, n0 = (ptg - tilesBefore[q]) * BN + nl0;
        const int N = p.info[CBS_INFO_TP + CBS_MAXSEQ + q];
        if (n0 >= N) continue;      // (uniform: a tile's last groups may be empty)
        float* outq = p.seq[q].out;
        // fine-grained frame (accumulate: out += W * delta, cbconv2d_fg_backend.cu:37-66): the sum of the slabs JOINS
        // what the output holds, the relu'd copy (reluOut, if the layer has one) is kept beside it, and the tail's input
        // is that copy -- what the next module of the network is handed (conv2d.py:169-173)
        float* relq = p.accumulate ? p.seq[q].reluOut : nullptr;
        int pixMine = -1;           // (thread px of every 16: the same 16 indices, no LDS hop before the scatter)
        if (n0 + px < N) pixMine = p.seq[q].listOut[n0 + px];
        // (everything below that depends only on the thread index is the same in every round of this loop, and the
        //  compiler would keep it all -- some eighty registers of addresses -- alive across the loop: an opaque copy
        //  of the index makes it recompute them, a few integer instructions per use)
        int tq = t >> 4;
        asm volatile("" : "+v"(tq));
        if ((unsigned)pixMine >= (unsigned)HW) pixMine = -1;
        const int pixLd = max(pixMine, 0);      // (loads are never predicated -- a predicated load is a branch, and
                                                //  sixteen branches in a row are sixteen round trips: clamped addresses,
                                                //  predicated USES)
        if (SK > 1) {
            // X[c][px] from the slabs (SK = CBS_CHUNKS of them): thread -> (px, channel quads t/16 + i NT/16); all loads
            // of a thread in flight.  The same number of rounds for every thread, so that all reach the barrier.
            const float4* sl0 = slabs + (long)ptg * CMB * TILE4 + nl0 + px;
            const int QN = C0P / 4, rounds = (QN + 4 * cstep - 1) / (4 * cstep);
            for (int rd = 0; rd < rounds; ++rd) {
                const int cq0 = tq + rd * 4 * cstep;
                float4 v[4][CBS_CHUNKS], bq[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int cq = min(cq0 + u * cstep, QN - 1), mt = cq / (BM / 4), mq = cq - mt * (BM / 4);
#pragma unroll
                    for (int j = 0; j < CBS_CHUNKS; ++j) v[u][j] = sl0[((long)j * MT + mt) * TILE4 + mq * BN];
                    bq[u] = *(const float4*)(biasp + (hasBias ? min(4 * cq, p.K - 4) : 0));
                }
                __builtin_amdgcn_sched_barrier(0);      // (all loads requested before anything of the second half)
                if (rd == 0) {
                    __syncthreads();   // previous group's Hs / s_pix / Xs reads are done
                    if (t < CB_TAIL_PX) s_pix[t] = pixMine;
                    CB_TAIL_STAMP(2);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int cq = cq0 + u * cstep;
                    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
                    for (int j = 0; j < CBS_CHUNKS; ++j) s0 += v[u][j].x, s1 += v[u][j].y, s2 += v[u][j].z, s3 += v[u][j].w;
                    const float sv[4] = {s0, s1, s2, s3}, bv[4] = {bq[u].x, bq[u].y, bq[u].z, bq[u].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int m = 4 * cq + e;
                        float r = fmaf(sv[e], p.outScale, hasBias ? bv[e] : 0.f);
                        if (p.relu) r = r <= 0.f ? 0.f : r;
                        if (cq < QN) {
                            if (p.accumulate) {      // (uniform; the old outputs: one more round trip, as cbs_reduce_kernel's)
                                if (pixMine >= 0 && m < p.K) {
                                    r += outq[(long)m * HW + pixMine];
                                    outq[(long)m * HW + pixMine] = r;
                                    if (relq) {
                                        r = r <= 0.f ? 0.f : r;
                                        relq[(long)m * HW + pixMine] = r;
                                    }
                                }
                            } else if (pixMine >= 0 && m < p.K) {
                                outq[(long)m * HW + pixMine] = r;
                            }
                            L.Xs[m * CB_TAIL_PX + px] = (m < p.K && pixMine >= 0) ? r : 0.f;
                        }
                    }
                }
            }
        } else {
            // unsplit: the contraction's epilogue has written prevOutput; gather as the tail kernel does
            __syncthreads();
            if (t < CB_TAIL_PX) s_pix[t] = pixMine;
            const float* gsrc = relq ? relq : outq;      // (fine-grained with a relu'd copy: the tail reads the copy)
            for (int c0 = tq; c0 < C0P; c0 += 16 * cstep) {
                float v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = gsrc[(long)min(c0 + u * cstep, C0 - 1) * HW + pixLd];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int c = c0 + u * cstep;
                    if (c < C0P) L.Xs[c * CB_TAIL_PX + px] = (pixMine >= 0 && c < C0) ? v[u] : 0.f;
                }
            }
        }
        // this wave's rows of W1: requested once the slab registers are free (requested earlier -- beside the slabs,
        // or behind their barrier -- the launch came out longer: the barrier waits for them too, and the stores of the
        // sums queue behind them), per group -- few workgroups see a second one -- rather than kept across the loop
        asm volatile("" ::: "memory");
        CbTailPre P;
        cb_tail_preload(P, ta.w1p, ta.b1, C0P, C1);
        CB_TAIL_STAMP(3);
        cb_tail_tile(L, s_pix, P, ta.w1p, ta.out[q], C0P, C1, C2, HW, ta.relu1, ta.relu2);
        CB_TAIL_STAMP(7);
    }
}

int cbs_num_cus() {
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

template <int BM, int BN, int WM, int WN, int PRE_CAP, bool MASK_LDS, int RING, int AR = 0>
int cbs_launch_conv(const CbsParams& p, int perCU, const CbsTailArgs* tail, hipStream_t s,
                    const typename CbsExtOf<AR>::type& ext = typename CbsExtOf<AR>::type()) {
    if ((long)p.nSeq * p.maskWords > PRE_CAP) return CB_ERR_UNSUPPORTED;
    const bool second = p.slabs && (p.nStages >= 48 || p.forceSK > 0);
    if (tail && !second) return CB_ERR_UNSUPPORTED;      // (the fused tail reads the launch info of a deep contraction)
    dim3 grid((unsigned)(perCU * cbs_num_cus())), block(64 * WM * WN);
    hipLaunchKernelGGL((cbs_conv_kernel<BM, BN, WM, WN, PRE_CAP, MASK_LDS, RING, AR>), grid, block, 0, s, p, ext);
    int st = cb_launch_status();
    if (st != CB_OK) return st;
    if constexpr (AR == 1) {
        if (second) {      // (fp16: the group-aware second launch, which also runs the consumers' detection)
            hipLaunchKernelGGL((cbh_reduce_kernel<BM, BN>), dim3(2 * cbs_num_cus()), dim3(256), 0, s, p, ext);
            st = cb_launch_status();
        }
        return st;
    }
    if (tail) {
        const int waves = (tail->C1 + 15) / 16;
        hipLaunchKernelGGL((cbs_reduce_tail_kernel<BM, BN>), dim3(2 * cbs_num_cus()), dim3(64 * waves),
                           cb_tail_lds_bytes(p.K, tail->C1, tail->C2), s, p, *tail);
        st = cb_launch_status();
    } else if (second) {
        hipLaunchKernelGGL(cbs_reduce_kernel, dim3(4 * cbs_num_cus()), dim3(256), 0, s, p, BM, BN);
        st = cb_launch_status();
    }
    return st;
}

}  // namespace cbs
using namespace cbs;

#ifdef CBS_STAMP
extern "C" int cbinfer_debug_split_stamps(void* host, long bytes, int clear) {
    if (clear) {
        void* d = nullptr;
        if (hipGetSymbolAddress(&d, HIP_SYMBOL(cbs_stamp_buf)) != hipSuccess) return -1;
        return (int)hipMemset(d, 0, sizeof(unsigned long long) * 4 * 2048 * 16);
    }
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(cbs_stamp_buf), (size_t)bytes);
}
#endif

extern "C" {

int cbinfer_split_supported(int C, int K, int kH, int kW) { return cbs_supported(C, K, kH, kW) ? 1 : 0; }
int cbinfer_split_max_sequences(void) { return CBS_MAXSEQ; }
// mask words (cbinfer_mask_words(H,W), summed over the sequences of a launch) the contraction of a K-channel layer takes
long cbinfer_split_max_mask_words(int K) { return cbs_bm(K) >= 128 ? CBS_PRE_BIG : CBS_PRE_MID; }

long cbinfer_split_state_bytes(int C, int H, int W, int kH, int kW) {
    const CbsGeom g = cbs_geom(C, H, W, kH, kW);
    return CBS_SPAD + (long)g.Hp * g.Wp * g.rec;
}
long cbinfer_split3_state_bytes(int C, int H, int W, int kH, int kW) {
    const CbsGeom g = cbs_geom(C, H, W, kH, kW, 3);
    return CBS_SPAD + (long)g.Hp * g.Wp * g.rec;
}

// prepared weights: [A fragments | stage table | pad to 16 B | plain f32 filter bank]
static long cbs_plain_offset(long aBytes, int nStages) { return (aBytes + (long)nStages * 4 + 15) / 16 * 16; }
long cbinfer_split_prepared_bytes(int C, int K, int kH, int kW) {
    const CbsGeom g = cbs_geom(C, 64, 64, kH, kW);
    return cbs_plain_offset((long)g.nStages * (cbs_kp(K) / 32) * 4096, g.nStages) + (long)K * C * kH * kW * 4;
}
// bf16-triple form: [A fragments, 6 KB per stage and 32-row tile | stage table | pad to 16 B] (no plain filter bank: no exact path)
long cbinfer_split3_prepared_bytes(int C, int K, int kH, int kW) {
    const CbsGeom g = cbs_geom(C, 64, 64, kH, kW, 3);
    return cbs_plain_offset((long)g.nStages * (cbs_kp(K) / 32) * 6144, g.nStages);
}

// Workspace of a deep contraction (>= 48 stages; 0 bytes otherwise): 64 ints of launch info + one BM x BN partial
// tile per work item of a split contraction (at most a grid of them; sized for one per tile of every sequence, which
// an earlier form of the unsplit case used for its running sums)
static long cbs_slab_capacity(int nSeq, int H, int W, int K) {
    const int bm = cbs_bm(K), bn = bm >= 128 ? 128 : 64;
    const long tiles = (long)nSeq * (((long)H * W + bn - 1) / bn) * (cbs_kp(K) / bm);
    return tiles > 2l * cbs_num_cus() ? tiles : 2l * cbs_num_cus();
}
long cbinfer_split_workspace_bytes(int nSeq, int C, int H, int W, int K, int kH, int kW) {
    if (!cbs_supported(C, K, kH, kW)) return 0;
    const CbsGeom g = cbs_geom(C, H, W, kH, kW);
    if (g.nStages < 48) return 0;
    const int bm = cbs_bm(K), bn = bm >= 128 ? 128 : 64;
    // (+ one arrival counter per partial tile behind the slabs: the last-arriver experiment, CBINFER_SPLIT_LAST=1)
    return 256 + cbs_slab_capacity(nSeq, H, W, K) * bm * bn * 4 + cbs_slab_capacity(nSeq, H, W, K) * 4;
}

int cbinfer_split_prep_weights(const float* weight, void* prepared, int K, int C, int kH, int kW, int H, int W,
                               float weightScale, cbStream_t stream) {
    CB_REQUIRE(weight && prepared && H > 0 && W > 0 && weightScale > 0.f);
    if (!cbs_supported(C, K, kH, kW)) return CB_ERR_UNSUPPORTED;
    const CbsGeom g = cbs_geom(C, H, W, kH, kW);
    if ((long)g.Hp * g.Wp * g.rec >= (1l << 31)) return CB_ERR_UNSUPPORTED;
    const int KP = cbs_kp(K);
    const long total = (long)g.nStages * (KP / 32) * 4 * 64;
    const long aBytes = (long)g.nStages * (KP / 32) * 4096;
    int* stageOff = (int*)((char*)prepared + aBytes);
    float* wPlain = (float*)((char*)prepared + cbs_plain_offset(aBytes, g.nStages));
    hipLaunchKernelGGL(cbs_prep_kernel, dim3(cb_div_up(total, 256)), dim3(256), 0, (hipStream_t)stream, weight,
                       (halfx8*)prepared, stageOff, wPlain, g, K, KP, weightScale);
    return cb_launch_status();
}
int cbinfer_split3_prep_weights(const float* weight, void* prepared, int K, int C, int kH, int kW, int H, int W,
                                cbStream_t stream) {
    CB_REQUIRE(weight && prepared && H > 0 && W > 0);
    if (!cbs_supported(C, K, kH, kW)) return CB_ERR_UNSUPPORTED;
    const CbsGeom g = cbs_geom(C, H, W, kH, kW, 3);
    if ((long)g.Hp * g.Wp * g.rec >= (1l << 31)) return CB_ERR_UNSUPPORTED;
    const int KP = cbs_kp(K);
    const long total = (long)g.nStages * (KP / 32) * 6 * 64;
    const long aBytes = (long)g.nStages * (KP / 32) * 6144;
    int* stageOff = (int*)((char*)prepared + aBytes);
    hipLaunchKernelGGL(cbs_prep_kernel, dim3(cb_div_up(total > g.nStages ? total : g.nStages, 256)), dim3(256), 0,
                       (hipStream_t)stream, weight, (halfx8*)prepared, stageOff, (float*)nullptr, g, K, KP, 1.0f);
    return cb_launch_status();
}

static int cbs_state_init(void* splitState, int C, int H, int W, int kH, int kW, int planes, cbStream_t stream) {
    CB_REQUIRE(splitState && H > 0 && W > 0);
    if (!cbs_supported(C, 1, kH, kW)) return CB_ERR_UNSUPPORTED;
    const CbsGeom g = cbs_geom(C, H, W, kH, kW, planes);
    hipLaunchKernelGGL(cbs_state_init_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, (uint4*)splitState, g);
    return cb_launch_status();
}
int cbinfer_split_state_init(void* splitState, int C, int H, int W, int kH, int kW, cbStream_t stream) {
    return cbs_state_init(splitState, C, H, W, kH, kW, 2, stream);
}
int cbinfer_split3_state_init(void* splitState, int C, int H, int W, int kH, int kW, cbStream_t stream) {
    return cbs_state_init(splitState, C, H, W, kH, kW, 3, stream);
}

// splitState <- split(state): for a state the caller wrote itself (restored states); the zero border must exist
// already (cbinfer_split_state_init)
int cbinfer_split_state_rebuild(const float* state, void* splitState, int C, int H, int W, int kH, int kW,
                                int32_t* rangeFlag, cbStream_t stream) {
    CB_REQUIRE(state && splitState && H > 0 && W > 0);
    if (!cbs_supported(C, 1, kH, kW)) return CB_ERR_UNSUPPORTED;
    const CbsGeom g = cbs_geom(C, H, W, kH, kW);
    hipLaunchKernelGGL(cbs_state_rebuild_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, state,
                       (char*)splitState, g, rangeFlag);
    return cb_launch_status();
}
int cbinfer_split3_state_rebuild(const float* state, void* splitState, int C, int H, int W, int kH, int kW,
                                 cbStream_t stream) {
    CB_REQUIRE(state && splitState && H > 0 && W > 0);
    if (!cbs_supported(C, 1, kH, kW)) return CB_ERR_UNSUPPORTED;
    const CbsGeom g = cbs_geom(C, H, W, kH, kW, 3);
    hipLaunchKernelGGL(cbs_state_rebuild_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, state,
                       (char*)splitState, g, (int*)nullptr);
    return cb_launch_status();
}

// Detection of up to CBS_MAXSEQ sequences in one launch (see cbSplitSeq in the header).  pooled != 0: `input` is
// the tensor in front of a 2x2/stride-2 max pool, [C,pH,pW].
int cbinfer_split_detect(const cbSplitSeq* seqs, int nSeq, int mode, int pH, int pW, int C, int H, int W,
                         int kH, int kW, float threshold, cbStream_t stream) {
    CB_REQUIRE(seqs && nSeq >= 1 && nSeq <= CBS_MAXSEQ && H > 0 && W > 0 && mode >= 0 && mode <= 15);
    const int x3 = (mode >> 3) & 1;   // bit 3 (CBINFER_SPLIT_X3): the records are bf16 triples (cbinfer_split3_state_bytes)
    const int pooled = mode & 1;      // bit 1 (CBINFER_SPLIT_COPY_ALL): not in feedback mode, every value goes to the states
    const int fg = (mode >> 2) & 1;   // bit 2 (CBINFER_SPLIT_FG): fine-grained frame, the pre-split copy takes the differences
    if (!cbs_supported(C, 1, kH, kW) || H > 65535) return CB_ERR_UNSUPPORTED;
    if (pooled) CB_REQUIRE((H == pH / 2 || H == (pH + 1) / 2) && (W == pW / 2 || W == (pW + 1) / 2));
    const CbsGeom g = cbs_geom(C, H, W, kH, kW, x3 ? 3 : 2);
    CbsDetArgs a;
    a.planes = g.planes;
    for (int q = 0; q < nSeq; ++q) {
        CB_REQUIRE(seqs[q].input && seqs[q].state && seqs[q].splitState && seqs[q].frameMasks);
        a.seq[q].in = seqs[q].input;
        a.seq[q].state = seqs[q].state;
        a.seq[q].S = (char*)seqs[q].splitState;
        a.seq[q].masks = (unsigned long long*)seqs[q].frameMasks;
        // (fine-grained: every record is rewritten every frame -- a segment the producer did not touch holds last frame's
        //  differences and must be zeroed: no producer-mask shortcut)
        a.seq[q].prodMask = (pooled && !fg) ? (const unsigned long long*)seqs[q].producerMask : nullptr;
        a.seq[q].rangeFlag = x3 ? nullptr : seqs[q].rangeFlag;      // (bf16 triples have f32's range)
        a.seq[q].delta = seqs[q].delta;
        if (fg) CB_REQUIRE(seqs[q].delta != nullptr);
    }
    a.W = W, a.H = H, a.C = C, a.kHH = (kH - 1) / 2, a.kWH = (kW - 1) / 2;
    a.wpr = cbinfer_mask_words_per_row(W), a.pH = pH, a.pW = pW;
    a.Wp = g.Wp, a.rec = g.rec, a.padY = g.padY, a.padXL = g.padXL;
    a.words = cbinfer_mask_words(H, W);
    a.th = threshold;
    a.copyAll = ((mode >> 1) & 1) | fg;
    a.fg = fg;
    dim3 grid(a.wpr, H, nSeq), block(64 * (C / 4));
    if (pooled)
        hipLaunchKernelGGL(cbs_detect_kernel<true>, grid, block, 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(cbs_detect_kernel<false>, grid, block, 0, (hipStream_t)stream, a);
    return cb_launch_status();
}

// The contraction of up to CBS_MAXSEQ sequences in one launch (+ the reduce launch of a split contraction).
// outScale = 1 / (weightScale * 2^-4).  forceSplit > 0 overrides the k-split decision (tests, tuning).
// weightScale == 0 selects the bf16-TRIPLE form (x3: prepared weights of cbinfer_split3_prep_weights, split states of
// cbinfer_split3_state_bytes; no scale anywhere, outScale = 1, no range flag).
// Layers whose contraction launch can run in WINDOW order and carry the pooled detection of the layer behind the 2x2 pool
// (AR = 3): all output channels in one 64-row tile (K <= 64, a multiple of 16), a shallow contraction (one item per tile:
// no slabs), mask words and the three unit prefixes beside a five-stage ring; the consumer a split-state layer on K channels.
static bool cbs_next_supported(int C, int K, int kH, int kW, int H, int W, const cbNextDetect* next) {
    if (!next || !cbs_supported(C, K, kH, kW) || K > 64 || K % 16 != 0 || !cbs_supported(K, 1, next->kH, next->kW)) return false;
    if (cbs_geom(C, H, W, kH, kW, 3).nStages >= 48) return false;
    if (!((next->H == H / 2 || next->H == (H + 1) / 2) && (next->W == W / 2 || next->W == (W + 1) / 2))) return false;
    if (next->arith != 0 && next->arith != 1) return false;
    const long MW = cbinfer_mask_words(H, W), EU = (long)((H + 1) / 2) * cbinfer_mask_words_per_row(W);
    if (MW > CBS_PRE_SMALL || 2 * EU + 2 > CBS_PRE_SMALL + 1 || EU + 1 > CBS_PRE_SMALL / 2 + 2) return false;
    const CbsGeom g2 = cbs_geom(K, next->H, next->W, next->kH, next->kW, next->arith ? 3 : 2);
    return (long)g2.Hp * g2.Wp * g2.rec < (1l << 31) && (long)H * W < (1l << 20);
}

// Layers whose pixel-order contraction can carry a side refresh (AR = 4): the single-sequence 64-row instance with the mask
// words in LDS, a shallow contraction (no second launch)
static bool cbs_refresh_supported(int C, int K, int kH, int kW, int H, int W) {
    return cbs_supported(C, K, kH, kW) && K <= 64 && cbs_geom(C, H, W, kH, kW, 3).nStages < 48 &&
           cbinfer_mask_words(H, W) <= CBS_PRE_SMALL;
}

static int cbs_split_conv(const cbSplitSeq* seqs, int nSeq, const void* prepared, const float* bias, int C, int H,
                          int W, int K, int kH, int kW, float weightScale, int relu, void* workspace, int forceSplit,
                          const CbsTailArgs* tail, cbStream_t stream, int accumulate = 0,
                          const cbNextDetect* next = nullptr, const cbSideRefresh* side = nullptr) {
    if (workspace == nullptr && cbs_supported(C, K, kH, kW) && cbs_geom(C, H, W, kH, kW).nStages >= 48)
        return CB_ERR_BADARG;      // a deep contraction needs its workspace (cbinfer_split_workspace_bytes)
    CB_REQUIRE(seqs && nSeq >= 1 && nSeq <= CBS_MAXSEQ && prepared && H > 0 && W > 0 && weightScale >= 0.f);
    if (!cbs_supported(C, K, kH, kW)) return CB_ERR_UNSUPPORTED;
    const bool x3 = weightScale == 0.f;
    const CbsGeom g = cbs_geom(C, H, W, kH, kW, x3 ? 3 : 2);
    const int KP = cbs_kp(K), BM = cbs_bm(K);
    const long MW = cbinfer_mask_words(H, W);
    if ((long)nSeq * MW > (BM >= 128 ? CBS_PRE_BIG : CBS_PRE_MID) || (long)g.Hp * g.Wp * g.rec >= (1l << 31) ||
        (long)H * W * W >= (1l << 32))
        return CB_ERR_UNSUPPORTED;
    CbsParams p;
    for (int q = 0; q < nSeq; ++q) {
        CB_REQUIRE(seqs[q].splitState && seqs[q].output && seqs[q].frameMasks && seqs[q].idxOut && seqs[q].countOut);
        p.seq[q].S = (const char*)seqs[q].splitState;
        // (what the exact path contracts in plain f32: the layer state -- or, fine-grained, the delta tensor)
        p.seq[q].state = accumulate ? seqs[q].delta : seqs[q].state;
        p.seq[q].reluOut = accumulate ? seqs[q].reluOut : nullptr;
        p.seq[q].rangeFlag = (p.seq[q].state && !x3) ? seqs[q].rangeFlag : nullptr;      // (no f32 operand handed in: no exact path)
        p.seq[q].out = seqs[q].output;
        p.seq[q].masks = (unsigned long long*)seqs[q].frameMasks;
        p.seq[q].listOut = seqs[q].idxOut;
        p.seq[q].countOut = seqs[q].countOut;
        p.seq[q].maskCopy = (unsigned long long*)seqs[q].maskCopy;
    }
    p.nSeq = nSeq;
    p.aBytes = (long)g.nStages * (KP / 32) * (x3 ? 6144 : 4096);
    p.A = (const char*)prepared;
    p.stageOff = (const int*)((const char*)prepared + p.aBytes);
    p.wPlain = x3 ? nullptr : (const float*)((const char*)prepared + cbs_plain_offset(p.aBytes, g.nStages));
    p.bias = bias;
    p.info = workspace ? (int*)workspace : nullptr;
    p.slabs = workspace ? (float*)((char*)workspace + 256) : nullptr;
    p.K = K, p.KP = KP, p.H = H, p.W = W, p.Wp = g.Wp, p.rec = g.rec, p.nStages = g.nStages, p.kH = kH, p.kW = kW;
    p.maskWords = (int)MW, p.wpr = cbinfer_mask_words_per_row(W), p.relu = relu, p.dummyBase = g.dummyBase;
    p.stateBytes = (long)g.Hp * g.Wp * g.rec;
    p.outScale = x3 ? 1.0f : 1.0f / (weightScale * CBS_XSCALE);
    p.magicMW = (1ull << 32) / (unsigned long long)MW + 1ull;
    p.magicWpr = (1ull << 32) / (unsigned long long)p.wpr + 1ull;
    p.magicW = (1ull << 32) / (unsigned long long)W + 1ull;
    p.magicMT = (1ull << 32) / (unsigned long long)(KP / BM) + 1ull;
    p.forceSK = forceSplit;
    p.accumulate = accumulate;
    p.halfOut = 0;
    p.upstream = nullptr;
    {
        static int rounds = -1;      // CBINFER_SPLIT_ROUNDS (tuning aid; default 2)
        if (rounds < 0) {
            const char* e = getenv("CBINFER_SPLIT_ROUNDS");
            rounds = e && atoi(e) > 0 ? atoi(e) : 2;
        }
        p.splitRounds = rounds;
    }
    p.arriveShards = (int)(MW / 16 < 8 ? MW / 16 : 8);
    p.dbg = 0;
    p.maxChunks = CBS_CHUNKS;
#ifdef CBS_LAST_ARRIVER
    {
        static int last = -1;      // CBINFER_SPLIT_LAST=1: the last-arriver experiment (x3, 128-row tile, no accumulate form)
        if (last < 0) {
            const char* e = getenv("CBINFER_SPLIT_LAST");
            last = e ? atoi(e) : 0;
        }
        p.lastArrive = (last && x3 && BM == 128 && !accumulate && workspace) ? 1 : 0;
        p.arriveTiles = p.lastArrive ? (int*)((char*)workspace + 256 + cbs_slab_capacity(nSeq, H, W, K) * (long)BM * 128 * 4)
                                     : nullptr;
    }
#endif
#ifdef CBS_DBG
    if (const char* e = getenv("CBINFER_SPLIT_DBG")) p.dbg = atoi(e);
#endif
    hipStream_t s = (hipStream_t)stream;
    // the workspace was sized by cbinfer_split_workspace_bytes for this very geometry: a split the kernel decides on
    // itself (items <= grid <= 2 CUs) always fits, a forced one (forceSplit: tests, tuning) is refused beyond it
    const long cap = cbs_slab_capacity(nSeq, H, W, K);
    p.slabCap = (int)(cap > 0x7fffffffl ? 0x7fffffffl : cap);
    if (side && side->frame && !next) {
        // pixel order + another layer's state refresh on the idle workgroups (AR = 4)
        if (!cbs_refresh_supported(C, K, kH, kW, H, W) || nSeq != 1 || !x3 || accumulate || tail) return CB_ERR_UNSUPPORTED;
        CB_REQUIRE(side->state && side->C >= 1 && side->H >= 1 && side->W >= 1 && (long)side->H * side->W < (1l << 30));
        CbsSideExt e;
        e.sideFrame = side->frame, e.sideState = side->state, e.sideC = side->C, e.sideHW = side->H * side->W;
        e.sideTh = side->threshold;
        return cbs_launch_conv<64, 64, 2, 2, CBS_PRE_SMALL, true, 5, 4>(p, 1, nullptr, s, e);
    }
    if (next) {
        // window order + the pooled detection of the layer behind the 2x2 pool in this launch (AR = 3)
        if (!cbs_next_supported(C, K, kH, kW, H, W, next) || nSeq != 1 || !x3 || accumulate || tail) return CB_ERR_UNSUPPORTED;
        CB_REQUIRE(next->state && next->splitState && next->frameMasks);
        const CbsGeom g2 = cbs_geom(K, next->H, next->W, next->kH, next->kW, next->arith ? 3 : 2);
        CbsWinExt e;
        e.nstate = next->state, e.nS = (char*)next->splitState, e.nmasks = (unsigned long long*)next->frameMasks;
        e.nflag = next->arith ? nullptr : next->rangeFlag;      // (bf16 triples have f32's range)
        e.H2 = next->H, e.W2 = next->W, e.wpr2 = cbinfer_mask_words_per_row(next->W);
        e.kHH = (next->kH - 1) / 2, e.kWH = (next->kW - 1) / 2;
        e.Wp = g2.Wp, e.rec = g2.rec, e.padY = g2.padY, e.padXL = g2.padXL, e.planes = g2.planes;
        e.th = next->threshold;
        e.magicW2 = (1ull << 32) / (unsigned long long)next->W + 1ull;
        e.sideFrame = nullptr, e.sideState = nullptr, e.sideC = 0, e.sideHW = 0, e.sideTh = 0.f;
        if (side && side->frame) {
            CB_REQUIRE(side->state && side->C >= 1 && side->H >= 1 && side->W >= 1 && (long)side->H * side->W < (1l << 30));
            e.sideFrame = side->frame, e.sideState = side->state, e.sideC = side->C, e.sideHW = side->H * side->W;
            e.sideTh = side->threshold;
        }
        return cbs_launch_conv<64, 64, 2, 2, CBS_PRE_SMALL, true, 5, 3>(p, 1, nullptr, s, e);
    }
    if (x3) {
        // bf16 triples: 48 KB stages (128-row tile, three in the ring, the mask words in LDS while they and their prefix
        // fit beside it: six 80x120 sequences) / 24 KB stages (64-row tile: five alone on a CU, three when two
        // workgroups share one)
        if (BM == 128) {
            if ((long)nSeq * MW <= CBS_PRE_X3) return cbs_launch_conv<128, 128, 4, 2, CBS_PRE_X3, true, 3, 2>(p, 1, tail, s);
            return cbs_launch_conv<128, 128, 4, 2, CBS_PRE_BIG, false, 3, 2>(p, 1, tail, s);
        }
        if (nSeq == 1 && MW <= CBS_PRE_SMALL) return cbs_launch_conv<64, 64, 2, 2, CBS_PRE_SMALL, true, 5, 2>(p, 1, tail, s);
        if ((long)nSeq * MW <= CBS_PRE_SMALL) return cbs_launch_conv<64, 64, 2, 2, CBS_PRE_SMALL, false, 3, 2>(p, 2, tail, s);
        return cbs_launch_conv<64, 64, 2, 2, CBS_PRE_MID, false, 5, 2>(p, 1, tail, s);
    }
    if (BM == 128) return cbs_launch_conv<128, 128, 4, 2, CBS_PRE_BIG, true, 4>(p, 1, tail, s);
    // The 64-row tile moves 16 KB per stage, and what bounds its stage rate is the DMA in flight on the CU (bytes in
    // flight / latency): one sequence rarely has more tiles than there are CUs, so it runs one workgroup per CU with
    // a ring of eight stages (seven in flight, 112 KB); several sequences run two workgroups per CU with four-stage
    // rings (96 KB in flight between them) while the mask words and their prefix fit beside two rings.
    if (nSeq == 1 && MW <= CBS_PRE_SMALL) return cbs_launch_conv<64, 64, 2, 2, CBS_PRE_SMALL, true, 8>(p, 1, tail, s);
    if ((long)nSeq * MW <= CBS_PRE_SMALL) return cbs_launch_conv<64, 64, 2, 2, CBS_PRE_SMALL, true, 4>(p, 2, tail, s);
    if ((long)nSeq * MW <= CBS_PRE_MID2) return cbs_launch_conv<64, 64, 2, 2, CBS_PRE_MID2, false, 4>(p, 2, tail, s);
    {
        // (five and more sequences: two workgroups per CU with three-stage rings beside the 20 KB prefix -- the
        //  workgroups' prologues and epilogues overlap each other's stage loops; CBINFER_SPLIT_BIG2=0: one per CU, eight stages)
        static int big2 = -1;
        if (big2 < 0) {
            const char* e = getenv("CBINFER_SPLIT_BIG2");
            big2 = e ? atoi(e) : 1;
        }
        if (big2) return cbs_launch_conv<64, 64, 2, 2, CBS_PRE_MID, false, 3>(p, 2, tail, s);
    }
    return cbs_launch_conv<64, 64, 2, 2, CBS_PRE_MID, false, 8>(p, 1, tail, s);
}

int cbinfer_split_conv(const cbSplitSeq* seqs, int nSeq, const void* prepared, const float* bias, int C, int H,
                       int W, int K, int kH, int kW, float weightScale, int relu, void* workspace, int forceSplit,
                       cbStream_t stream) {
    return cbs_split_conv(seqs, nSeq, prepared, bias, C, H, W, K, kH, kW, weightScale, relu, workspace, forceSplit,
                          nullptr, stream);
}

// 1 if cbinfer_split_forward_tail takes this layer + tail: a deep contraction (its second launch exists anyway) and
// a tail the one-launch tail kernel would take
int cbinfer_split_tail_supported(int C, int K, int kH, int kW, int C1, int C2) {
    return cbs_supported(C, K, kH, kW) && cbs_geom(C, 64, 64, kH, kW).nStages >= 48 && K % 16 == 0 && C1 >= 4 && C2 >= 1 &&
           C1 <= 16 * CB_TAIL_MAXW && cb_tail_lds_bytes(K, C1, C2) <= 60 * 1024;
}

// cbinfer_split_conv with the fused 1x1 tail in the second launch (the two launches of cbinfer_split_forward_tail behind
// its detection)
int cbinfer_split_conv_tail(const cbSplitSeq* seqs, int nSeq, const void* prepared, const float* bias, int C, int H,
                            int W, int K, int kH, int kW, float weightScale, int relu, void* workspace, int forceSplit,
                            const cbSplitTail* tail, cbStream_t stream) {
    CB_REQUIRE(tail && tail->w1Prepared && tail->b1 && tail->w2 && tail->b2 && nSeq >= 1 && nSeq <= CBS_MAXSEQ);
    // (the second launch reads the layer's bias -- and, without one, the tail's first -- four values at a time)
    CB_REQUIRE((((size_t)bias | (size_t)tail->b1) & 15) == 0);
    if (!cbinfer_split_tail_supported(C, K, kH, kW, tail->C1, tail->C2)) return CB_ERR_UNSUPPORTED;
    CbsTailArgs ta;
    ta.w1p = tail->w1Prepared, ta.b1 = tail->b1, ta.w2 = tail->w2, ta.b2 = tail->b2;
    ta.C1 = tail->C1, ta.C2 = tail->C2, ta.relu1 = tail->relu1, ta.relu2 = tail->relu2;
    for (int q = 0; q < CBS_MAXSEQ; ++q) {
        if (q < nSeq) CB_REQUIRE(tail->output[q]);
        ta.out[q] = q < nSeq ? tail->output[q] : nullptr;
    }
    return cbs_split_conv(seqs, nSeq, prepared, bias, C, H, W, K, kH, kW, weightScale, relu, workspace, forceSplit, &ta,
                          stream);
}

// cbinfer_split_forward + the fused 1x1 tail behind the layer (cbinfer_tail1x1's arithmetic and weight layout) in
// the contraction's second launch: conv1x1 (K -> C1) -> [relu1] -> conv1x1 (C1 -> C2) -> [relu2] at the changed pixels,
// into tail->output[sequence] [C2,H,W].  forceSplit as for cbinfer_split_conv.
int cbinfer_split_forward_tail(const cbSplitSeq* seqs, int nSeq, int mode, int pH, int pW, const void* prepared,
                               const float* bias, int C, int H, int W, int K, int kH, int kW, float threshold,
                               float weightScale, int relu, void* workspace, int forceSplit, const cbSplitTail* tail,
                               cbStream_t stream) {
    CB_REQUIRE(tail);
    if (!cbinfer_split_tail_supported(C, K, kH, kW, tail->C1, tail->C2)) return CB_ERR_UNSUPPORTED;
    mode = (mode & ~CBINFER_SPLIT_X3) | (weightScale == 0.f ? CBINFER_SPLIT_X3 : 0);
    const int st = cbinfer_split_detect(seqs, nSeq, mode, pH, pW, C, H, W, kH, kW, threshold, stream);
    if (st != CB_OK) return st;
    return cbinfer_split_conv_tail(seqs, nSeq, prepared, bias, C, H, W, K, kH, kW, weightScale, relu, workspace,
                                   forceSplit, tail, stream);
}

// One FINE-GRAINED frame (CBConv2d.forward_fg, conv2d.py:160-176 -> conv2d_fg.py:75-85; cbinfer_cbconv2d_forward_fg's
// contract) on the split-state kernels: the detection writes the frame into prevInput, d = in - prev where |d| > th
// (0 elsewhere) into `delta` and, as f16 pairs, into the pre-split records of EVERY pixel, and the dilated any-channel
// mask; the contraction ADDS W * delta to `output` at the mask's pixels (reluOut, if given, kept at relu(output)
// there).  No bias, deterministic.  A |d| >= 2^20 trips the range flag: the launch then contracts `delta` in plain f32.
// pooled != 0: `input` is the tensor in FRONT of a 2x2/stride-2 max pool [C,pH,pW] (a CBPoolMax2d folded into the
// detection: configs[2] of BASELINE.json, fine-grained convs + CBPoolMax2d, without the pool's launch).
int cbinfer_split_forward_fg(const cbSplitSeq* seqs, int nSeq, int pooled, int pH, int pW, const void* prepared, int C,
                             int H, int W, int K, int kH, int kW, float threshold, float weightScale, void* workspace,
                             cbStream_t stream) {
    const int st = cbinfer_split_detect(seqs, nSeq, CBINFER_SPLIT_FG | (pooled ? CBINFER_SPLIT_POOLED : 0) |
                                        (weightScale == 0.f ? CBINFER_SPLIT_X3 : 0), pH, pW, C, H, W,
                                        kH, kW, threshold, stream);
    if (st != CB_OK) return st;
    return cbs_split_conv(seqs, nSeq, prepared, nullptr, C, H, W, K, kH, kW, weightScale, 0, workspace, 0, nullptr,
                          stream, 1);
}

// cbinfer_split_forward_fg with the fused 1x1 tail behind the layer evaluated by the contraction's second launch
// (cbinfer_split_forward_tail's arrangement for the fine-grained frame, round 5): that launch adds the slabs' sum to
// `output`, keeps `reluOut` and runs conv1x1 -> [relu1] -> conv1x1 on the finished columns -- on relu(output) when the
// layer has a relu'd copy (reluOut != NULL: what the network hands the next module), else on output.
int cbinfer_split_forward_fg_tail(const cbSplitSeq* seqs, int nSeq, int pooled, int pH, int pW, const void* prepared,
                                  int C, int H, int W, int K, int kH, int kW, float threshold, float weightScale,
                                  void* workspace, const cbSplitTail* tail, cbStream_t stream) {
    CB_REQUIRE(tail && tail->w1Prepared && tail->b1 && tail->w2 && tail->b2 && nSeq >= 1 && nSeq <= CBS_MAXSEQ);
    CB_REQUIRE(((size_t)tail->b1 & 15) == 0);
    if (!cbinfer_split_tail_supported(C, K, kH, kW, tail->C1, tail->C2)) return CB_ERR_UNSUPPORTED;
    CbsTailArgs ta;
    ta.w1p = tail->w1Prepared, ta.b1 = tail->b1, ta.w2 = tail->w2, ta.b2 = tail->b2;
    ta.C1 = tail->C1, ta.C2 = tail->C2, ta.relu1 = tail->relu1, ta.relu2 = tail->relu2;
    for (int q = 0; q < CBS_MAXSEQ; ++q) {
        if (q < nSeq) CB_REQUIRE(tail->output[q]);
        ta.out[q] = q < nSeq ? tail->output[q] : nullptr;
    }
    const int st = cbinfer_split_detect(seqs, nSeq, CBINFER_SPLIT_FG | (pooled ? CBINFER_SPLIT_POOLED : 0) |
                                        (weightScale == 0.f ? CBINFER_SPLIT_X3 : 0), pH, pW, C, H, W,
                                        kH, kW, threshold, stream);
    if (st != CB_OK) return st;
    return cbs_split_conv(seqs, nSeq, prepared, nullptr, C, H, W, K, kH, kW, weightScale, 0, workspace, 0, &ta,
                          stream, 1);
}

// One frame of a feedback-mode CBConv2d (conv2d.py:178-259) of every sequence: detection (+ pooling) + refresh of
// both states, then the contraction.
// The contraction in WINDOW order with the pooled change detection of the NEXT layer (behind a 2x2 / stride-2 max pool)
// in its epilogue -- cbp_rowpair_kernel's arrangement for a split-state producer (round 6): one sequence, the bf16-triple
// arithmetic (weightScale == 0), K <= 64 output channels, a shallow contraction.  `next`: the consumer's tensors, geometry
// and threshold (cbNextDetect).  cbinfer_split_next_supported tells whether a layer pair is taken.
int cbinfer_split_next_supported(int C, int K, int kH, int kW, int H, int W, const cbNextDetect* next) {
    return cbs_next_supported(C, K, kH, kW, H, W, next) ? 1 : 0;
}
int cbinfer_split_conv_next(const cbSplitSeq* seqs, int nSeq, const void* prepared, const float* bias, int C, int H,
                            int W, int K, int kH, int kW, float weightScale, int relu, void* workspace,
                            const cbNextDetect* next, cbStream_t stream) {
    CB_REQUIRE(next);
    return cbs_split_conv(seqs, nSeq, prepared, bias, C, H, W, K, kH, kW, weightScale, relu, workspace, 0, nullptr, stream,
                          0, next);
}
// cbinfer_split_conv_next + the feedback refresh of ANOTHER layer's state on the launch's idle workgroups (`side`: what
// cbinfer_refresh_state does in a launch of its own -- for the row-pair layer in front that ran cbinfer_conv_rowpairs_detect)
int cbinfer_split_conv_next_refresh(const cbSplitSeq* seqs, int nSeq, const void* prepared, const float* bias, int C, int H,
                                    int W, int K, int kH, int kW, float weightScale, int relu, void* workspace,
                                    const cbNextDetect* next, const cbSideRefresh* side, cbStream_t stream) {
    CB_REQUIRE(next && side);
    return cbs_split_conv(seqs, nSeq, prepared, bias, C, H, W, K, kH, kW, weightScale, relu, workspace, 0, nullptr, stream,
                          0, next, side);
}
// cbinfer_split_conv (pixel order) + the side refresh: for the consumer of a self-detecting row-pair layer that does not run
// in window order.  cbinfer_split_refresh_supported tells whether the layer's launch can carry it.
int cbinfer_split_refresh_supported(int C, int K, int kH, int kW, int H, int W) {
    return cbs_refresh_supported(C, K, kH, kW, H, W) ? 1 : 0;
}
int cbinfer_split_conv_refresh(const cbSplitSeq* seqs, int nSeq, const void* prepared, const float* bias, int C, int H, int W,
                               int K, int kH, int kW, float weightScale, int relu, void* workspace,
                               const cbSideRefresh* side, cbStream_t stream) {
    CB_REQUIRE(side && side->frame);
    return cbs_split_conv(seqs, nSeq, prepared, bias, C, H, W, K, kH, kW, weightScale, relu, workspace, 0, nullptr, stream,
                          0, nullptr, side);
}
int cbinfer_split_forward_next(const cbSplitSeq* seqs, int nSeq, int mode, int pH, int pW, const void* prepared,
                               const float* bias, int C, int H, int W, int K, int kH, int kW, float threshold,
                               float weightScale, int relu, void* workspace, const cbNextDetect* next,
                               cbStream_t stream) {
    CB_REQUIRE(next);
    if (!cbs_next_supported(C, K, kH, kW, H, W, next) || nSeq != 1 || weightScale != 0.f) return CB_ERR_UNSUPPORTED;
    mode = (mode & ~CBINFER_SPLIT_X3) | CBINFER_SPLIT_X3;
    const int st = cbinfer_split_detect(seqs, nSeq, mode, pH, pW, C, H, W, kH, kW, threshold, stream);
    if (st != CB_OK) return st;
    return cbinfer_split_conv_next(seqs, nSeq, prepared, bias, C, H, W, K, kH, kW, weightScale, relu, workspace, next,
                                   stream);
}

int cbinfer_split_forward(const cbSplitSeq* seqs, int nSeq, int mode, int pH, int pW, const void* prepared,
                          const float* bias, int C, int H, int W, int K, int kH, int kW, float threshold,
                          float weightScale, int relu, void* workspace, cbStream_t stream) {
    mode = (mode & ~CBINFER_SPLIT_X3) | (weightScale == 0.f ? CBINFER_SPLIT_X3 : 0);
    const int st = cbinfer_split_detect(seqs, nSeq, mode, pH, pW, C, H, W, kH, kW, threshold, stream);
    if (st != CB_OK) return st;
    return cbs_split_conv(seqs, nSeq, prepared, bias, C, H, W, K, kH, kW, weightScale, relu, workspace, 0, nullptr,
                          stream);
}

}  // extern "C"

// ===================================================================================================
// fp16 layers on the split-state machinery ("hsplit", round 4): the fused a1 + a5..a8 path of a CBConv2d in half
// precision (cbconv2d_cg_half_backend.cu:10-88, :146-197; conv2d.py:178-259) for layers whose input channels are a
// multiple of 64.  The state is kept a second time pixel-major, [Hp][Wp][C] f16 with the zero border and dummy rows
// of the f16-pair form -- no split: the values are f16 already -- so a stage is 64 channels of one tap = 128 bytes of
// a pixel's record, and cbs_conv_kernel<..., HALF> runs DMA, ring and fragment reads unchanged with one
// v_mfma_f32_32x32x16_f16 per 16 k (f32 accumulation, outputs rounded to f16 once: the arithmetic of rounds 1-2's
// fp16 list kernel, another summation order).
// ===================================================================================================
namespace cbs {

// Input channels that are not a multiple of 64 (OpenPose's 185 = 38 + 19 + 128 concatenated maps, PoseModel.py:122-137)
// are PADDED to the next one in the pixel-major copy and in the prepared weights -- zero records, zero weights (round 5);
// the [C,H,W] tensors keep their C.  g.C is the padded count.
__host__ __device__ inline int cbh_cpad(int C) { return (C + 63) / 64 * 64; }
__host__ __device__ inline CbsGeom cbh_geom(int Cin, int H, int W, int kH, int kW) {
    CbsGeom g;
    const int C = cbh_cpad(Cin);
    g.planes = 1;
    g.C = C, g.G = C / 64, g.H = H, g.W = W, g.kH = kH, g.kW = kW;      // (G: stages per tap)
    g.padY = kH / 2, g.padXL = kW / 2, g.padXR = kW / 2;
    g.pair = 0, g.kWs = kW;
    g.Wp = W + g.padXL + g.padXR;
    g.Hp = H + 4 * g.padY + 1;
    g.rec = C * 2;
    g.nStages = kH * kW * g.G;
    g.dummyBase = ((H + 2 * g.padY) * g.Wp) * g.rec;
    return g;
}
inline bool cbh_supported(int C, int K, int kH, int kW) {
    // (round 6: from a single k-stage up -- the 1x1 layers of OpenPose's stages on 128 and 512 channels; the ring is
    //  primed with dead stages past the item's last one, and the stage loop has a one-stage tail)
    return C >= 64 && C <= 1024 && K >= 1 && K <= 1024 && (kH & 1) && (kW & 1) && kH <= 15 && kW <= 15 &&
           cbh_geom(C, 64, 64, kH, kW).nStages >= 1;
}

// weights [K,C,kH,kW] f16 -> [stage][row tile of 32][k-step 0..3][lane][8 f16] (the A-fragment order of the f16-pair
// form with the (k-step, plane) index read as the k-step of four) + the stage table
__global__ __launch_bounds__(256) void cbh_prep_kernel(const _Float16* __restrict__ w, halfx8* __restrict__ A,
                                                      int* __restrict__ stageOff, CbsGeom g, int K, int KP, int Cin) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < g.nStages) {
        const int s = (int)idx, tap = s / g.G, sub = s % g.G;
        stageOff[s] = ((tap / g.kW) * g.Wp + tap % g.kW) * g.rec + sub * 128;
    }
    const int RT = KP / 32;
    const long total = (long)g.nStages * RT * 4 * 64;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    const long blk = idx >> 6;
    const int ks = (int)(blk & 3);
    const int rt = (int)((blk >> 2) % RT), stage = (int)((blk >> 2) / RT);
    const int m = rt * 32 + (lane & 31);
    const int tap = stage / g.G, sub = stage % g.G, ky = tap / g.kW, kx = tap % g.kW;
    halfx8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = sub * 64 + ks * 16 + 8 * (lane >> 5) + j;
        o[j] = (m < K && c < Cin) ? w[(((long)m * Cin + c) * g.kH + ky) * g.kW + kx] : (_Float16)0;
    }
    A[idx] = o;
}

// pixel-major state: zero everywhere, +inf at the image pixels (the f16 state starts as +inf, conv2d.py:192-199)
__global__ __launch_bounds__(256) void cbh_state_init_kernel(uint4* __restrict__ S, CbsGeom g, int Cin) {
    const long chunks = (long)g.Hp * g.Wp * (g.rec / 16);
    if (blockIdx.x == 0) S[threadIdx.x] = make_uint4(0u, 0u, 0u, 0u);     // the CBS_SPAD bytes in front (256 x 16)
    S += CBS_SPAD / 16;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < chunks; i += (long)gridDim.x * blockDim.x) {
        const long pix = i / (g.rec / 16);
        const int c0 = (int)(i % (g.rec / 16)) * 8;      // first channel of this 16-byte piece
        const int py = (int)(pix / g.Wp), px = (int)(pix % g.Wp);
        const bool inside = py >= g.padY && py < g.padY + g.H && px >= g.padXL && px < g.padXL + g.W;
        unsigned v[4];      // (the padding channels behind Cin stay zero: their weights are zero, and inf * 0 is NaN)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            v[j] = !inside ? 0u : ((c0 + 2 * j < Cin ? 0x7c00u : 0u) | (c0 + 2 * j + 1 < Cin ? 0x7c000000u : 0u));
        S[i] = make_uint4(v[0], v[1], v[2], v[3]);
    }
}

// the pixel-major copy made again from the [C,H,W] state (restored states; the border is left alone)
__global__ __launch_bounds__(256) void cbh_state_rebuild_kernel(const _Float16* __restrict__ state,
                                                               char* __restrict__ S, CbsGeom g, int Cin) {
    const long HW = (long)g.H * g.W;
    const int parts = g.C / 8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < HW * parts; i += (long)gridDim.x * blockDim.x) {
        const int part = (int)(i / HW);
        const long pix = i % HW;                       // (consecutive threads: consecutive pixels of one channel)
        const int y = (int)(pix / g.W), x = (int)(pix % g.W);
        halfx8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = part * 8 + j < Cin ? state[(long)(part * 8 + j) * HW + pix] : (_Float16)0;
        *(halfx8*)(S + CBS_SPAD + ((long)(y + g.padY) * g.Wp + (x + g.padXL)) * g.rec + part * 16) = v;
    }
}

// Change detection of an fp16 layer with the second state: one workgroup = one 64-pixel row segment, 8 waves, the
// channels in groups of 64 (wave g: eight of them -- sixteen 128-byte loads in flight per wave: with four, the
// 64-channel 368x654 layer of OpenPose ran at 2.9 TB/s), predicate and dilation as cb_detect_kernel<cb_half> (v_sub_f16,
// strict >, cbconv2d_cg_half_backend.cu:24-35).  copyAll (no feedback loop, copyInput): one pass -- every value
// compared goes straight into prevInput, and group by group through LDS into the pixel records.  Feedback mode: a
// first pass compares, a second one writes the changed pixels' values into both states.
struct CbhDetArgs {
    const _Float16* in;
    _Float16* state;
    char* S;
    unsigned long long* masks;
    int W, H, C, kHH, kWH, wpr, Wp, rec, padY, padXL;
    float th;
    int copyAll;
    const int* upstream;      // optional: the producing layer's change count of this frame (0: nothing to look at)
    int pH, pW;               // POOL: `in` is the tensor in front of a 2x2/stride-2 max pool, [C,pH,pW]
    const unsigned long long* prodMask;      // POOL, optional: the producing layer's change mask of this frame
};
// POOL: the value compared with (and written to) the states is the 2x2 max of `in`, computed on the fly with clamped
// window coordinates (cb_detect_kernel<T, BITS, POOL>); a segment none of whose window pixels the producer rewrote is
// not looked at (its pooled values are last frame's: nothing above the threshold, nothing to copy).
// (round 6: the detections of BOTH layers of a group in one launch -- blockIdx.z = layer x channel group)
struct CbhDetGroup {
    CbhDetArgs a[CBH_GROUP];
    int zPer;      // blockIdx.z values per layer
};
template <bool POOL>
__global__ __launch_bounds__(512) void cbh_detect_kernel(CbhDetGroup grp) {
    const int layer = (int)blockIdx.z / grp.zPer, zl = (int)blockIdx.z - layer * grp.zPer;
    const CbhDetArgs& a = grp.a[layer];
    if (a.upstream && *a.upstream == 0) return;
    cb_touch_kernarg<sizeof(CbhDetGroup)>();
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6, t = threadIdx.x;
    const int tx = blockIdx.x, y = blockIdx.y;
    const int W = a.W, H = a.H, C = a.C;
    if (POOL && a.prodMask) {
        const int pwpr = (a.pW + 63) >> 6;
        unsigned long long any = 0ull;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int yy = 2 * y + j, ww = 2 * tx + i;
                const unsigned long long v = a.prodMask[(long)min(yy, a.pH - 1) * pwpr + min(ww, pwpr - 1)];
                any |= (yy < a.pH && ww < pwpr) ? v : 0ull;
            }
        if (__builtin_amdgcn_readfirstlane((int)(any != 0ull)) == 0) return;
    }
    if (!POOL && a.prodMask) {
        // round 5: the layer is handed the OUTPUT BUFFER of another change-based layer of the same resolution (a chain,
        // CBConv2d._note_upstream) and that layer's change mask of this frame: a 64-pixel segment none of whose pixels
        // the producer rewrote holds bit for bit what this layer compared last frame -- nothing to detect, nothing to
        // copy -- and is not even read (the pooled form's shortcut, for the layers between the pools)
        if (__builtin_amdgcn_readfirstlane((int)(a.prodMask[(long)y * a.wpr + tx] != 0ull)) == 0) return;
    }
    const int x = tx * 64 + lane;
    const bool valid = x < W;
    const long HW = (long)H * W;
    const long p = (long)y * W + x;
    const _Float16* __restrict__ in = a.in;
    _Float16* state = a.state;
    const cb_half th = cb_threshold(a.th, (cb_half*)nullptr);
    const long pHW = (long)a.pH * a.pW;
    const int py0 = 2 * y, px0 = 2 * x;
    const int px1 = POOL ? min(px0 + 1, a.pW - 1) - px0 : 0, py1 = POOL ? (min(py0 + 1, a.pH - 1) - py0) * a.pW : 0;
    auto ldin = [&](int c) -> _Float16 {      // (a channel beyond C -- the padding of the last group -- reads channel C-1:
        c = min(c, C - 1);                    //  never used, see `real` below)
        if (!POOL) return in[(long)c * HW + p];
        const _Float16* q = in + (long)c * pHW + (long)py0 * a.pW + px0;
        return cb_max(cb_max(q[0], q[px1]), cb_max(q[py1], q[py1 + px1]));
    };
    __shared__ unsigned long long sm[8];
    __shared__ _Float16 T[64][66];
    const unsigned long long vm = cbs_valid_mask(W, tx);
    char* const recRow = a.S + CBS_SPAD + ((long)(y + a.padY) * a.Wp + (tx * 64 + a.padXL)) * a.rec;

    // one channel group: [compare] -> [state <- input at the lanes of upd] -> [records of the pixels of upd]
    // (copy-all form: a value that equals the state bit for bit is not written again, and a pixel none of whose 64
    //  values of the group differs keeps its record -- the input of a layer behind another change-based layer is
    //  bit-identical to last frame's wherever that layer recomputed nothing, i.e. almost everywhere)
    __shared__ unsigned long long smDiff[2][8];
    auto group = [&](int cg, bool compare, bool write, unsigned long long upd, bool& chg) {
        const int c0 = cg * 64 + g * 8;
        _Float16 xv[8];
        unsigned diffBits = 0;      // bit i: channel c0 + i of this lane's pixel differs from the state
#pragma unroll
        for (int i = 0; i < 8; ++i) xv[i] = (_Float16)0;
        // which of this wave's eight channels exist (all of them unless the group is the padded tail of C)
        const unsigned real = c0 + 8 <= C ? 0xffu : (c0 >= C ? 0u : (1u << (C - c0)) - 1u);
        if (valid) {
#pragma unroll
            for (int i = 0; i < 8; ++i) xv[i] = ldin(c0 + i);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (!((real >> i) & 1u)) xv[i] = (_Float16)0;      // (padding: zero records, never changed)
            if (compare) {
                _Float16 sv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) sv[i] = state[(long)min(c0 + i, C - 1) * HW + p];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (!((real >> i) & 1u)) continue;
                    chg |= cb_changed(sv[i], xv[i], th);
                    diffBits |= (__builtin_bit_cast(unsigned short, sv[i]) != __builtin_bit_cast(unsigned short, xv[i]))
                                << i;
                }
            } else {
                diffBits = real;
            }
        }
        if (!write) return;
        if ((upd >> lane) & 1ull) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if ((diffBits >> i) & 1u) state[(long)(c0 + i) * HW + p] = xv[i];
        }
        const unsigned long long db = __ballot(diffBits != 0u);
        if (lane == 0) smDiff[cg & 1][g] = db;
        __syncthreads();      // (the previous group's records have been read out of T)
#pragma unroll
        for (int i = 0; i < 8; ++i) T[g * 8 + i][lane] = xv[i];
        __syncthreads();
        {
            unsigned long long rd = 0ull;
#pragma unroll
            for (int i = 0; i < 8; ++i) rd |= smDiff[cg & 1][i];
            rd &= upd;
            const int pl = t >> 3, c8 = t & 7;
            if ((rd >> pl) & 1ull) {
                halfx8 v;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = T[c8 * 8 + j][pl];
                *(halfx8*)(recRow + (long)pl * a.rec + cg * 128 + c8 * 16) = v;
            }
        }
    };

    // (copy-all form: the channel groups are independent -- each compares, copies and ORs its own finds into the mask
    //  -- and go to workgroups of their own, blockIdx.z: a 512-channel 46x81 layer took eight dependent rounds, 15 us)
    const int NG = (C + 63) >> 6;
    const int cgBeg = a.copyAll ? zl : 0, cgEnd = a.copyAll ? cgBeg + 1 : NG;
    bool chg = false;
    for (int cg = cgBeg; cg < cgEnd; ++cg) group(cg, true, a.copyAll != 0, vm, chg);
    const unsigned long long b = __ballot(chg);
    if (lane == 0) sm[g] = b;
    __syncthreads();
    unsigned long long m = 0;
    for (int i = 0; i < 8; ++i) m |= sm[i];
    if (m == 0) return;      // uniform over the workgroup
    if (!a.copyAll) {        // feedback: both states take the changed pixels' values (.cu:74-80 of the half backend)
        bool dummy = false;
        for (int cg = 0; cg < NG; ++cg) group(cg, false, true, m, dummy);
    }

    unsigned long long D = m, SR = 0, SL = 0;
    for (int d = 1; d <= a.kWH; ++d) {
        D |= (m << d) | (m >> d);
        SR |= m >> (64 - d);
        SL |= m << (64 - d);
    }
    D &= vm;
    SR = (tx + 1 < a.wpr) ? (SR & cbs_valid_mask(W, tx + 1)) : 0ull;
    if (tx == 0) SL = 0;
    if (g == 0) {
        const int items = 3 * (2 * a.kHH + 1);
        for (int i = lane; i < items; i += 64) {
            const int yy = y + i / 3 - a.kHH;
            const int which = i % 3;
            if (yy < 0 || yy >= H) continue;
            const unsigned long long v = which == 0 ? D : (which == 1 ? SR : SL);
            const int t2 = which == 0 ? tx : (which == 1 ? tx + 1 : tx - 1);
            if (v) atomicOr(&a.masks[(long)yy * a.wpr + t2], v);
        }
    }
}

}  // namespace cbs

extern "C" {

int cbinfer_hsplit_supported(int C, int K, int kH, int kW) { return cbh_supported(C, K, kH, kW) ? 1 : 0; }
// mask words (cbinfer_mask_words(H,W)) the contraction of a K-channel fp16 layer takes
long cbinfer_hsplit_max_mask_words(int K) { (void)K; return CBS_PRE_MID; }
long cbinfer_hsplit_state_bytes(int C, int H, int W, int kH, int kW) {
    const CbsGeom g = cbh_geom(C, H, W, kH, kW);
    return CBS_SPAD + (long)g.Hp * g.Wp * g.rec;
}
long cbinfer_hsplit_prepared_bytes(int C, int K, int kH, int kW) {
    const CbsGeom g = cbh_geom(C, 64, 64, kH, kW);
    return cbs_plain_offset((long)g.nStages * (cbs_kp(K) / 32) * 4096, g.nStages);
}
long cbinfer_hsplit_workspace_bytes(int C, int H, int W, int K, int kH, int kW) {
    if (!cbh_supported(C, K, kH, kW)) return 0;
    if (cbh_geom(C, H, W, kH, kW).nStages < 48) return 0;
    const int bm = cbs_bm(K), bn = bm >= 128 ? 128 : 64;
    return 256 + cbs_slab_capacity(1, H, W, K) * bm * bn * 4;
}
int cbinfer_hsplit_prep_weights(const void* weight, void* prepared, int K, int C, int kH, int kW, int H, int W,
                                cbStream_t stream) {
    CB_REQUIRE(weight && prepared && H > 0 && W > 0);
    if (!cbh_supported(C, K, kH, kW)) return CB_ERR_UNSUPPORTED;
    const CbsGeom g = cbh_geom(C, H, W, kH, kW);
    if ((long)g.Hp * g.Wp * g.rec >= (1l << 31)) return CB_ERR_UNSUPPORTED;
    const int KP = cbs_kp(K);
    const long total = (long)g.nStages * (KP / 32) * 4 * 64;
    const long aBytes = (long)g.nStages * (KP / 32) * 4096;
    hipLaunchKernelGGL(cbh_prep_kernel, dim3(cb_div_up(total > g.nStages ? total : g.nStages, 256)), dim3(256), 0,
                       (hipStream_t)stream, (const _Float16*)weight, (halfx8*)prepared,
                       (int*)((char*)prepared + aBytes), g, K, KP, C);
    return cb_launch_status();
}
int cbinfer_hsplit_state_init(void* pixelState, int C, int H, int W, int kH, int kW, cbStream_t stream) {
    CB_REQUIRE(pixelState && H > 0 && W > 0);
    if (!cbh_supported(C, 1, kH, kW)) return CB_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(cbh_state_init_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, (uint4*)pixelState,
                       cbh_geom(C, H, W, kH, kW), C);
    return cb_launch_status();
}
int cbinfer_hsplit_state_rebuild(const void* state, void* pixelState, int C, int H, int W, int kH, int kW,
                                 cbStream_t stream) {
    CB_REQUIRE(state && pixelState && H > 0 && W > 0);
    if (!cbh_supported(C, 1, kH, kW)) return CB_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(cbh_state_rebuild_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream,
                       (const _Float16*)state, (char*)pixelState, cbh_geom(C, H, W, kH, kW), C);
    return cb_launch_status();
}

// One frame of a GROUP of fp16 CBConv2d layers of one geometry (conv2d.py:178-259 on the half backend; the two branches of
// an OpenPose stage, PoseModel.py:122-137): per layer with detect != 0 the detection launch (refresh of prevInput and
// of its pixel-major copy; feedbackLoop: at the changed pixels, else every value), then ONE contraction launch for all
// layers of the group (+ one reduce launch of a deep one), whose epilogue also runs the change detection of every
// layer's consumers (cbHalfLayer.next[]; copy-mode consumers only).  See include/cbinfer_hip.h.
int cbinfer_hsplit_forward_group(const cbHalfLayer* layers, int nLayers, int pooled, int pH, int pW, int C, int H, int W,
                                 int kH, int kW, int feedbackLoop, void* workspace, cbStream_t stream) {
    CB_REQUIRE(layers && nLayers >= 1 && nLayers <= CBH_GROUP && H > 0 && W > 0);
    if (pooled) CB_REQUIRE((H == pH / 2 || H == (pH + 1) / 2) && (W == pW / 2 || W == (pW + 1) / 2));
    const int K0 = layers[0].K;
    if (!cbh_supported(C, K0, kH, kW) || H > 65535) return CB_ERR_UNSUPPORTED;
    const CbsGeom g = cbh_geom(C, H, W, kH, kW);
    const int KP = cbs_kp(K0);
    int BM = cbs_bm(K0);
    const long MW = cbinfer_mask_words(H, W);
    if (MW > CBS_PRE_MID || (long)g.Hp * g.Wp * g.rec >= (1l << 31) || (long)H * W * W >= (1l << 32))
        return CB_ERR_UNSUPPORTED;
    if (workspace == nullptr && g.nStages >= 48) return CB_ERR_BADARG;
    for (int q = 0; q < nLayers; ++q) {
        const cbHalfLayer& L = layers[q];
        CB_REQUIRE(L.state && L.pixelState && L.frameMasks && L.output && L.idxOut && L.countOut && L.prepared);
        CB_REQUIRE(!L.detect || L.input);
        CB_REQUIRE(L.nNext >= 0 && L.nNext <= CBH_NEXT);
        // (one tile grid for the group: the padded row count and the tile height must agree)
        if (!cbh_supported(C, L.K, kH, kW) || cbs_kp(L.K) != KP || cbs_bm(L.K) != BM) return CB_ERR_UNSUPPORTED;
        for (int c = 0; c < L.nNext; ++c) {
            const cbHalfNext& n = L.next[c];
            CB_REQUIRE(n.state && n.pixelState && n.frameMasks && (n.kH & 1) && (n.kW & 1) && n.kH <= 15 && n.kW <= 15);
            if (L.K < 64) return CB_ERR_UNSUPPORTED;      // (a consumer on this machinery has >= 64 input channels)
        }
    }
    // The 128-row tile pays when there are tiles enough for every CU.  A shallow contraction on a map that would give
    // fewer than four 128 x 128 tiles per CU even if EVERY pixel changed -- at the change ratios this path is for it
    // then gives a fraction of one -- runs on 64 x 64 tiles instead (the prepared weights are per 32-row tile: either
    // tile height reads them): OpenPose's 128->128 @184x327 at 24 % change 21 -> 12 us, 128->256 @92x163 19 -> 10 us.
    {
        static int small = -1;
        if (small < 0) {
            const char* e = getenv("CBINFER_HSPLIT_SMALL_TILES");
            small = e ? atoi(e) : 1;
        }
        const long full128 = (((long)H * W + 127) / 128) * (KP / 128);
        if (small && BM == 128 && g.nStages < 48 && full128 < 4l * cbs_num_cus()) BM = 64;
    }
    hipStream_t s = (hipStream_t)stream;
    const int wpr = cbinfer_mask_words_per_row(W);
    {
        // the detection launch: the layers of the group that run their own detection, in ONE launch
        CbhDetGroup dg;
        int nd = 0;
        for (int q = 0; q < nLayers; ++q) {
            const cbHalfLayer& L = layers[q];
            if (!L.detect) continue;
            CbhDetArgs& a = dg.a[nd++];
            a.in = (const _Float16*)L.input, a.state = (_Float16*)L.state, a.S = (char*)L.pixelState;
            a.masks = (unsigned long long*)L.frameMasks;
            a.W = W, a.H = H, a.C = C, a.kHH = (kH - 1) / 2, a.kWH = (kW - 1) / 2, a.wpr = wpr;
            a.Wp = g.Wp, a.rec = g.rec, a.padY = g.padY, a.padXL = g.padXL, a.th = L.threshold;
            a.copyAll = feedbackLoop ? 0 : 1;
            a.upstream = nLayers == 1 ? L.upstreamCount : nullptr;
            a.pH = pH, a.pW = pW, a.prodMask = (const unsigned long long*)L.producerMask;   // (pooled: at pH x pW; else at H x W)
        }
        if (nd > 0) {
            for (int q = nd; q < CBH_GROUP; ++q) dg.a[q] = dg.a[0];
            dg.zPer = feedbackLoop ? 1 : g.C / 64;
            const dim3 grid(wpr, H, dg.zPer * nd);
            if (pooled)
                hipLaunchKernelGGL(cbh_detect_kernel<true>, grid, dim3(512), 0, s, dg);
            else
                hipLaunchKernelGGL(cbh_detect_kernel<false>, grid, dim3(512), 0, s, dg);
            const int st = cb_launch_status();
            if (st != CB_OK) return st;
        }
    }

    CbsParams p;
    CbhExt x;
    for (int q = 0; q < CBS_MAXSEQ; ++q) p.seq[q] = CbsSeq{};
    for (int q = 0; q < CBH_GROUP; ++q) {
        x.L[q] = CbhLayer{};
        const cbHalfLayer& L = layers[q < nLayers ? q : 0];
        if (q < nLayers) {
            p.seq[q].S = (const char*)L.pixelState;
            p.seq[q].out = (float*)L.output;
            p.seq[q].masks = (unsigned long long*)L.frameMasks;
            p.seq[q].listOut = L.idxOut, p.seq[q].countOut = L.countOut;
            p.seq[q].maskCopy = (unsigned long long*)L.maskCopy;
        }
        x.L[q].A = (const char*)L.prepared;
        x.L[q].bias = (const _Float16*)L.bias;
        x.L[q].K = L.K, x.L[q].relu = L.relu, x.L[q].nNext = q < nLayers ? L.nNext : 0;
        for (int c = 0; c < x.L[q].nNext; ++c) {
            const cbHalfNext& n = L.next[c];
            const CbsGeom gn = cbh_geom(L.K, H, W, n.kH, n.kW);
            CbhNext& d = x.L[q].next[c];
            d.state = (_Float16*)n.state, d.S = (char*)n.pixelState, d.masks = (unsigned long long*)n.frameMasks;
            d.kHH = (n.kH - 1) / 2, d.kWH = (n.kW - 1) / 2;
            d.Wp = gn.Wp, d.rec = gn.rec, d.padY = gn.padY, d.padXL = gn.padXL, d.th = n.threshold;
        }
    }
    p.nSeq = nLayers;
    p.aBytes = (long)g.nStages * (KP / 32) * 4096;
    p.A = (const char*)layers[0].prepared;
    p.stageOff = (const int*)((const char*)layers[0].prepared + p.aBytes);      // (the geometry's: one table for the group)
    p.wPlain = nullptr;
    p.bias = (const float*)layers[0].bias;      // (f16 values; the fp16 kernels read the per-layer pointers of CbhExt)
    p.info = workspace ? (int*)workspace : nullptr;
    p.slabs = workspace ? (float*)((char*)workspace + 256) : nullptr;
    p.K = K0, p.KP = KP, p.H = H, p.W = W, p.Wp = g.Wp, p.rec = g.rec, p.nStages = g.nStages, p.kH = kH, p.kW = kW;
    p.maskWords = (int)MW, p.wpr = wpr, p.relu = layers[0].relu, p.dummyBase = g.dummyBase;
    p.stateBytes = (long)g.Hp * g.Wp * g.rec;
    p.outScale = 1.0f;
    p.magicMW = (1ull << 32) / (unsigned long long)MW + 1ull;
    p.magicWpr = (1ull << 32) / (unsigned long long)p.wpr + 1ull;
    p.magicW = (1ull << 32) / (unsigned long long)W + 1ull;
    p.magicMT = (1ull << 32) / (unsigned long long)(KP / BM) + 1ull;
    p.forceSK = 0, p.accumulate = 0, p.halfOut = 1, p.splitRounds = 2, p.dbg = 0;
#ifdef CBS_LAST_ARRIVER
    p.lastArrive = 0, p.arriveTiles = nullptr;
#endif
    {
        static int mc = -1;      // CBINFER_HSPLIT_MAXCHUNKS (A/B aid; default 16: the device picks 4, 8 or 16 by the tile count)
        if (mc < 0) {
            const char* e = getenv("CBINFER_HSPLIT_MAXCHUNKS");
            mc = e && atoi(e) >= 4 ? atoi(e) : 16;
            // (the device loop doubles CH from 4 while 2 CH <= maxChunks, and decodes items with chShift in {2, 3, 4}:
            //  4, 8 or 16 chunks -- anything else is rounded down to one of them)
            mc = mc >= 16 ? 16 : (mc >= 8 ? 8 : 4);
        }
        p.maxChunks = mc;
    }
    p.upstream = nLayers == 1 ? layers[0].upstreamCount : nullptr;
    p.arriveShards = (int)(MW / 16 < 8 ? MW / 16 : 8);
    const long cap = cbs_slab_capacity(nLayers, H, W, K0), cap1 = cbs_slab_capacity(1, H, W, K0);
    p.slabCap = (int)(cap > 0x7fffffffl ? 0x7fffffffl : cap);
    x.slabCap1 = (int)(cap1 > 0x7fffffffl ? 0x7fffffffl : cap1);
    x.pad_ = 0;
    const long E = (long)nLayers * MW;
    if (BM == 128) {
        if (E <= CBS_PRE_BIG) return cbs_launch_conv<128, 128, 4, 2, CBS_PRE_BIG, true, 4, 1>(p, 1, nullptr, s, x);
        return cbs_launch_conv<128, 128, 4, 2, CBS_PRE_MID, false, 4, 1>(p, 1, nullptr, s, x);
    }
    if (E <= CBS_PRE_SMALL) {
        // (two workgroups per CU with four-stage rings: these contractions are shallow -- 9 to 36 stages -- and a layer
        //  sent here from the 128-row tile has two to four times the items)
        static int two = -1;
        if (two < 0) {
            const char* e = getenv("CBINFER_HSPLIT_TWO_PER_CU");
            two = e ? atoi(e) : 1;
        }
        if (two) return cbs_launch_conv<64, 64, 2, 2, CBS_PRE_SMALL, true, 4, 1>(p, 2, nullptr, s, x);
        return cbs_launch_conv<64, 64, 2, 2, CBS_PRE_SMALL, true, 8, 1>(p, 1, nullptr, s, x);
    }
    // (a large map -- OpenPose's 64-channel 368x654 layer: 541 tiles of nine stages at 14 % change -- has more tiles than
    //  CUs and little depth: two workgroups per CU with three-stage rings (68 KB each beside the 20 KB prefix))
    static int big2 = -1;
    if (big2 < 0) {
        const char* e = getenv("CBINFER_HSPLIT_BIG2");
        big2 = e ? atoi(e) : 1;
    }
    if (E > CBS_PRE_MID) return CB_ERR_UNSUPPORTED;
    if (big2) return cbs_launch_conv<64, 64, 2, 2, CBS_PRE_MID, false, 3, 1>(p, 2, nullptr, s, x);
    return cbs_launch_conv<64, 64, 2, 2, CBS_PRE_MID, false, 8, 1>(p, 1, nullptr, s, x);
}

// One frame of ONE fp16 CBConv2d: cbinfer_hsplit_forward_group with a single layer that runs its own detection and has no
// consumer folded in.  All tensors f16; frameMasks / idxOut / countOut / maskCopy as for cbinfer_split_forward;
// upstreamCount (optional) as for cbinfer_cbconv2d_forward_after.  pooled != 0: `input` is the tensor in FRONT of a
// 2x2/stride-2 max pool [C,pH,pW] (a CBPoolMax2d folded into the detection), producerMask (optional) the change mask the
// layer that produced it left this frame (cbinfer_mask_words(pH,pW) words): segments whose windows it did not touch are
// skipped.
int cbinfer_hsplit_forward(const int32_t* upstreamCount, const void* input, int pooled, int pH, int pW,
                           const uint64_t* producerMask, void* state, void* pixelState, uint64_t* frameMasks,
                           void* output, int32_t* idxOut, int32_t* countOut, uint64_t* maskCopy, const void* prepared,
                           const void* bias, int C, int H, int W, int K, int kH, int kW, float threshold,
                           int feedbackLoop, int relu, void* workspace, cbStream_t stream) {
    CB_REQUIRE(input);
    cbHalfLayer L = {};
    L.upstreamCount = upstreamCount, L.input = input, L.producerMask = producerMask;
    L.state = state, L.pixelState = pixelState, L.frameMasks = frameMasks, L.output = output;
    L.idxOut = idxOut, L.countOut = countOut, L.maskCopy = maskCopy, L.prepared = prepared, L.bias = bias;
    L.K = K, L.threshold = threshold, L.relu = relu, L.detect = 1, L.nNext = 0;
    return cbinfer_hsplit_forward_group(&L, 1, pooled, pH, pW, C, H, W, kH, kW, feedbackLoop, workspace, stream);
}

long cbinfer_hsplit_group_workspace_bytes(int nLayers, int C, int H, int W, int K, int kH, int kW) {
    if (nLayers < 1 || nLayers > CBH_GROUP || !cbh_supported(C, K, kH, kW)) return 0;
    if (cbh_geom(C, H, W, kH, kW).nStages < 48) return 0;
    const int bm = cbs_bm(K), bn = bm >= 128 ? 128 : 64;
    return 256 + cbs_slab_capacity(nLayers, H, W, K) * bm * bn * 4;
}

}  // extern "C"
