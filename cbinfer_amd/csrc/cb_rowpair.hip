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
//   schedule  a persistent grid of four four-wave workgroups per CU; workgroup b looks at the candidate units b,
//             b + grid, b + 2 grid, ... (at most four; neighbouring units -- a changed block -- land on different
//             workgroups), reads their two mask words in one burst and works through the non-empty ones.  (Round 4's
//             first form let every workgroup copy the whole mask into LDS and scan it for an ordered list of the
//             non-empty units: 13 us for 170 units, nearly all of it that prologue.)  The last workgroup out zeroes
//             the mask for the next frame (arrival counter; nobody waits) -- second form; now every workgroup zeroes
//             the words of its own units, which nobody else reads.
//   gather    the (kH + 1) x (64 + kW - 1) input rows under the pair are staged once per channel in LDS (coalesced
//             row loads); every B operand of every tap is a ds_read_b32 at `lane base + immediate tap offset`.
//   product   v_mfma_f32_16x16x4_f32 (the exact f32 fma chain), two 16-pixel tiles per wave (four independent
//             accumulation chains), two waves per row; the weights in MFMA fragment order straight from L2 into registers
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

#define CBP_NT 256
#define CBP_NW 4
#define CBP_MAXCAND 4               // candidate units per workgroup: the grid is at least units / 4
#define CBP_MAXWORDS (1 << 20)

struct PairNext {                   // geometry of the next layer's detection (fold != 0)
    int fold, H2, W2, wpr2, kHH, kWH, Wp, rec, padY, padXL;
    float th;
    int planes;                     // its records: 2 = f16 pairs, 3 = bf16 triples (cb_split_common.h)
};
struct PairSeq {                    // the tensors of ONE sequence (several sequences per launch: cbPairSeq)
    const float* frame;             // DET: this frame's input [C,H,W] (the layer's change detection rides in the launch)
    const float* state;             // prevInput [C,H,W]: what the gather reads (conv2d.py:242)
    float* out;                     // prevOutput [K,H,W]
    unsigned long long* bits;       // change mask of this frame (every workgroup zeroes the words of its units)
    unsigned long long* maskCopy;
    float* nstate;                  // the next layer's prevInput [K, H2, W2]
    char* nS;                       // its split state
    unsigned long long* nmasks;     // its frame mask (the detection ORs into it)
    int* nflag;                     // its range flag (may be null)
};
struct PairParams {
    const float* wq;                // cbinfer_rowconv_prep_weights layout
    const float* bias;
    int C, H, W, K, relu, wpr, MW, units, nSeq;
    float th;                       // DET: the layer's own threshold
    PairNext next;
    PairSeq seq[CBINFER_SPLIT_MAX_SEQUENCES];
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

#ifdef CBP_STAMP
// diagnostic build only (make EXTRA=-DCBP_STAMP; tools/pair_stamps.py): per-workgroup phase stamps of its FIRST non-empty
// unit, 100 MHz constant clock: 0 entry, 1 mask words known, 2 weights in LDS / requests issued, 3 patch staged,
// 4 k-loop + stores done, 5 pooled detection decided, 6 unit done, 7 kernel exit
__device__ unsigned long long cbp_stamp_buf[4096 * 8];
#define CBP_STAMP_AT(i)                                                                                  \
    do {                                                                                                 \
        if (threadIdx.x == 0 && blockIdx.x < 4096 && ((i) == 0 || (i) == 1 || (i) == 7 || cbp_first))    \
            cbp_stamp_buf[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime();                      \
    } while (0)
#else
#define CBP_STAMP_AT(i)
#endif

// DET (round 6; one sequence, one unit per workgroup): the layer's OWN change detection in this launch as well.  The unit's
// patch is what the detection of its two rows needs anyway -- rows ya - PH .. ya + 1 + PH, columns 64 tx - PW .. 64 tx + 63 + PW
// --, so every workgroup loads it from the frame and from the state, compares (strict >, any channel: cbconv2d_cg_backend.cu:
// 40-73), dilates the changed pixels of the patch into its own two mask words and, if they are not empty, multiplies the
// REFRESHED values (frame where the pixel changed, state elsewhere: .cu:74-80).  It does not write the state: a neighbour
// that starts later must still see the old one -- the refresh is left to a launch behind this one (cbs_conv_kernel's idle
// workgroups: cbinfer_refresh_state's loop).  No global mask, no mask round trip in front of the patch's, no detection launch.
template <int KH, int KW, bool DET = false>
__global__ __launch_bounds__(CBP_NT, 4) void cbp_rowpair_kernel(PairParams p) {
    cb_touch_kernarg<sizeof(PairParams)>();
#ifdef CBP_STAMP
    bool cbp_first = true;
#endif
    CBP_STAMP_AT(0);
    constexpr int RS = 64 + KW - 1, PR = KH + 1, CS = cbp_plane_stride(KH, KW), S = KH * KW, G = (S + 3) / 4;
    constexpr int PH = (KH - 1) / 2, PW = (KW - 1) / 2;
    __shared__ float s_patch[4 * CS];
    __shared__ __attribute__((aligned(16))) float s_w[G * 256];     // the weights, MFMA fragment order: [group][lane][4]
    __shared__ __attribute__((aligned(16))) float s_out[2 * 16 * 64];          // [row][channel][x]: the pair's outputs after this frame
    __shared__ float s_P[16 * 33];                // pooled values [channel][xo]
    __shared__ unsigned s_chg[CBP_NW];
    __shared__ unsigned long long s_rb[DET ? 2 * (KH + 1) : 1];      // DET: changed pixels of the patch, [row][64 + KW - 1 bits]

    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wpr = p.wpr, H = p.H, W = p.W, HW = H * W;
    const bool fold = p.next.fold != 0;
    const int unitsSeq = p.units, units = p.units * p.nSeq;       // candidate unit u: sequence u / unitsSeq
    // (the per-sequence tensors come out of the kernel-argument table by the unit's sequence index: scalar loads)
    const PairSeq* tab = ((const PairParams*)__builtin_amdgcn_kernarg_segment_ptr())->seq;

    // ---- requested first, before anybody knows whether this workgroup has work: the weights (fragment order, every
    //      wave the same 16 output channels; they go through LDS, 16 bytes per lane and group of four k-steps -- held
    //      in registers, 52 of them for 7x7, they cost the third workgroup per CU) and the bias.  Their round trip runs
    //      beside the mask words' (1.5 us: the detection's atomics leave them at the memory side).
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.wq, 0, G * 1024, 0x00020000);
    constexpr int WPT = (G * 64 + CBP_NT - 1) / CBP_NT;
    floatx4 wv[WPT];
    float bv[4];

    // ---- this workgroup's candidate units: blockIdx.x, + gridDim.x, ... (neighbouring units go to different
    //      workgroups); their mask words are requested in one burst and are the same for every lane ---------------
    unsigned long long cwA[CBP_MAXCAND], cwB[CBP_MAXCAND];
    constexpr int PROWS = 4 * PR;                       // patch rows: plane x row
    constexpr int PX4 = (PROWS * 16 + CBP_NT - 1) / CBP_NT, PED = (PROWS * (KW - 1) + CBP_NT - 1) / CBP_NT;
    floatx4 xv4[DET ? PX4 : 1], sv4[DET ? PX4 : 1];
    float xe[DET ? PED : 1], se[DET ? PED : 1];
    if constexpr (DET) {
        const int ul = blockIdx.x;
        const int yo = ul / wpr, tx = ul - yo * wpr, ya = 2 * yo;
        const __amdgpu_buffer_rsrc_t frsrc =
            __builtin_amdgcn_make_buffer_rsrc((void*)tab[0].frame, 0, (int)min((long)p.C * HW * 4, (long)0x7fffffff), 0x00020000);
        const __amdgpu_buffer_rsrc_t srsrc0 =
            __builtin_amdgcn_make_buffer_rsrc((void*)tab[0].state, 0, (int)min((long)p.C * HW * 4, (long)0x7fffffff), 0x00020000);
        if (t < 2 * PR) s_rb[t] = 0ull;
#pragma unroll
        for (int i = 0; i < PX4; ++i) {
            const int e = t + CBP_NT * i;
            const int r = e >> 4, j4 = (e & 15) * 4;
            const int c = r / PR, pr = r - c * PR;
            const int yy = ya + pr - PH, xx = tx * 64 + j4;
            const bool ok = (r < PROWS) & (c < p.C) & (yy >= 0) & (yy < H) & (xx < W);
            const int off = ok ? ((c * H + yy) * W + xx) * 4 : (1 << 30);      // (out of range: the buffer load returns 0)
            xv4[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(frsrc, off, 0, 0));
            sv4[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(srsrc0, off, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < PED; ++i) {
            const int e = t + CBP_NT * i;
            const int r = e / (KW - 1), j = e - r * (KW - 1);
            const int c = r / PR, pr = r - c * PR;
            const int yy = ya + pr - PH, xx = j < PW ? tx * 64 - PW + j : tx * 64 + 64 + (j - PW);
            const bool ok = (r < PROWS) & (c < p.C) & (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W);
            const int off = ok ? ((c * H + yy) * W + xx) * 4 : (1 << 30);
            xe[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(frsrc, off, 0, 0));
            se[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srsrc0, off, 0, 0));
        }
    }
#pragma unroll
    for (int k = 0; k < CBP_MAXCAND; ++k) {
        if constexpr (DET) {
            cwA[k] = 0ull, cwB[k] = 0ull;
            continue;
        }
        const int u = blockIdx.x + k * gridDim.x;
        const int uc = min(u, units - 1);
        const int q = uc / unitsSeq, ul = uc - q * unitsSeq;
        const int yo = ul / wpr, tx = ul - yo * wpr, ya = 2 * yo;
        const unsigned long long* bq = tab[q].bits;
        const unsigned long long a0 = bq[ya * wpr + tx];
        const unsigned long long b0 = bq[min(ya + 1, H - 1) * wpr + tx];
        cwA[k] = u < units ? a0 : 0ull;
        cwB[k] = (u < units && ya + 1 < H) ? b0 : 0ull;
    }
#pragma unroll
    for (int i = 0; i < WPT; ++i)
        wv[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(
                                                wrsrc, min(t + CBP_NT * i, G * 64 - 1) * 16, 0, 0));
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = p.bias ? p.bias[min(4 * (lane >> 4) + r, p.K - 1)] : 0.f;
    if constexpr (DET) {
        // (the weights to LDS right away -- held in registers through the detection they cost the kernel its occupancy)
#pragma unroll
        for (int i = 0; i < WPT; ++i)
            if (t + CBP_NT * i < G * 64) *(floatx4*)(s_w + (t + CBP_NT * i) * 4) = wv[i];
        // phase 1: the changed pixels of the patch (any channel) as bits of s_rb[patch row]: bit = patch column
        const int ul = blockIdx.x;
        const int yo = ul / wpr, tx = ul - yo * wpr, ya = 2 * yo;
        __syncthreads();      // (s_rb is zero)
#pragma unroll
        for (int i = 0; i < PX4; ++i) {
            const int e = t + CBP_NT * i;
            const int r = e >> 4, j4 = (e & 15) * 4;
            const int c = r / PR, pr = r - c * PR;
            const int yy = ya + pr - PH, xx = tx * 64 + j4;
            const bool ok = (r < PROWS) & (c < p.C) & (yy >= 0) & (yy < H);
            unsigned b4 = 0u;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)      // (a piece cut by the map's right edge holds the next row's first pixels)
                b4 |= (unsigned)(ok && xx + jj < W && cb_changed(sv4[i][jj], xv4[i][jj], p.th)) << jj;
            if (b4) {
                const int col = PW + j4;      // (<= 63 + PW: the four bits may straddle the first word's end)
                atomicOr(&s_rb[2 * pr], (unsigned long long)b4 << col);
                if (col + 3 >= 64) atomicOr(&s_rb[2 * pr + 1], (unsigned long long)b4 >> (64 - col));
            }
        }
#pragma unroll
        for (int i = 0; i < PED; ++i) {
            const int e = t + CBP_NT * i;
            const int r = e / (KW - 1), j = e - r * (KW - 1);
            const int c = r / PR, pr = r - c * PR;
            const int yy = ya + pr - PH, xx = j < PW ? tx * 64 - PW + j : tx * 64 + 64 + (j - PW);
            const bool ok = (r < PROWS) & (c < p.C) & (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W);
            const int col = j < PW ? j : 64 + j;
            if (ok && cb_changed(se[i], xe[i], p.th)) {
                if (col < 64) atomicOr(&s_rb[2 * pr], 1ull << col);
                else atomicOr(&s_rb[2 * pr + 1], 1ull << (col - 64));
            }
        }
        __syncthreads();
        // the unit's two mask words: output pixel (row r, column x) <- patch rows r .. r + KH - 1, columns x .. x + KW - 1
        unsigned long long w2[2] = {0ull, 0ull};
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            unsigned long long lo = 0ull, hi = 0ull;
#pragma unroll
            for (int dr = 0; dr < KH; ++dr) lo |= s_rb[2 * (r + dr)], hi |= s_rb[2 * (r + dr) + 1];
            unsigned long long D = 0ull;
#pragma unroll
            for (int dx = 0; dx < KW; ++dx) D |= dx ? ((lo >> dx) | (hi << (64 - dx))) : lo;
            const int rem = W - tx * 64;
            if (rem < 64) D &= (1ull << rem) - 1ull;
            w2[r] = (ya + r < H) ? D : 0ull;
        }
        cwA[0] = w2[0], cwB[0] = w2[1];
    }
    auto uniform64 = [](unsigned long long v) -> unsigned long long {
        return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
               (unsigned)__builtin_amdgcn_readfirstlane((int)v);
    };
    bool any = false;
#pragma unroll
    for (int k = 0; k < CBP_MAXCAND; ++k) {
        cwA[k] = uniform64(cwA[k]), cwB[k] = uniform64(cwB[k]);
        if (k == CBP_MAXCAND - 1) CBP_STAMP_AT(1);
        any |= (cwA[k] | cwB[k]) != 0ull;
    }
    // A unit's two mask words belong to this workgroup alone: it leaves the frame's copy at its fixed address and zeroes
    // them for the next frame's detection itself, AT THE END (no arrival counter: 768 workgroups counting themselves on
    // one word took 9 of this launch's first 19 us -- a returning atomic per 11 ns).  Why at the end: with the stores
    // issued here, right behind the words' loads, workgroups that read their inputs a few microseconds into the launch
    // (late starters of a grid beyond the resident four per CU; the same with a deliberate delay) sporadically computed
    // whole units from wrong operands -- stale state or weights, by the pattern of the errors -- while identical code
    // with the stores behind the last load of the workgroup never did (tmp experiments of round 4, DESIGN A.8).  The
    // rule kept here: no store is issued while a load of this wave is still being waited for by a COUNTED vmcnt.
    auto finish = [&]() {
        if (t == 0) {
#pragma unroll
            for (int k = 0; k < CBP_MAXCAND; ++k) {
                const int u = blockIdx.x + k * gridDim.x;
                if (u < units) {
                    const int q = u / unitsSeq, ul = u - q * unitsSeq;
                    const int yo = ul / wpr, tx = ul - yo * wpr, ya = 2 * yo;
                    unsigned long long* mc = tab[q].maskCopy;
                    unsigned long long* bq = tab[q].bits;
                    if (mc) {
                        mc[ya * wpr + tx] = cwA[k];
                        if (ya + 1 < H) mc[(ya + 1) * wpr + tx] = cwB[k];
                    }
                    if (!DET && cwA[k]) bq[ya * wpr + tx] = 0ull;
                    if (!DET && cwB[k]) bq[(ya + 1) * wpr + tx] = 0ull;
                }
            }
        }
    };
    if (!any) {
        // (the weight and bias loads requested above are waited for before the wave stores anything or ends)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        finish();
        CBP_STAMP_AT(7);
        return;
    }
    if constexpr (!DET) {
#pragma unroll
        for (int i = 0; i < WPT; ++i)
            if (t + CBP_NT * i < G * 64) *(floatx4*)(s_w + (t + CBP_NT * i) * 4) = wv[i];
        // (visible to every wave behind the first barrier of the unit loop)
    }

#pragma unroll 1
    for (int k = 0; k < CBP_MAXCAND; ++k) {
        unsigned long long wordA = cwA[0], wordB = cwB[0];
#pragma unroll
        for (int j = 1; j < CBP_MAXCAND; ++j)
            if (j == k) wordA = cwA[j], wordB = cwB[j];
        if ((wordA | wordB) == 0ull) continue;      // (uniform)
        const int u = blockIdx.x + k * gridDim.x;
        const int q = u / unitsSeq, ul = u - q * unitsSeq;
        const int yo = ul / wpr, tx = ul - yo * wpr, ya = 2 * yo;
        const bool hasB = ya + 1 < H;
        const PairSeq sq = tab[q];
        const __amdgpu_buffer_rsrc_t srsrc =
            __builtin_amdgcn_make_buffer_rsrc((void*)sq.state, 0, (int)min((long)p.C * HW * 4, (long)0x7fffffff), 0x00020000);
        const __amdgpu_buffer_rsrc_t orsrc =
            __builtin_amdgcn_make_buffer_rsrc((void*)sq.out, 0, (int)min((long)p.K * HW * 4, (long)0x7fffffff), 0x00020000);
        // (the previous unit's stores -- outputs, the next layer's state and mask -- have left before this unit's loads
        //  are counted: see the rule above)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ---- requests, all in one burst: the patch, the pair's old outputs, the next layer's state.  16 bytes per
        //      lane wherever rows allow it (the 64 columns of the word; the kW - 1 columns beside them by dword): seven
        //      load instructions per thread instead of nineteen -- a CU takes in about 11 bytes per cycle however they
        //      are requested, but every instruction is a slot in its memory pipe.
        floatx4 pv4[PX4];
        float pe[PED];
#pragma unroll
        for (int i = 0; i < PX4; ++i) {
            if constexpr (DET) {      // (the refreshed value: the frame's where the pixel changed, the state's elsewhere)
                const int e = t + CBP_NT * i;
                const int r = e >> 4, j4 = (e & 15) * 4;
                const int pr = r - (r / PR) * PR, col = PW + j4;
                const unsigned long long rb0 = s_rb[2 * (pr % PR)], rb1 = s_rb[2 * (pr % PR) + 1];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int cc = col + jj;
                    const bool ch = cc < 64 ? ((rb0 >> cc) & 1ull) != 0ull : ((rb1 >> (cc - 64)) & 1ull) != 0ull;
                    // (columns beyond the map's right edge are the next row's first pixels, not padding)
                    pv4[i][jj] = tx * 64 + j4 + jj >= W ? 0.f : (ch ? xv4[i][jj] : sv4[i][jj]);
                }
                continue;
            }
            const int e = t + CBP_NT * i;
            const int r = e >> 4, j4 = (e & 15) * 4;
            const int c = r / PR, pr = r - c * PR;
            const int yy = ya + pr - PH, xx = tx * 64 + j4;
            const bool ok = (r < PROWS) & (c < p.C) & (yy >= 0) & (yy < H) & (xx < W);      // (no short circuit: a branch each)
            // (an invalid piece gets an out-of-range offset: the buffer load returns 0 for it)
            floatx4 v = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(
                                                        srsrc, ok ? ((c * H + yy) * W + xx) * 4 : (1 << 30), 0, 0));
            // (columns beyond the map's right edge are the next row's first pixels, not padding)
            if (xx + 1 >= W) v[1] = 0.f;
            if (xx + 2 >= W) v[2] = 0.f;
            if (xx + 3 >= W) v[3] = 0.f;
            pv4[i] = v;
        }
#pragma unroll
        for (int i = 0; i < PED; ++i) {
            const int e = t + CBP_NT * i;
            const int r = e / (KW - 1), j = e - r * (KW - 1);      // j < PW: left of the word, else right of it
            const int c = r / PR, pr = r - c * PR;
            if constexpr (DET) {
                const int col = j < PW ? j : 64 + j;
                const bool ch = col < 64 ? ((s_rb[2 * (pr % PR)] >> col) & 1ull) != 0ull
                                         : ((s_rb[2 * (pr % PR) + 1] >> (col - 64)) & 1ull) != 0ull;
                pe[i] = ch ? xe[i] : se[i];
                continue;
            }
            const int yy = ya + pr - PH, xx = j < PW ? tx * 64 - PW + j : tx * 64 + 64 + (j - PW);
            const bool ok = (r < PROWS) & (c < p.C) & (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W);
            pe[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                  srsrc, ok ? ((c * H + yy) * W + xx) * 4 : (1 << 30), 0, 0));
        }
        constexpr int OX4 = 2 * 16 * 16 / CBP_NT;       // old outputs: [row][channel][16 pieces of 4 columns]
        floatx4 ov4[OX4];
        float s2[2] = {0.f, 0.f};
        const int c2 = t >> 5, xo = t & 31, gx = tx * 32 + xo;        // pooled pixel xo, channels c2 and c2 + 8
        const bool valid2 = fold && yo < p.next.H2 && gx < p.next.W2;
        const long H2W2 = fold ? (long)p.next.H2 * p.next.W2 : 0;
        if (fold) {
#pragma unroll
            for (int i = 0; i < OX4; ++i) {
                const int e = t + CBP_NT * i;
                const int x4 = (e & 15) * 4, m = (e >> 4) & 15, r = e >> 8;
                const int yy = ya + r, xx = tx * 64 + x4;
                const bool ok = (yy < H) & (xx < W) & (m < p.K);
                // (columns beyond the map's edge come back as the next row's pixels: no window of a valid pooled pixel
                //  reads them -- cx1 below)
                if constexpr (DET) {
                    // straight into the LDS tile by LDS-DMA (lane t's 16 bytes land at s_out + 16 e): no registers held
                    // through the multiplication, nothing waited for here -- they land while the unit is multiplied, and
                    // the wait stands in front of the first store into the tile.  (Inline asm: a DMA the compiler knows of
                    // it would wait for in front of the next LDS access.)
                    typedef __attribute__((address_space(3))) void* lds_ptr_t;
                    const unsigned base = (unsigned)(size_t)(lds_ptr_t)(s_out + (wave * 64 + CBP_NT * i) * 4);
                    const int voff = ok ? (m * HW + yy * W + xx) * 4 : (1 << 30);
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                                 ::"s"(base), "v"(voff), "s"(orsrc) : "memory", "m0");
                    continue;
                }
                ov4[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(
                                                         orsrc, ok ? (m * HW + yy * W + xx) * 4 : (1 << 30), 0, 0));
            }
            // (clamped addresses, predicated uses)
            const long pos2 = (long)min(yo, p.next.H2 - 1) * p.next.W2 + min(gx, p.next.W2 - 1);
            s2[0] = sq.nstate[(long)min(c2, p.K - 1) * H2W2 + pos2];
            s2[1] = sq.nstate[(long)min(c2 + 8, p.K - 1) * H2W2 + pos2];
        }
        CBP_STAMP_AT(2);
        __syncthreads();        // (the previous unit's readers of s_patch / s_out / s_P are done)
#pragma unroll
        for (int i = 0; i < PX4; ++i) {
            const int e = t + CBP_NT * i;
            const int r = e >> 4, j4 = (e & 15) * 4;
            const int c = r / PR, pr = r - c * PR;
            if (r < PROWS) {
                float* d = s_patch + c * CS + pr * RS + PW + j4;      // (PW columns in: not 16-byte aligned)
                d[0] = pv4[i][0], d[1] = pv4[i][1], d[2] = pv4[i][2], d[3] = pv4[i][3];
            }
        }
#pragma unroll
        for (int i = 0; i < PED; ++i) {
            const int e = t + CBP_NT * i;
            const int r = e / (KW - 1), j = e - r * (KW - 1);
            const int c = r / PR, pr = r - c * PR;
            if (r < PROWS) s_patch[c * CS + pr * RS + (j < PW ? j : 64 + j)] = pe[i];
        }
        if (fold && !DET) {
#pragma unroll
            for (int i = 0; i < OX4; ++i) *(floatx4*)(s_out + (t + CBP_NT * i) * 4) = ov4[i];
        }
        __syncthreads();

        CBP_STAMP_AT(3);
        // ---- two 16-pixel tiles per wave: waves 0, 1 the first row's tiles (0,1), (2,3); waves 2, 3 the second row's.
        //      Four independent accumulation chains per wave keep the matrix pipe fed. ---------------------------------
        const int row = wave >> 1, tile0 = 2 * (wave & 1);
        const unsigned long long word = row ? wordB : wordA;
        const int pc = __popcll(word);
        const int n0 = tile0 * 16 + (lane & 15), n1 = n0 + 16;
        int xl0 = 0, xl1 = 0;
        floatx4 accA = {0.f, 0.f, 0.f, 0.f}, accB = accA;
        if (tile0 * 16 < pc) {
            xl0 = cbp_nth_bit(word, n0 < pc ? n0 : 0), xl1 = cbp_nth_bit(word, n1 < pc ? n1 : 0);
            const float* pl0 = s_patch + (lane >> 4) * CS + row * RS + xl0;
            const float* pl1 = s_patch + (lane >> 4) * CS + row * RS + xl1;
            // (the second tile is multiplied whether it exists or not -- its lanes then read the first changed pixel's
            //  columns, and nothing of it is stored: no branch inside the k-loop, which otherwise came out as read ->
            //  wait -> one matrix instruction -> branch, 98 times over: 5 of this kernel's first 18 us)
            floatx4 acc00 = {0.f, 0.f, 0.f, 0.f}, acc01 = acc00, acc10 = acc00, acc11 = acc00;
            const floatx4* wl = (const floatx4*)s_w + lane;
            // (DET: the weight fragments of ONE filter row at a time, read with the B operands of that row -- all G of
            //  them in registers, 52 for 7x7, do not fit beside the detection's patches)
            constexpr int AR = (KW + 6) / 4 + 1;      // fragments a filter row can touch
            floatx4 ag[DET ? 1 : G], arow[DET ? AR : 1], nrow[DET ? AR : 1];
            if constexpr (!DET) {
#pragma unroll
                for (int g = 0; g < G; ++g) ag[g] = wl[g * 64];
            } else {
#pragma unroll
                for (int j = 0; j < AR; ++j) arow[j] = wl[min(j, G - 1) * 64];
            }
            // one filter row at a time: the B operands of row ky + 1 are read while row ky is multiplied
            float b0[KW], b1[KW], nb0[KW], nb1[KW];
#pragma unroll
            for (int kx = 0; kx < KW; ++kx) b0[kx] = pl0[kx], b1[kx] = pl1[kx];
#pragma unroll
            for (int ky = 0; ky < KH; ++ky) {
                const int g0 = (ky * KW) >> 2;      // (compile-time: the loop is unrolled)
                if (ky + 1 < KH) {
#pragma unroll
                    for (int kx = 0; kx < KW; ++kx) nb0[kx] = pl0[(ky + 1) * RS + kx], nb1[kx] = pl1[(ky + 1) * RS + kx];
                    if constexpr (DET) {
                        const int gn = ((ky + 1) * KW) >> 2;
#pragma unroll
                        for (int j = 0; j < AR; ++j) nrow[j] = wl[min(gn + j, G - 1) * 64];
                    }
                }
#pragma unroll
                for (int kx = 0; kx < KW; ++kx) {
                    const int s = ky * KW + kx;
                    const float av = DET ? arow[DET ? (s >> 2) - g0 : 0][s & 3] : ag[DET ? 0 : (s >> 2)][s & 3];
                    if (s & 1) {
                        acc01 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0[kx], acc01, 0, 0, 0);
                        acc11 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1[kx], acc11, 0, 0, 0);
                    } else {
                        acc00 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0[kx], acc00, 0, 0, 0);
                        acc10 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1[kx], acc10, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int kx = 0; kx < KW; ++kx) b0[kx] = nb0[kx], b1[kx] = nb1[kx];
                if constexpr (DET) {
#pragma unroll
                    for (int j = 0; j < AR; ++j) arow[j] = nrow[j];
                }
            }
            accA = acc00 + acc01, accB = acc10 + acc11;
        }
        if constexpr (DET) {
            // (EVERY wave's DMA of the old outputs has landed in the LDS tile before anybody stores a new value over one of
            //  them: each wave waits for its own, the barrier makes it everybody's)
            if (fold) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
        }
        if (tile0 * 16 < pc) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int n = h2 ? n1 : n0, xl = h2 ? xl1 : xl0;
                if (n < pc) {
                    const int pix = (ya + row) * W + tx * 64 + xl;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int m = 4 * (lane >> 4) + r;
                        float v = (h2 ? accB[r] : accA[r]) + bv[r];
                        if (p.relu) v = v <= 0.f ? 0.f : v;
                        if (m < p.K) {
                            sq.out[(long)m * HW + pix] = v;
                            s_out[(row * 16 + m) * 64 + xl] = v;
                        }
                    }
                }
            }
        }
        if (!fold) continue;
        __syncthreads();
        CBP_STAMP_AT(4);

        // ---- the next layer's detection for the 32 pooled pixels of this unit: thread = (pooled xo; channels c2, c2+8)
        const unsigned long long tw = wordA | wordB;
        const bool touched = ((tw >> (2 * xo)) & 3ull) != 0ull;
        const int cx1 = (tx * 64 + 2 * xo + 1 < W) ? 2 * xo + 1 : 2 * xo, r1 = hasB ? 16 * 64 : 0;
        float pooled[2];
        bool chg = false;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const float* so = s_out + (c2 + 8 * h2) * 64;
            pooled[h2] = fmaxf(fmaxf(so[2 * xo], so[cx1]), fmaxf(so[r1 + 2 * xo], so[r1 + cx1]));
            chg |= (c2 + 8 * h2 < p.K) && cb_changed(s2[h2], pooled[h2], p.next.th);
            s_P[(c2 + 8 * h2) * 33 + xo] = pooled[h2];
        }
        chg = chg && valid2 && touched;
        const unsigned long long bal = __ballot(chg);
        if (lane == 0) s_chg[wave] = (unsigned)(bal | (bal >> 32));
        __syncthreads();
        unsigned m32 = 0;
#pragma unroll
        for (int w = 0; w < CBP_NW; ++w) m32 |= s_chg[w];
        m32 = __builtin_amdgcn_readfirstlane(m32);
        CBP_STAMP_AT(5);
        if (m32 == 0u) continue;        // (uniform)
        // feedback: refresh the f32 state at the changed pooled pixels only (.cu:74-80) ...
        if (valid2 && ((m32 >> xo) & 1u)) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
                if (c2 + 8 * h2 < p.K)
                    sq.nstate[(long)(c2 + 8 * h2) * H2W2 + (long)yo * p.next.W2 + gx] = pooled[h2];
        }
        // ... and its pre-split pixel-major copy: thread = (pooled pixel, 8-channel half), whole 16-byte pieces
        if (t < 64) {
            const int pl2 = t >> 1, half = t & 1;
            if ((m32 >> pl2) & 1u) {
                float v8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int c = half * 8 + j;
                    v8[j] = c < p.K ? s_P[c * 33 + pl2] : 0.f;
                }
                char* rec = sq.nS + CBS_SPAD +
                            ((long)(yo + p.next.padY) * p.next.Wp + (tx * 32 + pl2 + p.next.padXL)) * p.next.rec;
                const bool over = cbs::cbs_store_part(rec, 0, half, p.next.planes, v8);
                if (over && sq.nflag) *sq.nflag = 1;
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
                if (v) atomicOr(&sq.nmasks[(long)yy * wpr2 + t2], v);
            }
        }
        CBP_STAMP_AT(6);
#ifdef CBP_STAMP
        cbp_first = false;
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    finish();
    CBP_STAMP_AT(7);
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

#ifdef CBP_STAMP
extern "C" int cbinfer_debug_pair_stamps(void* host, long bytes, int clear) {
    if (clear) {
        void* d = nullptr;
        if (hipGetSymbolAddress(&d, HIP_SYMBOL(cbp_stamp_buf)) != hipSuccess) return -1;
        return (int)hipMemset(d, 0, sizeof(cbp_stamp_buf));
    }
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(cbp_stamp_buf), (size_t)bytes);
}
#endif

extern "C" {

// Layers the row-pair kernel takes: fp32, at most 4 input and 16 output channels, a square odd filter of 3, 5 or 7,
// a change mask of at most CBP_MAXWORDS words.
int cbinfer_rowpairs_supported(int C, int K, int kH, int kW, int H, int W) {
    if (C < 1 || C > 4 || K < 1 || K > 16 || kH != kW || (kH != 3 && kH != 5 && kH != 7) || H < 1 || W < 1) return 0;
    if (cbinfer_mask_words(H, W) > CBP_MAXWORDS || (long)16 * H * W * 4 >= (1l << 30)) return 0;
    return 1;
}

static int cbp_launch(const cbPairSeq* seqs, int nSeq, const void* prepared, const float* bias, int C, int H, int W,
                      int K, int kH, int kW, int relu, const cbNextDetect* next, cbStream_t stream,
                      const float* frame = nullptr, float threshold = 0.f) {
    CB_REQUIRE(seqs && nSeq >= 1 && nSeq <= CBINFER_SPLIT_MAX_SEQUENCES && prepared);
    if (!cbinfer_rowpairs_supported(C, K, kH, kW, H, W)) return CB_ERR_UNSUPPORTED;
    PairParams p;
    p.wq = (const float*)prepared, p.bias = bias;
    p.C = C, p.H = H, p.W = W, p.K = K, p.relu = relu, p.nSeq = nSeq;
    p.wpr = cbinfer_mask_words_per_row(W), p.MW = (int)cbinfer_mask_words(H, W);
    p.units = ((H + 1) / 2) * p.wpr;
    p.next.fold = 0, p.next.planes = 2;
    const bool fold = next != nullptr && next->H > 0 && seqs[0].nextState != nullptr;
    if (fold) {
        // the layer behind the 2x2/stride-2 pool: K channels at (H/2 or (H+1)/2) x (W/2 or (W+1)/2), split-state form
        CB_REQUIRE((next->H == H / 2 || next->H == (H + 1) / 2) && (next->W == W / 2 || next->W == (W + 1) / 2));
        if (K != 16 || !cbs::cbs_supported(K, 1, next->kH, next->kW)) return CB_ERR_UNSUPPORTED;
        CB_REQUIRE(next->arith == 0 || next->arith == 1);
        const cbs::CbsGeom g = cbs::cbs_geom(K, next->H, next->W, next->kH, next->kW, next->arith ? 3 : 2);
        p.next.fold = 1;
        p.next.planes = g.planes;
        p.next.H2 = next->H, p.next.W2 = next->W, p.next.wpr2 = cbinfer_mask_words_per_row(next->W);
        p.next.kHH = (next->kH - 1) / 2, p.next.kWH = (next->kW - 1) / 2;
        p.next.Wp = g.Wp, p.next.rec = g.rec, p.next.padY = g.padY, p.next.padXL = g.padXL;
        p.next.th = next->threshold;
    }
    p.th = threshold;
    for (int q = 0; q < nSeq; ++q) {
        CB_REQUIRE(seqs[q].state && seqs[q].output && (seqs[q].bits || frame));
        p.seq[q].frame = frame;
        p.seq[q].state = seqs[q].state, p.seq[q].out = seqs[q].output;
        p.seq[q].bits = (unsigned long long*)seqs[q].bits, p.seq[q].maskCopy = (unsigned long long*)seqs[q].maskCopy;
        p.seq[q].nstate = nullptr, p.seq[q].nS = nullptr, p.seq[q].nmasks = nullptr, p.seq[q].nflag = nullptr;
        if (fold) {
            CB_REQUIRE(seqs[q].nextState && seqs[q].nextSplitState && seqs[q].nextFrameMasks);
            p.seq[q].nstate = seqs[q].nextState, p.seq[q].nS = (char*)seqs[q].nextSplitState;
            p.seq[q].nmasks = (unsigned long long*)seqs[q].nextFrameMasks;
            p.seq[q].nflag = next->arith ? nullptr : seqs[q].nextRangeFlag;      // (bf16 triples have f32's range)
        }
    }
    const long total = (long)p.units * nSeq;
    if (frame) {      // DET: one unit per workgroup, the layer's own detection inside (one sequence, 7x7)
        if (nSeq != 1 || kH != 7) return CB_ERR_UNSUPPORTED;
        hipLaunchKernelGGL((cbp_rowpair_kernel<7, 7, true>), dim3((unsigned)total), dim3(CBP_NT), 0, (hipStream_t)stream, p);
        return cb_launch_status();
    }
    // four workgroups of four waves per CU are resident (<= 128 registers, 32 KB of LDS each); a workgroup's units are
    // worked through one after the other (8 us each, mostly round trips to memory), so up to eight workgroups per CU
    // are started -- one candidate unit each at 480x320: the empty ones are gone after 1.8 us -- and at most
    // CBP_MAXCAND candidates each beyond that
    long grid = 8l * cbp_num_cus();
    {
        static int perCU = -1;
        if (perCU < 0) {
            const char* e = getenv("CBINFER_PAIR_WGS_PER_CU");      // tuning aid
            perCU = e ? atoi(e) : 0;
        }
        if (perCU > 0) grid = (long)perCU * cbp_num_cus();
    }
    if (grid > total) grid = total;
    if (grid * CBP_MAXCAND < total) grid = (total + CBP_MAXCAND - 1) / CBP_MAXCAND;
    if (kH == 7)
        hipLaunchKernelGGL((cbp_rowpair_kernel<7, 7>), dim3((unsigned)grid), dim3(CBP_NT), 0, (hipStream_t)stream, p);
    else if (kH == 5)
        hipLaunchKernelGGL((cbp_rowpair_kernel<5, 5>), dim3((unsigned)grid), dim3(CBP_NT), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL((cbp_rowpair_kernel<3, 3>), dim3((unsigned)grid), dim3(CBP_NT), 0, (hipStream_t)stream, p);
    return cb_launch_status();
}

int cbinfer_conv_changed_rowpairs(const float* state, uint64_t* bits, int32_t* ctl, uint64_t* maskCopy,
                                  const void* prepared, const float* bias, float* output, int C, int H, int W, int K,
                                  int kH, int kW, int relu, const cbNextDetect* next, cbStream_t stream) {
    (void)ctl;      // (was: an arrival counter; every workgroup now zeroes the mask words of its own units)
    cbPairSeq sq;
    sq.state = state, sq.output = output, sq.bits = bits, sq.maskCopy = maskCopy;
    sq.nextState = nullptr, sq.nextSplitState = nullptr, sq.nextFrameMasks = nullptr, sq.nextRangeFlag = nullptr;
    if (next && next->state) {
        CB_REQUIRE(next->splitState && next->frameMasks);
        sq.nextState = next->state, sq.nextSplitState = next->splitState;
        sq.nextFrameMasks = next->frameMasks, sq.nextRangeFlag = next->rangeFlag;
    }
    return cbp_launch(&sq, 1, prepared, bias, C, H, W, K, kH, kW, relu, sq.nextState ? next : nullptr, stream);
}

// nSeq sequences in one launch (own tensors each, shared weights); `next` gives the next layer's geometry and threshold
// (its pointer fields are ignored: the per-sequence ones of cbPairSeq count), NULL or a sequence table without
// nextState: no folding
int cbinfer_conv_changed_rowpairs_batched(const cbPairSeq* seqs, int nSeq, const void* prepared, const float* bias,
                                          int C, int H, int W, int K, int kH, int kW, int relu,
                                          const cbNextDetect* next, cbStream_t stream) {
    return cbp_launch(seqs, nSeq, prepared, bias, C, H, W, K, kH, kW, relu, next, stream);
}

// The feedback refresh of a layer's state (cbconv2d_cg_backend.cu:74-80) as a launch of its own: every pixel of which some
// channel differs from the state by more than the threshold takes the frame's values in all channels.  What
// cbinfer_conv_rowpairs_detect leaves undone -- its workgroups must all see the OLD state --; the split-state contraction
// behind it can carry this loop on its idle workgroups instead (cbSideRefresh).
__global__ __launch_bounds__(256) void cbp_refresh_kernel(const float* __restrict__ frame, float* state, int C, long HW,
                                                          float th) {
    for (long i = blockIdx.x * 256l + threadIdx.x; i < HW; i += (long)gridDim.x * 256) {
        bool chg = false;
        for (int c = 0; c < C; ++c) chg |= cb_changed(state[c * HW + i], frame[c * HW + i], th);
        if (chg)
            for (int c = 0; c < C; ++c) state[c * HW + i] = frame[c * HW + i];
    }
}
int cbinfer_refresh_state(const float* frame, float* state, int C, int H, int W, float threshold, cbStream_t stream) {
    CB_REQUIRE(frame && state && C >= 1 && H >= 1 && W >= 1);
    const long HW = (long)H * W;
    hipLaunchKernelGGL(cbp_refresh_kernel, dim3((unsigned)((HW + 255) / 256)), dim3(256), 0, (hipStream_t)stream, frame, state, C,
                       HW, threshold);
    return cb_launch_status();
}

// cbinfer_cbconv2d_forward_rowpairs WITHOUT its detection launch (round 6): the row-pair kernel detects the layer's changes
// itself, from `input` and the UNTOUCHED prevInput, and multiplies the refreshed values; prevInput is refreshed by the
// caller afterwards -- cbinfer_refresh_state, or the refresh loop of the next layer's contraction.  maskCopy takes the
// frame's dilated change mask (there is no other mask).  One sequence, 7x7.
int cbinfer_conv_rowpairs_detect(const float* input, const float* prevInput, float* prevOutput, uint64_t* maskCopy,
                                 const void* prepared, const float* bias, int C, int H, int W, int K, int kH, int kW,
                                 float threshold, int relu, const cbNextDetect* next, cbStream_t stream) {
    CB_REQUIRE(input && prevInput && prevOutput && maskCopy);
    cbPairSeq sq;
    sq.state = prevInput, sq.output = prevOutput, sq.bits = nullptr, sq.maskCopy = maskCopy;
    sq.nextState = nullptr, sq.nextSplitState = nullptr, sq.nextFrameMasks = nullptr, sq.nextRangeFlag = nullptr;
    if (next && next->state) {
        CB_REQUIRE(next->splitState && next->frameMasks);
        sq.nextState = next->state, sq.nextSplitState = next->splitState;
        sq.nextFrameMasks = next->frameMasks, sq.nextRangeFlag = next->rangeFlag;
    }
    return cbp_launch(&sq, 1, prepared, bias, C, H, W, K, kH, kW, relu, sq.nextState ? next : nullptr, stream, input,
                      threshold);
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
