// Change detection (+dilation, +feedback update), mask dilation and changed-index compaction for
// gfx950.  HBM-bound byte/flag work: coalesced 256-B row segments per wave over the channel axis,
// wave64 ballots instead of per-pixel flags, dilation on 64-bit words, order-preserving compaction by
// popcount prefix.  Reference behaviour restated from cbconv2d_cg_backend.cu:6-136 and
// conv2d_cg.py:200-209; see include/cbinfer_hip.h for the entry-point contracts.
#include "cb_common.h"

namespace {

__device__ __forceinline__ unsigned long long cb_valid_mask(int W, int tile) {
    const int rem = W - tile * 64;
    return rem >= 64 ? ~0ull : ((1ull << rem) - 1ull);
}

// One workgroup = one 64-pixel row segment x all channels.  blockDim.x = 64*G: wave g scans channels
// g, g+G, ... (each wave-load is one coalesced 256-B (fp32) / 128-B (fp16) segment of a channel row),
// the per-wave ballots are OR-reduced through LDS, and the resulting 64-bit word is dilated with
// shifts.  BITS=true : atomicOr of the dilated words into the row-padded bit mask;
// BITS=false: byte stores of 1 into the (pre-zeroed) [H,W] map -- same-value races are benign, as in
// the reference (cbconv2d_cg_backend.cu:69).
// POOL: `in` is the tensor BEFORE a 2x2/stride-2 max pool ([C, pH, pW]); the value compared with the
// state at (c, y, x) is the pooled one, computed on the fly (windows clipped at the border as the pool
// kernel clips them).  The pooled map itself is never written: a feedback-mode layer only needs it in
// its state, which this kernel refreshes at the changed pixels.
// (several sequences in one launch -- batch.nSeq > 1: blockIdx.y = sequence * H + row, input / state / mask from
//  the table)
struct DetBatch {
    int nSeq;
    struct {
        const void* in;
        void* state;
        unsigned long long* bits;
    } seq[CBINFER_SPLIT_MAX_SEQUENCES];
};

template <typename T, bool BITS, bool POOL = false>
__global__ __launch_bounds__(1024) void cb_detect_kernel(const T* __restrict__ in, T* state,
                                                        int8_t* __restrict__ map,
                                                        unsigned long long* __restrict__ bits, int W,
                                                        int H, int C, int kHH, int kWH, float thf,
                                                        int update, int wpr,
                                                        const int* __restrict__ parity, long altWords,
                                                        int pH, int pW,
                                                        const unsigned long long* __restrict__ prodMask,
                                                        const int* __restrict__ upstream, DetBatch batch) {
    cb_touch_kernarg<112 + sizeof(DetBatch)>();
    // `upstream`: the change count the layer that PRODUCED `in` published this frame.  Zero: it rewrote no output
    // pixel, `in` is bit for bit what this layer compared (and, not in feedback mode, copied) last frame, nothing can
    // exceed the threshold -- the launch is over after one scalar load (cbinfer_cbconv2d_forward_after).
    if (upstream && *upstream == 0) return;
    int yb = blockIdx.y;
    if (batch.nSeq > 1) {
        const int q = (int)blockIdx.y / H;
        yb -= q * H;
        in = (const T*)batch.seq[q].in, state = (T*)batch.seq[q].state, bits = batch.seq[q].bits;
    }
    // POOL with the producer's change mask (pre-pool resolution, this frame): a pooled pixel none of whose
    // window pixels was rewritten by the producing layer compares exactly as it did last frame, i.e. not
    // above the threshold -- the 64 pooled pixels of this workgroup lie under four words of that mask
    if (POOL && prodMask) {
        const int pwpr = (pW + 63) >> 6;
        unsigned long long any = 0ull;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int yy = 2 * yb + j, ww = 2 * (int)blockIdx.x + i;
                const unsigned long long v = prodMask[(long)min(yy, pH - 1) * pwpr + min(ww, pwpr - 1)];
                any |= (yy < pH && ww < pwpr) ? v : 0ull;      // (clamped, not predicated: one round trip for the four)
            }
        if (__builtin_amdgcn_readfirstlane((int)(any != 0ull)) == 0) return;
    }
    // frame pipeline: two masks alternate by a device-side parity (flipped by the consumer kernel)
    if (BITS && parity && *parity) bits += altWords;
    const int lane = threadIdx.x & 63;
    const int g = threadIdx.x >> 6;
    const int G = blockDim.x >> 6;
    const int tx = blockIdx.x;
    const int y = yb;
    const int x = tx * 64 + lane;
    const bool valid = x < W;
    const long HW = (long)H * W;
    const long p = (long)y * W + x;
    const T th = cb_threshold(thf, (T*)nullptr);

    // input value of channel c at this lane's pixel
    const long pHW = (long)pH * pW;
    const int py0 = 2 * y, px0 = 2 * x;
    // (four unconditional loads with clamped coordinates -- a window cut off by the map's edge reads a pixel twice,
    //  max(a, a) = a: a per-lane branch around the loads makes the compiler wait for one channel's window before it
    //  requests the next)
    const int px1 = min(px0 + 1, pW - 1) - px0, py1 = (min(py0 + 1, pH - 1) - py0) * pW;
    auto ldin = [&](int c) -> T {
        if (!POOL) return in[(long)c * HW + p];
        const T* q = in + (long)c * pHW + (long)py0 * pW + px0;
        return cb_max(cb_max(q[0], q[px1]), cb_max(q[py1], q[py1 + px1]));
    };

    bool chg = false;
    // (a wave that owns exactly four channels -- C == 4 G, the 16- and 64-channel layers -- keeps their input
    //  values for the state refresh below instead of loading them again)
    const bool keep = C <= 4 * G;
    T k0 = T(0), k1 = T(0), k2 = T(0), k3 = T(0);
    if (valid) {
        int c = g;
        // update == 2 (round 4): the layer is NOT in feedback mode and keeps a copy of its input (conv2d.py:234-236,
        // prevInput.copy_(input)): every value read for the comparison is written to the state right here -- same
        // thread, behind its own read of the old value -- instead of by a full-tensor copy launch behind this one.
        const bool copyAll = update == 2;
#pragma unroll 1
        for (; c + 3 * G < C; c += 4 * G) {  // 8 independent loads in flight per lane
            const T s0 = state[(long)c * HW + p], x0 = ldin(c);
            const T s1 = state[(long)(c + G) * HW + p], x1 = ldin(c + G);
            const T s2 = state[(long)(c + 2 * G) * HW + p], x2 = ldin(c + 2 * G);
            const T s3 = state[(long)(c + 3 * G) * HW + p], x3 = ldin(c + 3 * G);
            chg |= cb_changed(s0, x0, th) | cb_changed(s1, x1, th) | cb_changed(s2, x2, th) |
                   cb_changed(s3, x3, th);
            k0 = x0, k1 = x1, k2 = x2, k3 = x3;
            if (copyAll) {      // (only what differs: behind another change-based layer that is almost nothing)
                if (cb_differs(s0, x0)) state[(long)c * HW + p] = x0;
                if (cb_differs(s1, x1)) state[(long)(c + G) * HW + p] = x1;
                if (cb_differs(s2, x2)) state[(long)(c + 2 * G) * HW + p] = x2;
                if (cb_differs(s3, x3)) state[(long)(c + 3 * G) * HW + p] = x3;
            }
        }
        if (keep && c == g) {   // fewer than four channels for this wave: the same, value by value
            if (c < C) k0 = ldin(c);
            if (c + G < C) k1 = ldin(c + G);
            if (c + 2 * G < C) k2 = ldin(c + 2 * G);
            T t0 = T(0), t1 = T(0), t2 = T(0);
            if (c < C) t0 = state[(long)c * HW + p], chg |= cb_changed(t0, k0, th);
            if (c + G < C) t1 = state[(long)(c + G) * HW + p], chg |= cb_changed(t1, k1, th);
            if (c + 2 * G < C) t2 = state[(long)(c + 2 * G) * HW + p], chg |= cb_changed(t2, k2, th);
            if (copyAll) {
                if (c < C && cb_differs(t0, k0)) state[(long)c * HW + p] = k0;
                if (c + G < C && cb_differs(t1, k1)) state[(long)(c + G) * HW + p] = k1;
                if (c + 2 * G < C && cb_differs(t2, k2)) state[(long)(c + 2 * G) * HW + p] = k2;
            }
        } else {
            for (; c < C; c += G) {
                const T xv = ldin(c), sv = state[(long)c * HW + p];
                chg |= cb_changed(sv, xv, th);
                if (copyAll && cb_differs(sv, xv)) state[(long)c * HW + p] = xv;
            }
        }
    }

    __shared__ unsigned long long sm[16];
    const unsigned long long b = __ballot(chg);
    if (lane == 0) sm[g] = b;
    __syncthreads();
    unsigned long long m = 0;
    for (int i = 0; i < G; ++i) m |= sm[i];
    if (m == 0) return;  // uniform over the workgroup

    // feedback: refresh the state at the (pre-dilation) changed pixels only (.cu:74-80)
    if (update == 1 && ((m >> lane) & 1ull)) {
        if (keep) {
            if (g < C) state[(long)g * HW + p] = k0;
            if (g + G < C) state[(long)(g + G) * HW + p] = k1;
            if (g + 2 * G < C) state[(long)(g + 2 * G) * HW + p] = k2;
            if (g + 3 * G < C) state[(long)(g + 3 * G) * HW + p] = k3;
        } else {
            for (int c = g; c < C; c += G) state[(long)c * HW + p] = ldin(c);
        }
    }

    // horizontal dilation of the 64-pixel word, with the parts spilling into the neighbour words
    unsigned long long D = m, SR = 0, SL = 0;
    for (int d = 1; d <= kWH; ++d) {
        D |= (m << d) | (m >> d);
        SR |= m >> (64 - d);
        SL |= m << (64 - d);
    }
    D &= cb_valid_mask(W, tx);
    SR = (tx + 1 < wpr) ? (SR & cb_valid_mask(W, tx + 1)) : 0ull;
    if (tx == 0) SL = 0;

    if (BITS) {
        if (g == 0) {
            const int items = 3 * (2 * kHH + 1);
            for (int i = lane; i < items; i += 64) {
                const int yy = y + i / 3 - kHH;
                const int which = i % 3;
                if (yy < 0 || yy >= H) continue;
                const unsigned long long v = which == 0 ? D : (which == 1 ? SR : SL);
                const int t2 = which == 0 ? tx : (which == 1 ? tx + 1 : tx - 1);
                if (v) atomicOr(&bits[(long)yy * wpr + t2], v);
            }
        }
    } else {
        for (int r = g; r <= 2 * kHH; r += G) {
            const int yy = y + r - kHH;
            if (yy < 0 || yy >= H) continue;
            int8_t* row = map + (long)yy * W;
            if ((D >> lane) & 1ull) row[x] = 1;
            if (lane < kWH) {
                if ((SR >> lane) & 1ull) row[(tx + 1) * 64 + lane] = 1;
                if ((SL >> (64 - kWH + lane)) & 1ull) row[tx * 64 - kWH + lane] = 1;
            }
        }
    }
}

// Fine-grained frame detection (cbconv2d_fg_backend.cu:7-23 + the any-channel dilated pixel mask the
// contraction needs).  Same decomposition as cb_detect_kernel: one workgroup = one 64-pixel row segment x
// all channels.  Per value d = in - prev; delta = |d| > th ? d : 0 is written for EVERY value (the gather
// of the contraction reads unchanged neighbours too); with refresh != 0 prev <- in wherever they differ at
// all, which is what the reference's `self.prevInput = input` amounts to (conv2d.py:175: sub-threshold
// differences are absorbed, not accumulated).  The pixels any of whose values changed are dilated by the
// filter support and ORed into the frame mask the parity selects.
__global__ __launch_bounds__(1024) void cb_detect_fg_frame_kernel(
    const float* __restrict__ in, float* prev, float* __restrict__ delta,
    unsigned long long* __restrict__ bits, int W, int H, int C, int kHH, int kWH, float th, int refresh,
    int wpr, const int* __restrict__ parity, long altWords) {
    if (parity && *parity) bits += altWords;
    const int lane = threadIdx.x & 63;
    const int g = threadIdx.x >> 6;
    const int G = blockDim.x >> 6;
    const int tx = blockIdx.x;
    const int y = blockIdx.y;
    const int x = tx * 64 + lane;
    const long HW = (long)H * W;
    const long p = (long)y * W + x;
    bool chg = false;
    if (x < W) {
        int c = g;
#pragma unroll 1
        for (; c + 3 * G < C; c += 4 * G) {  // 8 independent loads in flight per lane
            float s[4], v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                s[u] = prev[(long)(c + u * G) * HW + p];
                v[u] = in[(long)(c + u * G) * HW + p];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float d = v[u] - s[u];
                const bool pred = fabsf(d) > th;
                chg |= pred;
                delta[(long)(c + u * G) * HW + p] = pred ? d : 0.f;
                if (refresh && d != 0.f) prev[(long)(c + u * G) * HW + p] = v[u];
            }
        }
        for (; c < C; c += G) {
            const float s0 = prev[(long)c * HW + p], v0 = in[(long)c * HW + p];
            const float d = v0 - s0;
            const bool pred = fabsf(d) > th;
            chg |= pred;
            delta[(long)c * HW + p] = pred ? d : 0.f;
            if (refresh && d != 0.f) prev[(long)c * HW + p] = v0;
        }
    }
    __shared__ unsigned long long sm[16];
    const unsigned long long b = __ballot(chg);
    if (lane == 0) sm[g] = b;
    __syncthreads();
    if (g != 0) return;
    unsigned long long m = 0;
    for (int i = 0; i < G; ++i) m |= sm[i];
    if (m == 0) return;
    unsigned long long D = m, SR = 0, SL = 0;
    for (int d = 1; d <= kWH; ++d) {
        D |= (m << d) | (m >> d);
        SR |= m >> (64 - d);
        SL |= m << (64 - d);
    }
    D &= cb_valid_mask(W, tx);
    SR = (tx + 1 < wpr) ? (SR & cb_valid_mask(W, tx + 1)) : 0ull;
    if (tx == 0) SL = 0;
    const int items = 3 * (2 * kHH + 1);
    for (int i = lane; i < items; i += 64) {
        const int yy = y + i / 3 - kHH;
        const int which = i % 3;
        if (yy < 0 || yy >= H) continue;
        const unsigned long long v = which == 0 ? D : (which == 1 ? SR : SL);
        const int t2 = which == 0 ? tx : (which == 1 ? tx + 1 : tx - 1);
        if (v) atomicOr(&bits[(long)yy * wpr + t2], v);
    }
}

// Stand-alone gather-form dilation of a byte map (cbconv2d_cg_backend.cu:101-124).
__global__ __launch_bounds__(256) void cb_propagate_kernel(const int8_t* __restrict__ in,
                                                          int8_t* __restrict__ out, int W, int H,
                                                          int kHH, int kWH) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (long)H * W) return;
    const int xo = (int)(p % W), yo = (int)(p / W);
    bool change = false;
    for (int k = -kHH; k <= kHH; ++k) {
        const int yi = yo + k;
        if (yi < 0 || yi >= H) continue;
        for (int l = -kWH; l <= kWH; ++l) {
            const int xi = xo + l;
            if (xi >= 0 && xi < W) change |= in[(long)yi * W + xi] != 0;
        }
    }
    out[p] = change;
}

// bytes -> flat bit words (one ballot per 64 map bytes)
__global__ __launch_bounds__(256) void cb_bytes_to_bits_kernel(const int8_t* __restrict__ map,
                                                              long numel,
                                                              unsigned long long* __restrict__ words) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool v = i < numel && map[i] != 0;
    const unsigned long long b = __ballot(v);
    if ((threadIdx.x & 63) == 0 && (i >> 6) < ((numel + 63) >> 6)) words[i >> 6] = b;
}

__device__ __forceinline__ int cb_wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Order-preserving compaction of a bit mask into ascending pixel indices.  Each workgroup owns
// CB_CW consecutive words; its output offset is the popcount of every word before them, which it
// recomputes itself from the (L2-resident, <=32 KB) mask -- no inter-workgroup hand-off, no second
// launch.  Inside the workgroup a wave turns one word into indices with ballot-free rank
// arithmetic: rank(lane) = popcount(word & lanemask_lt).
#define CB_CW 64
__global__ __launch_bounds__(256) void cb_compact_kernel(
    const unsigned long long* __restrict__ bits, long nWords, int wpr, int W, int H,
    int32_t* __restrict__ idx, int32_t* __restrict__ count, unsigned long long* __restrict__ clearBits,
    int8_t* __restrict__ mapOut) {
    const int lane = threadIdx.x & 63;
    const int g = threadIdx.x >> 6;
    const long w0 = (long)blockIdx.x * CB_CW;

    __shared__ int red[4];
    __shared__ unsigned long long sw[CB_CW];
    __shared__ int soff[CB_CW + 1];

    // popcount of every earlier word: independent loads, 8 in flight per thread
    int part = 0;
    {
        long i = threadIdx.x;
        for (; i + 7 * 256 < w0; i += 8 * 256) {
            unsigned long long w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = bits[i + u * 256];
#pragma unroll
            for (int u = 0; u < 8; ++u) part += __popcll(w[u]);
        }
        for (; i < w0; i += 256) part += __popcll(bits[i]);
    }
    part = cb_wave_sum(part);
    if (lane == 0) red[g] = part;

    if (g == 0) {
        const long wi = w0 + lane;
        const unsigned long long word = wi < nWords ? bits[wi] : 0ull;
        const int pc = __popcll(word);
        int incl = pc;  // inclusive scan over the 64 lanes
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        sw[lane] = word;
        soff[lane] = incl - pc;
        if (lane == 63) soff[CB_CW] = incl;
    }
    __syncthreads();
    const int base = red[0] + red[1] + red[2] + red[3];

    for (int j = g; j < CB_CW; j += 4) {
        const long wi = w0 + j;
        if (wi >= nWords) break;
        const unsigned long long word = sw[j];
        const int row = (int)(wi / wpr), tile = (int)(wi % wpr);
        const int x = tile * 64 + lane;
        const bool bit = (word >> lane) & 1ull;
        if (bit) {
            const int rank = __popcll(word & ((1ull << lane) - 1ull));
            idx[base + soff[j] + rank] = row * W + x;
        }
        if (mapOut && x < W) mapOut[(long)row * W + x] = bit;
        if (clearBits && lane == 0) clearBits[wi] = 0ull;
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) count[0] = base + soff[CB_CW];
}

int detect_groups(int C) {
    int G = C / 4;   // waves per 64-pixel row segment (C/8 measured 2 % slower per frame: too few loads in flight)
    if (G < 1) G = 1;
    if (G > 16) G = 16;
    return G;
}

template <typename T, bool BITS, bool POOL = false>
int launch_detect(const void* input, void* state, int8_t* map, uint64_t* bits, int W, int H, int C,
                  int kHH, int kWH, float th, int update, hipStream_t s, const int* parity = nullptr,
                  long altWords = 0, int pH = 0, int pW = 0, const uint64_t* prodMask = nullptr,
                  const DetBatch* batch = nullptr, const int* upstream = nullptr) {
    const int wpr = cbinfer_mask_words_per_row(W);
    const int G = detect_groups(C);
    DetBatch b;
    b.nSeq = 1;
    if (batch) b = *batch;
    if ((long)H * b.nSeq > 65535) return CB_ERR_UNSUPPORTED;
    dim3 grid(wpr, H * b.nSeq), block(64 * G);
    hipLaunchKernelGGL((cb_detect_kernel<T, BITS, POOL>), grid, block, 0, s, (const T*)input, (T*)state,
                       map, (unsigned long long*)bits, W, H, C, kHH, kWH, th, update, wpr, parity, altWords,
                       pH, pW, (const unsigned long long*)prodMask, upstream, b);
    return cb_launch_status();
}

}  // namespace

extern "C" {

int cbinfer_mask_words_per_row(int W) { return (W + 63) / 64; }
long cbinfer_mask_words(int H, int W) { return (long)H * ((W + 63) / 64); }

int cbinfer_change_detection(const void* input, void* state, int8_t* changeMap, int W, int H, int C,
                             int kHHalf, int kWHalf, float threshold, int updateInputState,
                             int dtype, cbStream_t stream) {
    CB_REQUIRE(input && state && changeMap && W > 0 && H > 0 && C > 0 && kHHalf >= 0 && kWHalf >= 0);
    if (kWHalf > 63 || H > 65535) return CB_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(changeMap, 0, (size_t)H * W, s);
    if (e != hipSuccess) return (int)e;
    if (dtype == CB_F32)
        return launch_detect<float, false>(input, state, changeMap, nullptr, W, H, C, kHHalf, kWHalf,
                                           threshold, updateInputState, s);
    if (dtype == CB_F16)
        return launch_detect<cb_half, false>(input, state, changeMap, nullptr, W, H, C, kHHalf, kWHalf,
                                             threshold, updateInputState, s);
    return CB_ERR_BADARG;
}

int cbinfer_change_detection_bits(const void* input, void* state, uint64_t* bitsOut, int W, int H,
                                  int C, int kHHalf, int kWHalf, float threshold,
                                  int updateInputState, int dtype, cbStream_t stream) {
    CB_REQUIRE(input && state && bitsOut && W > 0 && H > 0 && C > 0 && kHHalf >= 0 && kWHalf >= 0);
    if (kWHalf > 63 || H > 65535) return CB_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == CB_F32)
        return launch_detect<float, true>(input, state, nullptr, bitsOut, W, H, C, kHHalf, kWHalf,
                                          threshold, updateInputState, s);
    if (dtype == CB_F16)
        return launch_detect<cb_half, true>(input, state, nullptr, bitsOut, W, H, C, kHHalf, kWHalf,
                                            threshold, updateInputState, s);
    return CB_ERR_BADARG;
}

// cbinfer_change_detection_bits (fp32) for nSeq sequences in one launch: inputs[q] / states[q] / bitsOut[q]
int cbinfer_change_detection_bits_batched(const float* const* inputs, float* const* states, uint64_t* const* bitsOut,
                                          int nSeq, int W, int H, int C, int kHHalf, int kWHalf, float threshold,
                                          int updateInputState, cbStream_t stream) {
    CB_REQUIRE(inputs && states && bitsOut && nSeq >= 1 && nSeq <= CBINFER_SPLIT_MAX_SEQUENCES && W > 0 && H > 0 &&
               C > 0 && kHHalf >= 0 && kWHalf >= 0);
    if (kWHalf > 63) return CB_ERR_UNSUPPORTED;
    DetBatch b;
    b.nSeq = nSeq;
    for (int q = 0; q < nSeq; ++q) {
        CB_REQUIRE(inputs[q] && states[q] && bitsOut[q]);
        b.seq[q].in = inputs[q], b.seq[q].state = states[q], b.seq[q].bits = (unsigned long long*)bitsOut[q];
    }
    return launch_detect<float, true>(inputs[0], states[0], nullptr, bitsOut[0], W, H, C, kHHalf, kWHalf, threshold,
                                      updateInputState, (hipStream_t)stream, nullptr, 0, 0, 0, nullptr, &b);
}

// Single-mask form of the pooled detection (see cbinfer_change_detection_frame_pooled): ORs into bitsOut.
int cbinfer_change_detection_bits_pooled(const void* prePool, int pH, int pW, const uint64_t* producerMask,
                                         void* state, uint64_t* bitsOut, int W, int H, int C, int kHHalf,
                                         int kWHalf, float threshold, int dtype, cbStream_t stream) {
    CB_REQUIRE(prePool && state && bitsOut && W > 0 && H > 0 && C > 0 && kHHalf >= 0 && kWHalf >= 0);
    CB_REQUIRE((H == pH / 2 || H == (pH + 1) / 2) && (W == pW / 2 || W == (pW + 1) / 2));
    if (kWHalf > 63 || H > 65535) return CB_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == CB_F32)
        return launch_detect<float, true, true>(prePool, state, nullptr, bitsOut, W, H, C, kHHalf, kWHalf,
                                                threshold, 1, s, nullptr, 0, pH, pW, producerMask);
    if (dtype == CB_F16)
        return launch_detect<cb_half, true, true>(prePool, state, nullptr, bitsOut, W, H, C, kHHalf, kWHalf,
                                                  threshold, 1, s, nullptr, 0, pH, pW, producerMask);
    return CB_ERR_BADARG;
}

// Frame-pipeline form: frameMasks = [2][words] masks followed by {parity, done}; the detection ORs into
// the mask the parity selects (cbinfer_conv_changed_from_mask consumes it and flips the parity).
// upstreamCount (optional, device): see cb_detect_kernel -- a zero there ends the launch at once.
int cbinfer_change_detection_frame_after(const int32_t* upstreamCount, const void* input, void* state,
                                         uint64_t* frameMasks, int W, int H, int C, int kHHalf, int kWHalf,
                                         float threshold, int updateInputState, int dtype, cbStream_t stream) {
    CB_REQUIRE(input && state && frameMasks && W > 0 && H > 0 && C > 0 && kHHalf >= 0 && kWHalf >= 0);
    if (kWHalf > 63 || H > 65535) return CB_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const long words = cbinfer_mask_words(H, W);
    const int* parity = (const int*)(frameMasks + 2 * words);
    if (dtype == CB_F32)
        return launch_detect<float, true>(input, state, nullptr, frameMasks, W, H, C, kHHalf, kWHalf,
                                          threshold, updateInputState, s, parity, words, 0, 0, nullptr, nullptr,
                                          upstreamCount);
    if (dtype == CB_F16)
        return launch_detect<cb_half, true>(input, state, nullptr, frameMasks, W, H, C, kHHalf, kWHalf,
                                            threshold, updateInputState, s, parity, words, 0, 0, nullptr, nullptr,
                                            upstreamCount);
    return CB_ERR_BADARG;
}

int cbinfer_change_detection_frame(const void* input, void* state, uint64_t* frameMasks, int W, int H,
                                   int C, int kHHalf, int kWHalf, float threshold,
                                   int updateInputState, int dtype, cbStream_t stream) {
    return cbinfer_change_detection_frame_after(nullptr, input, state, frameMasks, W, H, C, kHHalf, kWHalf, threshold,
                                                updateInputState, dtype, stream);
}

// The same with the 2x2/stride-2 max pool in front of the layer folded in: `prePool` is [C, pH, pW],
// H x W the pooled size (floor: pH/2, ceil: (pH+1)/2).  The state is always refreshed (feedback mode).
int cbinfer_change_detection_frame_pooled(const void* prePool, int pH, int pW, void* state,
                                          uint64_t* frameMasks, int W, int H, int C, int kHHalf,
                                          int kWHalf, float threshold, int dtype, cbStream_t stream) {
    CB_REQUIRE(prePool && state && frameMasks && W > 0 && H > 0 && C > 0 && kHHalf >= 0 && kWHalf >= 0);
    CB_REQUIRE((H == pH / 2 || H == (pH + 1) / 2) && (W == pW / 2 || W == (pW + 1) / 2));
    if (kWHalf > 63 || H > 65535) return CB_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const long words = cbinfer_mask_words(H, W);
    const int* parity = (const int*)(frameMasks + 2 * words);
    if (dtype == CB_F32)
        return launch_detect<float, true, true>(prePool, state, nullptr, frameMasks, W, H, C, kHHalf,
                                                kWHalf, threshold, 1, s, parity, words, pH, pW);
    if (dtype == CB_F16)
        return launch_detect<cb_half, true, true>(prePool, state, nullptr, frameMasks, W, H, C, kHHalf,
                                                  kWHalf, threshold, 1, s, parity, words, pH, pW);
    return CB_ERR_BADARG;
}

// Fine-grained frame detection (see cb_detect_fg_frame_kernel); frameMasks as in
// cbinfer_change_detection_frame.
int cbinfer_change_detection_fg_frame(const float* input, float* prevInput, float* delta,
                                      uint64_t* frameMasks, int W, int H, int C, int kHHalf, int kWHalf,
                                      float threshold, int refreshState, cbStream_t stream) {
    CB_REQUIRE(input && prevInput && delta && frameMasks && W > 0 && H > 0 && C > 0 && kHHalf >= 0 &&
               kWHalf >= 0);
    if (kWHalf > 63 || H > 65535) return CB_ERR_UNSUPPORTED;
    const long words = cbinfer_mask_words(H, W);
    const int wpr = cbinfer_mask_words_per_row(W);
    dim3 grid(wpr, H), block(64 * detect_groups(C));
    hipLaunchKernelGGL(cb_detect_fg_frame_kernel, grid, block, 0, (hipStream_t)stream, input, prevInput,
                       delta, (unsigned long long*)frameMasks, W, H, C, kHHalf, kWHalf, threshold,
                       refreshState, wpr, (const int*)(frameMasks + 2 * words), words);
    return cb_launch_status();
}

int cbinfer_change_detection_fg_bits(const float* input, float* prevInput, float* delta, uint64_t* bits, int W,
                                     int H, int C, int kHHalf, int kWHalf, float threshold, int refreshState,
                                     cbStream_t stream) {
    CB_REQUIRE(input && prevInput && delta && bits && W > 0 && H > 0 && C > 0 && kHHalf >= 0 && kWHalf >= 0);
    if (kWHalf > 63 || H > 65535) return CB_ERR_UNSUPPORTED;
    const int wpr = cbinfer_mask_words_per_row(W);
    dim3 grid(wpr, H), block(64 * detect_groups(C));
    hipLaunchKernelGGL(cb_detect_fg_frame_kernel, grid, block, 0, (hipStream_t)stream, input, prevInput,
                       delta, (unsigned long long*)bits, W, H, C, kHHalf, kWHalf, threshold, refreshState, wpr,
                       (const int*)nullptr, 0l);
    return cb_launch_status();
}

int cbinfer_change_propagation(const int8_t* mapIn, int8_t* mapOut, int W, int H, int kHHalf,
                               int kWHalf, cbStream_t stream) {
    CB_REQUIRE(mapIn && mapOut && W > 0 && H > 0 && kHHalf >= 0 && kWHalf >= 0);
    const long n = (long)H * W;
    hipLaunchKernelGGL(cb_propagate_kernel, dim3(cb_div_up(n, 256)), dim3(256), 0,
                       (hipStream_t)stream, mapIn, mapOut, W, H, kHHalf, kWHalf);
    return cb_launch_status();
}

int cbinfer_change_indexes_extr(const int8_t* changeMap, long numel, uint64_t* scratchWords,
                                int32_t* idxOut, int32_t* countDev, cbStream_t stream) {
    CB_REQUIRE(changeMap && scratchWords && idxOut && countDev && numel > 0 && numel < (1l << 31));
    hipStream_t s = (hipStream_t)stream;
    const long nWords = (numel + 63) / 64;
    hipLaunchKernelGGL(cb_bytes_to_bits_kernel, dim3(cb_div_up(nWords * 64, 256)), dim3(256), 0, s,
                       changeMap, numel, (unsigned long long*)scratchWords);
    // flat layout: a single "row" of numel pixels
    hipLaunchKernelGGL(cb_compact_kernel, dim3(cb_div_up(nWords, CB_CW)), dim3(256), 0, s,
                       (const unsigned long long*)scratchWords, nWords, (int)nWords, (int)numel, 1,
                       idxOut, countDev, (unsigned long long*)nullptr, (int8_t*)nullptr);
    return cb_launch_status();
}

int cbinfer_compact_bits(const uint64_t* bits, int W, int H, int32_t* idxOut, int32_t* countDev,
                         uint64_t* clearBits, int8_t* mapOut, cbStream_t stream) {
    CB_REQUIRE(bits && idxOut && countDev && W > 0 && H > 0);
    CB_REQUIRE((const void*)clearBits != (const void*)bits);
    const int wpr = cbinfer_mask_words_per_row(W);
    const long nWords = (long)H * wpr;
    hipLaunchKernelGGL(cb_compact_kernel, dim3(cb_div_up(nWords, CB_CW)), dim3(256), 0,
                       (hipStream_t)stream, (const unsigned long long*)bits, nWords, wpr, W, H, idxOut,
                       countDev, (unsigned long long*)clearBits, mapOut);
    return cb_launch_status();
}

}  // extern "C"
