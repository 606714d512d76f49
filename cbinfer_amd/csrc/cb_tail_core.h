// The arithmetic of the fused 1x1 tail (cb_tail.hip) for ONE tile of CB_TAIL_PX changed pixels whose input columns
// are (being) staged in LDS -- shared by cb_tail1x1_kernel, which gathers them from the producing layer's output,
// and cbs_reduce_tail_kernel (cb_split.hip), which makes them from the partial tiles of a split contraction: one
// body, so both give the same bits.
#pragma once

typedef float cb_tail_floatx4 __attribute__((ext_vector_type(4)));

#ifndef CB_TAIL_STAMP
#define CB_TAIL_STAMP(i)
#endif

#define CB_TAIL_PX 16
#define CB_TAIL_MAXW 8          // waves per workgroup = ceil(C1/16) <= 8  -> C1 <= 128

// dynamic LDS of a tail workgroup: Xs[C0P][16] | Hs[16 waves][17] | W2s[C2][C1] | b2s[C2]
struct CbTailLds {
    float *Xs, *Hs, *W2s, *b2s;
};
__device__ __forceinline__ CbTailLds cb_tail_lds(float* sm, int C0P, int C1, int C2, int NT) {
    CbTailLds l;
    l.Xs = sm;
    l.Hs = sm + (long)C0P * CB_TAIL_PX;
    l.W2s = l.Hs + (NT >> 6) * 16 * (CB_TAIL_PX + 1);
    l.b2s = l.W2s + C2 * C1;
    return l;
}
__host__ __device__ inline size_t cb_tail_lds_bytes(int C0, int C1, int C2) {
    const int C0P = (C0 + 15) / 16 * 16;
    const int waves = (C1 + 15) / 16;
    return ((size_t)C0P * CB_TAIL_PX + (size_t)waves * 16 * (CB_TAIL_PX + 1) + (size_t)C2 * C1 + C2) * 4;
}

// What a wave needs from memory for every tile it evaluates, the same for all of them: the first 16 fragment groups
// of its 16 rows of W1 (all of them up to 256 input channels) and its rows' biases.  Loaded ONCE per workgroup, at
// the top of the kernel, so that the request travels together with everything else the kernel asks for first.
struct CbTailPre {
    cb_tail_floatx4 a[16];
    float b1v[4];
};
__device__ __forceinline__ void cb_tail_preload(CbTailPre& P, const float* __restrict__ w1p,
                                                const float* __restrict__ b1, int C0P, int C1) {
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int groups = C0P / 16;
    const cb_tail_floatx4* ap = (const cb_tail_floatx4*)w1p + ((long)wave * groups) * 64 + lane;
#pragma unroll
    for (int i = 0; i < 16; ++i) P.a[i] = ap[(long)min(i, groups - 1) * 64];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = 16 * wave + 4 * (lane >> 4) + r;
        P.b1v[r] = m < C1 ? b1[m] : 0.f;
    }
}

// Called by every thread of the workgroup once its part of the X tile is WRITTEN to l.Xs (the barrier that makes the
// tile complete is inside).  s_pix[px]: pixel index or -1.
__device__ __forceinline__ void cb_tail_tile(const CbTailLds& l, const int* s_pix, const CbTailPre& P,
                                             const float* __restrict__ w1p, float* out, int C0P, int C1, int C2, int HW,
                                             int relu1, int relu2) {
    typedef cb_tail_floatx4 floatx4;
    const int t = threadIdx.x, NT = blockDim.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    float* Xs = l.Xs;
    float* Hs = l.Hs;
    const float* W2s = l.W2s;
    const float* b2s = l.b2s;
    // H tile of this wave: rows 16 wave .. +15, cols = 16 px.  A: lane holds W1[16w + l%16][4s + l/16],
    // B: lane holds X[4s + l/16][l%16].  (Fetched one by one in the MFMA loop the weight fragments cost sixteen
    // L2 round trips per tile: 15.8 us per launch in the frame, round 2.)
    // Four accumulation chains (fragment group i feeds chain i % 4), summed at the end: a single chain of 4 C0P/16
    // dependent matrix instructions, each waiting for its predecessor and for an LDS read, took 2 us of a 12 us launch.
    floatx4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = floatx4{0.f, 0.f, 0.f, 0.f};
    const floatx4* ap = (const floatx4*)w1p + ((long)wave * (C0P / 16)) * 64 + lane;
    const float* bp = Xs + (lane >> 4) * CB_TAIL_PX + (lane & 15);
    const int groups = C0P / 16;
    __syncthreads();   // the X tile is complete
    CB_TAIL_STAMP(4);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (i < groups) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(P.a[i][j], bp[(16 * i + 4 * j) * CB_TAIL_PX], acc[i & 3],
                                                                  0, 0, 0);
        }
        if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // (at most 16 LDS operands requested ahead)
    }
    for (int g0 = 16; g0 < groups; g0 += 16) {      // (more than 256 input channels)
        floatx4 a[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = ap[(long)min(g0 + i, groups - 1) * 64];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (g0 + i < groups) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][j], bp[(16 * (g0 + i) + 4 * j) * CB_TAIL_PX],
                                                                      acc[i & 3], 0, 0, 0);
            }
        }
    }
    const floatx4 accs = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    // C/D map of the 16x16 tile: col = lane%16, row = 4*(lane/16) + r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = 16 * wave + 4 * (lane >> 4) + r;
        float v = accs[r] + P.b1v[r];
        if (relu1) v = v <= 0.f ? 0.f : v;
        Hs[m * (CB_TAIL_PX + 1) + (lane & 15)] = v;
    }
    CB_TAIL_STAMP(5);
    __syncthreads();
    CB_TAIL_STAMP(6);
    // second layer: out[c2][px] = b2[c2] + sum_j W2[c2][j] * H[j][px].  With threads to spare the hidden channels are
    // dealt over PARTS of them (part k: the k-th quarter, ...), the partial sums meet in LDS (the X tile's space: its
    // last reader was the first layer) and are added in part order.
    const int outs = C2 * CB_TAIL_PX;
    int parts = 1;
    while (parts < 4 && 2 * parts * outs <= NT && 2 * parts * outs <= C0P * CB_TAIL_PX && C1 % (2 * parts) == 0) parts *= 2;
    float* Ps = Xs;
    const int o = t % outs, part = t / outs;
    const int c2 = o >> 4, p2 = o & 15;
    if (part < parts) {
        const int j0 = part * (C1 / parts), j1 = j0 + C1 / parts;
        float v = part == 0 ? b2s[c2] : 0.f;
        const float* wr = W2s + c2 * C1;
#pragma unroll 8
        for (int j = j0; j < j1; ++j) v = fmaf(wr[j], Hs[j * (CB_TAIL_PX + 1) + p2], v);
        if (parts > 1) Ps[part * outs + o] = v;
        if (parts == 1) {
            const int pix = s_pix[p2];
            if (relu2) v = v <= 0.f ? 0.f : v;
            if (pix >= 0) out[(long)c2 * HW + pix] = v;
        }
    }
    if (parts > 1) {
        __syncthreads();
        if (t < outs) {
            float v = Ps[o];
            for (int k = 1; k < parts; ++k) v += Ps[k * outs + o];
            const int pix = s_pix[p2];
            if (relu2) v = v <= 0.f ? 0.f : v;
            if (pix >= 0) out[(long)c2 * HW + pix] = v;
        }
    }
    for (int o2 = t + NT; o2 < outs; o2 += NT) {      // (more outputs than threads: the plain form for the rest)
        const int c3 = o2 >> 4, p3 = o2 & 15;
        const int pix = s_pix[p3];
        if (pix < 0) continue;
        float v = b2s[c3];
        const float* wr = W2s + c3 * C1;
#pragma unroll 8
        for (int j = 0; j < C1; ++j) v = fmaf(wr[j], Hs[j * (CB_TAIL_PX + 1) + p3], v);
        if (relu2) v = v <= 0.f ? 0.f : v;
        out[(long)c3 * HW + pix] = v;
    }
}
