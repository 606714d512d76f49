// Micro-benchmark for an 8-wave, 256-register form of the bf16x3 list contraction: 256 channels x 64 pixels per
// workgroup, every wave a 64 x 64 output tile (four 32x32 accumulators), weight fragments straight from global
// memory (fragment order, prefetched one stage ahead), the pixel operand through LDS (gathered, split and stored
// by the four k-group 0 waves only).  One barrier per stage.  What does a stage cost against its 0.73 us of MFMA?
// Build:  hipcc -O3 --offload-arch=gfx950 -o tools/micro/direct tools/micro/direct.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sub(float a, float b) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) void k(float* out, const float* in,
                                                                             const u32x4* wts, int stages, int mode) {
    __shared__ __attribute__((aligned(16))) char lds[2 * 64 * 208];
    const int t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
    const int ks = wave >> 2, wm = wave & 3;
    floatx16 a00 = {}, a01 = {}, a10 = {}, a11 = {};
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)wts, 0, 1 << 30, 0x00020000);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, 1 << 28, 0x00020000);
    // weight fragments of this wave: 2 row tiles x 3 planes x 16 B per lane and stage = 6 KB per wave
    const int woff = ((wm * 2) * 2 + ks) * 3 * 1024 + lane * 16;   // + stage * (8 row tiles * 2 * 3 KB)
    u32x4 w[6], wn[6];
    auto loadw = [&](u32x4 (&d)[6], int s) {
#pragma unroll
        for (int i = 0; i < 6; ++i)
            d[i] = __builtin_amdgcn_raw_buffer_load_b128(wr, woff + (i / 3) * 6144 + (i % 3) * 1024, (s & 63) * 49152, 0);
    };
    loadw(w, 0);
    const int px = t & 63, kq = (t >> 6) & 3;   // k-group 0 waves: pixel px, k 8 kq .. 8 kq + 7 of the stage
    float g[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) g[i] = 1.0f + i;
    const int pbase = (blockIdx.x * 64 + px) * 4;
    for (int s = 0; s < stages; ++s) {
        const char* bp = lds + (s & 1) * 64 * 208 + (lane & 31) * 208 + ks * 32 + (lane >> 5) * 16;
        if (mode & 1) loadw(wn, s + 1);
        if (ks == 0 && (mode & 2)) {
            // split the 8 gathered values, store into the other buffer, gather the next 8
            char* row = lds + ((s + 1) & 1) * 64 * 208 + px * 208 + kq * 16;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                unsigned u[4], v[4];
                float r1[4], r2[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    u[e] = __builtin_bit_cast(unsigned, g[4 * q + e]);
                    r1[e] = sub(g[4 * q + e], __builtin_bit_cast(float, u[e] & 0xffff0000u));
                    v[e] = __builtin_bit_cast(unsigned, r1[e]);
                    r2[e] = sub(r1[e], __builtin_bit_cast(float, v[e] & 0xffff0000u));
                }
                *(uint2*)(row + q * 8) = make_uint2(__builtin_amdgcn_perm(u[1], u[0], 0x07060302u),
                                                    __builtin_amdgcn_perm(u[3], u[2], 0x07060302u));
                *(uint2*)(row + 64 + q * 8) = make_uint2(__builtin_amdgcn_perm(v[1], v[0], 0x07060302u),
                                                         __builtin_amdgcn_perm(v[3], v[2], 0x07060302u));
                *(uint2*)(row + 128 + q * 8) =
                    make_uint2(__builtin_bit_cast(unsigned, r2[0]) >> 16 | (__builtin_bit_cast(unsigned, r2[1]) & 0xffff0000u),
                               __builtin_bit_cast(unsigned, r2[2]) >> 16 | (__builtin_bit_cast(unsigned, r2[3]) & 0xffff0000u));
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
                g[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, pbase, ((s * 8 + i) & 1023) * 38400, 0));
        }
        if (mode & 4) {
            const bf16x8 bh0 = *(const bf16x8*)bp, bm0 = *(const bf16x8*)(bp + 64), bl0 = *(const bf16x8*)(bp + 128);
            const bf16x8 bh1 = *(const bf16x8*)(bp + 32 * 208), bm1 = *(const bf16x8*)(bp + 32 * 208 + 64),
                         bl1 = *(const bf16x8*)(bp + 32 * 208 + 128);
            __builtin_amdgcn_sched_barrier(0);
#define SIX(ACC, WH, WM, WL, BH, BM, BL)                                                       \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, WL), BH, ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, WH), BL, ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, WM), BM, ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, WM), BH, ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, WH), BM, ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, WH), BH, ACC, 0, 0, 0);
            SIX(a00, w[0], w[1], w[2], bh0, bm0, bl0)
            SIX(a01, w[0], w[1], w[2], bh1, bm1, bl1)
            SIX(a10, w[3], w[4], w[5], bh0, bm0, bl0)
            SIX(a11, w[3], w[4], w[5], bh1, bm1, bl1)
            __builtin_amdgcn_sched_barrier(0);
        }
        if (mode & 1) {
#pragma unroll
            for (int i = 0; i < 6; ++i) w[i] = wn[i];
        }
        __syncthreads();
    }
    float r = g[0] + g[7];
    for (int i = 0; i < 16; ++i) r += a00[i] + a01[i] + a10[i] + a11[i];
    out[blockIdx.x * 512 + t] = r + (float)w[0].x;
}
int main() {
    float *in, *out;
    u32x4* w;
    hipMalloc(&in, 256 << 20);
    hipMalloc(&out, 4 << 20);
    hipMalloc(&w, 64 << 20);
    hipMemset(in, 0, 256 << 20);
    hipMemset(w, 0, 64 << 20);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int stages = 1000;
    for (int mode : {4, 5, 6, 7, 3, 1, 2}) {
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, in, w, stages, mode);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("mode %d (%s%s%s): %.3f us per stage\n", mode, mode & 4 ? "MFMA+Bfrag " : "", mode & 1 ? "Wdirect " : "",
               mode & 2 ? "Bwork" : "", best * 1e3 / stages);
    }
    return 0;
}
