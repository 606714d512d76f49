// Micro-benchmark: do a workgroup's MFMA waves and its VALU/LDS/VMEM waves overlap on a CU?
// Build and run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -o tools/micro/overlap tools/micro/overlap.hip
// 1024-thread workgroups, one per CU, one barrier per iteration.  mode bit 0: waves 8-15 run an MFMA chain of
// 12 v_mfma_f32_32x32x16_bf16 after 9 ds_read_b128; bit 1: waves 0-7 run ~60 VALU + 4 LDS stores + 6 loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4))) void k(float* out, const float* in, int iters, int mode,
                                                                               int allwaves) {
    __shared__ __attribute__((aligned(16))) float lds[24576];
    const int t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
    floatx16 acc = {}, acc1 = {};
    float v0 = in[t], v1 = in[t + 1024], v2 = in[t + 2048], v3 = in[t + 3072];
    const bool mf = allwaves ? true : wave >= 8;
    const bool st = allwaves ? true : wave < 8;
    const float* src = in + (blockIdx.x * 1024 + t) * 4;
    for (int i = 0; i < iters; ++i) {
        if (mf && (mode & 1)) {
            const char* bp = (const char*)lds + lane * 208;
            const bf16x8 a0 = *(const bf16x8*)bp, a1 = *(const bf16x8*)(bp + 64), a2 = *(const bf16x8*)(bp + 128);
            const bf16x8 b0 = *(const bf16x8*)(bp + 13312), b1 = *(const bf16x8*)(bp + 13376), b2 = *(const bf16x8*)(bp + 13440);
            const bf16x8 c0 = *(const bf16x8*)(bp + 26624), c1 = *(const bf16x8*)(bp + 26688), c2 = *(const bf16x8*)(bp + 26752);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c0, b0, acc1, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c1, b1, acc1, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c2, b2, acc1, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (st && (mode & 2)) {
            // ~60 dependent-ish VALU ops on four values, as the bf16x3 split does
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                v0 = v0 * 1.0001f + v1;
                v1 = v1 * 0.9999f - v2;
                v2 = v2 * 1.0002f + v3;
                v3 = v3 * 0.9998f - v0;
                v0 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v0) & 0xffffff00u);
                v1 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v1) & 0xffffff00u);
                v2 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v2) & 0xffffff00u);
                v3 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v3) & 0xffffff00u);
                v0 += v2;
                v1 += v3;
                v2 -= v1;
                v3 -= v0;
            }
            if (mode & 4) {   // LDS stores: two 16-byte + three 8-byte per thread
                *(float4*)(lds + 12288 + t * 4) = make_float4(v0, v1, v2, v3);
                *(float4*)(lds + 16384 + t * 4) = make_float4(v1, v2, v3, v0);
                *(float2*)(lds + 20480 + (t & 511) * 2) = make_float2(v0, v1);
                *(float2*)(lds + 21504 + (t & 511) * 2) = make_float2(v2, v3);
                *(float2*)(lds + 22528 + (t & 511) * 2) = make_float2(v1, v3);
            }
            if (mode & 8) {   // global loads: two 16-byte + four 4-byte per thread (L2 resident)
                const float4 g0 = *(const float4*)(src + ((i & 7) << 14));
                const float4 g1 = *(const float4*)(src + ((i & 7) << 14) + 4096 * 4);
                v0 += g0.x + g1.y;
                v1 += in[(t * 7 + i) & 65535];
                v2 += in[(t * 13 + i) & 65535];
                v3 += in[(t * 29 + i) & 65535] + in[(t * 31 + i) & 65535];
            }
        }
        __syncthreads();
    }
    float s = v0 + v1 + v2 + v3;
    for (int r = 0; r < 16; ++r) s += acc[r] + acc1[r];
    out[blockIdx.x * 1024 + t] = s;
}
int main() {
    float *in, *out;
    hipMalloc(&in, 64 << 20);
    hipMalloc(&out, 4 << 20);
    hipMemset(in, 0, 64 << 20);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 2000;
    for (int allw = 0; allw < 2; ++allw)
        for (int mode : {1, 2, 3, 6, 7, 10, 11, 14, 15}) {
            float best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, out, in, iters, mode, allw);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("%s mode %2d (%s%s%s%s): %.3f us per iteration\n", allw ? "all waves do both   " : "waves 8-15 MFMA, 0-7 rest",
                   mode, mode & 1 ? "MFMA " : "", mode & 2 ? "VALU " : "", mode & 4 ? "LDSst " : "", mode & 8 ? "VMEM" : "",
                   best * 1e3 / iters);
        }
    return 0;
}
