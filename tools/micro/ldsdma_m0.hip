// Micro-test (round 3): back-to-back LDS-DMA instructions (buffer_load_dwordx4 ... lds) whose LDS address register
// M0 is rewritten between them, under memory-system load from other kernels.  Variant A rewrites M0 before every
// DMA (what hipcc emits for __builtin_amdgcn_raw_ptr_buffer_load_lds with different LDS pointers); variant B keeps
// ONE M0 for a group of four DMAs and moves the LDS address with the instruction's immediate offset (which the
// memory address gets too, so the per-lane offset is reduced by it).  Each workgroup DMAs a pattern that names
// (block, iteration, position) into 64 KB of its LDS and checks every word.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr;
template <int OFF>
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, char* lds, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)lds, 16, voff, soff, OFF, 0);
}
template <bool ONE_M0>
__global__ __launch_bounds__(256) void k(const unsigned* src, unsigned* bad, int iters) {
    __shared__ __attribute__((aligned(1024))) char lds[64 * 1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 1 << 30, 0x00020000);
    unsigned errs = 0;
    for (int it = 0; it < iters; ++it) {
        const int rowid = (blockIdx.x * 7 + it) & 1023;
        const int row = rowid * 65536 + 4096;          // (+4096: room for the negative per-lane offsets of variant B)
        for (int g = 0; g < 4; ++g) {                  // 4 groups of 4 blocks of 1 KB per wave
            char* base = lds + (wave * 16 + g * 4) * 1024;
            const int v = lane * 16 + (wave * 16 + g * 4) * 1024;
            if (ONE_M0) {
                dma16<0>(r, base, v, row);
                dma16<1024>(r, base, v, row);          // LDS +1024 and memory +1024 by the immediate
                dma16<2048>(r, base, v, row);
                dma16<3072>(r, base, v, row);
            } else {
                dma16<0>(r, base, v, row);
                dma16<0>(r, base + 1024, v + 1024, row);
                dma16<0>(r, base + 2048, v + 2048, row);
                dma16<0>(r, base + 3072, v + 3072, row);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int i = threadIdx.x; i < 16384; i += 256) {
            const unsigned v = ((volatile unsigned*)lds)[i];
            errs += v != (((unsigned)rowid << 16) | (unsigned)i);
        }
        __builtin_amdgcn_s_barrier();
    }
    if (errs) atomicAdd(bad, errs);
}
// occupies CUs with workgroups whose LDS allocation is an odd number of 512-byte granules, for a while
template <int BYTES>
__global__ void occupy(unsigned* sink, int spins) {
    __shared__ char x[BYTES];
    x[threadIdx.x] = (char)threadIdx.x;
    unsigned acc = 0;
    for (int i = 0; i < spins; ++i) {
        __syncthreads();
        acc += x[(threadIdx.x * 7 + i) % BYTES];
        x[(threadIdx.x + i) % BYTES] = (char)acc;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void hog(const float4* a, float4* b, long n, int reps) {
    for (int r = 0; r < reps; ++r)
        for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) b[i] = a[(i * 97) % n];
}
int main() {
    std::vector<unsigned> h(1024 * 16384 + 1024);
    for (int r = 0; r < 1024; ++r)
        for (int i = 0; i < 16384; ++i) h[1024 + r * 16384 + i] = (r << 16) | i;
    unsigned *d, *bad;
    (void)hipMalloc(&d, h.size() * 4);
    (void)hipMalloc(&bad, 4);
    (void)hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    float4 *ha, *hb;
    const long hn = 1 << 24;
    (void)hipMalloc(&ha, hn * 16);
    (void)hipMalloc(&hb, hn * 16);
    hipStream_t s1, s2, s3;
    (void)hipStreamCreate(&s1); (void)hipStreamCreate(&s2); (void)hipStreamCreate(&s3);
    for (int load = 0; load < 2; ++load)
        for (int variant = 0; variant < 2; ++variant) {
            (void)hipMemset(bad, 0, 4);
            for (int rep = 0; rep < 30; ++rep) {
                if (load) {
                    hog<<<2048, 256, 0, s2>>>(ha, hb, hn, 1);
                    hog<<<2048, 256, 0, s3>>>(hb, ha, hn, 1);
                }
                if (variant)
                    k<true><<<256, 256, 0, s1>>>(d, bad, 40);
                else
                    k<false><<<256, 256, 0, s1>>>(d, bad, 40);
            }
            (void)hipDeviceSynchronize();
            unsigned b;
            (void)hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost);
            printf("%s, %s: %u mismatching words\n", load ? "memory hogs on two other streams" : "alone",
                   variant ? "one M0 per four DMAs (immediate offsets)" : "M0 rewritten before every DMA", b);
        }
    // co-resident workgroups of ANOTHER kernel with an odd LDS size first on every CU
    for (int variant = 0; variant < 2; ++variant) {
        (void)hipMemset(bad, 0, 4);
        for (int rep = 0; rep < 30; ++rep) {
            occupy<16896><<<256, 64, 0, s2>>>(bad + 0, 20000);      // (bad is only written on a value never reached)
            occupy<8704><<<256, 64, 0, s3>>>(bad + 0, 20000);
            if (variant)
                k<true><<<256, 256, 0, s1>>>(d, bad, 40);
            else
                k<false><<<256, 256, 0, s1>>>(d, bad, 40);
        }
        (void)hipDeviceSynchronize();
        unsigned b;
        (void)hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost);
        printf("behind co-resident workgroups with 16.5 KB / 8.5 KB of LDS, %s: %u mismatching words\n",
               variant ? "one M0 per four DMAs" : "M0 rewritten before every DMA", b);
    }
    return 0;
}
