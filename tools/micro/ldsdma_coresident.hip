// Micro-test: does an LDS-DMA (buffer_load_dwordx4 ... lds) of a workgroup that is NOT the first on its CU land in
// its own LDS allocation?  Every workgroup DMAs a pattern that names it into its LDS, reads it back with ds_read
// and counts mismatches.  Run with 1 and with 2 workgroups per CU (LDS 70 KB each).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr;
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, char* lds, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)lds, 16, voff, soff, 0, 0);
}
template <int LDSB>
__global__ __launch_bounds__(256) void k(const unsigned* src, unsigned* bad, int iters) {
    __shared__ __attribute__((aligned(1024))) char lds[LDSB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 1 << 30, 0x00020000);
    unsigned errs = 0;
    for (int it = 0; it < iters; ++it) {
        // 64 KB of the LDS: block (wave, i) of 1 KB <- src[(blockIdx * 7 + it) & 1023][...]
        const int row = ((blockIdx.x * 7 + it) & 1023) * 65536;
        for (int i = 0; i < 16; ++i) dma16(r, lds + (wave * 16 + i) * 1024, lane * 16 + (wave * 16 + i) * 1024, row);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int i = threadIdx.x; i < 16384; i += 256) {
            const unsigned v = ((volatile unsigned*)lds)[i];
            const unsigned want = (((blockIdx.x * 7 + it) & 1023) << 16) | (i & 0xffff);
            errs += v != want;
        }
        __builtin_amdgcn_s_barrier();
    }
    if (errs) atomicAdd(bad, errs);
}
int main() {
    std::vector<unsigned> h(1024 * 16384);
    for (int r = 0; r < 1024; ++r)
        for (int i = 0; i < 16384; ++i) h[r * 16384 + i] = (r << 16) | i;
    unsigned *d, *bad;
    hipMalloc(&d, h.size() * 4);
    hipMalloc(&bad, 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int percu = 1; percu <= 2; ++percu) {
        hipMemset(bad, 0, 4);
        k<70 * 1024><<<256 * percu, 256>>>(d, bad, 200);
        hipDeviceSynchronize();
        unsigned b;
        hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost);
        printf("LDS 70 KB per workgroup, %d per CU: %u mismatching words (%s)\n", percu, b, hipGetErrorString(hipGetLastError()));
    }
    // two different streams, each 1 per CU
    hipStream_t s1, s2;
    hipStreamCreate(&s1);
    hipStreamCreate(&s2);
    hipMemset(bad, 0, 4);
    for (int rep = 0; rep < 20; ++rep) {
        k<70 * 1024><<<256, 256, 0, s1>>>(d, bad, 50);
        k<70 * 1024><<<256, 256, 0, s2>>>(d, bad, 50);
    }
    hipDeviceSynchronize();
    unsigned b;
    hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost);
    printf("two streams, 256 workgroups each: %u mismatching words\n", b);
    return 0;
}
