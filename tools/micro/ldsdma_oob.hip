// Micro-test (round 3): an LDS-DMA instruction whose buffer resource has num_records = 0 (every access out of
// range).  Question: does it complete like any other (vmcnt counts it, the wave goes on), and what does it leave in
// LDS?  cb_split.hip's stage loop issues such DMAs at the end of an item so that the count of DMA instructions in
// flight -- which its s_waitcnt vmcnt(N) relies on -- is the same in every step.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lds_ptr;
template <int OFF>
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, char* lds, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)lds, 16, voff, soff, OFF, 0);
}
__global__ __launch_bounds__(64) void k(const unsigned* src, unsigned* out, int live) {
    __shared__ __attribute__((aligned(1024))) char lds[4096];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) ((unsigned*)lds)[i] = 0xAAAAAAAAu;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t dead = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, live ? 4096 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t good = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 8192, 0x00020000);
    dma16<0>(dead, lds, lane * 16, 0);            // out of range (live == 0)
    dma16<1024>(dead, lds, lane * 16, 0);         // out of range, immediate offset
    dma16<2048>(good, lds, lane * 16, 0);         // in range: source bytes 2048.. -> LDS 2048..
    dma16<3072>(dead, lds, lane * 16, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int i = lane; i < 1024; i += 64) out[i] = ((volatile unsigned*)lds)[i];
}
int main() {
    unsigned h[2048], o[1024];
    for (int i = 0; i < 2048; ++i) h[i] = 0x10000u + i;
    unsigned *d, *out;
    (void)hipMalloc(&d, sizeof h);
    (void)hipMalloc(&out, sizeof o);
    (void)hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    for (int live = 0; live < 2; ++live) {
        k<<<1, 64>>>(d, out, live);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
        (void)hipMemcpy(o, out, sizeof o, hipMemcpyDeviceToHost);
        for (int b = 0; b < 4; ++b) {
            int zeros = 0, kept = 0, src = 0, other = 0;
            for (int i = 0; i < 256; ++i) {
                const unsigned v = o[b * 256 + i];
                if (v == 0) ++zeros; else if (v == 0xAAAAAAAAu) ++kept; else if (v == 0x10000u + b * 256 + i) ++src; else ++other;
            }
            printf("num_records %s, block %d: %d zero, %d untouched, %d source words, %d other\n",
                   live ? "4096" : "0", b, zeros, kept, src, other);
        }
    }
    return 0;
}
