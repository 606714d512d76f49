// Change-based 2x2 max pooling and the fine-grained (per-value delta) path for gfx950.
// Reference behaviour restated from cbconv2d_cg_backend.cu:199-240 and cbconv2d_fg_backend.cu:7-112;
// entry-point contracts in include/cbinfer_hip.h.
#include <math.h>

#include "cb_common.h"

namespace {


// One thread per (changed input pixel, channel); the pixel index is the fastest-varying coordinate so
// neighbouring lanes read neighbouring window columns of one channel plane and write neighbouring
// outputs.  Several changed pixels of one window recompute the same value (same-value race, as in the
// reference, .cu:224).  Windows outside the output (odd size, floor mode) are skipped.
template <typename T>
__global__ __launch_bounds__(256) void cb_maxpool_kernel(const T* __restrict__ in, T* out,
                                                        const int32_t* __restrict__ list, int nHost,
                                                        const int32_t* __restrict__ countDev, int C,
                                                        int iH, int iW, int oH, int oW) {
    const int N = countDev ? min(*countDev, nHost) : nHost;
    const long total = (long)N * C;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long)gridDim.x * blockDim.x) {
        const int n = (int)(e % N), ch = (int)(e / N);
        const int pos = list[n];
        const int y = pos / iW, x = pos - y * iW;
        const int yo = y >> 1, xo = x >> 1;
        if (yo >= oH || xo >= oW) continue;
        T v = cb_neg_inf((T*)nullptr);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int yi = yo * 2 + j, xi = xo * 2 + i;
                if (yi < iH && xi < iW) v = cb_max(v, in[((long)ch * iH + yi) * iW + xi]);
            }
        out[((long)ch * oH + yo) * oW + xo] = v;
    }
}

// f4: change indexes carried THROUGH a 2x2/stride-2 pool.  Every changed input pixel marks
// its output window in a row-padded bit mask of the pooled map (the reference hands the input-resolution
// list on unchanged, conv2d.py:80-83, which no consumer can use); cbinfer_compact_bits then turns the
// mask into the ascending, duplicate-free output-resolution list.
__global__ __launch_bounds__(256) void cb_pool_indexes_kernel(const int32_t* __restrict__ list, int nHost,
                                                             const int32_t* __restrict__ countDev,
                                                             int iW, int oH, int oW, int wpr,
                                                             unsigned long long* __restrict__ bits) {
    const int N = countDev ? min(*countDev, nHost) : nHost;
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
        const int pos = list[n];
        const int y = pos / iW, x = pos - y * iW;
        const int yo = y >> 1, xo = x >> 1;
        if (yo >= oH || xo >= oW) continue;
        // the four pixels of a window are neighbours in the list order only along x: let the even-x
        // (or first) one of a pair do the atomic when its partner is the next list entry
        if ((x & 1) && n > 0 && list[n - 1] == pos - 1) continue;
        atomicOr(bits + (long)yo * wpr + (xo >> 6), 1ull << (xo & 63));
    }
}

__global__ __launch_bounds__(256) void cb_detect_fg_kernel(const float* __restrict__ in,
                                                          const float* __restrict__ prev,
                                                          float* __restrict__ diffs,
                                                          int8_t* __restrict__ map, long numVals,
                                                          float th, int zeroUnchanged) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < numVals;
         i += (long)gridDim.x * blockDim.x) {
        const float d = in[i] - prev[i];
        const bool pred = fabsf(d) > th;
        map[i] = pred;
        if (pred)
            diffs[i] = d;
        else if (zeroUnchanged)
            diffs[i] = 0.f;
    }
}

// grid.x over changed values (consecutive lanes = consecutive list entries = mostly consecutive x, so
// one atomic wave-instruction touches a contiguous run of one output row), grid.y over output
// channels.  f32 atomic adds execute at the memory side on gfx950; summation order is unspecified,
// as with the reference's atomicAdd (cbconv2d_fg_backend.cu:61).
__global__ __launch_bounds__(256) void cb_update_fg_kernel(const float* __restrict__ diffs,
                                                          const float* __restrict__ weight,
                                                          float* output,
                                                          const int64_t* __restrict__ coords, int K,
                                                          int C, int H, int W, int kH, int kW,
                                                          long numChanges) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= numChanges) return;
    const int co = blockIdx.y;
    const int pos = (int)coords[t];
    const int ci = pos / (H * W);
    const int y = (pos / W) % H;
    const int x = pos % W;
    const float d = diffs[pos];
    const float* w = weight + ((long)co * C + ci) * kH * kW;
    float* o = output + (long)co * H * W;
    for (int iky = 0; iky < kH; ++iky) {
        const int ytot = y - iky + kH / 2;
        if (ytot < 0 || ytot >= H) continue;
        for (int ikx = 0; ikx < kW; ++ikx) {
            const int xtot = x - ikx + kW / 2;
            if (xtot < 0 || xtot >= W) continue;
            atomicAdd(o + (long)ytot * W + xtot, w[iky * kW + ikx] * d);
        }
    }
}

}  // namespace

extern "C" {

int cbinfer_abi_version(void) { return CBINFER_ABI_VERSION; }

const char* cbinfer_status_string(int status) {
    if (status == CB_OK) return "ok";
    if (status == CB_ERR_BADARG) return "cbinfer: bad argument (null pointer, size or dtype)";
    if (status == CB_ERR_UNSUPPORTED) return "cbinfer: shape not supported by the gfx950 kernels";
    if (status > 0) return hipGetErrorString((hipError_t)status);
    return "cbinfer: unknown status";
}

int cbinfer_max_pool2d(const void* input, void* output, const int32_t* changeIndexes, int numChanges,
                       const int32_t* countDev, int C, int iH, int iW, int oH, int oW, int dtype,
                       cbStream_t stream) {
    CB_REQUIRE(input && output && changeIndexes && numChanges >= 0 && C > 0 && iH > 0 && iW > 0 &&
               oH > 0 && oW > 0);
    if (numChanges == 0) return CB_OK;
    const long total = (long)numChanges * C;
    long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    dim3 grid((unsigned)blocks), block(256);
    if (dtype == CB_F32)
        hipLaunchKernelGGL((cb_maxpool_kernel<float>), grid, block, 0, (hipStream_t)stream,
                           (const float*)input, (float*)output, changeIndexes, numChanges, countDev, C,
                           iH, iW, oH, oW);
    else if (dtype == CB_F16)
        hipLaunchKernelGGL((cb_maxpool_kernel<cb_half>), grid, block, 0, (hipStream_t)stream,
                           (const cb_half*)input, (cb_half*)output, changeIndexes, numChanges,
                           countDev, C, iH, iW, oH, oW);
    else
        return CB_ERR_BADARG;
    return cb_launch_status();
}

int cbinfer_pool_change_indexes(const int32_t* changeIndexes, int numChanges, const int32_t* countDev,
                                int iW, int oH, int oW, uint64_t* bitsOut, cbStream_t stream) {
    CB_REQUIRE(changeIndexes && bitsOut && numChanges >= 0 && iW > 0 && oH > 0 && oW > 0);
    if (numChanges == 0) return CB_OK;
    long blocks = ((long)numChanges + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(cb_pool_indexes_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       changeIndexes, numChanges, countDev, iW, oH, oW, (oW + 63) / 64,
                       (unsigned long long*)bitsOut);
    return cb_launch_status();
}

int cbinfer_change_detection_fg(const float* input, const float* prevInput, float* diffs,
                                int8_t* changeMap, long numVals, float threshold, int zeroUnchanged,
                                cbStream_t stream) {
    CB_REQUIRE(input && prevInput && diffs && changeMap && numVals > 0);
    long blocks = (numVals + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(cb_detect_fg_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       input, prevInput, diffs, changeMap, numVals, threshold, zeroUnchanged);
    return cb_launch_status();
}

int cbinfer_update_output_fg(const float* diffs, const float* weight, float* output,
                             const int64_t* changeCoords, int K, int C, int H, int W, int kH, int kW,
                             long numChanges, cbStream_t stream) {
    CB_REQUIRE(diffs && weight && output && changeCoords && K > 0 && C > 0 && H > 0 && W > 0 &&
               kH > 0 && kW > 0 && numChanges >= 0);
    if ((long)C * H * W >= (1l << 31) || K > 65535) return CB_ERR_UNSUPPORTED;
    if (numChanges == 0) return CB_OK;
    dim3 grid(cb_div_up(numChanges, 256), K), block(256);
    hipLaunchKernelGGL(cb_update_fg_kernel, grid, block, 0, (hipStream_t)stream, diffs, weight, output,
                       changeCoords, K, C, H, W, kH, kW, numChanges);
    return cb_launch_status();
}

// Host fine-grained delta convolution (cbconv2d_fg_backend.cu:81-112).  Parallel over OUTPUT
// channels, which own disjoint output planes -- the reference's `omp for` over input channels races.
void cbinfer_conv2d_fg_cpu(const float* input, const float* prevInput, float* output,
                           const float* weight, float threshold, int no, int ni, int h, int w,
                           int kh, int kw) {
    const int khhalf = kh / 2, kwhalf = kw / 2;
#pragma omp parallel for schedule(static)
    for (int co = 0; co < no; ++co)
        for (int ci = 0; ci < ni; ++ci)
            for (int y = 0; y < h; ++y)
                for (int x = 0; x < w; ++x) {
                    const long iidx = ((long)ci * h + y) * w + x;
                    const float diff = input[iidx] - prevInput[iidx];
                    if (fabsf(diff) < threshold) continue;
                    for (int iky = 0; iky < kh; ++iky) {
                        const int oy = y - iky + khhalf;
                        if (oy < 0 || oy >= h) continue;
                        for (int ikx = 0; ikx < kw; ++ikx) {
                            const int ox = x - ikx + kwhalf;
                            if (ox < 0 || ox >= w) continue;
                            output[((long)co * h + oy) * w + ox] +=
                                diff * weight[(((long)co * ni + ci) * kh + iky) * kw + ikx];
                        }
                    }
                }
}

}  // extern "C"
