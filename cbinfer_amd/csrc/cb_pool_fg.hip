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

// SURVEY 8f-4, k > 1 consumers of propagated change indexes: every listed pixel (y, x) of an H x W map marks its
// filter support -- rows y-kHH..y+kHH, columns x-kWH..x+kWH, clipped to the map -- in a row-padded bit mask; that is
// the set of OUTPUT pixels of a kH x kW convolution the listed input pixels reach, i.e. what the layer's own change
// detection (cbconv2d_cg_backend.cu:62-72) would mark if every listed pixel had changed.  One thread per (list entry,
// row of the support); a run of set bits crosses at most one word boundary for kWH < 32.
__global__ __launch_bounds__(256) void cb_dilate_indexes_kernel(const int32_t* __restrict__ list, int nHost,
                                                               const int32_t* __restrict__ countDev, int H, int W,
                                                               int kHH, int kWH, int wpr,
                                                               unsigned long long* __restrict__ bits) {
    const int N = countDev ? min(*countDev, nHost) : nHost;
    const int rows = 2 * kHH + 1;
    const long total = (long)N * rows;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i / rows), r = (int)(i - (long)n * rows);
        const int pos = list[n];
        if ((unsigned)pos >= (unsigned)(H * W)) continue;      // (an out-of-map entry is dropped, as the contraction does)
        const int y = pos / W, x = pos - y * W;
        const int yy = y + r - kHH;
        if (yy < 0 || yy >= H) continue;
        const int x0 = max(x - kWH, 0), x1 = min(x + kWH, W - 1);
        const int w0 = x0 >> 6, w1 = x1 >> 6;
        const unsigned long long lo = ~0ull << (x0 & 63), hi = ~0ull >> (63 - (x1 & 63));
        if (w0 == w1) {
            atomicOr(bits + (long)yy * wpr + w0, lo & hi);
        } else {
            atomicOr(bits + (long)yy * wpr + w0, lo);
            for (int w = w0 + 1; w < w1; ++w) atomicOr(bits + (long)yy * wpr + w, ~0ull);
            atomicOr(bits + (long)yy * wpr + w1, hi);
        }
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

// Per-value delta scatter (a12, API-parity form of cbconv2d_fg_backend.cu:37-66).  grid.x strides over
// the changed values (consecutive lanes = consecutive list entries = mostly consecutive x, so one atomic
// wave-instruction touches a contiguous run of one output row -- the shape the memory-side f32 atomic
// units take at full rate), grid.y over output channels.  The list is int64 (torch.nonzero,
// conv2d_fg.py:82) or int32 with a device-side length (cbinfer_change_indexes_extr: no host round trip).
// Summation order is unspecified, as with the reference's atomicAdd (.cu:61).  The modules do not use this
// kernel: their fine-grained frame runs cbinfer_cbconv2d_forward_fg (no atomics, see cb_frame.hip).
template <typename IDX>
__global__ __launch_bounds__(256) void cb_update_fg_kernel(const float* __restrict__ diffs,
                                                          const float* __restrict__ weight,
                                                          float* output, const IDX* __restrict__ coords,
                                                          const int32_t* __restrict__ countDev, int C,
                                                          int H, int W, int kH, int kW, long numChanges) {
    const long N = countDev ? min((long)*countDev, numChanges) : numChanges;
    const int co = blockIdx.y;
    const int HW = H * W, ph = kH / 2, pw = kW / 2;
    float* o = output + (long)co * HW;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < N; t += (long)gridDim.x * blockDim.x) {
        const int pos = (int)coords[t];
        const int ci = pos / HW, pix = pos - ci * HW;
        const int y = pix / W, x = pix - y * W;
        const float d = diffs[pos];
        const float* w = weight + ((long)co * C + ci) * kH * kW;
        // tap (ky,kx) of the value at (y,x) lands on output (y - ky + ph, x - kx + pw)
        const int ky0 = max(0, y + ph - (H - 1)), ky1 = min(kH - 1, y + ph);
        const int kx0 = max(0, x + pw - (W - 1)), kx1 = min(kW - 1, x + pw);
        for (int ky = ky0; ky <= ky1; ++ky)
            for (int kx = kx0; kx <= kx1; ++kx)
                atomicAdd(o + (long)(y - ky + ph) * W + (x - kx + pw), w[ky * kW + kx] * d);
    }
}

// Channel concatenation of batch-1 tensors [Ci,H,W] (torch.cat(dim=1); poseDetection/openPose/PoseModel.py:131: the input
// of a refinement stage is cat(branch 1, branch 2, features)): the sources are contiguous blocks that follow each other in
// the destination -- one launch copies them all, in the widest unit every block boundary allows.
struct CbConcatArgs {
    const char* src[CBINFER_CONCAT_MAX];
    long begin[CBINFER_CONCAT_MAX + 1];      // byte offsets of the blocks in the destination
    int n;
};
template <typename U>
__global__ __launch_bounds__(256) void cb_concat_kernel(CbConcatArgs a, char* __restrict__ dst) {
    const long total = a.begin[a.n] / (long)sizeof(U);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long b = i * (long)sizeof(U);
        int k = 0;
#pragma unroll
        for (int u = 1; u < CBINFER_CONCAT_MAX; ++u)
            if (u < a.n && b >= a.begin[u]) k = u;
        ((U*)dst)[i] = *(const U*)(a.src[k] + (b - a.begin[k]));
    }
}

}  // namespace

extern "C" {

int cbinfer_abi_version(void) { return CBINFER_ABI_VERSION; }

int cbinfer_concat_channels(const void* const* sources, const int32_t* channels, int n, void* output, long HW, int dtype,
                            cbStream_t stream) {
    CB_REQUIRE(sources && channels && output && n >= 1 && n <= CBINFER_CONCAT_MAX && HW > 0 &&
               (dtype == CB_F32 || dtype == CB_F16));
    const long es = dtype == CB_F32 ? 4 : 2;
    CbConcatArgs a;
    a.n = n;
    a.begin[0] = 0;
    unsigned long long align = (unsigned long long)(size_t)output;
    for (int k = 0; k < CBINFER_CONCAT_MAX; ++k) {
        a.src[k] = k < n ? (const char*)sources[k] : nullptr;
        if (k < n) {
            CB_REQUIRE(sources[k] && channels[k] > 0);
            a.begin[k + 1] = a.begin[k] + (long)channels[k] * HW * es;
            align |= (unsigned long long)(size_t)sources[k] | (unsigned long long)a.begin[k + 1];
        } else {
            a.begin[k + 1] = a.begin[k];
        }
    }
    const long bytes = a.begin[n];
    const int unit = (align & 15) == 0 ? 16 : ((align & 7) == 0 ? 8 : ((align & 3) == 0 ? 4 : 2));
    const int blocks = (int)((bytes / unit + 255) / 256 > 4096 ? 4096 : (bytes / unit + 255) / 256);
    hipStream_t s = (hipStream_t)stream;
    if (unit == 16)
        hipLaunchKernelGGL(cb_concat_kernel<uint4>, dim3(blocks), dim3(256), 0, s, a, (char*)output);
    else if (unit == 8)
        hipLaunchKernelGGL(cb_concat_kernel<uint2>, dim3(blocks), dim3(256), 0, s, a, (char*)output);
    else if (unit == 4)
        hipLaunchKernelGGL(cb_concat_kernel<unsigned>, dim3(blocks), dim3(256), 0, s, a, (char*)output);
    else
        hipLaunchKernelGGL(cb_concat_kernel<unsigned short>, dim3(blocks), dim3(256), 0, s, a, (char*)output);
    return cb_launch_status();
}

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

int cbinfer_dilate_change_indexes(const int32_t* changeIndexes, int numChanges, const int32_t* countDev, int H, int W,
                                  int kHHalf, int kWHalf, uint64_t* bitsOut, cbStream_t stream) {
    CB_REQUIRE(changeIndexes && bitsOut && numChanges >= 0 && H > 0 && W > 0 && kHHalf >= 0 && kWHalf >= 0);
    if (numChanges == 0) return CB_OK;
    long blocks = ((long)numChanges * (2 * kHHalf + 1) + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(cb_dilate_indexes_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       changeIndexes, numChanges, countDev, H, W, kHHalf, kWHalf, (W + 63) / 64,
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

static int cb_update_fg_launch(const float* diffs, const float* weight, float* output,
                               const void* coords, int coords64, const int32_t* countDev, int K, int C,
                               int H, int W, int kH, int kW, long numChanges, cbStream_t stream) {
    CB_REQUIRE(diffs && weight && output && coords && K > 0 && C > 0 && H > 0 && W > 0 && kH > 0 &&
               kW > 0 && numChanges >= 0);
    if ((long)C * H * W >= (1l << 31) || K > 65535) return CB_ERR_UNSUPPORTED;
    if (numChanges == 0) return CB_OK;
    long bx = (numChanges + 255) / 256;
    if (bx > 4096) bx = 4096;   // grid-stride: the capacity may be the whole tensor when the count is on the device
    dim3 grid((unsigned)bx, K), block(256);
    if (coords64)
        hipLaunchKernelGGL((cb_update_fg_kernel<int64_t>), grid, block, 0, (hipStream_t)stream, diffs, weight,
                           output, (const int64_t*)coords, countDev, C, H, W, kH, kW, numChanges);
    else
        hipLaunchKernelGGL((cb_update_fg_kernel<int32_t>), grid, block, 0, (hipStream_t)stream, diffs, weight,
                           output, (const int32_t*)coords, countDev, C, H, W, kH, kW, numChanges);
    return cb_launch_status();
}

int cbinfer_update_output_fg(const float* diffs, const float* weight, float* output,
                             const int64_t* changeCoords, int K, int C, int H, int W, int kH, int kW,
                             long numChanges, cbStream_t stream) {
    return cb_update_fg_launch(diffs, weight, output, changeCoords, 1, nullptr, K, C, H, W, kH, kW,
                               numChanges, stream);
}

int cbinfer_update_output_fg_list(const float* diffs, const float* weight, float* output,
                                  const int32_t* changeCoords, long capacity, const int32_t* countDev,
                                  int K, int C, int H, int W, int kH, int kW, cbStream_t stream) {
    return cb_update_fg_launch(diffs, weight, output, changeCoords, 0, countDev, K, C, H, W, kH, kW,
                               capacity, stream);
}

// Host fine-grained delta convolution, the library's counterpart of conv2d_fg_cpu
// (cbconv2d_fg_backend.cu:81-112): every input value whose difference is not below the threshold
// (`!(|d| < th)`, i.e. >=, as there) adds d * w[:, c, ky, kx] to the outputs its taps reach.  One OpenMP
// task per OUTPUT plane (planes are disjoint, so no two threads ever touch the same float -- the reference
// parallelises over input channels, which all accumulate into the same planes), and the in-image tap range
// of a value is computed once instead of testing every tap.
void cbinfer_conv2d_fg_cpu(const float* input, const float* prevInput, float* output,
                           const float* weight, float threshold, int no, int ni, int h, int w,
                           int kh, int kw) {
    const int ph = kh / 2, pw = kw / 2;
    const long hw = (long)h * w;
#pragma omp parallel for schedule(static)
    for (int plane = 0; plane < no; ++plane) {
        float* const dst = output + plane * hw;
        for (long v = 0; v < (long)ni * hw; ++v) {
            const float d = input[v] - prevInput[v];
            if (fabsf(d) < threshold) continue;
            const int c = (int)(v / hw);
            const int pix = (int)(v - c * hw), y = pix / w, x = pix - y * w;
            const float* taps = weight + ((long)plane * ni + c) * kh * kw;
            const int ky0 = y + ph - (h - 1) > 0 ? y + ph - (h - 1) : 0, ky1 = y + ph < kh - 1 ? y + ph : kh - 1;
            const int kx0 = x + pw - (w - 1) > 0 ? x + pw - (w - 1) : 0, kx1 = x + pw < kw - 1 ? x + pw : kw - 1;
            for (int ky = ky0; ky <= ky1; ++ky) {
                float* row = dst + (long)(y - ky + ph) * w + (x + pw);
                for (int kx = kx0; kx <= kx1; ++kx) row[-kx] += d * taps[ky * kw + kx];
            }
        }
    }
}

}  // extern "C"
