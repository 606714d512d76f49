/*
 * cb_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of CBinfer's change-based convolution hot path, written from the
 * behaviour of the reference's CUDA kernels and Python twins.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product (cbinfer_amd/) never does.
 *
 * Parity status: PINNED.  Checked against (i) the reference's own known-answer vectors
 * (genTestData 15-index KAT, changeIndexesExtr_test1 6-index KAT, cbconvFG_test1), (ii) golden
 * fixtures emitted by the reference's pure-torch *_python ops imported in the build container
 * (tests/golden/gen_golden.py), and (iii) on the GPU box, the reference's own kernels compiled from
 * /root/reference by oracle/Makefile into oracle/_ref/.
 *
 * Every function cites the reference lines it restates (paths relative to /root/reference/pycbinfer).
 * All tensors are NCHW with batch 1, contiguous.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* comparison modes for the change predicate */
#define ORC_CMP_GT 0 /* CUDA kernels: fabs(d) >  th  (cbconv2d_cg_backend.cu:22,56; fg :19) */
#define ORC_CMP_GE 1 /* python twin / conv2d_fg_cpu: fabs(d) >= th (conv2d_cg.py:126; fg.cu:93) */

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static inline int changed_f32(float a, float b, float th, int cmp) {
    float d = fabsf(a - b);
    return cmp == ORC_CMP_GE ? (d >= th) : (d > th);
}

/* changeDetection: cbconv2d_cg_backend.cu:6-37 (1x1) and :40-81 (dilating); launcher :83-100.
 * Per input pixel: change = OR_c |state[c,p]-in[c,p]| > th; a changed pixel marks its
 * (2kHHalf+1)x(2kWHalf+1) neighbourhood in changeMap (which the caller pre-zeroes, conv2d_cg.py:105)
 * and, if updateInputState, copies in[:,p] -> state[:,p] for that (pre-dilation) pixel only.
 * The per-pixel work only touches the pixel's own column of state, so a sequential loop is
 * equivalent to the parallel kernel. */
void orc_changeDetection(const float* input, float* state, int8_t* changeMap, int width, int height,
                         int nInputPlane, int kHHalf, int kWHalf, float th, int updateInputState,
                         int cmp) {
    const long HW = (long)width * height;
#pragma omp parallel for schedule(static)
    for (long p = 0; p < HW; ++p) {
        int change = 0;
        for (int c = 0; c < nInputPlane; ++c)
            change |= changed_f32(state[c * HW + p], input[c * HW + p], th, cmp);
        if (!change) continue;
        int xIn = (int)(p % width), yIn = (int)(p / width);
        for (int k = -kHHalf; k <= kHHalf; ++k) {
            int yOut = yIn + k;
            for (int l = -kWHalf; l <= kWHalf; ++l) {
                int xOut = xIn + l;
                if (yOut >= 0 && yOut < height && xOut >= 0 && xOut < width)
                    changeMap[(long)yOut * width + xOut] = 1; /* benign same-value race, .cu:69 */
            }
        }
        if (updateInputState)
            for (int c = 0; c < nInputPlane; ++c) state[c * HW + p] = input[c * HW + p];
    }
}

/* changePropagation: cbconv2d_cg_backend.cu:101-124 (gather-form dilation of a bool map). */
void orc_changePropagation(const int8_t* in, int8_t* out, int width, int height, int kHHalf,
                           int kWHalf) {
#pragma omp parallel for schedule(static)
    for (int yOut = 0; yOut < height; ++yOut)
        for (int xOut = 0; xOut < width; ++xOut) {
            int change = 0;
            for (int k = -kHHalf; k <= kHHalf; ++k) {
                int yIn = yOut + k;
                for (int l = -kWHalf; l <= kWHalf; ++l) {
                    int xIn = xOut + l;
                    if (yIn >= 0 && yIn < height && xIn >= 0 && xIn < width)
                        change = change || in[(long)yIn * width + xIn];
                }
            }
            out[(long)yOut * width + xOut] = (int8_t)change;
        }
}

/* changeIndexesExtr: conv2d_cg.py:200-209, torch.nonzero(map.view(-1)).int() -- ascending flat
 * indices y*W+x of the non-zero map entries.  Returns N. */
int orc_changeIndexesExtr(const int8_t* changeMap, long numel, int32_t* idx) {
    /* chunked two-pass form (count per chunk, exclusive prefix, ordered write): the result is the
     * sequential scan's, the chunks only let the host cores share the work when this port is timed as
     * the CPU baseline */
    enum { CHUNKS = 256 };
    long cnt[CHUNKS + 1];
    const long per = (numel + CHUNKS - 1) / CHUNKS;
#pragma omp parallel for schedule(static)
    for (int c = 0; c < CHUNKS; ++c) {
        long lo = c * per, hi = lo + per < numel ? lo + per : numel, n = 0;
        for (long p = lo; p < hi; ++p) n += changeMap[p] != 0;
        cnt[c + 1] = n;
    }
    cnt[0] = 0;
    for (int c = 0; c < CHUNKS; ++c) cnt[c + 1] += cnt[c];
#pragma omp parallel for schedule(static)
    for (int c = 0; c < CHUNKS; ++c) {
        long lo = c * per, hi = lo + per < numel ? lo + per : numel, n = cnt[c];
        for (long p = lo; p < hi; ++p)
            if (changeMap[p]) idx[n++] = (int32_t)p;
    }
    return (int)cnt[CHUNKS];
}

/* genXMatrix: cbconv2d_cg_backend.cu:138-161; python twin conv2d_cg.py:263-281.
 * X[n, (c*kH+ky)*kW+kx] = in[c, y+ky-(kH-1)/2, x+kx-(kW-1)/2], zero outside the image. */
void orc_genXMatrix(float* columns, const float* input, const int32_t* changeList, int kW, int kH,
                    int nInputPlane, int width, int height, int numChanges) {
    const long rowLen = (long)kW * kH * nInputPlane;
#pragma omp parallel for schedule(static)
    for (int n = 0; n < numChanges; ++n) {
        int pos = changeList[n];
        for (int ky = 0; ky < kH; ++ky)
            for (int kx = 0; kx < kW; ++kx) {
                int ix = pos % width + kx - (kW - 1) / 2;
                int iy = pos / width + ky - (kH - 1) / 2;
                int inside = ix >= 0 && ix < width && iy >= 0 && iy < height;
                float* dst = columns + n * rowLen + ky * kW + kx;
                for (int c = 0; c < nInputPlane; ++c)
                    dst[(long)c * kH * kW] =
                        inside ? input[((long)c * height + iy) * width + ix] : 0.0f;
            }
    }
}

/* matrixMult: conv2d_cg.py:342-349, Y = X . W.view(K,-1)^T + bias, Y is [N,K].
 * accMode 0: accumulate in double and round once (the "truth" the 1e-4 bar is measured against).
 * accMode 1: k-ordered float fmaf chain starting from 0, bias added last (what an exact-f32 MFMA
 *            chain produces; used to check how close to bit-exact the device GEMM is). */
void orc_matrixMult(const float* X, const float* Wt, const float* bias, float* Y, int N, int Ckk,
                    int K, int accMode) {
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n) {
        const float* x = X + (long)n * Ckk;
        for (int k = 0; k < K; ++k) {
            const float* w = Wt + (long)k * Ckk;
            if (accMode == 0) {
                double acc = 0.0;
                for (int j = 0; j < Ckk; ++j) acc += (double)x[j] * (double)w[j];
                Y[(long)n * K + k] = (float)(acc + (double)bias[k]);
            } else {
                float acc = 0.0f;
                for (int j = 0; j < Ckk; ++j) acc = fmaf(x[j], w[j], acc);
                Y[(long)n * K + k] = acc + bias[k];
            }
        }
    }
}

/* updateOutput: cbconv2d_cg_backend.cu:175-189.  Yt is the [K,N] (transposed, contiguous) result;
 * out[k*HW + changeList[n]] = relu ? (v <= 0 ? 0 : v) : v. */
void orc_updateOutput(const float* Yt, float* output, const int32_t* changeList, int numOutputPixel,
                      int numChanges, int nOutputPlane, int relu) {
#pragma omp parallel for schedule(static)
    for (int k = 0; k < nOutputPlane; ++k)
        for (int n = 0; n < numChanges; ++n) {
            float v = Yt[(long)k * numChanges + n];
            v = (relu && v <= 0) ? 0 : v;
            output[(long)k * numOutputPixel + changeList[n]] = v;
        }
}

/* maxPool2d: cbconv2d_cg_backend.cu:199-227.  One work item per changed INPUT pixel index; its 2x2
 * (stride x stride) window is recomputed for every channel with the yi<H && xi<W guard.
 * guardOutput=0 reproduces the reference exactly -- including its out-of-bounds write when
 * yo>=oheight or xo>=owidth (odd size in floor mode) -- and must only be used where
 * that cannot happen; guardOutput=1 skips such windows (what the product does). */
void orc_maxPool2d(const float* input, float* output, const int32_t* changeIndexes, int numChanges,
                   int numCh, int iheight, int iwidth, int oheight, int owidth, int stridey,
                   int stridex, int guardOutput) {
    /* channel planes are disjoint, so the host cores share them; within a plane the list is walked in
     * order, exactly as a single thread of the reference's grid would */
#pragma omp parallel for schedule(static)
    for (int ch = 0; ch < numCh; ++ch)
        for (int t = 0; t < numChanges; ++t) {
            int pxIdx = changeIndexes[t];
            int y = pxIdx / iwidth, x = pxIdx % iwidth;
            int yo = y / stridey, xo = x / stridex;
            if (guardOutput && (yo >= oheight || xo >= owidth)) continue;
            float v = -INFINITY;
            for (int j = 0; j < stridey; ++j)
                for (int i = 0; i < stridex; ++i) {
                    int yi = yo * stridey + j, xi = xo * stridex + i;
                    if (yi < iheight && xi < iwidth)
                        v = fmaxf(v, input[((long)ch * iheight + yi) * iwidth + xi]);
                }
            output[((long)ch * oheight + yo) * owidth + xo] = v;
        }
}

/* changeDetectionFG: cbconv2d_fg_backend.cu:7-23.  Per value: d = in - prev; pred = |d| > th;
 * changeMap = pred; diffs written only where pred (left untouched elsewhere, as the kernel does). */
void orc_changeDetectionFG(const float* input, const float* prevInput, float* diffs,
                           int8_t* changeMap, long numVals, float th, int cmp) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < numVals; ++i) {
        float d = input[i] - prevInput[i];
        int pred = cmp == ORC_CMP_GE ? (fabsf(d) >= th) : (fabsf(d) > th);
        changeMap[i] = (int8_t)pred;
        if (pred) diffs[i] = d;
    }
}

/* updateOutputFG: cbconv2d_fg_backend.cu:37-66.  For each changed value (flat coordinate into
 * [C,H,W], int64 as torch.nonzero yields) add w[co,ci,ky,kx]*d into out[co, y-ky+kH/2, x-kx+kW/2].
 * The GPU uses atomicAdd (order unspecified); the oracle adds in list order. */
void orc_updateOutputFG(const float* diffs, const float* weight, float* output,
                        const int64_t* changeCoords, int numOut, int numIn, int height, int width,
                        int kH, int kW, long numChanges) {
    for (long t = 0; t < numChanges; ++t) {
        int pos = (int)changeCoords[t];
        int ci = pos / (height * width);
        int y = (pos / width) % height;
        int x = pos % width;
        float d = diffs[((long)ci * height + y) * width + x];
        for (int co = 0; co < numOut; ++co)
            for (int iky = 0; iky < kH; ++iky)
                for (int ikx = 0; ikx < kW; ++ikx) {
                    int ytot = y - iky + kH / 2, xtot = x - ikx + kW / 2;
                    float w = weight[(((long)co * numIn + ci) * kH + iky) * kW + ikx];
                    if (0 <= ytot && ytot < height && 0 <= xtot && xtot < width)
                        output[((long)co * height + ytot) * width + xtot] += w * d;
                }
    }
}

/* conv2d_fg_cpu: cbconv2d_fg_backend.cu:81-112, run single-threaded (the reference's
 * `omp for` over ci races on `output` for ni>1).  Processes a value unless |d| < th. */
void orc_conv2d_fg_cpu(const float* input, const float* prevInput, float* output,
                       const float* weight, float th, int no, int ni, int h, int w, int kh, int kw) {
    const int khhalf = kh / 2, kwhalf = kw / 2;
    for (int ci = 0; ci < ni; ++ci)
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                long iidx = ((long)ci * h + y) * w + x;
                float diff = input[iidx] - prevInput[iidx];
                if (fabsf(diff) < th) continue;
                for (int co = 0; co < no; ++co)
                    for (int iky = 0; iky < kh; ++iky) {
                        int oy = y - iky + khhalf;
                        if (oy < 0 || oy >= h) continue;
                        for (int ikx = 0; ikx < kw; ++ikx) {
                            int ox = x - ikx + kwhalf;
                            if (ox < 0 || ox >= w) continue;
                            output[((long)co * h + oy) * w + ox] +=
                                diff * weight[(((long)co * ni + ci) * kh + iky) * kw + ikx];
                        }
                    }
            }
}

/* Dense reference used as ground truth for whole-layer tests: out = conv2d(in, w, pad=k/2) + bias
 * accumulated in double (what CBConv2d must equal wherever a pixel is recomputed; conv2d.py:163-167
 * uses F.conv2d for the fine-grained first frame). */
void orc_conv2d_dense(const float* input, const float* weight, const float* bias, float* output,
                      int no, int ni, int h, int w, int kh, int kw, int relu) {
    const int ph = (kh - 1) / 2, pw = (kw - 1) / 2;
#pragma omp parallel for collapse(2) schedule(static)
    for (int co = 0; co < no; ++co)
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                double acc = bias ? (double)bias[co] : 0.0;
                for (int ci = 0; ci < ni; ++ci)
                    for (int ky = 0; ky < kh; ++ky) {
                        int iy = y + ky - ph;
                        if (iy < 0 || iy >= h) continue;
                        for (int kx = 0; kx < kw; ++kx) {
                            int ix = x + kx - pw;
                            if (ix < 0 || ix >= w) continue;
                            acc += (double)input[((long)ci * h + iy) * w + ix] *
                                   (double)weight[(((long)co * ni + ci) * kh + ky) * kw + kx];
                        }
                    }
                float v = (float)acc;
                output[((long)co * h + y) * w + x] = (relu && v <= 0) ? 0 : v;
            }
}
