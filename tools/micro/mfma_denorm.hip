// Micro-test (round 4): does v_mfma_f32_32x32x16_f16 honour f16 SUBNORMAL operands on gfx950, or flush them?
// cb_split.hip drops every term of an f16 pair that would be subnormal before it forms the residual, so that the
// result does not depend on the answer -- at the price of an absolute floor (2^-21 per activation) instead of a
// relative bound for small values.  If the matrix unit honours subnormals the drop can go.
//   A = a (every element), B = b (every element): D = 16 a b per element if nothing is flushed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(64) void k(float a, float b, float* out) {
    halfx8 A, B;
    for (int i = 0; i < 8; ++i) A[i] = (_Float16)a, B[i] = (_Float16)b;
    floatx16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = acc[0];
}
int main() {
    float* out;
    (void)hipMalloc(&out, 4);
    const float cases[][2] = {{ldexpf(1.f, -20), 1.f},          // subnormal x normal
                              {1.f, ldexpf(1.f, -24)},          // normal x smallest subnormal
                              {ldexpf(1.f, -16), ldexpf(1.f, -16)},   // subnormal x subnormal (2^-32 per product)
                              {ldexpf(1.f, -14), 1.f},          // smallest normal (control)
                              {ldexpf(3.f, -24), 1024.f}};
    int honoured = 1;
    for (auto& c : cases) {
        k<<<1, 64>>>(c[0], c[1], out);
        float h = 0.f;
        if (hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("failed\n"); return 1; }
        const double want = 16.0 * (double)(float)(_Float16)c[0] * (double)(float)(_Float16)c[1];
        printf("a = %.3e  b = %.3e : D = %.9e  (unflushed: %.9e) %s\n", c[0], c[1], h, want,
               h == (float)want ? "exact" : "DIFFERS");
        honoured &= h == (float)want;
    }
    printf("f16 subnormal operands %s\n", honoured ? "HONOURED" : "FLUSHED (or rounded)");
    return 0;
}
