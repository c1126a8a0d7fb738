// Probe: does v_mfma_f32_16x16x32_f16 honour fp16 subnormal inputs, and does the float->half conversion keep subnormals?
// (Decides whether the split-precision hi/lo activations need an exponent offset on the lo part.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(float* out) {
    const int lane = threadIdx.x;
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (f16)0.f; b[j] = (f16)0.f; }
    // A row (lane&15), k = 8*(lane>>4)+j ; B col (lane&15), same k
    const float tiny = 9.5367431640625e-07f;       // 2^-20: an fp16 subnormal
    if (lane == 0) { a[0] = (f16)tiny; b[0] = (f16)1024.f; }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    if (lane == 0) {
        out[0] = acc[0];                           // expect 2^-10 = 9.765625e-4 when subnormals are honoured
        out[1] = (float)(f16)tiny;                 // conversion keeps the subnormal?
        const float x = 0.0123456789f;
        const f16 hi = (f16)x; const f16 lo = (f16)(x - (float)hi);
        out[2] = (float)hi + (float)lo - x;        // residual of a hi/lo split of a small value
        out[3] = (float)lo;
    }
}

int main() {
    float* d; float h[4];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mfma(subnormal 2^-20 * 1024) = %.9g (expect 9.765625e-04)\n", h[0]);
    printf("f16(2^-20) = %.9g (expect 9.53674316e-07)\n", h[1]);
    printf("hi+lo-x = %.9g  lo = %.9g\n", h[2], h[3]);
    return 0;
}
