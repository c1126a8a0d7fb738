/* A plain-C host of libmarconet_hip.so (no Python, no PyTorch): packs an OIHW fp32 weight with the library's own packer
 * (spectral-norm fold + scale + layout change, mnet_pack_weights) and runs one 3x3 convolution with a fused bias + LeakyReLU
 * epilogue through mnet_conv2d_nhwc, in the exact-fp32 mode and in the split-half (fp16x3) mode, then checks both against a
 * scalar loop on the host.  This is the binding a non-Python integrator writes against include/marconet_hip.h.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_host_conv.c \
 *       -Lmarconet_amd/lib -lmarconet_hip -L/opt/rocm/lib -lamdhip64 -lm -o c_host_conv
 *   LD_LIBRARY_PATH=marconet_amd/lib:/opt/rocm/lib ./c_host_conv
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "marconet_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_MNET(x) do { int r_ = (x); if (r_ != MNET_OK) { fprintf(stderr, "%s: %d %s\n", #x, r_, mnet_last_error()); return 3; } } while (0)

enum { N = 2, H = 12, W = 20, CIN = 64, COUT = 96, K = 3 };

static float frand(unsigned* s) { *s = *s * 1664525u + 1013904223u; return (float)((*s >> 8) & 0xffff) / 65536.0f - 0.5f; }

int main(void) {
    const size_t nx = (size_t)N * H * W * CIN, nw = (size_t)COUT * CIN * K * K, ny = (size_t)N * H * W * COUT;
    float* x = malloc(nx * 4), *w = malloc(nw * 4), *u = malloc(COUT * 4), *v = malloc(CIN * K * K * 4), *b = malloc(COUT * 4);
    float* ref = malloc(ny * 4), *got = malloc(ny * 4);
    unsigned seed = 7;
    for (size_t i = 0; i < nx; ++i) x[i] = frand(&seed);                     /* NHWC */
    for (size_t i = 0; i < nw; ++i) w[i] = frand(&seed) * 0.1f;              /* OIHW, like the checkpoint */
    for (int i = 0; i < COUT; ++i) { u[i] = frand(&seed); b[i] = frand(&seed); }
    for (int i = 0; i < CIN * K * K; ++i) v[i] = frand(&seed);
    /* host reference: sigma = u^T (W_mat v); y = lrelu_0.2(conv(x, W / sigma) + b) */
    double sigma = 0.0;
    for (int o = 0; o < COUT; ++o) { double r = 0.0; for (int k = 0; k < CIN * K * K; ++k) r += (double)w[(size_t)o * CIN * K * K + k] * v[k]; sigma += u[o] * r; }
    for (int n = 0; n < N; ++n) for (int oh = 0; oh < H; ++oh) for (int ow = 0; ow < W; ++ow) for (int o = 0; o < COUT; ++o) {
        double acc = 0.0;
        for (int r = 0; r < K; ++r) for (int s = 0; s < K; ++s) {
            const int ih = oh + r - 1, iw = ow + s - 1;
            if (ih < 0 || ih >= H || iw < 0 || iw >= W) continue;
            for (int c = 0; c < CIN; ++c) acc += (double)x[(((size_t)n * H + ih) * W + iw) * CIN + c] * (w[(((size_t)o * CIN + c) * K + r) * K + s] / (float)sigma);
        }
        acc += b[o];
        ref[(((size_t)n * H + oh) * W + ow) * COUT + o] = (float)(acc > 0 ? acc : 0.2 * acc);
    }
    if (mnet_abi_version() != MNET_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
    float *dx, *dw, *du, *dv, *db, *dy32; void *dwp32, *dwp3, *dx3, *dy3; double* ws; hipStream_t st;
    CHECK_HIP(hipStreamCreate(&st));
    CHECK_HIP(hipMalloc((void**)&dx, nx * 4)); CHECK_HIP(hipMalloc((void**)&dw, nw * 4)); CHECK_HIP(hipMalloc((void**)&du, COUT * 4));
    CHECK_HIP(hipMalloc((void**)&dv, CIN * K * K * 4)); CHECK_HIP(hipMalloc((void**)&db, COUT * 4)); CHECK_HIP(hipMalloc((void**)&dy32, ny * 4));
    CHECK_HIP(hipMalloc(&dwp32, nw * 4)); CHECK_HIP(hipMalloc(&dwp3, nw * 4)); CHECK_HIP(hipMalloc(&dx3, nx * 4)); CHECK_HIP(hipMalloc(&dy3, ny * 4));
    CHECK_HIP(hipMalloc((void**)&ws, (COUT + 1) * 8));
    CHECK_HIP(hipMemcpy(dx, x, nx * 4, hipMemcpyHostToDevice)); CHECK_HIP(hipMemcpy(dw, w, nw * 4, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(du, u, COUT * 4, hipMemcpyHostToDevice)); CHECK_HIP(hipMemcpy(dv, v, CIN * K * K * 4, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(db, b, COUT * 4, hipMemcpyHostToDevice));
    double worst[2] = {0.0, 0.0};
    for (int mode = 0; mode < 2; ++mode) {
        const int dt = mode == 0 ? MNET_F32 : MNET_F16X2;
        void* wp = mode == 0 ? dwp32 : dwp3;
        CHECK_MNET(mnet_pack_weights(dw, COUT, CIN, K, K, du, dv, 1.0f, dt, COUT, CIN, wp, ws, st));
        const void* xin = dx; void* yout = dy32;
        if (mode == 1) { CHECK_MNET(mnet_convert(dx, MNET_F32, dx3, MNET_F16X2, (int64_t)nx, st)); xin = dx3; yout = dy3; }
        mnet_conv_desc d = {0};
        d.dtype = dt; d.x0 = xin; d.c0 = CIN; d.n = N; d.h = H; d.w = W; d.wgt = wp; d.cout = COUT; d.kh = d.kw = K;
        d.stride_h = d.stride_w = 1; d.pad_h = d.pad_w = 1; d.ho = H; d.wo = W; d.bias = db; d.act = MNET_ACT_LRELU; d.y = yout;
        CHECK_MNET(mnet_conv2d_nhwc(&d, st));
        if (mode == 1) CHECK_MNET(mnet_convert(dy3, MNET_F16X2, dy32, MNET_F32, (int64_t)ny, st));
        CHECK_HIP(hipStreamSynchronize(st));
        CHECK_HIP(hipMemcpy(got, dy32, ny * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < ny; ++i) { const double e = fabs((double)got[i] - ref[i]); if (e > worst[mode]) worst[mode] = e; }
    }
    printf("c_host_conv: max-abs vs host loop  fp32 %.3e  fp16x3 %.3e\n", worst[0], worst[1]);
    return (worst[0] <= 1e-4 && worst[1] <= 1e-4) ? 0 : 4;
}
