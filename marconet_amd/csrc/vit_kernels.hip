// TextViT helper kernels (fp32): LayerNorm, token-axis LayerNorm+Linear, small-sequence attention, argmax.
// The ViT is 0.2 % of the path's FLOPs; its GEMMs run on the fp32 MFMA path of
// conv_igemm.hip, these kernels keep all statistics / softmax in fp32 so that argmax(logits) stays
// bit-exact with the CPU reference.
#include "common.h"

// ---------------------------------------------------------------------------- LayerNorm: one wave per row
// D <= 1024: each lane keeps D/64 values in registers, two-pass mean / biased variance like ATen's CPU kernel.
template <int PER>
__global__ void __launch_bounds__(256) layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ y,
                                                        int rows, int D, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (size_t)row * D;
    float v[PER];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) { const int c = lane + 64 * i; v[i] = c < D ? xr[c] : 0.f; s += v[i]; }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) { const int c = lane + 64 * i; const float d = c < D ? v[i] - mean : 0.f; q += d * d; }
    const float rstd = 1.f / sqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = lane + 64 * i;
        if (c < D) y[(size_t)row * D + c] = (v[i] - mean) * rstd * gamma[c] + beta[c];
    }
}

extern "C" int mnet_layernorm(const float* x, const float* gamma, const float* beta, float* y, int32_t rows,
                              int32_t d, float eps, void* stream) {
    MNET_CHECK_ARG(x && gamma && beta && y && rows > 0 && d > 0 && d <= 1024, "layernorm: bad args (d<=1024)");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    dim3 grid((rows + 3) / 4), block(256);
    if (d <= 64) hipLaunchKernelGGL(layernorm_kernel<1>, grid, block, 0, st, x, gamma, beta, y, rows, d, eps);
    else if (d <= 512) hipLaunchKernelGGL(layernorm_kernel<8>, grid, block, 0, st, x, gamma, beta, y, rows, d, eps);
    else hipLaunchKernelGGL(layernorm_kernel<16>, grid, block, 0, st, x, gamma, beta, y, rows, d, eps);
    MNET_LAUNCH_CHECK("layernorm");
    return MNET_OK;
}

// ---------------------------------------------------------------------------- token-axis LN + Linear
// thread per (b, d): the T (<=64) tokens of channel d are strided by D in memory, so consecutive threads
// (consecutive d) read consecutive addresses — coalesced without a transpose.
__global__ void __launch_bounds__(256) token_mix_kernel(const float* __restrict__ x, const float* __restrict__ ln_g,
                                                        const float* __restrict__ ln_b, const float* __restrict__ wgt,
                                                        const float* __restrict__ bias, float* __restrict__ y,
                                                        int B, int T, int D, int J, float eps) {
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= B * D) return;
    const int b = id / D, d = id - b * D;
    float v[64];
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 64; ++t) { v[t] = t < T ? x[((size_t)b * T + t) * D + d] : 0.f; s += v[t]; }
    const float mean = s / (float)T;
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < 64; ++t) { const float e = t < T ? v[t] - mean : 0.f; q += e * e; }
    const float rstd = 1.f / sqrtf(q / (float)T + eps);
#pragma unroll
    for (int t = 0; t < 64; ++t) v[t] = t < T ? (v[t] - mean) * rstd * ln_g[t] + ln_b[t] : 0.f;
    for (int j = 0; j < J; ++j) {
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < 64; ++t) acc = fmaf(v[t], t < T ? wgt[j * T + t] : 0.f, acc);
        y[((size_t)b * J + j) * D + d] = acc + bias[j];
    }
}

extern "C" int mnet_token_mix(const float* x, const float* ln_g, const float* ln_b, const float* wgt,
                              const float* bias, float* y, int32_t B, int32_t T, int32_t D, int32_t J, float eps,
                              void* stream) {
    MNET_CHECK_ARG(x && ln_g && ln_b && wgt && bias && y, "token_mix: null pointer");
    MNET_CHECK_ARG(B > 0 && T > 0 && T <= 64 && D > 0 && J > 0, "token_mix: bad geometry (T<=64)");
    const int tot = B * D;
    hipLaunchKernelGGL(token_mix_kernel, dim3((tot + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       x, ln_g, ln_b, wgt, bias, y, B, T, D, J, eps);
    MNET_LAUNCH_CHECK("token_mix");
    return MNET_OK;
}

// ---------------------------------------------------------------------------- attention, N<=64, d=64
// one workgroup per (batch, head): Q,K,V rows in LDS (row stride 65 floats → conflict-free column walks),
// thread (i = t/4, quarter = t%4) owns 16 score columns of query row i; row max / sum reduce over the
// 4 neighbouring lanes with xor-shuffles; P goes back through LDS for the P·V product.
__global__ void __launch_bounds__(256) attention_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                        int N, int H, float scale) {
    __shared__ float sq[64][65], sk[64][65], sv[64][65];
    float (*sp)[65] = sq;   // P overwrites Q once every thread is done with Q (barrier below)
    const int b = blockIdx.x / H, h = blockIdx.x % H, t = threadIdx.x;
    const int HD = H * 64;
    const float* base = qkv + (size_t)b * N * 3 * HD + h * 64;
    for (int e = t; e < N * 64; e += 256) {
        const int r = e >> 6, c = e & 63;
        const float* row = base + (size_t)r * 3 * HD;
        sq[r][c] = row[c]; sk[r][c] = row[HD + c]; sv[r][c] = row[2 * HD + c];
    }
    __syncthreads();
    const int i = t >> 2, qd = t & 3;
    float s[16];
    float mx = -INFINITY;
    if (i < N) {
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
            const int j = qd * 16 + jj;
            float acc = 0.f;
            if (j < N) {
                for (int c = 0; c < 64; ++c) acc = fmaf(sq[i][c], sk[j][c], acc);
                acc *= scale;
                mx = fmaxf(mx, acc);
            }
            s[jj] = acc;
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
    float sum = 0.f;
    if (i < N) {
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
            const int j = qd * 16 + jj;
            const float e = j < N ? expf(s[jj] - mx) : 0.f;
            s[jj] = e; sum += e;
        }
    }
    sum += __shfl_xor(sum, 1, 64);
    sum += __shfl_xor(sum, 2, 64);
    __syncthreads();
    if (i < N) {
        const float inv = 1.f / sum;
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) sp[i][qd * 16 + jj] = s[jj] * inv;
    }
    __syncthreads();
    if (i < N) {
        float* o = out + ((size_t)b * N + i) * HD + h * 64 + qd * 16;
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) {
            const int c = qd * 16 + cc;
            float acc = 0.f;
            for (int j = 0; j < N; ++j) acc = fmaf(sp[i][j], sv[j][c], acc);
            o[cc] = acc;
        }
    }
}

extern "C" int mnet_attention(const float* qkv, float* out, int32_t B, int32_t N, int32_t H, float scale, void* stream) {
    MNET_CHECK_ARG(qkv && out && B > 0 && N > 0 && N <= 64 && H > 0, "attention: bad args (N<=64)");
    hipLaunchKernelGGL(attention_kernel, dim3(B * H), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), qkv, out, N, H, scale);
    MNET_LAUNCH_CHECK("attention");
    return MNET_OK;
}

// ---------------------------------------------------------------------------- argmax (first max index)
__global__ void __launch_bounds__(256) argmax_kernel(const float* __restrict__ x, int64_t* __restrict__ idx, int rows, int D) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (size_t)row * D;
    float best = -INFINITY; int bi = 0x7fffffff;
    for (int i = lane; i < D; i += 64) { const float v = xr[i]; if (v > best) { best = v; bi = i; } }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) idx[row] = bi;
}

extern "C" int mnet_argmax_rows(const float* x, int64_t* idx, int32_t rows, int32_t d, void* stream) {
    MNET_CHECK_ARG(x && idx && rows > 0 && d > 0, "argmax: bad args");
    hipLaunchKernelGGL(argmax_kernel, dim3((rows + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, idx, rows, d);
    MNET_LAUNCH_CHECK("argmax");
    return MNET_OK;
}
