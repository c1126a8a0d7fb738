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

// ---------------------------------------------------------------------------- attention, N<=64, d=64, on the matrix cores
// softmax(q k^T * scale) v per (batch, head) with v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulate: the character
// indices downstream must stay bit-exact, so no reduced precision here).  One workgroup per (batch, head); Q, K, V rows in LDS
// with a row stride of 68 floats (the MFMA operand reads — 16 rows x 4 consecutive k — then hit 64 distinct banks).
// Wave w owns query rows 16w..16w+15:
//   S = Q K^T : A = Q[row l16][k = 4 ks + g], B = K[key l16][k = 4 ks + g]  → acc[kb][q] = S[row 4g+q][key 16 kb + l16]
//   softmax   : a query row lives in the 16 lanes of one g-group x 4 key blocks → in-thread + 4 xor-shuffles (1,2,4,8)
//   O = P V   : P goes back through the wave's own Q rows (D layout → A layout), B = V[key 4 ks + g][col 16 cb + l16]
__global__ void __launch_bounds__(256) attention_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                        int N, int H, float scale) {
    constexpr int LD = 68;
    __shared__ float sq[64 * LD], sk[64 * LD], sv[64 * LD];
    const int b = blockIdx.x / H, h = blockIdx.x % H, t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6, l16 = lane & 15, g = lane >> 4;
    const int HD = H * 64;
    const float* base = qkv + (size_t)b * N * 3 * HD + h * 64;
    for (int e = t; e < 64 * 64; e += 256) {
        const int r = e >> 6, c = e & 63;
        float q = 0.f, k = 0.f, v = 0.f;
        if (r < N) { const float* row = base + (size_t)r * 3 * HD; q = row[c]; k = row[HD + c]; v = row[2 * HD + c]; }
        sq[r * LD + c] = q; sk[r * LD + c] = k; sv[r * LD + c] = v;
    }
    __syncthreads();
    const int nb = (N + 15) >> 4;                        // 16-row blocks of queries == of keys
    const bool live = wave < nb;                         // wave-uniform
    f32x4 s[4];
    if (live) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) s[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* qa = sq + (wave * 16 + l16) * LD + g;
        const float* ka = sk + l16 * LD + g;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const float a = qa[4 * ks];
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
                if (kb < nb) s[kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, ka[kb * 16 * LD + 4 * ks], s[kb], 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float mx = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const bool ok = kb * 16 + l16 < N;
                s[kb][q] = ok ? s[kb][q] * scale : -INFINITY;
                mx = fmaxf(mx, s[kb][q]);
            }
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
            float sum = 0.f;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const float e = kb * 16 + l16 < N ? expf(s[kb][q] - mx) : 0.f;
                s[kb][q] = e; sum += e;
            }
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) sum += __shfl_xor(sum, o, 64);
            const float inv = 1.f / sum;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) s[kb][q] *= inv;
        }
    }
    __syncthreads();                                     // every wave is done reading Q (only its own rows, but keep it simple)
    if (live) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) sq[(wave * 16 + 4 * g + q) * LD + kb * 16 + l16] = s[kb][q];
    }
    __syncthreads();
    if (!live) return;
    f32x4 o[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) o[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* pa = sq + (wave * 16 + l16) * LD + g;
    const float* va = sv + g * LD + l16;
    for (int ks = 0; ks < nb * 4; ++ks) {                // keys beyond N carry P == 0 and V == 0
        const float a = pa[4 * ks];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) o[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, va[4 * ks * LD + cb * 16], o[cb], 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = wave * 16 + 4 * g + q;
        if (row >= N) continue;
        float* op = out + ((size_t)b * N + row) * HD + h * 64 + l16;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) op[cb * 16] = o[cb][q];
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
