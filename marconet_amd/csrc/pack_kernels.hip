// One-time weight packing on the device (SURVEY.md §2a K18, §8b `mnet_pack_weights`) and the small row gathers of the generator's
// style plumbing.  Nothing here is on the per-pixel hot path; it exists so that a host in any language can go from the
// checkpoint's tensors (fp32, OIHW, spectral-norm u / v vectors) to the layouts the conv kernels read without PyTorch or a
// BLAS: the spectral-norm fold the reference redoes on every forward (models/networks.py:14 — W_orig / (u^T W_mat v), 211
// addmv + 275 div launches per SR forward) happens once, here.
#include "common.h"

// ---------------------------------------------------------------------------- sigma = u^T (W_mat v), fp64 sums, fixed order
// stage 1: one workgroup per output channel o: partial[o] = u[o] * sum_k W[o][k] v[k]
__global__ void __launch_bounds__(256) sn_rowdot_kernel(const float* __restrict__ w, const float* __restrict__ u,
                                                        const float* __restrict__ v, double* __restrict__ partial, int K) {
    __shared__ double red[4];
    const int o = blockIdx.x, t = threadIdx.x;
    const float* row = w + (size_t)o * K;
    double s = 0.0;
    for (int k = t; k < K; k += 256) s += (double)row[k] * (double)v[k];
    s = wave_sum_d(s);
    if ((t & 63) == 0) red[t >> 6] = s;
    __syncthreads();
    if (t == 0) partial[o] = (double)u[o] * ((red[0] + red[1]) + (red[2] + red[3]));
}
// stage 2: one thread folds the rows in order → sigma (fp32, like the reference's parametrisation)
__global__ void sn_fold_kernel(const double* __restrict__ partial, int cout, float* __restrict__ sigma) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0.0;
    for (int o = 0; o < cout; ++o) s += partial[o];
    sigma[0] = (float)s;
}

// ---------------------------------------------------------------------------- OIHW fp32 → [cout_pad][kh][kw][cin_pad] T
// one thread per packed element (coalesced writes; the strided reads of a few-MB tensor do not matter once per load)
template <typename T>
__global__ void __launch_bounds__(256) pack_weights_kernel(const float* __restrict__ w, T* __restrict__ dst, int cout, int cin,
                                                           int kh, int kw, int cout_pad, int cin_pad, float scale,
                                                           const float* __restrict__ sigma, long long total) {
    const float sg = sigma ? sigma[0] : 1.0f;
    for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long long)gridDim.x * 256) {
        const int i = (int)(id % cin_pad);
        long long r = id / cin_pad;
        const int s = (int)(r % kw); r /= kw;
        const int rr = (int)(r % kh);
        const int o = (int)(r / kh);
        float v = 0.f;
        if (o < cout && i < cin) {
            v = w[(((size_t)o * cin + i) * kh + rr) * kw + s];
            if (sigma) v = v / sg;                     // W_orig / sigma first (the reference's weight, an fp32 division), then the layer's constant scale
            v *= scale;
        }
        st_elem<T>(dst + (id - i), i, v);              // dst + first element of this (o, r, s) row; st_elem handles the split layout
    }
}

// fp16+8 (MNET_F16M) conv weights, layout in include/marconet_hip.h: one workgroup per (padded) output channel — pass 1 the row's
// max |f16(256 W)| → the row scale s = 2^(floor(log2 max) - 7); pass 2 one thread per 8-channel chunk: hi halves, lo8 = e4m3(lo * 2^11 / s),
// hi8 = e4m3(hi / s); then the row's byte E8M0(s * 2^-11) after the cout_pad * K elements
__global__ void __launch_bounds__(256) pack_weights_mx_kernel(const float* __restrict__ w, unsigned char* __restrict__ dst, int cout, int cin,
                                                              int kh, int kw, int cout_pad, int cin_pad, float scale,
                                                              const float* __restrict__ sigma) {
    __shared__ float red[4];
    const int o = blockIdx.x, t = threadIdx.x;
    const float sg = sigma ? sigma[0] : 1.0f;
    const int K = kh * kw * cin_pad;
    auto value = [&](int k) -> float {
        const int i = k % cin_pad, tap = k / cin_pad;
        if (o >= cout || i >= cin) return 0.f;
        float v = w[(((size_t)o * cin + i) * kh + tap / kw) * kw + tap % kw];
        if (sigma) v = v / sg;
        return v * scale;
    };
    float m = 0.f;
    for (int k = t; k < K; k += 256) m = fmaxf(m, fabsf((float)(f16)value(k)));
    m = wave_max(m);
    if ((t & 63) == 0) red[t >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const int e8 = min(254, max(11, hm_e8_raw(m)));
    const float inv_hi = __builtin_bit_cast(float, (unsigned)(254 - e8) << 23);         // 1 / s
    unsigned char* row = dst + (size_t)o * K * 4;
    for (int ch = t; ch < K / 8; ch += 256) {
        const int s = ch & 3;
        unsigned char* blk = row + (size_t)(ch >> 2) * 128;
        float v[8];
        f16x8 h;
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[j] = value(ch * 8 + j); h[j] = (f16)v[j]; }
        stg16(blk + s * 16, bitcast<u32x4>(h));
        u32x2 hi8;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            int q = 0;
            q = __builtin_amdgcn_cvt_pk_fp8_f32((float)h[4 * d] * inv_hi, (float)h[4 * d + 1] * inv_hi, q, false);
            q = __builtin_amdgcn_cvt_pk_fp8_f32((float)h[4 * d + 2] * inv_hi, (float)h[4 * d + 3] * inv_hi, q, true);
            hi8[d] = (unsigned)q;
        }
        unsigned char* f8 = blk + 64 + (s & 1) * 32 + (s >> 1) * 8;
        *reinterpret_cast<u32x2*>(f8) = hm_encode_lo(v, h, e8);
        *reinterpret_cast<u32x2*>(f8 + 16) = hi8;
    }
    if (t == 0) dst[(size_t)cout_pad * K * 4 + o] = (unsigned char)(e8 - 11);
}

extern "C" int mnet_pack_weights(const float* w_oihw, int32_t cout, int32_t cin, int32_t kh, int32_t kw, const float* sn_u,
                                 const float* sn_v, float scale, int32_t dtype, int32_t cout_pad, int32_t cin_pad, void* packed,
                                 double* workspace, void* stream) {
    MNET_CHECK_ARG(w_oihw && packed && cout > 0 && cin > 0 && kh > 0 && kw > 0, "pack_weights: bad args");
    MNET_CHECK_ARG(cout_pad >= cout && cin_pad >= cin, "pack_weights: padded sizes smaller than the tensor");
    MNET_CHECK_ARG(dtype == MNET_F32 || dtype == MNET_F16 || dtype == MNET_F16X2 || dtype == MNET_F16M, "pack_weights: bad dtype");
    MNET_CHECK_ARG((sn_u == nullptr) == (sn_v == nullptr), "pack_weights: sn_u and sn_v go together");
    MNET_CHECK_ARG(!sn_u || workspace, "pack_weights: the spectral-norm fold needs a workspace of cout + 1 doubles");
    MNET_CHECK_ALIGN((dtype != MNET_F16X2 && dtype != MNET_F16M) || (cin_pad % 32 == 0 && aligned128(packed)), "pack_weights: split-half needs cin_pad %% 32 == 0, 128-byte aligned");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    float* sigma = nullptr;
    if (sn_u) {
        sigma = reinterpret_cast<float*>(workspace + cout);
        hipLaunchKernelGGL(sn_rowdot_kernel, dim3(cout), dim3(256), 0, st, w_oihw, sn_u, sn_v, workspace, cin * kh * kw);
        MNET_LAUNCH_CHECK("sn_rowdot");
        hipLaunchKernelGGL(sn_fold_kernel, dim3(1), dim3(64), 0, st, workspace, cout, sigma);
        MNET_LAUNCH_CHECK("sn_fold");
    }
    if (dtype == MNET_F16M) {           // `packed` holds cout_pad * kh * kw * cin_pad * 4 + cout_pad bytes
        hipLaunchKernelGGL(pack_weights_mx_kernel, dim3(cout_pad), dim3(256), 0, st, w_oihw, (unsigned char*)packed, cout, cin, kh, kw, cout_pad,
                           cin_pad, scale * MNET_SPLIT_WSCALE, sigma);
        MNET_LAUNCH_CHECK("pack_weights_mx");
        return MNET_OK;
    }
    const long long total = (long long)cout_pad * kh * kw * cin_pad;
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    if (dtype == MNET_F16X2) {
        // hi / lo of MNET_SPLIT_WSCALE * W (exponent offset, undone by the conv epilogue)
        hipLaunchKernelGGL(pack_weights_kernel<hs>, dim3(blocks), dim3(256), 0, st, w_oihw, (hs*)packed, cout, cin, kh, kw, cout_pad, cin_pad,
                           scale * MNET_SPLIT_WSCALE, sigma, total);
    } else if (dtype == MNET_F16) {
        hipLaunchKernelGGL(pack_weights_kernel<f16>, dim3(blocks), dim3(256), 0, st, w_oihw, (f16*)packed, cout, cin, kh, kw, cout_pad, cin_pad, scale, sigma, total);
    } else {
        hipLaunchKernelGGL(pack_weights_kernel<float>, dim3(blocks), dim3(256), 0, st, w_oihw, (float*)packed, cout, cin, kh, kw, cout_pad, cin_pad, scale, sigma, total);
    }
    MNET_LAUNCH_CHECK("pack_weights");
    return MNET_OK;
}

// ---------------------------------------------------------------------------- demodulation table
// wsq_t[i][o] = scale^2 * sum_{r,s} W[o][i][r][s]^2   (ModulatedConv2d, models/networks.py:284-287, for activation-side modulation)
__global__ void __launch_bounds__(256) pack_wsq_kernel(const float* __restrict__ w, float* __restrict__ wsq_t, int cout, int cin,
                                                       int khw, float scale2) {
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= cout * cin) return;
    const int o = id % cout, i = id / cout;              // consecutive threads → consecutive o: coalesced writes
    const float* p = w + ((size_t)o * cin + i) * khw;
    float s = 0.f;
    for (int k = 0; k < khw; ++k) s = fmaf(p[k] * scale2, p[k], s);   // (scale*w)^2 summed; scale2 = scale^2
    wsq_t[(size_t)i * cout + o] = s;
}

extern "C" int mnet_pack_wsq(const float* w_oihw, int32_t cout, int32_t cin, int32_t khw, float scale, float* wsq_t, void* stream) {
    MNET_CHECK_ARG(w_oihw && wsq_t && cout > 0 && cin > 0 && khw > 0, "pack_wsq: bad args");
    const int tot = cout * cin;
    hipLaunchKernelGGL(pack_wsq_kernel, dim3((tot + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w_oihw, wsq_t,
                       cout, cin, khw, scale * scale);
    MNET_LAUNCH_CHECK("pack_wsq");
    return MNET_OK;
}

// ---------------------------------------------------------------------------- row gather with a column window
// dst[r][0..ncols) = src[idx ? idx[r] : r][col0 .. col0 + ncols)   (fp32; src row stride ld).  The generator computes the style
// MLP / modulations / demodulation once per distinct style (one per image) and hands every glyph its image's row, and every
// StyledConv its own column window of the one batched modulation GEMM (models/networks.py:141,283).
__global__ void __launch_bounds__(256) gather_rows_kernel(const float* __restrict__ src, int ld, int col0, int ncols,
                                                          const int64_t* __restrict__ idx, float* __restrict__ dst, long long total) {
    for (long long id = (long long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long long)gridDim.x * 256) {
        const int c = (int)(id % ncols);
        const long long r = id / ncols;
        const long long sr = idx ? idx[r] : r;
        dst[id] = src[(size_t)sr * ld + col0 + c];
    }
}

// ---------------------------------------------------------------------------- style rows, normalised by a power of two
// dst[r][:] = src[idx ? idx[r] : r][col0 .. col0+ncols) * 2^-e[r],  e[r] = exponent of max|row window| (max * 2^-e in [0.5, 1); e = 0 for
// an all-zero row).  The modulated activations x * s are then bounded by |x| whatever the style's magnitude — the half-precision
// hazard of StyleGAN-type generators — and nothing else changes: the demodulation rsqrt(sum (W s)^2 + eps) absorbs the factor
// exactly when its eps is scaled by 4^-e (eps_scale), a conv without demodulation (ToRGB) multiplies its accumulator by 2^e
// (scale_b, broadcast over `bcast` output channels so that it can be passed as out_scale [rows][bcast]).  Power-of-two factors:
// every fp32 product and sum is scaled exactly, results are bit-identical to the un-normalised evaluation.
__global__ void __launch_bounds__(256) style_rows_kernel(const float* __restrict__ src, int ld, int col0, int ncols,
                                                         const int64_t* __restrict__ idx, float* __restrict__ dst,
                                                         float* __restrict__ eps_scale, float* __restrict__ scale_b, int bcast) {
    __shared__ float red[4];
    const int r = blockIdx.x, t = threadIdx.x;
    const float* row = src + (size_t)(idx ? idx[r] : r) * ld + col0;
    float m = 0.f;
    for (int c = t; c < ncols; c += 256) m = fmaxf(m, fabsf(row[c]));
    m = wave_max(m);
    if ((t & 63) == 0) red[t >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    int e = 0;
    if (m > 0.f && m < INFINITY) (void)frexpf(m, &e);
    const float down = ldexpf(1.f, -e), up = ldexpf(1.f, e);
    for (int c = t; c < ncols; c += 256) dst[(size_t)r * ncols + c] = row[c] * down;
    if (t == 0 && eps_scale) eps_scale[r] = down * down;
    if (scale_b) for (int c = t; c < bcast; c += 256) scale_b[(size_t)r * bcast + c] = up;
}

extern "C" int mnet_style_rows(const float* src, int32_t src_rows, int32_t ld, int32_t col0, int32_t ncols, const int64_t* idx,
                               int32_t rows, float* dst, float* eps_scale, float* scale_b, int32_t bcast, void* stream) {
    MNET_CHECK_ARG(src && dst && src_rows > 0 && rows > 0 && ncols > 0 && col0 >= 0 && col0 + ncols <= ld, "style_rows: bad args");
    MNET_CHECK_ARG(!scale_b || bcast > 0, "style_rows: bcast must be positive with scale_b");
    hipLaunchKernelGGL(style_rows_kernel, dim3(rows), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, ld, col0, ncols, idx, dst,
                       eps_scale, scale_b, bcast);
    MNET_LAUNCH_CHECK("style_rows");
    return MNET_OK;
}

extern "C" int mnet_gather_rows(const float* src, int32_t src_rows, int32_t ld, int32_t col0, int32_t ncols, const int64_t* idx,
                                int32_t rows, float* dst, void* stream) {
    MNET_CHECK_ARG(src && dst && src_rows > 0 && rows > 0 && ncols > 0 && col0 >= 0 && col0 + ncols <= ld, "gather_rows: bad args");
    const long long total = (long long)rows * ncols;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, ld, col0, ncols, idx, dst, total);
    MNET_LAUNCH_CHECK("gather_rows");
    return MNET_OK;
}
