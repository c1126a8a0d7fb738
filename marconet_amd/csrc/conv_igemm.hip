// Implicit-GEMM convolution for gfx950 (MI355X, CDNA4), the GENERAL (register-staged) kernel, and the dispatch of
// mnet_conv2d_nhwc between it and the two LDS-DMA kernels (conv_igemm_dma.hip, conv_strip_dma.hip), which carry ~97 % of
// the f16 FLOPs.  This kernel serves what those cannot: every fp32 launch (parity mode, the TextViT linears), input-side
// transforms (style modulation of the first StyledConv / ToRGB, GroupNorm prologue), channel counts that are not multiples
// of 64, cout < 64, tanh / GELU / sigmoid epilogues.
//
// GEMM view (computed transposed so that every lane ends up owning 4 *consecutive output channels*
// of one pixel, which makes bias / demod / residual loads and the NHWC store 8- or 16-byte vector ops):
//     D[cout][pixel] = sum_k  W[cout][k] * X[pixel][k],     k = (r, s, cin)  — NHWC makes cin contiguous
//   MFMA A operand = weight fragment (16 cout x K), B operand = activation fragment (K x 16 pixels),
//   v_mfma_f32_16x16x32_f16 (f16 storage, fp32 accumulate) or v_mfma_f32_16x16x4_f32 (exact fp32).
//
// Tiling: one 256-thread workgroup (4 waves) computes a BC(cout) x BP(pixel) tile; K is walked in
// 128-byte slabs (64 halves / 32 floats).  Both operands are staged through LDS as [rows][128 B] with a
// 16-byte-chunk XOR swizzle (chunk ^= (row>>1)&7) so that the ds_read_b128 fragment reads — 16 rows at
// one logical chunk per lane group — hit 16 distinct 16-byte slots of the 256-byte bank row, and the
// ds_write_b128 staging writes (8 consecutive lanes = one 128-byte row) stay conflict-free too.
// Global->register loads of slab t+1 are issued before the MFMAs of slab t (register prefetch), the
// LDS is double-buffered, one barrier per slab.  The input-side transforms (modulation scale,
// GroupNorm affine + swish, zero padding, ragged valid width, channel concat) run on the staged
// registers between the global load and the LDS write, so they cost no extra HBM pass.
// Workgroup ids are remapped so that each XCD (private 4 MiB L2) owns a contiguous run of tiles and the
// cout-tiles of one pixel tile are neighbours (the activation slab is then re-read from L2, not HBM).
#include <cstdlib>
#include "common.h"
#include "conv_args.h"

#ifndef MNET_MX_SCALED_DECODE
#define MNET_MX_SCALED_DECODE 0      // 1: decode fp16+8 lo bytes with v_cvt_scalef32_pk_f16_fp8 (A/B build, EXTRA_HIPCC_FLAGS=-DMNET_MX_SCALED_DECODE=1)
#endif

template <typename T> struct Mma;
template <> struct Mma<f16> {
    static __device__ __forceinline__ void run(f32x4& acc, const u32x4& a, const u32x4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(bitcast<f16x8>(a), bitcast<f16x8>(b), acc, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    // lane (l16, g) holds k = 4g..4g+3 of a 16-wide k-step; MFMA j consumes component j of A and B:
    // the k permutation is the same on both operands, so the dot product is unchanged.
    static __device__ __forceinline__ void run(f32x4& acc, const u32x4& a, const u32x4& b) {
        const f32x4 af = bitcast<f32x4>(a), bf = bitcast<f32x4>(b);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0], bf[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1], bf[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[2], bf[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[3], bf[3], acc, 0, 0, 0);
    }
};

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// "physical" element of a tensor as the k-loop sees it.  A split-half tensor (MNET_F16X2, T = hs) is walked as an f16 tensor
// with TWICE the channels: its 128-byte k-slab is one 32-channel block — chunks 0-3 the hi halves, chunks 4-7 the lo halves of
// the same 32 channels — and the host passes c0 / c1 / cin / K in physical (doubled) units.
template <typename T> struct Phys { typedef T type; };
template <> struct Phys<hs> { typedef f16 type; };
template <> struct Phys<hm> { typedef f16 type; };      // fp16+8: staged into LDS as a split-half slab (the lo bytes decoded to halves)

// fp16+8 operands of the register-staged kernel: chunk cc (0-3: hi halves, 4-7: the lo part of channels 8 (cc-4) ..) of the
// 128-byte block at `blk`, in split-half form — the lo bytes become halves (lo8 * s * 2^-11 is exact in f16 unless it underflows),
// and the k-loop then runs the three-product split-half MFMA sequence unchanged.
__device__ __forceinline__ u32x4 hm_lo_halves(u32x2 lo8, float sl) {
#if MNET_MX_SCALED_DECODE
    // v_cvt_scalef32_pk_f16_fp8: two e4m3 bytes -> two halves times 2^(exponent of sl), one instruction per pair
    u32x4 r;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        r[2 * d] = bitcast<unsigned>(__builtin_amdgcn_cvt_scalef32_pk_f16_fp8((int)lo8[d], sl, false));
        r[2 * d + 1] = bitcast<unsigned>(__builtin_amdgcn_cvt_scalef32_pk_f16_fp8((int)lo8[d], sl, true));
    }
    return r;
#else
    float l[8];
    hm_decode_lo(lo8, sl, l);
    return Vec<f16>::pack(l);
#endif
}
__device__ __forceinline__ u32x4 ld_hm_act_chunk(const unsigned char* blk, int cc) {
    if (cc < 4) return ldg16(blk + cc * 16);
    return hm_lo_halves(*reinterpret_cast<const u32x2*>(blk + 64 + hm_lo_slot(cc - 4) * 8), hm_lo_scale(blk[96]));
}
__device__ __forceinline__ u32x4 ld_hm_wgt_chunk(const unsigned char* blk, int cc, int wexp) {      // wexp: E8M0 of s_w * 2^-11
    if (cc < 4) return ldg16(blk + cc * 16);
    const int s = cc - 4;
    return hm_lo_halves(*reinterpret_cast<const u32x2*>(blk + 64 + (s & 1) * 32 + (s >> 1) * 8),
                        wexp > 0 ? __builtin_bit_cast(float, (unsigned)wexp << 23) : 0.f);
}

template <typename T>
__device__ __forceinline__ u32x4 in_transform(u32x4 raw, const float* sc, const float* sh, bool swish) {
    constexpr int N = Vec<T>::N;
    float v[N];
    Vec<T>::unpack(raw, v);
#pragma unroll
    for (int j = 0; j < N; j += 4) {
        const f32x4 s4 = *reinterpret_cast<const f32x4*>(sc + j);
        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
        if (sh) b4 = *reinterpret_cast<const f32x4*>(sh + j);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[j + q] = v[j + q] * s4[q] + b4[q];
    }
    if (swish) {          // one wave-uniform branch, not one per element
#pragma unroll
        for (int j = 0; j < N; ++j) v[j] = v[j] * (1.f / (1.f + expf(-v[j])));
    }
    return Vec<T>::pack(v);
}

// split-half input transform: this thread staged 16-byte chunk cc of a slab row (`own`) and its partner chunk cc ^ 4 (`other`):
// the hi and lo halves of the same 8 channels.  value = hi + lo → affine (+ swish) in fp32 → split again → the thread keeps
// its own half.  sc / sh point at the 8 LOGICAL channels of the chunk.
__device__ __forceinline__ u32x4 in_transform_split(u32x4 own, u32x4 other, bool own_is_lo, const float* sc, const float* sh, bool swish) {
    Raw<hs> r;
    r.hi = own_is_lo ? other : own;
    r.lo = own_is_lo ? own : other;
    float v[8];
    unpackr<hs>(r, v);
#pragma unroll
    for (int j = 0; j < 8; j += 4) {
        const f32x4 s4 = *reinterpret_cast<const f32x4*>(sc + j);
        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
        if (sh) b4 = *reinterpret_cast<const f32x4*>(sh + j);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[j + q] = v[j + q] * s4[q] + b4[q];
    }
    if (swish) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = v[j] * (1.f / (1.f + expf(-v[j])));
    }
    const Raw<hs> o = packr<hs>(v);
    return own_is_lo ? o.lo : o.hi;
}

template <typename T, int BC, int BP, int WC, int WP>
__global__ void __launch_bounds__(256, 2) conv_igemm_kernel(const ConvArgs p) {
    typedef typename Phys<T>::type PT;            // element type the k-loop addresses (f16 for split-half)
    constexpr bool SPLIT = sizeof(T) != sizeof(PT);
    constexpr bool MXS = __is_same(T, hm);         // fp16+8 storage on both sides of a split-half k-loop
    constexpr int KCH = 16 / (int)sizeof(PT);     // elements per 16-byte chunk
    constexpr int BK = 8 * KCH;                   // elements per 128-byte k-slab
    constexpr int FC = BC / WC / 16, FP = BP / WP / 16;
    constexpr int WROWS = (BC + 31) / 32;         // staged chunks per thread, weight tile
    constexpr int XROWS = BP / 32;                // staged chunks per thread, activation tile
    constexpr int STAGE = (BC + BP) * 128;
    static_assert(WC * WP == 4, "4 waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave / WP, wp = wave % WP;
    const int l16 = lane & 15, g = lane >> 4;

    // ---- XCD-aware, bijective tile remap: XCD x (= blockIdx % 8) owns a contiguous run of tiles
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int tc = wg % p.tilesC, tp = wg / p.tilesC;
    const int co0 = tc * BC, pix0 = tp * BP;

    const int cc = tid & 7;          // chunk column inside the 128-byte slab
    const int r0 = tid >> 3;         // first staged row (0..31); further rows at +32

    // ---- per-thread geometry of the staged activation rows (fixed over the K loop)
    int xn[XROWS], xih[XROWS], xiw[XROWS], xvw[XROWS];
#pragma unroll
    for (int i = 0; i < XROWS; ++i) {
        const int pix = pix0 + r0 + 32 * i;
        if (pix < p.npix) {
            const int n = pix / p.howo, rem = pix - n * p.howo;
            const int oh = rem / p.wo, ow = rem - oh * p.wo;
            xn[i] = n; xih[i] = oh * p.sh - p.ph; xiw[i] = ow * p.sw - p.pw;
            xvw[i] = p.valid_w ? min(p.valid_w[n], p.w) : p.w;
        } else { xn[i] = -1; xih[i] = 0; xiw[i] = 0; xvw[i] = 0; }
    }

    u32x4 wreg[WROWS], xreg[XROWS];
    u32x4 xreg2[SPLIT ? XROWS : 1];               // split-half + input transform: the partner chunk (other half of the same channels)
    int xc = 0;             // input channel of this thread's chunk in the slab being staged
    unsigned xok = 0;       // per-row validity bits of the slab being staged
    const u32x4 zero4 = {0u, 0u, 0u, 0u};

    auto load_slab = [&](int kt) {
        const int k0 = kt * BK;                       // wave-uniform
        const int tap0 = k0 / p.cin;                  // wave-uniform division (scalar unit)
        int c = k0 - tap0 * p.cin + cc * KCH;
        int fr = tap0 / p.kw, fs = tap0 - fr * p.kw;
        while (c >= p.cin) { c -= p.cin; if (++fs == p.kw) { fs = 0; ++fr; } }
        const int k = k0 + cc * KCH;
        const bool kok = k < p.K;
        // weights
#pragma unroll
        for (int i = 0; i < WROWS; ++i) {
            const int row = r0 + 32 * i, co = co0 + row;
            const bool ok = kok && row < BC && co < p.cout;
            if constexpr (MXS) {
                const unsigned char* wb = reinterpret_cast<const unsigned char*>(p.wgt);
                wreg[i] = ok ? ld_hm_wgt_chunk(wb + ((size_t)co * p.K + (k & ~63)) * 2, cc, wb[(size_t)p.cout * p.K * 2 + co]) : zero4;
            } else
            wreg[i] = ok ? ldg16(reinterpret_cast<const PT*>(p.wgt) + (size_t)co * p.K + k) : zero4;
        }
        // activations (implicit im2col gather; channel concat of two sources)
        const PT* src; int cs, cl;
        if (c < p.c0) { src = reinterpret_cast<const PT*>(p.x0); cs = p.c0; cl = c; }
        else { src = reinterpret_cast<const PT*>(p.x1); cs = p.c1; cl = c - p.c0; }
        xc = c; xok = 0;
#pragma unroll
        for (int i = 0; i < XROWS; ++i) {
            const int ih = xih[i] + fr, iw = xiw[i] + fs;
            const bool ok = kok && xn[i] >= 0 && (unsigned)ih < (unsigned)p.h && (unsigned)iw < (unsigned)xvw[i];
            if constexpr (MXS) {
                const unsigned char* blk = reinterpret_cast<const unsigned char*>(src + ((size_t)(xn[i] * p.h + ih) * p.w + iw) * cs + (cl & ~63));
                xreg[i] = ok ? ld_hm_act_chunk(blk, cc) : zero4;
                if (p.in_scale) xreg2[i] = ok ? ld_hm_act_chunk(blk, cc ^ 4) : zero4;
            } else {
            xreg[i] = ok ? ldg16(src + ((size_t)(xn[i] * p.h + ih) * p.w + iw) * cs + cl) : zero4;
            if constexpr (SPLIT) {
                if (p.in_scale) xreg2[i] = ok ? ldg16(src + ((size_t)(xn[i] * p.h + ih) * p.w + iw) * cs + (cl ^ 32)) : zero4;
            }
            }
            xok |= (ok ? 1u : 0u) << i;
        }
    };

    auto store_slab = [&](int stage) {
        unsigned char* sw_ = smem + stage * STAGE;
        unsigned char* sx_ = sw_ + BC * 128;
#pragma unroll
        for (int i = 0; i < WROWS; ++i) {
            const int row = r0 + 32 * i;
            if (row < BC) stg16(sw_ + swz(row, cc), wreg[i]);
        }
#pragma unroll
        for (int i = 0; i < XROWS; ++i) {
            const int row = r0 + 32 * i;
            u32x4 v = xreg[i];
            if (p.in_scale && ((xok >> i) & 1u)) {
                if constexpr (SPLIT) {
                    // physical channel xc = 64*block + 32*half + 8*sub → logical channel 32*block + 8*sub; [n][cin/2] tables
                    const size_t o = (size_t)xn[i] * (p.cin >> 1) + ((xc >> 6) << 5) + (xc & 31);
                    v = in_transform_split(v, xreg2[i], (xc & 32) != 0, p.in_scale + o, p.in_shift ? p.in_shift + o : nullptr, p.in_swish != 0);
                } else {
                    const size_t o = (size_t)xn[i] * p.cin + xc;
                    v = in_transform<T>(v, p.in_scale + o, p.in_shift ? p.in_shift + o : nullptr, p.in_swish != 0);
                }
            }
            stg16(sx_ + swz(row, cc), v);
        }
    };

    f32x4 acc[FC][FP];
#pragma unroll
    for (int a = 0; a < FC; ++a)
#pragma unroll
        for (int b = 0; b < FP; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute_slab = [&](int stage) {
        const unsigned char* sw_ = smem + stage * STAGE;
        const unsigned char* sx_ = sw_ + BC * 128;
        if constexpr (SPLIT) {
            // one 32-channel block: x*w = hi*hi + hi*lo + lo*hi (the lo*lo term is below fp32 resolution), three
            // v_mfma_f32_16x16x32_f16 per fragment pair into the same fp32 accumulator
            u32x4 ah[FC], al[FC], bh[FP], bl[FP];
#pragma unroll
            for (int f = 0; f < FC; ++f) {
                const int row = wc * (BC / WC) + f * 16 + l16;
                ah[f] = *reinterpret_cast<const u32x4*>(sw_ + swz(row, g));
                al[f] = *reinterpret_cast<const u32x4*>(sw_ + swz(row, 4 + g));
            }
#pragma unroll
            for (int f = 0; f < FP; ++f) {
                const int row = wp * (BP / WP) + f * 16 + l16;
                bh[f] = *reinterpret_cast<const u32x4*>(sx_ + swz(row, g));
                bl[f] = *reinterpret_cast<const u32x4*>(sx_ + swz(row, 4 + g));
            }
#pragma unroll
            for (int fa = 0; fa < FC; ++fa)
#pragma unroll
                for (int fb = 0; fb < FP; ++fb) {
                    Mma<f16>::run(acc[fa][fb], ah[fa], bh[fb]);
                    Mma<f16>::run(acc[fa][fb], ah[fa], bl[fb]);
                    Mma<f16>::run(acc[fa][fb], al[fa], bh[fb]);
                }
            return;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int chunk = ks * 4 + g;
            u32x4 a[FC], b[FP];
#pragma unroll
            for (int f = 0; f < FC; ++f) {
                const int row = wc * (BC / WC) + f * 16 + l16;
                a[f] = *reinterpret_cast<const u32x4*>(sw_ + swz(row, chunk));
            }
#pragma unroll
            for (int f = 0; f < FP; ++f) {
                const int row = wp * (BP / WP) + f * 16 + l16;
                b[f] = *reinterpret_cast<const u32x4*>(sx_ + swz(row, chunk));
            }
#pragma unroll
            for (int fa = 0; fa < FC; ++fa)
#pragma unroll
                for (int fb = 0; fb < FP; ++fb) Mma<PT>::run(acc[fa][fb], a[fa], b[fb]);
        }
    };

    // ---- main loop: register prefetch of slab t+1 over the MFMAs of slab t, double-buffered LDS
    load_slab(0);
    store_slab(0);
    __syncthreads();
    for (int kt = 0; kt < p.ktiles; ++kt) {
        const bool more = kt + 1 < p.ktiles;
        if (more) load_slab(kt + 1);
        compute_slab(kt & 1);
        if (more) store_slab((kt + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue: lane owns channels co..co+3 (fragment fa) of one pixel (fragment fb).  Whole-register-set passes, each
    // behind ONE wave-uniform branch (a per-element `switch (act)` costs a scalar branch chain per value).
    T* yo = reinterpret_cast<T*>(p.y);
    int epix[FP], eco[FC];
    if constexpr (SPLIT) {      // weights hold 256*W (exponent offset of the lo halves): exact power-of-two rescale
#pragma unroll
        for (int fa = 0; fa < FC; ++fa)
#pragma unroll
            for (int fb = 0; fb < FP; ++fb) acc[fa][fb] *= MNET_SPLIT_WSCALE_INV;
    }
#pragma unroll
    for (int fb = 0; fb < FP; ++fb) epix[fb] = pix0 + wp * (BP / WP) + fb * 16 + l16;
#pragma unroll
    for (int fa = 0; fa < FC; ++fa) eco[fa] = co0 + wc * (BC / WC) + fa * 16 + g * 4;
    const int last_pix = p.npix - 1;
    if (p.out_scale) {
#pragma unroll
        for (int fb = 0; fb < FP; ++fb) {
            const float* sp = p.out_scale + (size_t)(min(epix[fb], last_pix) / p.howo) * p.cout;
#pragma unroll
            for (int fa = 0; fa < FC; ++fa)
                if (eco[fa] < p.cout) acc[fa][fb] *= *reinterpret_cast<const f32x4*>(sp + eco[fa]);
        }
    }
    if (p.bias) {
#pragma unroll
        for (int fa = 0; fa < FC; ++fa) {
            if (eco[fa] >= p.cout) continue;
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + eco[fa]);
#pragma unroll
            for (int fb = 0; fb < FP; ++fb) acc[fa][fb] += b4;
        }
    }
    if (p.res) {
        const T* rs = reinterpret_cast<const T*>(p.res);
#pragma unroll
        for (int fb = 0; fb < FP; ++fb) {
            if (epix[fb] >= p.npix) continue;
            const int rpix = p.res_mod > 0 ? epix[fb] % p.res_mod : epix[fb];
#pragma unroll
            for (int fa = 0; fa < FC; ++fa) {
                if (eco[fa] >= p.cout) continue;
                const T* rp = rs + (size_t)rpix * p.cout + eco[fa];
                if constexpr (MXS) {
                    const unsigned char* rb = reinterpret_cast<const unsigned char*>(rs + (size_t)rpix * p.cout) + (eco[fa] >> 5) * 128;
                    const int ci = eco[fa] & 31;
                    const f16x4 h4 = *reinterpret_cast<const f16x4*>(rb + ci * 2);
                    const int b4 = *reinterpret_cast<const int*>(rb + 64 + hm_lo_slot(ci >> 3) * 8 + (ci & 7));
                    const float sl = hm_lo_scale(rb[96]);
                    const f32x2 la = __builtin_amdgcn_cvt_pk_f32_fp8(b4, false), lb = __builtin_amdgcn_cvt_pk_f32_fp8(b4, true);
                    acc[fa][fb] += f32x4{(float)h4[0] + la[0] * sl, (float)h4[1] + la[1] * sl, (float)h4[2] + lb[0] * sl, (float)h4[3] + lb[1] * sl};
                } else if constexpr (SPLIT) {
                    const f16* q = reinterpret_cast<const f16*>(rs + (size_t)rpix * p.cout) + (eco[fa] >> 5) * 64 + (eco[fa] & 31);
                    const f16x4 h4 = *reinterpret_cast<const f16x4*>(q), l4 = *reinterpret_cast<const f16x4*>(q + 32);
                    acc[fa][fb] += f32x4{(float)h4[0] + (float)l4[0], (float)h4[1] + (float)l4[1], (float)h4[2] + (float)l4[2], (float)h4[3] + (float)l4[3]};
                } else if constexpr (sizeof(T) == 4) {
                    acc[fa][fb] += *reinterpret_cast<const f32x4*>(rp);
                } else {
                    const f16x4 r4 = *reinterpret_cast<const f16x4*>(rp);
                    acc[fa][fb] += f32x4{(float)r4[0], (float)r4[1], (float)r4[2], (float)r4[3]};
                }
            }
        }
    }
    {
        float ev[FC * FP * 4];
#pragma unroll
        for (int fa = 0; fa < FC; ++fa)
#pragma unroll
            for (int fb = 0; fb < FP; ++fb)
#pragma unroll
                for (int q = 0; q < 4; ++q) ev[(fa * FP + fb) * 4 + q] = acc[fa][fb][q];
        act_apply_vec<FC * FP * 4>(ev, p.act);
#pragma unroll
        for (int fa = 0; fa < FC; ++fa)
#pragma unroll
            for (int fb = 0; fb < FP; ++fb)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[fa][fb][q] = ev[(fa * FP + fb) * 4 + q];
    }
    if (p.post_scale) {
#pragma unroll
        for (int fb = 0; fb < FP; ++fb) {
            const float* sp = p.post_scale + (size_t)(min(epix[fb], last_pix) / p.howo) * p.cout;
#pragma unroll
            for (int fa = 0; fa < FC; ++fa)
                if (eco[fa] < p.cout) acc[fa][fb] *= *reinterpret_cast<const f32x4*>(sp + eco[fa]);
        }
    }
    if constexpr (MXS) {
        // fp16+8 store: a 32-channel block of one pixel is held by the lanes l16 + 16 g (g = 0..3: 4 channels each) in the fragment
        // pair (2b, 2b+1): block max over the lane's 8 values, then over g (lanes ^16, ^32); FC is even for every tile that takes a
        // split launch (cout % 32 == 0, wave tiles of >= 32 channels)
        if constexpr (FC % 2 == 0)          // (the 16-channel tile never takes a split launch: cout % 32 == 0)
#pragma unroll
        for (int fb = 0; fb < FP; ++fb)
#pragma unroll
            for (int b = 0; b < FC / 2; ++b) {
                const f32x4 v0 = acc[2 * b][fb], v1 = acc[2 * b + 1][fb];
                f16x4 h0, h1;
                float m = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) { h0[q] = (f16)v0[q]; h1[q] = (f16)v1[q]; m = fmaxf(m, fmaxf(fabsf((float)h0[q]), fabsf((float)h1[q]))); }
                m = fmaxf(m, __shfl_xor(m, 16, 64));
                m = fmaxf(m, __shfl_xor(m, 32, 64));
                const int e8 = hm_e8_of(m);
                const float inv = e8 >= 11 ? __builtin_bit_cast(float, (unsigned)(265 - e8) << 23) : 0.f;
                if (epix[fb] >= p.npix || eco[2 * b] >= p.cout) continue;
                unsigned char* yb = reinterpret_cast<unsigned char*>(yo + (size_t)epix[fb] * p.cout) + (eco[2 * b] >> 5) * 128;
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const f32x4 v = f ? v1 : v0;
                    const f16x4 h4 = f ? h1 : h0;
                    const int ci = eco[2 * b + f] & 31;
                    *reinterpret_cast<f16x4*>(yb + ci * 2) = h4;
                    int w = 0;
                    w = __builtin_amdgcn_cvt_pk_fp8_f32((v[0] - (float)h4[0]) * inv, (v[1] - (float)h4[1]) * inv, w, false);
                    w = __builtin_amdgcn_cvt_pk_fp8_f32((v[2] - (float)h4[2]) * inv, (v[3] - (float)h4[3]) * inv, w, true);
                    *reinterpret_cast<int*>(yb + 64 + hm_lo_slot(ci >> 3) * 8 + (ci & 7)) = w;
                }
                if (g == 0) *reinterpret_cast<int*>(yb + 96) = e8;
                if (g == 1) { *reinterpret_cast<int*>(yb + 100) = 0; *reinterpret_cast<u32x2*>(yb + 104) = u32x2{0u, 0u}; }
                if (g == 2) stg16(yb + 112, u32x4{0u, 0u, 0u, 0u});
            }
        return;
    }
#pragma unroll
    for (int fb = 0; fb < FP; ++fb) {
        if (epix[fb] >= p.npix) continue;
#pragma unroll
        for (int fa = 0; fa < FC; ++fa) {
            if (eco[fa] >= p.cout) continue;
            T* yp = yo + (size_t)epix[fb] * p.cout + eco[fa];
            const f32x4 v = acc[fa][fb];
            if constexpr (SPLIT) {
                f16* q = reinterpret_cast<f16*>(yo + (size_t)epix[fb] * p.cout) + (eco[fa] >> 5) * 64 + (eco[fa] & 31);
                const f16x4 h4 = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                const f16x4 l4 = {(f16)(v[0] - (float)h4[0]), (f16)(v[1] - (float)h4[1]), (f16)(v[2] - (float)h4[2]), (f16)(v[3] - (float)h4[3])};
                *reinterpret_cast<f16x4*>(q) = h4;
                *reinterpret_cast<f16x4*>(q + 32) = l4;
            } else if constexpr (sizeof(T) == 4) {
                *reinterpret_cast<f32x4*>(yp) = v;
            } else {
                const f16x4 o4 = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                *reinterpret_cast<f16x4*>(yp) = o4;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ host
template <typename T, int BC, int BP, int WC, int WP>
static int launch_cfg(const ConvArgs& a, hipStream_t st) {
    constexpr int LDS = 2 * (BC + BP) * 128;
    auto kern = conv_igemm_kernel<T, BC, BP, WC, WP>;
    static thread_local DeviceOnce attr_once;      // per instantiation, per thread, per device
    if (!attr_once.done()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return mnet_fail(MNET_E_LAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_once.mark();
    }
    ConvArgs b = a;
    b.tilesC = (a.cout + BC - 1) / BC;
    const int tilesP = (a.npix + BP - 1) / BP;
    hipLaunchKernelGGL(kern, dim3((unsigned)(b.tilesC * tilesP)), dim3(256), LDS, st, b);
    MNET_LAUNCH_CHECK("conv_igemm_kernel");
    return MNET_OK;
}

template <typename T>
static int launch_dtype(const ConvArgs& a, hipStream_t st) {
    // Small launches (the TextViT linears: 4096 x 512 outputs) would leave CUs idle with 128-channel tiles: halve the cout
    // tile until the grid covers the CUs.  Every tile shape accumulates each output in the same k order, so
    // the choice never changes a result bit.
    const long long tilesP = (a.npix + 127) / 128;
    static const int force = [] { const char* e = getenv("MNET_REG_TILE"); return e ? atoi(e) : 0; }();   // A/B knob: cout tile
    if (force == 32 && a.cout >= 32) return launch_cfg<T, 32, 128, 1, 4>(a, st);
    if (force == 64 && a.cout >= 64) return launch_cfg<T, 64, 128, 2, 2>(a, st);
    if (force == 128 && a.cout >= 128) return launch_cfg<T, 128, 128, 2, 2>(a, st);
    if (a.cout >= 128 && tilesP * ((a.cout + 127) / 128) >= 512) return launch_cfg<T, 128, 128, 2, 2>(a, st);
    if (a.cout >= 64 && (a.cout < 128 || tilesP * ((a.cout + 63) / 64) >= 256)) return launch_cfg<T, 64, 128, 2, 2>(a, st);
    if (a.cout >= 128) return launch_cfg<T, 32, 128, 1, 4>(a, st);
    if (a.cout >= 64) return launch_cfg<T, 64, 128, 2, 2>(a, st);
    if (a.cout >= 32) return launch_cfg<T, 32, 128, 1, 4>(a, st);
    return launch_cfg<T, 16, 128, 1, 4>(a, st);
}

extern "C" double mnet_conv2d_flops(const mnet_conv_desc* d) {
    if (!d) return 0.0;
    return 2.0 * (double)d->n * d->ho * d->wo * d->cout * (double)d->kh * d->kw * (d->c0 + d->c1);
}

// argument validation shared by the launch and the planning entry points; fills the kernel-argument block
static int conv_prepare(const mnet_conv_desc* d, int32_t algo, ConvArgs& a) {
    a.one_tile_per_wg = (algo & MNET_CONV_ALGO_FLAG_ONE_TILE) ? 1 : 0;
    static const int env_fetch_pad = [] { const char* e = getenv("MNET_MX_FETCH_PAD"); return e ? atoi(e) : 0; }();   // A/B knob, see ConvArgs
    a.mx_fetch_pad = env_fetch_pad;
    a.x1_center = (algo & MNET_CONV_ALGO_FLAG_X1_CENTER) ? 1 : 0;
    algo &= ~(MNET_CONV_ALGO_FLAG_ONE_TILE | MNET_CONV_ALGO_FLAG_X1_CENTER);
    MNET_CHECK_ARG((algo >= 0 && algo <= 3) || (algo >= MNET_CONV_ALGO_DMA_CFG0 && algo < MNET_CONV_ALGO_DMA_CFG0 + 16) ||
                   (algo >= MNET_CONV_ALGO_STRIP_CFG0 && algo < MNET_CONV_ALGO_STRIP_CFG0 + 3) ||
                   (algo >= MNET_CONV_ALGO_DMA_CFG16 && algo < MNET_CONV_ALGO_DMA_CFG16 + 16), "conv: bad algo %d", algo);
    MNET_CHECK_ARG(d != nullptr, "conv: null descriptor");
    MNET_CHECK_ARG(d->dtype == MNET_F32 || d->dtype == MNET_F16 || d->dtype == MNET_F16X2 || d->dtype == MNET_F16M, "conv: bad dtype %d", d->dtype);
    MNET_CHECK_ARG(d->x0 && d->wgt && d->y, "conv: null tensor pointer");
    MNET_CHECK_ARG(d->n > 0 && d->h > 0 && d->w > 0 && d->ho > 0 && d->wo > 0, "conv: bad geometry");
    MNET_CHECK_ARG(d->c0 > 0 && d->c1 >= 0 && d->cout > 0, "conv: bad channel counts");
    MNET_CHECK_ARG((d->c1 == 0) == (d->x1 == nullptr), "conv: x1/c1 mismatch");
    MNET_CHECK_ARG(d->kh > 0 && d->kw > 0 && d->stride_h > 0 && d->stride_w > 0 && d->pad_h >= 0 && d->pad_w >= 0,
                   "conv: bad filter geometry");
    MNET_CHECK_ARG(d->ho == (d->h + 2 * d->pad_h - d->kh) / d->stride_h + 1 &&
                   d->wo == (d->w + 2 * d->pad_w - d->kw) / d->stride_w + 1 &&
                   d->h + 2 * d->pad_h >= d->kh && d->w + 2 * d->pad_w >= d->kw,
                   "conv: ho/wo (%d,%d) inconsistent with h,w,k,stride,pad", d->ho, d->wo);
    MNET_CHECK_ARG(d->act >= MNET_ACT_NONE && d->act <= MNET_ACT_SIGMOID, "conv: bad act %d", d->act);
    MNET_CHECK_ARG(!(d->in_shift || d->in_swish) || d->in_scale, "conv: in_shift/in_swish need in_scale");
    MNET_CHECK_ARG(d->res_mod >= 0, "conv: res_mod < 0");
    MNET_CHECK_ALIGN(d->c0 % 8 == 0 && d->c1 % 8 == 0, "conv: c0=%d c1=%d must be multiples of 8", d->c0, d->c1);
    MNET_CHECK_ALIGN(d->cout % 4 == 0, "conv: cout=%d must be a multiple of 4", d->cout);
    MNET_CHECK_ALIGN(aligned16(d->x0) && aligned16(d->x1) && aligned16(d->wgt) && aligned16(d->y) &&
                     aligned16(d->residual) && aligned16(d->in_scale) && aligned16(d->in_shift) &&
                     aligned16(d->out_scale) && aligned16(d->bias) && aligned16(d->post_scale), "conv: pointers must be 16-byte aligned");
    const long long npix = (long long)d->n * d->ho * d->wo;
    MNET_CHECK_ARG(npix < (1ll << 31) && (long long)d->n * d->h * d->w < (1ll << 31), "conv: too many pixels");
    const bool split = d->dtype == MNET_F16X2 || d->dtype == MNET_F16M;
    if (split) {
        MNET_CHECK_ALIGN(d->c0 % 32 == 0 && d->c1 % 32 == 0 && d->cout % 32 == 0, "conv: split-half tensors need c0, c1, cout %% 32 == 0 (got %d, %d, %d)", d->c0, d->c1, d->cout);
        MNET_CHECK_ALIGN(aligned128(d->x0) && aligned128(d->x1) && aligned128(d->wgt) && aligned128(d->y) && aligned128(d->residual),
                         "conv: split-half tensors must be 128-byte aligned");
    }
    const int cm = split ? 2 : 1;        // the k-loop walks a split-half tensor as f16 with twice the channels (hi block | lo block)

    a.x0 = d->x0; a.x1 = d->x1; a.wgt = d->wgt; a.y = d->y; a.res = d->residual;
    a.in_scale = d->in_scale; a.in_shift = d->in_shift; a.out_scale = d->out_scale; a.bias = d->bias;
    a.post_scale = d->post_scale;
    a.valid_w = d->valid_w;
    a.c0 = d->c0 * cm; a.c1 = d->c1 * cm; a.cin = a.c0 + a.c1; a.split = d->dtype == MNET_F16M ? 2 : (split ? 1 : 0);
    a.n = d->n; a.h = d->h; a.w = d->w; a.ho = d->ho; a.wo = d->wo; a.cout = d->cout;
    a.kh = d->kh; a.kw = d->kw; a.sh = d->stride_h; a.sw = d->stride_w; a.ph = d->pad_h; a.pw = d->pad_w;
    a.K = d->kh * d->kw * a.cin; a.npix = (int)npix; a.howo = d->ho * d->wo;
    a.in_swish = d->in_swish; a.act = d->act; a.res_mod = d->res_mod;
    const int bk = d->dtype == MNET_F32 ? 32 : 64;
    a.ktiles = (a.K + bk - 1) / bk; a.tilesC = 0; a.ntiles = 0;
    a.gn_partial = d->gn_partial;
    if (a.gn_partial) {
        MNET_CHECK_ARG(d->dtype == MNET_F16M && d->stride_h == 1 && d->stride_w == 1 && d->cout % 32 == 0 && (d->ho * d->wo) % 32 == 0 && d->ho == d->h && d->wo == d->w,
                       "conv: gn_partial needs an MNET_F16M launch with stride 1, 'same' size, cout %% 32 == 0 and ho*wo %% 32 == 0");
        MNET_CHECK_ALIGN((reinterpret_cast<uintptr_t>(d->gn_partial) & 7u) == 0, "conv: gn_partial must be 8-byte aligned");
    }
    a.howo_shift = a.wo_shift = -1;      // "not a power of two" until an LDS-DMA launcher says otherwise (0 would read as a shift by 0)
    a.center_tap = (d->kh / 2) * d->kw + d->kw / 2;
    a.center_tpx = (d->kh / 2) * d->w + d->kw / 2;
    if (a.x1_center) {
        MNET_CHECK_ARG(d->x1 && d->c1 > 0 && (d->kh & 1) && (d->kw & 1) && d->stride_h == 1 && d->stride_w == 1 &&
                       d->pad_h == d->kh / 2 && d->pad_w == d->kw / 2 && d->dtype != MNET_F32,
                       "conv: MNET_CONV_ALGO_FLAG_X1_CENTER needs a second source, an odd filter, stride 1, 'same' padding and a half-range dtype");
        MNET_CHECK_ARG(a.c0 % 64 == 0 && a.c1 % 64 == 0, "conv: MNET_CONV_ALGO_FLAG_X1_CENTER needs c0, c1 in whole k-slabs");
        a.ktiles = d->kh * d->kw * (a.c0 / 64) + a.c1 / 64;      // (physical channels: 64 halves per slab row)
    }
    return MNET_OK;
}

// resolves `algo` to the kernel that runs: MNET_CONV_ALGO_REG_STAGED, MNET_CONV_ALGO_SKINNY, MNET_CONV_ALGO_DMA_CFG0 + id or
// MNET_CONV_ALGO_STRIP_CFG0 + id (negative: error)
static int conv_resolve(const mnet_conv_desc* d, int32_t algo, const ConvArgs& a) {
    algo &= ~(MNET_CONV_ALGO_FLAG_ONE_TILE | MNET_CONV_ALGO_FLAG_X1_CENTER);
    static const bool no_strip = [] { const char* e = getenv("MNET_DMA_NO_STRIP"); return e && atoi(e) != 0; }();   // A/B knob
    const bool dma_ok = conv_dma_eligible(a, d->dtype);
    if (a.x1_center) {          // only the LDS-DMA kernels walk the second source at one tap
        if (!dma_ok || algo == MNET_CONV_ALGO_REG_STAGED || algo == MNET_CONV_ALGO_SKINNY || (algo >= MNET_CONV_ALGO_STRIP_CFG0 && algo < MNET_CONV_ALGO_DMA_CFG16))
            return mnet_fail(MNET_E_ARG, "conv: MNET_CONV_ALGO_FLAG_X1_CENTER needs a launch the LDS-DMA kernel takes");
        if (algo >= MNET_CONV_ALGO_DMA_CFG0) return algo;
        const int id = conv_dma_pick(a);
        return id < 16 ? MNET_CONV_ALGO_DMA_CFG0 + id : MNET_CONV_ALGO_DMA_CFG16 + (id - 16);
    }
    const int strip = conv_strip_pick(a, d->dtype, algo >= MNET_CONV_ALGO_STRIP_CFG0);
    if (algo >= MNET_CONV_ALGO_DMA_CFG16) {
        if (!dma_ok) return mnet_fail(MNET_E_ARG, "conv: this launch is not eligible for the LDS-DMA kernel");
        return algo;
    }
    if (algo >= MNET_CONV_ALGO_STRIP_CFG0) {
        if (strip != algo - MNET_CONV_ALGO_STRIP_CFG0)
            return mnet_fail(MNET_E_ARG, "conv: strip configuration %d is not the one this launch is eligible for (%d)",
                             algo - MNET_CONV_ALGO_STRIP_CFG0, strip);
        return algo;
    }
    if (algo == MNET_CONV_ALGO_SKINNY) {
        if (!conv_skinny_eligible(a, d->dtype))
            return mnet_fail(MNET_E_ARG, "conv: the skinny kernel needs fp32, 1x1 / stride 1 / no padding, one source, cin %% 16 == 0, "
                                         "no input transform and <= 512 pixels");
        return algo;
    }
    if (algo == MNET_CONV_ALGO_AUTO && conv_skinny_eligible(a, d->dtype)) return MNET_CONV_ALGO_SKINNY;
    if (algo >= MNET_CONV_ALGO_LDS_DMA && !dma_ok)
        return mnet_fail(MNET_E_ARG, "conv: LDS-DMA algo needs f16 (cin %% 64 == 0) or split-half (cin %% 32 == 0), cout >= 64, cout %% 8 == 0 and no input transform");
    if (algo >= MNET_CONV_ALGO_DMA_CFG0) return algo;
    if (algo != MNET_CONV_ALGO_REG_STAGED && strip >= 0 && !no_strip) return MNET_CONV_ALGO_STRIP_CFG0 + strip;
    if (dma_ok && algo != MNET_CONV_ALGO_REG_STAGED) {
        const int id = conv_dma_pick(a);
        return id < 16 ? MNET_CONV_ALGO_DMA_CFG0 + id : MNET_CONV_ALGO_DMA_CFG16 + (id - 16);
    }
    return MNET_CONV_ALGO_REG_STAGED;
}

// conv_resolve + the one constraint that depends on its answer: only the fp16+8 LDS-DMA / strip epilogue (dma_epilogue_mx) writes gn_partial
static int conv_resolve_checked(const mnet_conv_desc* d, int32_t algo, const ConvArgs& a) {
    const int k = conv_resolve(d, algo, a);
    if (k >= 0 && a.gn_partial && k < MNET_CONV_ALGO_DMA_CFG0)
        return mnet_fail(MNET_E_ARG, "conv: gn_partial needs a launch the LDS-DMA / strip kernels take (this one resolves to kernel %d)", k);
    return k;
}

extern "C" int mnet_conv2d_plan(const mnet_conv_desc* d, int32_t algo) {
    ConvArgs a;
    const int rc = conv_prepare(d, algo, a);
    if (rc != MNET_OK) return rc;
    return conv_resolve_checked(d, algo, a);
}

extern "C" int mnet_conv2d_nhwc_ex(const mnet_conv_desc* d, int32_t algo, void* stream) {
    ConvArgs a;
    const int rc = conv_prepare(d, algo, a);
    if (rc != MNET_OK) return rc;
    const int k = conv_resolve_checked(d, algo, a);
    if (k < 0) return k;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (k >= MNET_CONV_ALGO_DMA_CFG16) return launch_conv_dma(a, st, k - MNET_CONV_ALGO_DMA_CFG16 + 16);
    if (k >= MNET_CONV_ALGO_STRIP_CFG0) return launch_conv_strip(a, st, k - MNET_CONV_ALGO_STRIP_CFG0);
    if (k >= MNET_CONV_ALGO_DMA_CFG0) return launch_conv_dma(a, st, k - MNET_CONV_ALGO_DMA_CFG0);
    if (k == MNET_CONV_ALGO_SKINNY) return launch_conv_skinny(a, st);
    if (d->dtype == MNET_F16X2) return launch_dtype<hs>(a, st);
    if (d->dtype == MNET_F16M) return launch_dtype<hm>(a, st);
    return d->dtype == MNET_F16 ? launch_dtype<f16>(a, st) : launch_dtype<float>(a, st);
}

extern "C" int mnet_conv2d_splitk(const mnet_conv_desc* d, int32_t ksplit, float* workspace, void* stream) {
    ConvArgs a;
    const int rc = conv_prepare(d, MNET_CONV_ALGO_AUTO, a);
    if (rc != MNET_OK) return rc;
    MNET_CHECK_ARG(!a.gn_partial, "conv: gn_partial is not available on the split-K path");
    return launch_conv_skinny_splitk(a, d->dtype, ksplit, workspace, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int mnet_conv2d_nhwc(const mnet_conv_desc* d, void* stream) {
    return mnet_conv2d_nhwc_ex(d, MNET_CONV_ALGO_AUTO, stream);
}
