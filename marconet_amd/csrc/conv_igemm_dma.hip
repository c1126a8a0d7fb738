// Implicit-GEMM convolution, LDS-DMA variant (f16 storage, fp32 accumulate) — the fast path for every conv
// whose input-channel count is a multiple of 64 and that needs no input-side transform.
//
// What differs from conv_igemm.hip (the general, register-staged kernel):
//  * both operand tiles go HBM/L2 → LDS directly with `buffer_load_dwordx4 … lds` (1 KiB per wave-instruction):
//    no VGPR round trip, no ds_write pass, ~40 fewer VGPRs.  Zero padding, ragged valid width, the cout tail
//    and the pixel tail cost nothing: an invalid lane gets an out-of-range buffer offset and the hardware
//    writes zeros into its LDS slot (raw buffer bounds check against num_records).
//  * the 16-byte-chunk XOR swizzle of the LDS image is applied on the *source* side (lane (row, phys chunk)
//    fetches logical chunk phys ^ ((row>>1)&7)), because an LDS-DMA destination is always lane-linear.
//  * 8 waves per workgroup on a 256x128 / 128x256 / 64x256 (cout x pixel) tile, 64x64 (64x32) per wave;
//    3-stage LDS ring (3 x 48 KiB), one raw s_barrier per 64-deep k-slab, counted `s_waitcnt vmcnt(N)` so the
//    DMA of slab t+1 stays in flight across the barrier while slab t is multiplied (no vmcnt(0) in the loop).
//  * with cin % 64 == 0 a k-slab never straddles a filter tap or the concat boundary, so tap (r,s), source
//    tensor and channel offset are wave-uniform scalars advanced incrementally — no integer division in the loop.
#include <cstdlib>
#include "common.h"
#include "conv_args.h"

typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ int swz_dma(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

#define VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

template <int BC, int BP, int WC, int WP, int STAGES>
__global__ void __launch_bounds__(WC * WP * 64, WC * WP / 4) conv_dma_kernel(const ConvArgs p) {
    constexpr int NW = WC * WP;                          // waves per workgroup (8 or 16)
    constexpr int FC = BC / WC / 16, FP = BP / WP / 16;
    constexpr int WJ = BC / (8 * NW), XJ = BP / (8 * NW); // DMA instructions per wave per slab (weights / activations)
    constexpr int NDMA = WJ + XJ;
    constexpr int STAGE = (BC + BP) * 128;
    constexpr unsigned OOB = 0x80000000u;                // beyond every num_records used below
    static_assert(NW == 8 || NW == 16, "8 or 16 waves");
    static_assert(WJ >= 1 && XJ >= 1 && WJ * 8 * NW == BC && XJ * 8 * NW == BP, "tile / wave-count mismatch");
    static_assert(STAGES == 2 || (STAGES == 3 && (NDMA == 6 || NDMA == 5)), "vmcnt immediates below assume 5 or 6 DMAs per slab");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave / WP, wp = wave % WP;
    const int l16 = lane & 15, g = lane >> 4;

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int tc = wg % p.tilesC, tp = wg / p.tilesC;
    const int co0 = tc * BC, pix0 = tp * BP;

    // ---- buffer descriptors (wave-uniform).  Weights: rows co0.. ; bound = remaining rows → cout tail reads zeros.
    const long long wbytes = (long long)(p.cout - co0) * p.K * 2;
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<const f16*>(p.wgt) + (size_t)co0 * p.K), 0, (int)(wbytes < 0x7fffffffLL ? wbytes : 0x7fffffffLL), 0x00020000);
    // Activations: base = first image touched by this pixel tile; bound = the images the tile can touch.
    const int n_first = pix0 / p.howo;
    int n_last = (min(pix0 + BP, p.npix) - 1) / p.howo;
    const long long img0 = (long long)p.h * p.w * p.c0 * 2, img1 = (long long)p.h * p.w * p.c1 * 2;
    const int nimg = n_last - n_first + 1;
    const __amdgpu_buffer_rsrc_t rX0 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<const char*>(p.x0) + (size_t)n_first * img0), 0, (int)(img0 * nimg), 0x00020000);
    const __amdgpu_buffer_rsrc_t rX1 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.x1 ? reinterpret_cast<const char*>(p.x1) + (size_t)n_first * img1 : reinterpret_cast<const char*>(p.x0)),
        0, (int)(p.x1 ? img1 * nimg : 0), 0x00020000);

    // ---- per-lane DMA geometry: lane (rg = lane/8, pc = lane%8) of wave w fills LDS rows (w + 8j)*8 + rg
    const int rg = lane >> 3, pc = lane & 7;
    // Weight rows are permuted on their way into LDS (free: the DMA source address is per lane) so that the two MFMA
    // fragments 2t, 2t+1 of a lane together hold 8 CONSECUTIVE output channels → 16-byte epilogue loads/stores:
    //   LDS row 64b + 16f + i   holds channel   64b + 32(f/2) + 8(i/4) + 4(f%2) + i%4
    unsigned woff[WJ];                                   // byte offset of this lane's weight chunk at k-slab 0
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
        const int row = (wave + NW * j) * 8 + rg;
        const int f = (row >> 4) & 3, i = row & 15;
        const int ch = (row & ~63) + ((f >> 1) << 5) + ((i >> 2) << 3) + ((f & 1) << 2) + (i & 3);
        const int lc = pc ^ ((row >> 1) & 7);
        woff[j] = (co0 + ch < p.cout) ? (unsigned)(ch * p.K * 2 + lc * 16) : OOB;
    }
    // activation rows: byte offset of (tap (0,0), channel 0) in each concat source, and a bit mask of the filter taps
    // whose input pixel exists (inside the image and left of valid_w) — kh*kw <= 32 on this path
    unsigned xb0[XJ], xb1[XJ], xmask[XJ];
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
        const int row = (wave + NW * j) * 8 + rg;
        const int pix = pix0 + row;
        const int lcb = (pc ^ ((row >> 1) & 7)) * 16;
        xb0[j] = 0; xb1[j] = 0; xmask[j] = 0;
        if (pix < p.npix) {
            const int n = pix / p.howo, rem = pix - n * p.howo;
            const int oh = rem / p.wo, ow = rem - oh * p.wo;
            const int ih0 = oh * p.sh - p.ph, iw0 = ow * p.sw - p.pw;
            const int px = ((n - n_first) * p.h + ih0) * p.w + iw0;
            const int vw = p.valid_w ? min(p.valid_w[n], p.w) : p.w;
            xb0[j] = (unsigned)(px * p.c0 * 2 + lcb);
            xb1[j] = (unsigned)(px * p.c1 * 2 + lcb);
            // taps enumerated t = r*kw + q; fixed 8-trip loops (kh, kw <= 8) so everything stays in registers
            unsigned cm = 0, m = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (q < p.kw && (unsigned)(iw0 + q) < (unsigned)vw) cm |= 1u << q;
#pragma unroll
            for (int r = 0; r < 8; ++r)
                if (r < p.kh && (unsigned)(ih0 + r) < (unsigned)p.h) m |= cm << (r * p.kw);
            xmask[j] = m;
        }
    }

    // ---- wave-uniform k-slab cursor: filter tap index, channel offset inside the tap, tap pixel offset
    int cur_c = 0, cur_s = 0, cur_tap = 0, cur_tpx = 0, cur_k = 0;
    auto issue_slab = [&](int stage) __attribute__((always_inline)) {
        unsigned char* sw_ = smem + stage * STAGE;
        unsigned char* sx_ = sw_ + BC * 128;
        const unsigned kb = (unsigned)cur_k * 2u;
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            const unsigned vo = woff[j] == OOB ? OOB : woff[j] + kb;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_void*)(sw_ + (wave + NW * j) * 1024), 16, vo, 0, 0, 0);
        }
        const unsigned tapbit = 1u << cur_tap;
        if (cur_c >= p.c0) {                             // wave-uniform: second concat source
            const unsigned uni = (unsigned)(cur_tpx * p.c1 * 2 + (cur_c - p.c0) * 2);
#pragma unroll
            for (int j = 0; j < XJ; ++j) {
                const unsigned vo = (xmask[j] & tapbit) ? xb1[j] + uni : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rX1, (lds_void*)(sx_ + (wave + NW * j) * 1024), 16, vo, 0, 0, 0);
            }
        } else {
            const unsigned uni = (unsigned)(cur_tpx * p.c0 * 2 + cur_c * 2);
#pragma unroll
            for (int j = 0; j < XJ; ++j) {
                const unsigned vo = (xmask[j] & tapbit) ? xb0[j] + uni : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rX0, (lds_void*)(sx_ + (wave + NW * j) * 1024), 16, vo, 0, 0, 0);
            }
        }
        cur_k += 64; cur_c += 64;
        if (cur_c == p.cin) {
            cur_c = 0; ++cur_tap;
            if (++cur_s == p.kw) { cur_s = 0; cur_tpx += p.w - (p.kw - 1); } else { ++cur_tpx; }
        }
    };

    f32x4 acc[FC][FP];
#pragma unroll
    for (int a = 0; a < FC; ++a)
#pragma unroll
        for (int b = 0; b < FP; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute_half = [&](int stage, int ks) __attribute__((always_inline)) {
        const unsigned char* sw_ = smem + stage * STAGE;
        const unsigned char* sx_ = sw_ + BC * 128;
        {
            const int chunk = ks * 4 + g;
            u32x4 a[FC], b[FP];
#pragma unroll
            for (int f = 0; f < FC; ++f) a[f] = *reinterpret_cast<const u32x4*>(sw_ + swz_dma(wc * (BC / WC) + f * 16 + l16, chunk));
#pragma unroll
            for (int f = 0; f < FP; ++f) b[f] = *reinterpret_cast<const u32x4*>(sx_ + swz_dma(wp * (BP / WP) + f * 16 + l16, chunk));
#pragma unroll
            for (int fa = 0; fa < FC; ++fa)
#pragma unroll
                for (int fb = 0; fb < FP; ++fb)
                    acc[fa][fb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bitcast<f16x8>(a[fa]), bitcast<f16x8>(b[fb]), acc[fa][fb], 0, 0, 0);
        }
    };

    const int nk = p.ktiles;
    if constexpr (STAGES == 3) {
        // ---- 3-stage ring: slab t+1 stays in flight across the barrier of slab t (counted vmcnt, never 0 in the loop)
        issue_slab(0);
        if (nk > 1) issue_slab(1);
        int st_c = 0, st_i = 2;           // stage being computed / stage being filled next
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) { if constexpr (NDMA == 6) VMCNT(6); else VMCNT(5); }    // slab kt landed (this wave's DMAs)
            else VMCNT(0);
            __builtin_amdgcn_s_barrier();                                             // …everyone's; stage st_i is free
            asm volatile("" ::: "memory");
            if (kt + 2 < nk) issue_slab(st_i);
            compute_half(st_c, 0);
            compute_half(st_c, 1);
            st_c = st_c == 2 ? 0 : st_c + 1;
            st_i = st_i == 2 ? 0 : st_i + 1;
        }
    } else {
        // ---- 2-stage (256x256 tile, 16 waves): the DMA of slab t+1 is issued right after the barrier and overlaps the
        // whole multiply of slab t (4 waves per SIMD x 32 MFMAs each); drained with vmcnt(0) just before the next barrier
        issue_slab(0);
        for (int kt = 0; kt < nk; ++kt) {
            VMCNT(0);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + 1 < nk) issue_slab((kt + 1) & 1);
            compute_half(kt & 1, 0);
            compute_half(kt & 1, 1);
        }
    }

    // ---- epilogue (identical math to conv_igemm.hip; 8 consecutive channels per lane, see the weight-row permutation)
    static_assert((BC / WC) % 64 == 0, "the channel permutation works on 64-channel blocks of a wave tile");
    f16* yo = reinterpret_cast<f16*>(p.y);
    const f16* rs = reinterpret_cast<const f16*>(p.res);
#pragma unroll
    for (int fb = 0; fb < FP; ++fb) {
        const int pix = pix0 + wp * (BP / WP) + fb * 16 + l16;
        if (pix >= p.npix) continue;
        const int n = (p.out_scale || p.post_scale) ? pix / p.howo : 0;
        const int rpix = p.res_mod > 0 ? pix % p.res_mod : pix;
#pragma unroll
        for (int t = 0; t < FC / 2; ++t) {
            const int co = co0 + wc * (BC / WC) + t * 32 + g * 8;
            if (co >= p.cout) continue;
            float v[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) { v[q] = acc[2 * t][fb][q]; v[4 + q] = acc[2 * t + 1][fb][q]; }
            if (p.out_scale) {
                const f32x4 s0 = *reinterpret_cast<const f32x4*>(p.out_scale + (size_t)n * p.cout + co);
                const f32x4 s1 = *reinterpret_cast<const f32x4*>(p.out_scale + (size_t)n * p.cout + co + 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) { v[q] *= s0[q]; v[4 + q] *= s1[q]; }
            }
            if (p.bias) {
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + co);
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(p.bias + co + 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) { v[q] += b0[q]; v[4 + q] += b1[q]; }
            }
            if (rs) {
                const f16x8 r8 = bitcast<f16x8>(ldg16(rs + (size_t)rpix * p.cout + co));
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] += (float)r8[q];
            }
            if (p.act != MNET_ACT_NONE) {
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = act_apply(v[q], p.act);
            }
            if (p.post_scale) {
                const f32x4 s0 = *reinterpret_cast<const f32x4*>(p.post_scale + (size_t)n * p.cout + co);
                const f32x4 s1 = *reinterpret_cast<const f32x4*>(p.post_scale + (size_t)n * p.cout + co + 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) { v[q] *= s0[q]; v[4 + q] *= s1[q]; }
            }
            stg16(yo + (size_t)pix * p.cout + co, Vec<f16>::pack(v));
        }
    }
}

template <int BC, int BP, int WC, int WP, int STAGES>
static int launch_dma_cfg(const ConvArgs& a, hipStream_t st) {
    constexpr int LDS = STAGES * (BC + BP) * 128;
    auto kern = conv_dma_kernel<BC, BP, WC, WP, STAGES>;
    static thread_local bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return mnet_fail(MNET_E_LAUNCH, "hipFuncSetAttribute(dma): %s", hipGetErrorString(e));
        attr_set = true;
    }
    ConvArgs b = a;
    b.tilesC = (a.cout + BC - 1) / BC;
    const int tilesP = (a.npix + BP - 1) / BP;
    hipLaunchKernelGGL(kern, dim3((unsigned)(b.tilesC * tilesP)), dim3(WC * WP * 64), LDS, st, b);
    MNET_LAUNCH_CHECK("conv_dma_kernel");
    return MNET_OK;
}

// eligibility of the LDS-DMA path (see header comment); the caller falls back to the register-staged kernel
bool conv_dma_eligible(const ConvArgs& a, int dtype) {
    if (dtype != MNET_F16 || a.in_scale) return false;
    if (a.cin % 64 != 0 || a.c0 % 64 != 0 || a.cout < 64 || a.cout % 8 != 0 || a.kh * a.kw > 32 || a.kh > 8 || a.kw > 8) return false;
    // 31-bit buffer offsets: a pixel tile may touch ceil(256/howo)+1 images
    const long long imgs = 512 / a.howo + 2;   // largest pixel tile is 512
    const long long per_img = (long long)a.h * a.w * (a.c0 > a.c1 ? a.c0 : a.c1) * 2;
    if (per_img * imgs >= 0x7fffffffLL) return false;
    if ((long long)256 * a.K * 2 >= 0x40000000LL) return false;
    if ((long long)a.cout * a.K * 2 >= 0x7fffffffLL) return false;
    return true;
}

// tile configurations (BC x BP, waves, LDS stages); MNET_CONV_ALGO_DMA_CFG0 + id selects one explicitly
static int launch_dma_id(int id, const ConvArgs& a, hipStream_t st) {
    switch (id) {
        case 0: return launch_dma_cfg<256, 256, 4, 4, 2>(a, st);
        case 1: return launch_dma_cfg<256, 128, 4, 2, 3>(a, st);
        case 2: return launch_dma_cfg<128, 256, 2, 4, 3>(a, st);
        case 3: return launch_dma_cfg<64, 256, 1, 8, 3>(a, st);
        case 4: return launch_dma_cfg<128, 512, 2, 8, 2>(a, st);
        case 5: return launch_dma_cfg<64, 512, 1, 8, 2>(a, st);
        case 6: return launch_dma_cfg<256, 256, 2, 4, 2>(a, st);     // 8 waves, 128x64 per wave (experimental)
        default: return mnet_fail(MNET_E_ARG, "conv: unknown LDS-DMA tile configuration %d", id);
    }
}

int conv_dma_pick(const ConvArgs& a) {
    const bool big = a.npix >= 256 * 256;
    if (a.cout >= 256) return big ? 0 : 1;
    if (a.cout >= 128) return big ? 4 : 2;
    return big ? 5 : 3;
}

int launch_conv_dma(const ConvArgs& a, hipStream_t st, int cfg) {
    return launch_dma_id(cfg >= 0 ? cfg : conv_dma_pick(a), a, st);
}
