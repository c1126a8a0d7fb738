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

typedef float f32x16 __attribute__((ext_vector_type(16)));

// store 8 consecutive output channels co..co+7 of pixel `pix` (DBG: diagnostic variants, see launch_dma_id)
template <int DBG>
__device__ __forceinline__ void store8(const ConvArgs& p, const float* v, int pix, int co) {
    if constexpr (DBG == 5) {           // DIAGNOSTIC: no stores (values kept live)
        asm volatile("" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
        return;
    }
    if constexpr (DBG == 4) pix &= 255;  // DIAGNOSTIC: every tile stores over tile 0 (L2-resident writes)
    stg16(reinterpret_cast<f16*>(p.y) + (size_t)pix * p.cout + co, Vec<f16>::pack(v));
}

// MF = 16: v_mfma_f32_16x16x32_f16 — production (same k association as conv_igemm.hip → bit-identical to it);
// MF = 32: v_mfma_f32_32x32x16_f16 — experimental (half the matrix instructions per slab, measured ~10 % slower in this loop)
template <int BC, int BP, int WC, int WP, int STAGES, int MF = 16, int DBG = 0, int PIPE = 0>
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
    long long stamp[4] = {0, 0, 0, 0};                   // DBG == 3 only: wall-clock (100 MHz) at entry / loop start / loop end / exit
    if constexpr (DBG >= 3) stamp[0] = wall_clock64();

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
    //   LDS row 64b + 16f + i   holds channel   64b + 32(f/2) + 8(i/4) + 4(f%2) + i%4        (MF = 16)
    unsigned woff[WJ];                                   // byte offset of this lane's weight chunk at k-slab 0
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
        const int row = (wave + NW * j) * 8 + rg;
        int ch;
        if constexpr (MF == 16) {
            const int f = (row >> 4) & 3, i = row & 15;
            ch = (row & ~63) + ((f >> 1) << 5) + ((i >> 2) << 3) + ((f & 1) << 2) + (i & 3);
        } else {   // 32x32 tile: D row i = 8q + 4h + e of lane-half h  →  channel 16h + 4q + e (16 consecutive per lane)
            const int i = row & 31;
            ch = (row & ~31) + (((i >> 2) & 1) << 4) + ((i >> 3) << 2) + (i & 3);
        }
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
        if constexpr (DBG == 2) { if (cur_k > 0) { cur_k += 64; return; } }   // diagnostic: no DMA after the first slab (results wrong)
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
                unsigned vo = (xmask[j] & tapbit) ? xb0[j] + uni : OOB;
                if constexpr (DBG == 1) vo = (xb0[j] & 0x3ffffu) + (unsigned)(cur_c * 2);   // diagnostic: activations from a 256 KiB window (L2-resident; results wrong)
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
    f32x16 acc32[MF == 32 ? FC / 2 : 1][MF == 32 ? FP / 2 : 1];
    if constexpr (MF == 16) {
#pragma unroll
        for (int a = 0; a < FC; ++a)
#pragma unroll
            for (int b = 0; b < FP; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
        for (int a = 0; a < FC / 2; ++a)
#pragma unroll
            for (int b = 0; b < FP / 2; ++b)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc32[a][b][q] = 0.f;
    }

    // fragments of one half slab (32 k): FC weight + FP activation 16-byte chunks per lane (MF = 32: [k-step][fragment])
    auto read_half = [&](int stage, int ks, u32x4* a, u32x4* b) __attribute__((always_inline)) {
        const unsigned char* sw_ = smem + stage * STAGE;
        const unsigned char* sx_ = sw_ + BC * 128;
        if constexpr (MF == 16) {
            const int chunk = ks * 4 + g;
#pragma unroll
            for (int f = 0; f < FC; ++f) a[f] = *reinterpret_cast<const u32x4*>(sw_ + swz_dma(wc * (BC / WC) + f * 16 + l16, chunk));
#pragma unroll
            for (int f = 0; f < FP; ++f) b[f] = *reinterpret_cast<const u32x4*>(sx_ + swz_dma(wp * (BP / WP) + f * 16 + l16, chunk));
        } else {
            const int l32 = lane & 31, h = lane >> 5;
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {                 // two 16-deep k-steps per half slab
                const int chunk = ks * 4 + k2 * 2 + h;
#pragma unroll
                for (int f = 0; f < FC / 2; ++f) a[k2 * (FC / 2) + f] = *reinterpret_cast<const u32x4*>(sw_ + swz_dma(wc * (BC / WC) + f * 32 + l32, chunk));
#pragma unroll
                for (int f = 0; f < FP / 2; ++f) b[k2 * (FP / 2) + f] = *reinterpret_cast<const u32x4*>(sx_ + swz_dma(wp * (BP / WP) + f * 32 + l32, chunk));
            }
        }
    };
    auto mfma_half = [&](const u32x4* a, const u32x4* b) __attribute__((always_inline)) {
        if constexpr (MF == 16) {
#pragma unroll
            for (int fa = 0; fa < FC; ++fa)
#pragma unroll
                for (int fb = 0; fb < FP; ++fb)
                    acc[fa][fb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bitcast<f16x8>(a[fa]), bitcast<f16x8>(b[fb]), acc[fa][fb], 0, 0, 0);
        } else {
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int fa = 0; fa < FC / 2; ++fa)
#pragma unroll
                    for (int fb = 0; fb < FP / 2; ++fb)
                        acc32[fa][fb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bitcast<f16x8>(a[k2 * (FC / 2) + fa]), bitcast<f16x8>(b[k2 * (FP / 2) + fb]),
                                                                               acc32[fa][fb], 0, 0, 0);
        }
    };
    auto compute_half = [&](int stage, int ks) __attribute__((always_inline)) {
        u32x4 a[FC], b[FP];
        read_half(stage, ks, a, b);
        mfma_half(a, b);
    };

    const int nk = p.ktiles;
    long long cyc = 0;
    if constexpr (DBG >= 3) { stamp[1] = wall_clock64(); cyc = (long long)__builtin_readcyclecounter(); }
    if constexpr (PIPE >= 1) {
        // ---- staggered two-group loop (16 waves, 2 LDS stages).  Every wave alternates R(q) = read the fragments of half
        // slab q from LDS and M(q) = its 16 MFMAs, one step per barrier interval; the waves (w/4) odd run one interval
        // behind the others, so on every SIMD two waves multiply while the other two read (the matrix pipe never waits for
        // the post-barrier LDS burst).  Interval i % 4 == 0 of slab t: every wave issues its DMA share of slab t+1 into the
        // stage whose last reads ended two barriers earlier; interval i % 4 == 3: vmcnt(0), then the barrier publishes it.
        static_assert(STAGES == 2 && NW == 16, "staggered loop: 16 waves, 2 stages");
        issue_slab(0);
        VMCNT(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const bool lag = (wave >> 2) & 1;          // wave-uniform: this wave runs one interval behind
        u32x4 ra[FC], rb[FP];
        auto R = [&](int stage, int ks) __attribute__((always_inline)) {
            read_half(stage, ks, ra, rb);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };
        auto M = [&]() __attribute__((always_inline)) {
            if constexpr (PIPE == 2) __builtin_amdgcn_s_setprio(1);
            mfma_half(ra, rb);
            if constexpr (PIPE == 2) __builtin_amdgcn_s_setprio(0);
        };
        auto BAR = [&]() __attribute__((always_inline)) {     // raw barrier; sched_barrier pins the (register-only) MFMAs to their interval
            __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0);
        };
        // Both groups run the SAME instruction stream R(0) | M(0) | R(1) | M(1) ... (| = barrier); the lagging group takes one
        // extra barrier before it and the leading group one after it.  Only the DMA issue points and the vmcnt(0) differ
        // (they are tied to the global interval index): slab t+1 is issued in interval 4t, waited for in interval 4t+3.
        if (lag) { if (nk > 1) issue_slab(1); BAR(); }
        for (int t = 0; t < nk; ++t) {
            const int st = t & 1;
            if (!lag && t + 1 < nk) issue_slab(st ^ 1);        // lead: interval 4t
            R(st, 0);
            BAR();
            M();
            BAR();
            R(st, 1);
            if (lag) VMCNT(0);                                 // lag: interval 4t+3
            BAR();
            if (lag && t + 2 < nk) issue_slab(st);             // lag: interval 4(t+1); its own reads of stage st are done
            M();
            if (!lag) VMCNT(0);                                // lead: interval 4t+3
            BAR();
        }
        if (!lag) BAR();
    } else if constexpr (STAGES == 3) {
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

    if constexpr (DBG >= 3) { stamp[2] = wall_clock64(); cyc = (long long)__builtin_readcyclecounter() - cyc; }
    // ---- epilogue (identical math to conv_igemm.hip).  Thanks to the weight-row permutation every lane owns NG groups of
    // 8 consecutive output channels for each of its NPX pixels.  The epilogue runs as whole-register-set passes, each
    // behind ONE wave-uniform branch (out_scale / bias / residual / activation / post_scale), then 16-byte stores.
    static_assert((BC / WC) % 64 == 0 && (BP / WP) % 32 == 0, "the channel permutation works on 64-channel blocks of a wave tile");
    constexpr int NPX = MF == 16 ? FP : FP / 2;          // pixels per lane
    constexpr int NG = MF == 16 ? FC / 2 : FC;           // 8-channel groups per pixel per lane
    float ev[NPX][NG][8];
    int epix[NPX], eco[NG];
#pragma unroll
    for (int gi = 0; gi < NG; ++gi)
        eco[gi] = MF == 16 ? co0 + wc * (BC / WC) + gi * 32 + g * 8
                           : co0 + wc * (BC / WC) + (gi >> 1) * 32 + (lane >> 5) * 16 + (gi & 1) * 8;
#pragma unroll
    for (int px = 0; px < NPX; ++px) {
        epix[px] = MF == 16 ? pix0 + wp * (BP / WP) + px * 16 + l16 : pix0 + wp * (BP / WP) + px * 32 + (lane & 31);
#pragma unroll
        for (int gi = 0; gi < NG; ++gi)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if constexpr (MF == 16) ev[px][gi][q] = acc[2 * gi + (q >> 2)][px][q & 3];
                else ev[px][gi][q] = acc32[gi >> 1][px][(gi & 1) * 8 + q];
            }
    }
    const int last_pix = p.npix - 1;
    if (p.out_scale) {
#pragma unroll
        for (int px = 0; px < NPX; ++px) {
            const float* sp = p.out_scale + (size_t)(min(epix[px], last_pix) / p.howo) * p.cout;
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) {
                if (eco[gi] >= p.cout) continue;
                const f32x4 s0 = *reinterpret_cast<const f32x4*>(sp + eco[gi]), s1 = *reinterpret_cast<const f32x4*>(sp + eco[gi] + 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) { ev[px][gi][q] *= s0[q]; ev[px][gi][4 + q] *= s1[q]; }
            }
        }
    }
    if (p.bias) {
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            if (eco[gi] >= p.cout) continue;
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + eco[gi]), b1 = *reinterpret_cast<const f32x4*>(p.bias + eco[gi] + 4);
#pragma unroll
            for (int px = 0; px < NPX; ++px)
#pragma unroll
                for (int q = 0; q < 4; ++q) { ev[px][gi][q] += b0[q]; ev[px][gi][4 + q] += b1[q]; }
        }
    }
    if (p.res) {
        const f16* rs = reinterpret_cast<const f16*>(p.res);
#pragma unroll
        for (int px = 0; px < NPX; ++px) {
            if (epix[px] >= p.npix) continue;
            const int rpix = p.res_mod > 0 ? epix[px] % p.res_mod : epix[px];
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) {
                if (eco[gi] >= p.cout) continue;
                const f16x8 r8 = bitcast<f16x8>(ldg16(rs + (size_t)rpix * p.cout + eco[gi]));
#pragma unroll
                for (int q = 0; q < 8; ++q) ev[px][gi][q] += (float)r8[q];
            }
        }
    }
    act_apply_vec<NPX * NG * 8, true>(&ev[0][0][0], p.act);     // this path only takes the cheap (branch-free) activations
    if (p.post_scale) {
#pragma unroll
        for (int px = 0; px < NPX; ++px) {
            const float* sp = p.post_scale + (size_t)(min(epix[px], last_pix) / p.howo) * p.cout;
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) {
                if (eco[gi] >= p.cout) continue;
                const f32x4 s0 = *reinterpret_cast<const f32x4*>(sp + eco[gi]), s1 = *reinterpret_cast<const f32x4*>(sp + eco[gi] + 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) { ev[px][gi][q] *= s0[q]; ev[px][gi][4 + q] *= s1[q]; }
            }
        }
    }
#pragma unroll
    for (int px = 0; px < NPX; ++px) {
        if (epix[px] >= p.npix) continue;
#pragma unroll
        for (int gi = 0; gi < NG; ++gi)
            if (eco[gi] < p.cout) store8<DBG>(p, ev[px][gi], epix[px], eco[gi]);
    }
    if constexpr (DBG >= 3) {       // DIAGNOSTIC: overwrite the head of this workgroup's output tile with its time stamps
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp[3] = wall_clock64();
        if (tid == 0) {
            long long* o = DBG == 3 ? reinterpret_cast<long long*>(reinterpret_cast<f16*>(p.y) + (size_t)pix0 * p.cout + co0)
                                    : reinterpret_cast<long long*>(reinterpret_cast<f16*>(p.y) + (size_t)(256 + 32 * (size_t)wg) * p.cout);
            o[0] = 0x7157a3b5ll; o[1] = stamp[0]; o[2] = stamp[1]; o[3] = stamp[2]; o[4] = stamp[3];
            o[5] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_ID
            o[6] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // XCC_ID
            o[7] = cyc;                                          // shader cycles spent in the k-loop (s_memtime)
        }
    }
}

template <int BC, int BP, int WC, int WP, int STAGES, int MF = 16, int DBG = 0, int PIPE = 0>
static int launch_dma_cfg(const ConvArgs& a, hipStream_t st) {
    constexpr int LDS = STAGES * (BC + BP) * 128;
    auto kern = conv_dma_kernel<BC, BP, WC, WP, STAGES, MF, DBG, PIPE>;
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
    if (dtype != MNET_F16 || a.in_scale || a.act > MNET_ACT_LRELU_SQRT2) return false;
    if (a.cin % 64 != 0 || a.c0 % 64 != 0 || a.cout < 64 || a.cout % 8 != 0 || a.kh * a.kw > 32 || a.kh > 8 || a.kw > 8) return false;
    // 31-bit buffer offsets: a pixel tile may touch ceil(256/howo)+1 images
    const long long imgs = 512 / a.howo + 2;   // largest pixel tile is 512
    const long long per_img = (long long)a.h * a.w * (a.c0 > a.c1 ? a.c0 : a.c1) * 2;
    if (per_img * imgs >= 0x7fffffffLL) return false;
    if ((long long)256 * a.K * 2 >= 0x40000000LL) return false;
    if ((long long)a.cout * a.K * 2 >= 0x7fffffffLL) return false;
    return true;
}

// tile configurations (BC x BP, waves, LDS stages, MFMA shape); MNET_CONV_ALGO_DMA_CFG0 + id selects one explicitly.
// Production ids 0-6 all use v_mfma_f32_16x16x32_f16 with the register-staged kernel's k association, so every f16 conv
// launch gives the same bits whatever kernel / tile configuration its size selects (batch-size-invariant results).
// ids 7-9: v_mfma_f32_32x32x16_f16 forms (measured ~10 % slower here; fp32 sums associate differently).
// ids 11-15: DIAGNOSTIC builds that produce wrong results on purpose (tools/wg_timeline.py, tools/conv_bench.py).
static int launch_dma_id(int id, const ConvArgs& a, hipStream_t st) {
    switch (id) {
        case 0: return launch_dma_cfg<256, 256, 4, 4, 2>(a, st);
        case 1: return launch_dma_cfg<256, 128, 4, 2, 3>(a, st);
        case 2: return launch_dma_cfg<128, 256, 2, 4, 3>(a, st);
        case 3: return launch_dma_cfg<64, 256, 1, 8, 3>(a, st);
        case 4: return launch_dma_cfg<128, 512, 2, 8, 2>(a, st);
        case 5: return launch_dma_cfg<64, 512, 1, 8, 2>(a, st);
        case 6: return launch_dma_cfg<256, 256, 2, 4, 2>(a, st);          // 8 waves, 128x64 per wave
        case 7: return launch_dma_cfg<256, 256, 4, 4, 2, 32>(a, st);
        case 8: return launch_dma_cfg<256, 256, 4, 4, 2, 16, 0, 1>(a, st);  // staggered two-group loop
        case 9: return launch_dma_cfg<256, 256, 4, 4, 2, 16, 0, 2>(a, st);  //   + s_setprio around the MFMA phase
        case 10: return launch_dma_cfg<256, 256, 4, 4, 2, 32, 0, 1>(a, st); //   on 32x32x16
        case 11: return launch_dma_cfg<256, 256, 4, 4, 2, 16, 4>(a, st);  // DIAGNOSTIC: all tiles store over tile 0; stamps after tile 0
        case 12: return launch_dma_cfg<256, 256, 4, 4, 2, 16, 5>(a, st);  // DIAGNOSTIC: no output stores; stamps after tile 0
        case 13: return launch_dma_cfg<256, 256, 4, 4, 2, 16, 3>(a, st);  // DIAGNOSTIC: per-workgroup time stamps written over the output
        case 14: return launch_dma_cfg<256, 256, 4, 4, 2, 16, 1>(a, st);  // DIAGNOSTIC: activations read from a 256 KiB window
        case 15: return launch_dma_cfg<256, 256, 4, 4, 2, 16, 2>(a, st);  // DIAGNOSTIC: no DMA after the first k-slab
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
