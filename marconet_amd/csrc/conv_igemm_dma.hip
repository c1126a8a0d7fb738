// Implicit-GEMM convolution, LDS-DMA variant (f16 storage, fp32 accumulate) — the fast path for every conv
// whose input-channel count is a multiple of 64 and that needs no input-side transform.
//
// What differs from conv_igemm.hip (the general, register-staged kernel):
//  * both operand tiles go HBM/L2 → LDS directly with `buffer_load_dwordx4 … lds` (1 KiB per wave-instruction):
//    no VGPR round trip, no ds_write pass.  Zero padding, ragged valid width, the cout tail and the pixel tail cost
//    nothing: an invalid lane gets an out-of-range buffer offset and the hardware writes zeros into its LDS slot
//    (raw buffer bounds check against num_records).
//  * the 16-byte-chunk XOR swizzle of the LDS image is applied on the *source* side (lane (row, phys chunk)
//    fetches logical chunk phys ^ ((row>>1)&7)), because an LDS-DMA destination is always lane-linear.  Weight rows
//    are additionally permuted on the source side so that every lane ends up owning 8 consecutive output channels
//    (16-byte epilogue loads/stores).
//  * PERSISTENT workgroups: one workgroup per CU walks tiles v = blockIdx, blockIdx + grid, ...  The k-slabs of all its
//    tiles form ONE stream through the LDS ring (2 or 3 stages); the DMA of the next tile's first slab(s) is issued
//    during the last slab(s) of the current tile, so the epilogue, the tile set-up and the first-slab latency of the next
//    tile overlap (measured per 256x256 tile before this: 1.7 us prologue + 3.5 us inter-workgroup gap of 69 us).
//  * one raw s_barrier per 64-deep k-slab; 3-stage configs keep slab t+1 in flight across the barrier of slab t with a
//    counted `s_waitcnt vmcnt(N)`.
//  * with cin % 64 == 0 a k-slab never straddles a filter tap or the concat boundary, so tap (r,s), source
//    tensor and channel offset are wave-uniform scalars advanced incrementally — no integer division in the loop.
//  * the epilogue runs as whole-register-set passes behind wave-uniform branches (a per-element `switch (act)` compiled
//    to ~1000 scalar branches / 100 KB of code and cost 10 us per tile).
#include <type_traits>
#include "conv_dma_common.h"

// MF = 16: v_mfma_f32_16x16x32_f16 — production;
// MF = 32: v_mfma_f32_32x32x16_f16 — experimental (half the matrix instructions per slab; not faster here: the kernel is
//          power/clock limited)
// X3: split-half launch (MNET_F16X2, the fp16x3 precision mode).  The DMA side is unchanged — the tensors are walked as f16
//     with twice the channels, so one 128-byte slab row is one 32-channel block: chunks 0-3 hi, chunks 4-7 lo — and every slab is
//     multiplied three times (hi*hi, hi*lo, lo*hi) from the one LDS image: 1.5x the MFMA work per slab, per barrier and per byte
//     moved of the f16 kernel.
// SPREAD (hot iterations of 2-stage tiles): instead of issuing all of a slab's DMA pieces right after the slab barrier — when every
//     wave of the workgroup does the same and the matrix pipe idles — the weight pieces go out after the first third / half of
//     the slab's multiplies and the activation pieces after the second.
// MX (with X3, MF = 32): "fp16+8" launch (MNET_F16M) — same 128-byte block per 32 channels, but only the hi part is a half:
//     activations  chunks 0-3 hi (f16) | chunk 4 lo8 of channels {0-7,16-23} | chunk 5 lo8 of {8-15,24-31} | chunk 6 byte 0: E8M0 scale
//     weights      chunks 0-3 hi (f16) | chunk 4 lo8 {0-7,16-23} | chunk 5 hi8 {0-7,16-23} | chunk 6 lo8 {8-15,24-31} | chunk 7 hi8 {8-15,24-31}
//     (lo8 = e4m3(lo * 2^11 / s), hi8 = e4m3(hi / s), s = 2^(E - 127): per (pixel, block) for activations, per output channel for weights).
//     x*w = hi*hi on v_mfma_f32_32x32x16_f16 + (w_lo8*x_hi8 + w_hi8*x_lo8) as ONE v_mfma_scale_f32_32x32x64_f8f6f4 (2x rate, the
//     block scales s_w * s_x * 2^-11 applied by the instruction): 2 MFMA units per product instead of 3.  x_hi8 is converted from
//     the f16 fragments the lane already holds (v_cvt_scalef32_pk_fp8_f16).
// SWP (with MX, 8 waves, 2 stages; round 4): the slab loop SOFTWARE-PIPELINED across the slab barrier.  Measured on the lock-step loop
//     (tools/slab_phases.py, profiles/r4c_slab_phases*.txt): of ~3350 cycles per slab the matrix pipe works 2048 — the younger wave of a SIMD spends
//     1600 cycles on the 16 f16 MFMAs (LDS latency after the barrier, the pipe shared with its partner), 830 issuing the 8 DMA pieces in one
//     block and 680 on conversions + 8 scaled MFMAs, while its partner waits 900 cycles at the barrier.  Here a slab iteration is
//         barrier(s) | LDS reads of slab s's f16 operands | 8 scaled MFMAs of slab s-1, ONE DMA piece of slab s+1 behind each (64 pipe cycles
//         cover a piece's issue) | 16 f16 MFMAs of slab s with the fp8-side LDS reads and the fp8 conversions of slab s between them
//     so that the LDS latency sits under the previous slab's scaled MFMAs, the DMA pieces get the whole slab to land, and no phase is without
//     MFMAs.  Per output the order stays f16 k-step 0, f16 k-step 1, scaled fp8 of slab 0, then slab 1, ...: the bytes equal the other fp16+8 tiles'.
//     LDS hazards: slab s-1's stage is last read (fp8-side operands) during the f16 MFMAs of iteration s-1, before barrier(s); the DMA of slab
//     s+1 into that stage starts after barrier(s).
// SGN (with SWP): the software-pipelined tile WITH the GroupNorm-sum block in its epilogue.  Only conv_dma_swp_gn.hip instantiates it — a translation unit of its own,
//     compiled with `-mllvm -greedy-reverse-local-assignment=1`: with hipcc's default assignment order that block moves a spill of the slab loop onto its hot path (three
//     formulations tried), with the reverse order the tile is clean (31 spilled registers, none hot: tools/isa_hot_scratch.py).  The flag is kept away from every other tile.
template <int BC, int BP, int WC, int WP, int STAGES, int MF = 16, int DBG = 0, bool X3 = false, bool SPREAD = false, bool PIPE = false, bool MX = false, bool SWP = false, bool SGN = false>
__global__ void __launch_bounds__(WC * WP * 64, WC * WP / 4) conv_dma_kernel(const ConvArgs p) {
    constexpr int NW = WC * WP;                          // waves per workgroup (8 or 16)
    constexpr int FC = BC / WC / 16, FP = BP / WP / 16;
    constexpr int NWI = NW;                              // waves that issue DMA: all of them (an asymmetric form — the older wave of every SIMD issuing all
                                                         // the pieces — measured -27 %, a one-wave-per-SIMD 128x128 form spilled 450 registers: DESIGN.md §3.1e)
    constexpr int WJ = BC / (8 * NWI), XJ = BP / (8 * NWI); // DMA instructions per wave per slab (weights / activations)
    constexpr int NDMA = WJ + XJ;
    constexpr int STAGE = (BC + BP) * 128;
    constexpr unsigned OOB = 0x80000000u;                // beyond every num_records used below
    static_assert(NW == 8 || NW == 16, "8 or 16 waves");
    static_assert(!MX || (X3 && MF == 32), "MX: 4-byte storage, 32x32 MFMAs");
    static_assert(!SGN || SWP, "SGN: software-pipelined tiles only");
    static_assert(!SWP || (MX && NW == 8 && STAGES == 2 && (DBG == 0 || DBG == 6) && (FC / 2) * (FP / 2) <= 16 && NDMA <= 16), "SWP: fp16+8, 2 stages, <= 16 accumulator blocks per wave");
    static_assert(!SPREAD || (STAGES == 2 && (MF == 16 || X3) && (DBG == 0 || DBG == 6)), "SPREAD: production 2-stage tiles only");
    static_assert(WJ >= 1 && XJ >= 1 && WJ * 8 * NWI == BC && XJ * 8 * NWI == BP, "tile / wave-count mismatch");
    static_assert(STAGES == 2 || ((STAGES == 3 || STAGES == 4) && NDMA >= 4 && NDMA <= 6), "vmcnt immediates below cover 4-6 DMAs per slab, up to 3 slabs in flight");
    static_assert((BC / WC) % 64 == 0 && (BP / WP) % 32 == 0, "the channel permutation works on 64-channel blocks of a wave tile");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // fp16+8 tiles whose stages leave 4 KiB of LDS per wave: the epilogue's stores go through it (dma_epilogue_mx, conv_dma_common.h)
    constexpr int XB = dma_mx_xpose_bytes<NW, MX>(STAGES * STAGE);     // 4096, 1024 or 0 bytes per wave
    unsigned char* const xpose = XB ? smem + STAGES * STAGE + wave * XB : nullptr;
    const int wc = wave / WP, wp = wave % WP;
    const int l16 = lane & 15, g = lane >> 4;
    const int rg = lane >> 3, pc = lane & 7;             // DMA geometry: lane fills LDS row (w + NW j)*8 + rg, 16-byte slot pc
    const int G = gridDim.x, ntiles = p.ntiles, nk = p.ktiles;

    // XCD-aware bijective tile map: virtual block v (v % 8 = the XCD it runs on, G % 8 == 0 or a single pass) → tile
    const int q8 = ntiles >> 3, r8 = ntiles & 7;
    auto tile_coords = [&](int v, int& co0, int& pix0) __attribute__((always_inline)) {
        const int xcd = v & 7;
        const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (v >> 3);
        co0 = (t % p.tilesC) * BC;
        pix0 = (t / p.tilesC) * BP;
    };

    // ================================================================== DMA issue side: one slab stream over all tiles
    // wave-uniform base pointers / byte bounds of the tile being streamed; the buffer descriptors are rebuilt from them
    // (through readfirstlane, so that they provably live in SGPRs: a descriptor the compiler believes divergent makes it
    // wrap every DMA in a waterfall loop) at each issue
    unsigned long long bW = 0, bX0 = 0, bX1 = 0;
    int nW = 0, nX0 = 0, nX1 = 0;
    auto uni64 = [](unsigned long long v) __attribute__((always_inline)) -> unsigned long long {
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return ((unsigned long long)hi << 32) | lo;
    };
    unsigned woff[WJ];                                   // byte offset of this lane's weight chunk at k-slab 0
    int woff_co0 = -1;                                   // the channel tile woff[] was computed for
    unsigned xpx[XJ], xmask[XJ];                         // activation rows: input pixel of tap 0 (relative to the tile's first image); valid-tap bits
    // 16-byte slot inside the 128-byte slab row, XOR-swizzled on the source side.  Row (wave + NW j)*8 + rg → (row >> 1) & 7 =
    // 4*(wave & 1) + (rg >> 1) for every j (NW is even): ONE value per lane, not one per piece
    const unsigned lcb = (unsigned)((pc ^ (((wave & 1) << 2) + (rg >> 1))) << 4);
    static_assert(NWI % 2 == 0, "lcb is piece-independent only for an even issuing-wave count");
    int cur_c = 0, cur_s = 0, cur_tap = 0, cur_tpx = 0, cur_k = 0;   // wave-uniform k-slab cursor
    const long long img0 = (long long)p.h * p.w * p.c0 * 2, img1 = (long long)p.h * p.w * p.c1 * 2;

    unsigned rowrep = 0;                                 // bit r * kw for every filter row r (wave-uniform; the tap masks of setup() are products with it)
#pragma unroll
    for (int r = 0; r < 8; ++r)
        if (r < p.kh) rowrep |= 1u << (r * p.kw);
    auto setup = [&](int v) __attribute__((always_inline)) {
        int co0, pix0;
        tile_coords(v, co0, pix0);
        // Weights: rows co0.. ; bound = remaining rows → cout tail reads zeros.
        const long long wbytes = (long long)(p.cout - co0) * p.K * 2;
        bW = (unsigned long long)(reinterpret_cast<const f16*>(p.wgt) + (size_t)co0 * p.K);
        nW = (int)(wbytes < 0x7fffffffLL ? wbytes : 0x7fffffffLL);
        // Activations: base = first image touched by this pixel tile; bound = the images the tile can touch.
        const bool p2 = p.howo_shift >= 0 && p.wo_shift >= 0;          // (wave-uniform) every map of this network: shifts instead of divisions
        const int n_first = p2 ? pix0 >> p.howo_shift : pix0 / p.howo;
        const int n_last = p2 ? (min(pix0 + BP, p.npix) - 1) >> p.howo_shift : (min(pix0 + BP, p.npix) - 1) / p.howo;
        const int nimg = n_last - n_first + 1;
        bX0 = (unsigned long long)(reinterpret_cast<const char*>(p.x0) + (size_t)n_first * img0);
        nX0 = (int)(img0 * nimg);
        bX1 = (unsigned long long)(p.x1 ? reinterpret_cast<const char*>(p.x1) + (size_t)n_first * img1 : reinterpret_cast<const char*>(p.x0));
        nX1 = (int)(p.x1 ? img1 * nimg : 0);
        // Weight rows are permuted on their way into LDS (free: the DMA source address is per lane), see dma_weight_channel
        if (co0 != woff_co0) {                                          // (wave-uniform; one channel tile: computed once per launch)
            woff_co0 = co0;
#pragma unroll
            for (int j = 0; j < WJ; ++j) {
                const int row = (wave + NWI * j) * 8 + rg;
                const int ch = MX ? dma_weight_channel_mx(row) : dma_weight_channel<MF>(row);
                const int lc = pc ^ ((row >> 1) & 7);
                woff[j] = (co0 + ch < p.cout) ? (unsigned)(ch * p.K * 2 + lc * 16) : OOB;
            }
        }
        // activation rows: a bit mask of the filter taps whose input pixel exists (inside the image and left of valid_w); kh*kw <= 32 on this path.
        // This runs once per tile between a slab barrier and the slab's MFMAs, on both waves of every SIMD at once (measured: 12 900 cycles per
        // tile, 5 % of a 72-slab tile, tools/slab_phases.py): the tap mask is closed-form — the valid columns / rows of a window are RANGES, the
        // mask is (column range) * (row range of `rowrep`, one bit per filter row) — instead of 16 compare-and-or steps per row.
#pragma unroll
        for (int j = 0; j < XJ; ++j) {
            const int row = (wave + NWI * j) * 8 + rg;
            const int pix = pix0 + row;
            const int pixc = min(pix, p.npix - 1);                               // (clamped: rows beyond the tensor get an empty mask below)
            int n, rem, oh, ow;
            if (p2) { n = pixc >> p.howo_shift; rem = pixc & (p.howo - 1); oh = rem >> p.wo_shift; ow = rem & (p.wo - 1); }
            else { n = pixc / p.howo; rem = pixc - n * p.howo; oh = rem / p.wo; ow = rem - oh * p.wo; }
            const int xvwj = p.valid_w ? p.valid_w[n] : p.w;
            const int ih0 = oh * p.sh - p.ph, iw0 = ow * p.sw - p.pw;
            const int px = ((n - n_first) * p.h + ih0) * p.w + iw0;
            const int vw = min(xvwj, p.w);
            // taps enumerated t = r*kw + q: columns q in [qlo, qhi), rows r in [rlo, rhi)
            const int qlo = min(p.kw, max(0, -iw0)), qhi = min(p.kw, vw - iw0), rlo = min(p.kh, max(0, -ih0)), rhi = min(p.kh, p.h - ih0);   // (clamped: a padding beyond the filter must not reach the shifts below)
            const unsigned cm = ((1u << max(qhi, 0)) - 1u) & ~((1u << qlo) - 1u);                     // kw <= 8; qhi <= qlo gives 0
            const int sh_hi = max(rhi, 0) * p.kw, sh_lo = rlo * p.kw;                                    // <= 32 (eligibility: kh * kw <= 32)
            const unsigned rr = (sh_hi >= 32 ? rowrep : rowrep & ((1u << sh_hi) - 1u)) & (sh_lo >= 32 ? 0u : ~((1u << sh_lo) - 1u));
            const bool live = pix < p.npix && !(MX && (lcb >> 4) == 7u && !p.mx_fetch_pad);            // chunk 7 of an fp16+8 activation block is padding: not fetched
            xpx[j] = pix < p.npix ? (unsigned)px : 0u;           // may be negative (padding taps); |px| < 2^23 and |px * channels * 2| < 2^31 (eligibility): mul_i24 is exact
            xmask[j] = live ? cm * rr : 0u;                      // (cm < 2^kw and rr has one bit per filter row at multiples of kw: the product is the OR of the shifted column masks)
        }
        cur_c = 0; cur_s = 0; cur_tap = 0; cur_tpx = 0; cur_k = 0;
    };

    // one k-slab = WJ weight pieces + XJ activation pieces (1 KiB per wave-instruction each); issue_w / issue_x may be called apart
    // (SPREAD) — issue_x advances the slab cursor
    auto issue_w = [&](int stage) __attribute__((always_inline)) {
        unsigned char* sw_ = smem + stage * STAGE;
        // k-slab order: 64-channel slice OUTER, filter tap INNER — the 9 taps of a 3x3 filter re-read the same three input
        // rows of one 64-channel slice back to back (≈100 KB per tile) instead of coming back to them after a pass over all
        // channels, so the re-reads, and the rows shared with the vertically neighbouring tiles, hit in the XCD's L2.
        const unsigned kb = (unsigned)(cur_tap * p.cin + cur_c) * 2u;
        if constexpr (DBG == 2) { if (cur_k > 0) return; }   // DIAGNOSTIC: no DMA after a tile's first slab
        const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)uni64(bW), 0, __builtin_amdgcn_readfirstlane(nW), 0x00020000);
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            const unsigned vo = woff[j] == OOB ? OOB : woff[j] + kb;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_void*)(sw_ + (wave + NW * j) * 1024), 16, vo, 0, 0, 0);
        }
    };
    // k-slab cursor: filter taps inner, 64-channel slices outer; with p.x1_center the slices of the SECOND source have one slab each, at the
    // centre tap (the 1x1 skip convolution folded into the k-loop, MNET_CONV_ALGO_FLAG_X1_CENTER)
    auto advance_cursor = [&]() __attribute__((always_inline)) {
        cur_k += 64;
        if (p.x1_center && cur_c >= p.c0) { cur_c += 64; return; }
        ++cur_tap;
        if (++cur_s == p.kw) { cur_s = 0; cur_tpx += p.w - (p.kw - 1); } else { ++cur_tpx; }
        if (cur_tap == p.kh * p.kw) {
            cur_tap = 0; cur_s = 0; cur_tpx = 0; cur_c += 64;
            if (p.x1_center && cur_c >= p.c0) { cur_tap = p.center_tap; cur_tpx = p.center_tpx; }
        }
    };
    auto issue_x = [&](int stage) __attribute__((always_inline)) {
        unsigned char* sx_ = smem + stage * STAGE + BC * 128;
        bool skip = false;
        if constexpr (DBG == 2) skip = cur_k > 0;
        const unsigned tapbit = 1u << cur_tap;
        if (skip) {
        } else if (cur_c >= p.c0) {                      // wave-uniform: second concat source
            const unsigned uni = (unsigned)(cur_tpx * p.c1 * 2 + (cur_c - p.c0) * 2) + lcb, cb = (unsigned)p.c1 * 2u;
            const __amdgpu_buffer_rsrc_t rX1 = __builtin_amdgcn_make_buffer_rsrc((void*)uni64(bX1), 0, __builtin_amdgcn_readfirstlane(nX1), 0x00020000);
#pragma unroll
            for (int j = 0; j < XJ; ++j) {
                const unsigned vo = (xmask[j] & tapbit) ? (unsigned)__mul24((int)xpx[j], (int)cb) + uni : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rX1, (lds_void*)(sx_ + (wave + NW * j) * 1024), 16, vo, 0, 0, 0);
            }
        } else {
            const unsigned uni = (unsigned)(cur_tpx * p.c0 * 2 + cur_c * 2) + lcb, cb = (unsigned)p.c0 * 2u;
            const __amdgpu_buffer_rsrc_t rX0 = __builtin_amdgcn_make_buffer_rsrc((void*)uni64(bX0), 0, __builtin_amdgcn_readfirstlane(nX0), 0x00020000);
#pragma unroll
            for (int j = 0; j < XJ; ++j) {
                unsigned vo = (xmask[j] & tapbit) ? (unsigned)__mul24((int)xpx[j], (int)cb) + uni : OOB;
                if constexpr (DBG == 1) vo = (((unsigned)__mul24((int)xpx[j], (int)cb) + lcb) & 0x3ffffu) + (unsigned)(cur_c * 2);   // DIAGNOSTIC: activations from a 256 KiB window
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rX0, (lds_void*)(sx_ + (wave + NW * j) * 1024), 16, vo, 0, 0, 0);
            }
        }
        advance_cursor();
    };
    auto issue_slab = [&](int stage) __attribute__((always_inline)) { issue_w(stage); issue_x(stage); };

    int i_v = blockIdx.x, i_kt = 0, i_stage = 0;         // head of the slab stream: tile, slab, LDS stage
    bool i_live = true;
    auto issue_hot = [&]() __attribute__((always_inline)) {          // next slab of the SAME tile (caller guarantees i_kt < nk)
        issue_slab(i_stage);
        i_stage = i_stage == STAGES - 1 ? 0 : i_stage + 1;
        ++i_kt;
    };
    auto issue_next = [&]() __attribute__((always_inline)) -> int {  // next slab of the stream, crossing into the next tile if needed
        if (i_kt == nk) {
            i_kt = 0; i_v += G;
            i_live = i_v < ntiles;
            if (i_live) setup(i_v);
        }
        if (!i_live) return 0;
        issue_hot();
        return 1;
    };

    // ================================================================== compute side
    f32x4 acc[FC][FP];
    f32x16 acc32[MF == 32 ? FC / 2 : 1][MF == 32 ? FP / 2 : 1];

    // fragments of one half slab (32 k): FC weight + FP activation 16-byte chunks per lane (MF = 32: [k-step][fragment])
    auto compute_half = [&](int stage, int ks) __attribute__((always_inline)) {
        const unsigned char* sw_ = smem + stage * STAGE;
        const unsigned char* sx_ = sw_ + BC * 128;
        if constexpr (MF == 16) {
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
        } else {
            const int l32 = lane & 31, h = lane >> 5;
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {                 // two 16-deep k-steps per half slab
                const int chunk = ks * 4 + k2 * 2 + h;
                u32x4 a[FC / 2], b[FP / 2];
#pragma unroll
                for (int f = 0; f < FC / 2; ++f) a[f] = *reinterpret_cast<const u32x4*>(sw_ + swz_dma(wc * (BC / WC) + f * 32 + l32, chunk));
#pragma unroll
                for (int f = 0; f < FP / 2; ++f) b[f] = *reinterpret_cast<const u32x4*>(sx_ + swz_dma(wp * (BP / WP) + f * 32 + l32, chunk));
#pragma unroll
                for (int fa = 0; fa < FC / 2; ++fa)
#pragma unroll
                    for (int fb = 0; fb < FP / 2; ++fb)
                        acc32[fa][fb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bitcast<f16x8>(a[fa]), bitcast<f16x8>(b[fb]), acc32[fa][fb], 0, 0, 0);
            }
        }
    };

    // f16 slab with both half slabs' fragments requested up front (PIPE, 8-wave tiles: 256 VGPRs): the second half's LDS reads run
    // under the first half's MFMAs; the next slab's DMA pieces go out between the halves
    auto compute_f16_pipe = [&](int stage, auto&& between) __attribute__((always_inline)) {
        const unsigned char* sw_ = smem + stage * STAGE;
        const unsigned char* sx_ = sw_ + BC * 128;
        u32x4 a0[FC], b0[FP], a1[FC], b1[FP];
#pragma unroll
        for (int f = 0; f < FC; ++f) a0[f] = *reinterpret_cast<const u32x4*>(sw_ + swz_dma(wc * (BC / WC) + f * 16 + l16, g));
#pragma unroll
        for (int f = 0; f < FP; ++f) b0[f] = *reinterpret_cast<const u32x4*>(sx_ + swz_dma(wp * (BP / WP) + f * 16 + l16, g));
#pragma unroll
        for (int f = 0; f < FC; ++f) a1[f] = *reinterpret_cast<const u32x4*>(sw_ + swz_dma(wc * (BC / WC) + f * 16 + l16, 4 + g));
#pragma unroll
        for (int f = 0; f < FP; ++f) b1[f] = *reinterpret_cast<const u32x4*>(sx_ + swz_dma(wp * (BP / WP) + f * 16 + l16, 4 + g));
#pragma unroll
        for (int fa = 0; fa < FC; ++fa)
#pragma unroll
            for (int fb = 0; fb < FP; ++fb)
                acc[fa][fb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bitcast<f16x8>(a0[fa]), bitcast<f16x8>(b0[fb]), acc[fa][fb], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, FC + FP, 0);
#pragma unroll
        for (int i = 0; i < FC + FP; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, (FC * FP) / (FC + FP), 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, (FC * FP) - ((FC * FP) / (FC + FP)) * (FC + FP), 0);
        between();
#pragma unroll
        for (int fa = 0; fa < FC; ++fa)
#pragma unroll
            for (int fb = 0; fb < FP; ++fb)
                acc[fa][fb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bitcast<f16x8>(a1[fa]), bitcast<f16x8>(b1[fb]), acc[fa][fb], 0, 0, 0);
    };

    // split-half slab: 32 channels, hi halves in chunks 0-3 and lo halves in chunks 4-7 of every row; x*w = hi*hi + hi*lo + lo*hi
    // (lo*lo is below fp32 resolution) — the same fp32 accumulators take all three products
    auto compute_x3 = [&](int stage, auto&& after_first, auto&& after_second) __attribute__((always_inline)) {
        const unsigned char* sw_ = smem + stage * STAGE;
        const unsigned char* sx_ = sw_ + BC * 128;
        u32x4 a[FC], bh[FP], bl[FP];
#pragma unroll
        for (int f = 0; f < FC; ++f) a[f] = *reinterpret_cast<const u32x4*>(sw_ + swz_dma(wc * (BC / WC) + f * 16 + l16, g));
#pragma unroll
        for (int f = 0; f < FP; ++f) bh[f] = *reinterpret_cast<const u32x4*>(sx_ + swz_dma(wp * (BP / WP) + f * 16 + l16, g));
        if constexpr (PIPE) {           // lo activation fragments fetched under the first group's MFMAs (no more registers live than at the second group)
#pragma unroll
            for (int f = 0; f < FP; ++f) bl[f] = *reinterpret_cast<const u32x4*>(sx_ + swz_dma(wp * (BP / WP) + f * 16 + l16, 4 + g));
        }
#pragma unroll
        for (int fa = 0; fa < FC; ++fa)
#pragma unroll
            for (int fb = 0; fb < FP; ++fb)
                acc[fa][fb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bitcast<f16x8>(a[fa]), bitcast<f16x8>(bh[fb]), acc[fa][fb], 0, 0, 0);
        if constexpr (PIPE) {
            __builtin_amdgcn_sched_group_barrier(0x100, FC + FP, 0);       // all hi fragments up front (they are all needed before the pipe fills anyway)
#pragma unroll
            for (int i = 0; i < FP; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, FC, 0);        // FC MFMAs ...
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);         // ... one lo activation fragment
            }
        }
        after_first();
        if constexpr (!PIPE) {
#pragma unroll
            for (int f = 0; f < FP; ++f) bl[f] = *reinterpret_cast<const u32x4*>(sx_ + swz_dma(wp * (BP / WP) + f * 16 + l16, 4 + g));
        }
#pragma unroll
        for (int fa = 0; fa < FC; ++fa)
#pragma unroll
            for (int fb = 0; fb < FP; ++fb)
                acc[fa][fb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bitcast<f16x8>(a[fa]), bitcast<f16x8>(bl[fb]), acc[fa][fb], 0, 0, 0);
        after_second();
#pragma unroll
        for (int f = 0; f < FC; ++f) a[f] = *reinterpret_cast<const u32x4*>(sw_ + swz_dma(wc * (BC / WC) + f * 16 + l16, 4 + g));
#pragma unroll
        for (int fa = 0; fa < FC; ++fa)
#pragma unroll
            for (int fb = 0; fb < FP; ++fb)
                acc[fa][fb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bitcast<f16x8>(a[fa]), bitcast<f16x8>(bh[fb]), acc[fa][fb], 0, 0, 0);
    };

    // split-half slab on v_mfma_f32_32x32x16_f16: lane = (row lane & 31, k half h = lane >> 5); the 16-deep k-step k2 of the hi part
    // is chunk 2 k2 + h, of the lo part chunk 4 + 2 k2 + h.  Half the matrix instructions (and operand register reads) per FLOP of
    // the 16x16x32 form
    auto compute_x3_32 = [&](int stage, auto&& between) __attribute__((always_inline)) {
        const unsigned char* sw_ = smem + stage * STAGE;
        const unsigned char* sx_ = sw_ + BC * 128;
        const int l32 = lane & 31, h = lane >> 5;
        constexpr int FA = FC / 2, FB = FP / 2;
        u32x4 a[2][FA], bh[2][FB], bl[2][FB];
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
#pragma unroll
            for (int f = 0; f < FA; ++f) a[k2][f] = *reinterpret_cast<const u32x4*>(sw_ + swz_dma(wc * (BC / WC) + f * 32 + l32, 2 * k2 + h));
#pragma unroll
            for (int f = 0; f < FB; ++f) bh[k2][f] = *reinterpret_cast<const u32x4*>(sx_ + swz_dma(wp * (BP / WP) + f * 32 + l32, 2 * k2 + h));
        }
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int fa = 0; fa < FA; ++fa)
#pragma unroll
                for (int fb = 0; fb < FB; ++fb)
                    acc32[fa][fb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bitcast<f16x8>(a[k2][fa]), bitcast<f16x8>(bh[k2][fb]), acc32[fa][fb], 0, 0, 0);
        between();
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int f = 0; f < FB; ++f) bl[k2][f] = *reinterpret_cast<const u32x4*>(sx_ + swz_dma(wp * (BP / WP) + f * 32 + l32, 4 + 2 * k2 + h));
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int fa = 0; fa < FA; ++fa)
#pragma unroll
                for (int fb = 0; fb < FB; ++fb)
                    acc32[fa][fb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bitcast<f16x8>(a[k2][fa]), bitcast<f16x8>(bl[k2][fb]), acc32[fa][fb], 0, 0, 0);
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int f = 0; f < FA; ++f) a[k2][f] = *reinterpret_cast<const u32x4*>(sw_ + swz_dma(wc * (BC / WC) + f * 32 + l32, 4 + 2 * k2 + h));
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int fa = 0; fa < FA; ++fa)
#pragma unroll
                for (int fb = 0; fb < FB; ++fb)
                    acc32[fa][fb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bitcast<f16x8>(a[k2][fa]), bitcast<f16x8>(bh[k2][fb]), acc32[fa][fb], 0, 0, 0);
    };

    // DBG == 6 (DIAGNOSTIC, fp16+8 id 14; tools/slab_phases.py): shader cycles a wave spends in each phase of a hot slab, summed over the launch —
    //   0: LDS reads + the 16 f16 MFMAs   1: issue of the next slab's DMA pieces   2: fp8 conversions + the 8 scaled MFMAs
    //   3: s_waitcnt vmcnt(0)             4: s_barrier
    // (s_memtime needs lgkmcnt(0): each stamp also drains the wave's LDS reads — the stamps sit where the data is needed anyway)
    unsigned ph_sum[6] = {0u, 0u, 0u, 0u, 0u, 0u}, ph_prev = 0u, ph_slabs = 0u;      // (software-pipelined form: 0 = the tile-closing epilogue, 5 = the vmcnt(0) right after it)
    auto ph_stamp = [&](int i) __attribute__((always_inline)) {
        if constexpr (DBG == 6) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned t = (unsigned)__builtin_readcyclecounter();
            if (i >= 0) ph_sum[i] += t - ph_prev;
            ph_prev = t;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // fp16+8 slab (see the MX note above the kernel)
    int mx_sa[MF == 32 ? FC / 2 : 1];                      // per weight fragment: E8M0 byte of s_w * 2^-11 for this lane's row (constant over k)
    // PIPE: every LDS read of the slab is requested in program order up front and placed by scheduling hints — the f16 operands before
    // the first MFMA, the fp8-side operands one per MFMA under the f16 products (their registers are the ones the first k-step frees)
    auto compute_mx = [&](int stage, auto&& between) __attribute__((always_inline)) {
        const unsigned char* sw_ = smem + stage * STAGE;
        const unsigned char* sx_ = sw_ + BC * 128;
        const int l32 = lane & 31, h = lane >> 5;
        constexpr int FA = FC / 2, FB = FP / 2;
        u32x4 a[2][FA], bh[2][FB];
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
#pragma unroll
            for (int f = 0; f < FA; ++f) a[k2][f] = *reinterpret_cast<const u32x4*>(sw_ + swz_dma(wc * (BC / WC) + f * 32 + l32, 2 * k2 + h));
#pragma unroll
            for (int f = 0; f < FB; ++f) bh[k2][f] = *reinterpret_cast<const u32x4*>(sx_ + swz_dma(wp * (BP / WP) + f * 32 + l32, 2 * k2 + h));
        }
        i32x8 b8[FB], a8[FA];
        int eb[FB];
#pragma unroll
        for (int f = 0; f < FB; ++f) {
            const unsigned char* row = sx_ + (wp * (BP / WP) + f * 32 + l32) * 128;
            const int sw3 = ((wp * (BP / WP) + f * 32 + l32) >> 1) & 7;
            eb[f] = *(row + ((6 ^ sw3) << 4));                              // E8M0 of the block scale s_x
            const u32x4 lo8 = *reinterpret_cast<const u32x4*>(row + (((4 + h) ^ sw3) << 4));
            b8[f][4] = (int)lo8[0]; b8[f][5] = (int)lo8[1]; b8[f][6] = (int)lo8[2]; b8[f][7] = (int)lo8[3];
        }
        if constexpr (PIPE) {
#pragma unroll
            for (int f = 0; f < FA; ++f) {
                const u32x4 lo8 = *reinterpret_cast<const u32x4*>(sw_ + swz_dma(wc * (BC / WC) + f * 32 + l32, 4 + 2 * h));
                const u32x4 hi8 = *reinterpret_cast<const u32x4*>(sw_ + swz_dma(wc * (BC / WC) + f * 32 + l32, 5 + 2 * h));
                a8[f] = i32x8{(int)lo8[0], (int)lo8[1], (int)lo8[2], (int)lo8[3], (int)hi8[0], (int)hi8[1], (int)hi8[2], (int)hi8[3]};
            }
        }
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int fa = 0; fa < FA; ++fa)
#pragma unroll
                for (int fb = 0; fb < FB; ++fb)
                    acc32[fa][fb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bitcast<f16x8>(a[k2][fa]), bitcast<f16x8>(bh[k2][fb]), acc32[fa][fb], 0, 0, 0);
        if constexpr (PIPE) {
            constexpr int NF16 = 2 * (FA + FB), NF8 = 2 * FB + 2 * FA, NM = 2 * FA * FB;     // DS reads: f16 operands / fp8-side operands; f16 MFMAs
            __builtin_amdgcn_sched_group_barrier(0x100, NF16, 0);
#pragma unroll
            for (int i = 0; i < (NF8 < NM ? NF8 : NM); ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            if constexpr (NM > NF8) __builtin_amdgcn_sched_group_barrier(0x008, NM - NF8, 0);
        }
        ph_stamp(0);
        between();
        ph_stamp(1);
        // x_hi8 of this lane's 16 channels (the two f16 chunks it holds): e4m3(hi / s_x)
#pragma unroll
        for (int f = 0; f < FB; ++f) {
            const float sc = __builtin_bit_cast(float, (unsigned)eb[f] << 23);       // s_x = 2^(E - 127) (only the exponent field of the operand is used: E = 0, an all-zero / out-of-range row, gives 0)
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    // v_cvt_scalef32_pk_fp8_f16 DIVIDES by its scale operand (measured: tools/mx_spike.py, profiles/r3a_mx_spike.txt)
                    s16x2 r = {0, 0};
                    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, bitcast<f16x2>(bh[k2][f][2 * d]), sc, false);
                    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, bitcast<f16x2>(bh[k2][f][2 * d + 1]), sc, true);
                    b8[f][2 * k2 + d] = bitcast<int>(r);
                }
        }
        if constexpr (!PIPE) {
#pragma unroll
            for (int f = 0; f < FA; ++f) {
                const u32x4 lo8 = *reinterpret_cast<const u32x4*>(sw_ + swz_dma(wc * (BC / WC) + f * 32 + l32, 4 + 2 * h));
                const u32x4 hi8 = *reinterpret_cast<const u32x4*>(sw_ + swz_dma(wc * (BC / WC) + f * 32 + l32, 5 + 2 * h));
                a8[f] = i32x8{(int)lo8[0], (int)lo8[1], (int)lo8[2], (int)lo8[3], (int)hi8[0], (int)hi8[1], (int)hi8[2], (int)hi8[3]};
            }
        }
#pragma unroll
        for (int fa = 0; fa < FA; ++fa)
#pragma unroll
            for (int fb = 0; fb < FB; ++fb)
                acc32[fa][fb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[fa], b8[fb], acc32[fa][fb], 0, 0, 0, mx_sa[fa], 0, eb[fb]);
        ph_stamp(2);
    };
    auto compute_split = [&](int stage, auto&& between) __attribute__((always_inline)) {
        if constexpr (MX) compute_mx(stage, between);
        else if constexpr (MF == 32) compute_x3_32(stage, between);
        else compute_x3(stage, between, [] {});
    };

    // wait until at most `keep` (1 or 2) of this wave's slabs are still in flight: vmcnt(keep * NDMA), an immediate
    auto wait_keep = [&](int keep) {
        if (keep >= 2) { if constexpr (NDMA == 6) VMCNT(12); else if constexpr (NDMA == 5) VMCNT(10); else VMCNT(8); }
        else { if constexpr (NDMA == 6) VMCNT(6); else if constexpr (NDMA == 5) VMCNT(5); else VMCNT(4); }
    };


    if constexpr (SWP) {
        constexpr bool SWP_GN = SGN;                        // the GroupNorm partial sums in the software-pipelined tile's epilogue: see the SGN note above the kernel
        constexpr int FA = FC / 2, FB = FP / 2;
#if defined(MNET_SWP_PRIO) && MNET_SWP_PRIO == 1
        if (wave >= NW / 2) __builtin_amdgcn_s_setprio(1);              // A/B build: static priority for the later-dispatched wave of every SIMD
#elif defined(MNET_SWP_PRIO) && MNET_SWP_PRIO == 2
        if (wave < NW / 2) __builtin_amdgcn_s_setprio(1);               // A/B build: the other way round (control)
#endif
        const int l32 = lane & 31, h = lane >> 5;
        // slab issue in pieces (the same addresses as issue_w / issue_x): begin → the stream has another slab (crossing into the next tile if needed)
        auto sw_begin = [&]() __attribute__((always_inline)) -> bool {
            if (i_kt == nk) {
                i_kt = 0; i_v += G;
                i_live = i_v < ntiles;
                if (i_live) setup(i_v);
            }
            return i_live;
        };
        // a slab's pieces are straight-line code (no branch inside the scaled-MFMA group: LLVM sinks the MFMAs past control flow): the wave-uniform
        // parts of the addresses — descriptors, k offset, tap bit, concat source — are formed once per slab by sw_slab(), and a slab that does not
        // exist (end of the stream) is issued with out-of-range offsets (zeros into a stage nobody reads)
        __amdgpu_buffer_rsrc_t sl_rW, sl_rX;
        unsigned sl_kb = 0, sl_tap = 0, sl_cb = 0, sl_uni = 0, sl_dead = 0;
        auto sw_slab = [&](bool more) __attribute__((always_inline)) {
            const bool second = cur_c >= p.c0;                           // wave-uniform: second concat source
            sl_rW = __builtin_amdgcn_make_buffer_rsrc((void*)uni64(bW), 0, __builtin_amdgcn_readfirstlane(nW), 0x00020000);
            sl_rX = __builtin_amdgcn_make_buffer_rsrc((void*)uni64(second ? bX1 : bX0), 0, __builtin_amdgcn_readfirstlane(second ? nX1 : nX0), 0x00020000);
            sl_kb = (unsigned)(cur_tap * p.cin + cur_c) * 2u;
            sl_tap = (unsigned)cur_tap;
            sl_cb = (unsigned)(second ? p.c1 : p.c0) * 2u;
            sl_uni = (unsigned)(cur_tpx * (int)sl_cb + (second ? cur_c - p.c0 : cur_c) * 2) + lcb;
            sl_dead = more ? 0u : OOB;
        };
        auto sw_piece = [&](int idx) __attribute__((always_inline)) {
            unsigned char* sw_ = smem + i_stage * STAGE;
            if (idx < WJ) {
                const unsigned vo = (woff[idx] + sl_kb) | sl_dead;       // (an OOB row keeps bit 31 through the addition: sl_kb < 2^31)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(sl_rW, (lds_void*)(sw_ + (wave + NWI * idx) * 1024), 16, vo, 0, 0, 0);
            } else {
                const int j = idx - WJ;
                unsigned char* sx_ = sw_ + BC * 128;
                // branch-free and select-free (a `cond ? address : OOB` here is compiled to an exec-masked region, i.e. a basic-block split inside the
                // MFMA group): an invalid tap ORs bit 31 into the offset, which puts it beyond every num_records
                const unsigned inval = (((xmask[j] >> sl_tap) & 1u) - 1u) & OOB;
                const unsigned vo = ((unsigned)__mul24((int)xpx[j], (int)sl_cb) + sl_uni) | inval | sl_dead;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(sl_rX, (lds_void*)(sx_ + (wave + NWI * j) * 1024), 16, vo, 0, 0, 0);
            }
        };
        auto sw_end = [&]() __attribute__((always_inline)) {
            advance_cursor();
            i_stage ^= 1;
            ++i_kt;
        };
        auto load_scales = [&](int v) __attribute__((always_inline)) {   // E8M0 bytes of the weight scales of tile v (see the lock-step loop)
            int co0, pix0;
            tile_coords(v, co0, pix0);
            const unsigned char* wexp = reinterpret_cast<const unsigned char*>(p.wgt) + (size_t)p.cout * p.K * 2;
#pragma unroll
            for (int f = 0; f < FA; ++f) {
                const int ch = co0 + dma_weight_channel_mx(wc * (BC / WC) + f * 32 + l32);
                mx_sa[f] = ch < p.cout ? (int)wexp[ch] : 0;
            }
            // the loads must have RETURNED here, in the compiler's book-keeping too: left pending across the loop's back edge they would put an
            // `s_waitcnt vmcnt(0)` in front of the scaled MFMAs of EVERY slab, i.e. right behind the DMA pieces
#pragma unroll
            for (int f = 0; f < FA; ++f) asm volatile("" : "+v"(mx_sa[f]));
        };
        auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int a = 0; a < FA; ++a)
#pragma unroll
                for (int b = 0; b < FB; ++b)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc32[a][b][q] = 0.f;
        };
        auto bar = [&]() __attribute__((always_inline)) {
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        // per-lane LDS offsets of the operand fragments (swz_dma(row, chunk) of fragment 0; fragment f is f * 32 rows = f * 4096 bytes further; the
        // other chunks of a row are XORs on the ADDRESS: the swizzle is an XOR of the chunk index, rows are 128-byte aligned and fragment rows keep (row >> 1) & 7)
        const unsigned pa0 = (unsigned)swz_dma(wc * (BC / WC) + l32, h), pa8 = (unsigned)swz_dma(wc * (BC / WC) + l32, 4 + 2 * h);
        const unsigned pb0 = (unsigned)swz_dma(wp * (BP / WP) + l32, h), pbe = (unsigned)swz_dma(wp * (BP / WP) + l32, 6);

        u32x4 a[2][FA], bh[2][FB];                                       // f16 operands of the slab whose f16 MFMAs run in this iteration
        i32x8 a8[FA], b8[FB];                                            // fp8-side operands, carried to the NEXT iteration's scaled MFMAs
        int eb[FB], ebn[FB];                                             // E8M0 of s_x: of the carried slab / of the slab being read
        // ---- the three parts of an iteration
        auto front = [&](unsigned so) __attribute__((always_inline)) {   // LDS reads of slab s's first f16 k-step and its scale bytes (the second
            const unsigned aa = pa0 + so, ba = pb0 + so + BC * 128u, bea = pbe + so + BC * 128u;   // k-step's operands would not fit beside the carried fp8 operands)
#pragma unroll
            for (int f = 0; f < FA; ++f) a[0][f] = *reinterpret_cast<const u32x4*>(smem + (aa + f * 4096u));                  // chunk h
#pragma unroll
            for (int f = 0; f < FB; ++f) bh[0][f] = *reinterpret_cast<const u32x4*>(smem + (ba + f * 4096u));
#pragma unroll
            for (int f = 0; f < FB; ++f) ebn[f] = *(smem + (bea + f * 4096u));                                                 // chunk 6, byte 0
        };
        auto scaled_prev = [&](auto with_pieces) __attribute__((always_inline)) {   // 8 scaled MFMAs of the carried slab, one DMA piece of the next slab behind each
            constexpr bool pieces = decltype(with_pieces)::value;
#pragma unroll
            for (int fa = 0; fa < FA; ++fa)
#pragma unroll
                for (int fb = 0; fb < FB; ++fb) {
                    acc32[fa][fb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[fa], b8[fb], acc32[fa][fb], 0, 0, 0, mx_sa[fa], 0, eb[fb]);
                    if (pieces) asm volatile("" : "+v"(acc32[fa][fb]));  // pins the MFMA in front of its piece at the IR level (a pure intrinsic is otherwise sunk to its use)
                    __builtin_amdgcn_sched_barrier(0);
                    if (pieces) {                                        // compile-time
                        const int i = fa * FB + fb;
                        if (i < NDMA) sw_piece(i);
                        if (i == FA * FB - 1) {
#pragma unroll
                            for (int idx = FA * FB; idx < NDMA; ++idx) sw_piece(idx);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        };
        auto f16_part = [&](unsigned so) __attribute__((always_inline)) {   // 2 FA FB f16 MFMAs of slab s; its fp8-side reads and conversions between them
            const unsigned aa = pa0 + so, a8a = pa8 + so, ba = pb0 + so + BC * 128u;
            // up front: the second k-step's activation operands and the activations' lo bytes (the second k-step's weight fragments follow the
            // first k-step's out of the registers: fragment fa is read behind the last MFMA that uses a[0][fa])
#pragma unroll
            for (int f = 0; f < FB; ++f) bh[1][f] = *reinterpret_cast<const u32x4*>(smem + ((ba ^ 32u) + f * 4096u));         // chunk 2 + h
#pragma unroll
            for (int f = 0; f < FB; ++f) {
                eb[f] = ebn[f];
                const u32x4 lo8 = *reinterpret_cast<const u32x4*>(smem + ((ba ^ 64u) + f * 4096u));                           // chunk 4 + h
                b8[f][4] = (int)lo8[0]; b8[f][5] = (int)lo8[1]; b8[f][6] = (int)lo8[2]; b8[f][7] = (int)lo8[3];
            }
            __builtin_amdgcn_sched_barrier(0);
            // explicit placement (program order pinned by fences: scheduling hints did not hold the reads in front): behind MFMA i
            //   i < 2 FA      : one fp8-side weight read (lo8 / hi8 chunk of fragment i / 2) into the registers the scaled MFMAs have just released
            //   i < 4 FB      : one x_hi8 conversion group (2 v_cvt_scalef32_pk_fp8_f16 of 16 channels' halves; first k-step first)
            constexpr int NM = 2 * FA * FB;
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                const int k2 = i / (FA * FB), fa = (i % (FA * FB)) / FB, fb = i % FB;
                acc32[fa][fb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bitcast<f16x8>(a[k2][fa]), bitcast<f16x8>(bh[k2][fb]), acc32[fa][fb], 0, 0, 0);
                if (i < 2 * FA || i < 4 * FB) {
                    asm volatile("" : "+v"(acc32[fa][fb]));
                    __builtin_amdgcn_sched_barrier(0);
                    if (i < 2 * FA) {
                        const int f = i >> 1;
                        const u32x4 q = *reinterpret_cast<const u32x4*>(smem + (((i & 1) ? (a8a ^ 16u) : a8a) + f * 4096u));       // chunk 4 + 2 h / 5 + 2 h
                        const int o = (i & 1) * 4;
                        a8[f][o] = (int)q[0]; a8[f][o + 1] = (int)q[1]; a8[f][o + 2] = (int)q[2]; a8[f][o + 3] = (int)q[3];
                    }
                    if (i < FA * FB && fb == FB - 1)
                        a[1][fa] = *reinterpret_cast<const u32x4*>(smem + ((aa ^ 32u) + fa * 4096u));                          // chunk 2 + h
                    if (i < 4 * FB) {
                        const int kk = i / (2 * FB), f = (i >> 1) % FB, d = i & 1;
                        const float sc = __builtin_bit_cast(float, (unsigned)ebn[f] << 23);
                        s16x2 r = {0, 0};
                        r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, bitcast<f16x2>(bh[kk][f][2 * d]), sc, false);
                        r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, bitcast<f16x2>(bh[kk][f][2 * d + 1]), sc, true);
                        b8[f][2 * kk + d] = bitcast<int>(r);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };

        setup(i_v);
        {                                                                // slab 0 (a workgroup always has a tile)
            const bool more = sw_begin();
            sw_slab(more);
#pragma unroll
            for (int idx = 0; idx < NDMA; ++idx) sw_piece(idx);
            if (more) sw_end();
        }
        int c_v = blockIdx.x, c_kt = 0;                                  // tile / slab index inside it of the slab whose f16 MFMAs run in the iteration
        load_scales(c_v);
        zero_acc();
        const int total = ((ntiles - (int)blockIdx.x + G - 1) / G) * nk; // slabs in this workgroup's stream
        // ---- iteration 0: nothing carried yet
        VMCNT(0);
        bar();
        front(0u);
        __builtin_amdgcn_sched_barrier(0);
        {
            const bool more = sw_begin();
            sw_slab(more);
#pragma unroll
            for (int idx = 0; idx < NDMA; ++idx) sw_piece(idx);
            if (more) sw_end();
        }
        __builtin_amdgcn_sched_barrier(0);
        f16_part(0u);
        c_kt = 1;
        ph_stamp(-1);
        bool ph_after_epilogue = false;                                  // (DBG == 6 only)
        for (int s = 1; s < total; ++s) {
            VMCNT(0);                                                    // this wave's pieces of slab s (and the previous epilogue's stores) have landed ...
            ph_stamp(DBG == 6 && ph_after_epilogue ? 5 : 3);
            ph_after_epilogue = false;                                                 // (DBG == 6 only — phases: 1 front reads + scaled MFMAs + pieces, 2 f16 part, 3 vmcnt(0), 4 barrier)
            bar();                                                       // ... everyone's; nobody reads the stage of slab s-1 any more
            ph_stamp(4);
            ++ph_slabs;
            const unsigned so = (unsigned)(s & 1) * (unsigned)STAGE;
            front(so);
            __builtin_amdgcn_sched_barrier(0);
            const bool more = sw_begin();
            sw_slab(more);
            __builtin_amdgcn_sched_barrier(0);
            scaled_prev(std::true_type{});
            ph_stamp(1);
            if (more) sw_end();
            if (c_kt == nk) {                                            // slab s-1 closed its tile: epilogue, then the next tile's scales and a clean accumulator
                int co0, pix0;
                tile_coords(c_v, co0, pix0);
                dma_epilogue_mx<BC, BP, WC, WP, FC, FP, (XB == 1024 ? 16 : 64), SWP_GN>(p, acc32, co0, pix0, wc, wp, lane, xpose);
                c_kt = 0; c_v += G;
                if (p.tilesC > 1) load_scales(c_v);                      // (one channel tile: every tile has the same scales)
                zero_acc();
                ph_stamp(0);
                ph_after_epilogue = true;
            }
            __builtin_amdgcn_sched_barrier(0);
            f16_part(so);
            ph_stamp(2);
            ++c_kt;
        }
        __builtin_amdgcn_sched_barrier(0);
        scaled_prev(std::false_type{});                                  // the last slab's scaled MFMAs
        {
            int co0, pix0;
            tile_coords(c_v, co0, pix0);
            dma_epilogue_mx<BC, BP, WC, WP, FC, FP, (XB == 1024 ? 16 : 64), SWP_GN>(p, acc32, co0, pix0, wc, wp, lane, xpose);
        }
        if constexpr (DBG == 6) {       // DIAGNOSTIC: phase sums of this wave over the first bytes of the output (32 bytes per wave)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (lane == 0) {
                unsigned* o = reinterpret_cast<unsigned*>(p.y) + ((size_t)blockIdx.x * NW + wave) * 8;
                o[0] = 0x5157a3b6u; o[1] = ph_slabs; o[2] = ph_sum[0]; o[3] = ph_sum[1]; o[4] = ph_sum[2]; o[5] = ph_sum[3]; o[6] = ph_sum[4];
                o[7] = ph_sum[5];
            }
        }
        return;
    }
    // ---- prime the ring (STAGES-1 slabs ahead), then walk this workgroup's tiles
    setup(i_v);
    int inflight = 0;
#pragma unroll
    for (int d = 0; d < STAGES - 1; ++d) inflight += issue_next();
    int c_stage = 0;
    bool drain = false;                                  // the previous tile's epilogue stores share the vmcnt counter

    for (int c_v = blockIdx.x; c_v < ntiles; c_v += G) {
        long long stamp[4] = {0, 0, 0, 0};               // DBG >= 3 only: wall clock (100 MHz) at tile start / = / k-loop end / stores acked
        long long cyc = 0;
        if constexpr (DBG >= 3) { stamp[0] = stamp[1] = wall_clock64(); cyc = (long long)__builtin_readcyclecounter(); }
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

        if constexpr (MX) {             // E8M0 bytes of the weight scales (after the cout * K packed rows), one per output channel
            int co0, pix0;
            tile_coords(c_v, co0, pix0);
            const unsigned char* wexp = reinterpret_cast<const unsigned char*>(p.wgt) + (size_t)p.cout * p.K * 2;
#pragma unroll
            for (int f = 0; f < FC / 2; ++f) {
                const int ch = co0 + dma_weight_channel_mx(wc * (BC / WC) + f * 32 + (lane & 31));
                mx_sa[f] = ch < p.cout ? (int)wexp[ch] : 0;
            }
            drain = true;               // plain loads share the vmcnt counter with the DMA: the next wait is vmcnt(0)
        }

        // hot iterations: the slab issued stays inside this tile (same loop body as a non-persistent kernel)
        const int hot = nk - (STAGES - 1) > 0 ? nk - (STAGES - 1) : 0;
        ph_stamp(-1);
        for (int kt = 0; kt < hot; ++kt) {
            // the oldest slab in flight must have landed (this wave's share; the barrier extends it to everyone's)
            if (STAGES > 2 && !drain) wait_keep(STAGES - 2);
            else VMCNT(0);
            drain = false;
            if (kt > 0) ph_stamp(3);
            __builtin_amdgcn_s_barrier();                // ... and every wave is done reading the stage refilled next
            asm volatile("" ::: "memory");
            if (kt > 0) { ph_stamp(4); ++ph_slabs; } else ph_stamp(-1);
            if constexpr (SPREAD) {
                // the DMA pieces of the next slab ride between the multiplies of this one (sched barriers pin the placement)
                auto adv = [&]() __attribute__((always_inline)) { i_stage = i_stage == STAGES - 1 ? 0 : i_stage + 1; ++i_kt; };
                if constexpr (X3) {
                    compute_split(c_stage,
                                  [&]() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); issue_w(i_stage); issue_x(i_stage); adv(); __builtin_amdgcn_sched_barrier(0); });
                } else if constexpr (PIPE) {
                    compute_f16_pipe(c_stage, [&]() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); issue_w(i_stage); issue_x(i_stage); adv(); __builtin_amdgcn_sched_barrier(0); });
                } else {
                    compute_half(c_stage, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    issue_w(i_stage); issue_x(i_stage); adv();
                    __builtin_amdgcn_sched_barrier(0);
                    compute_half(c_stage, 1);
                }
            } else {
                issue_hot();
                if constexpr (X3) compute_split(c_stage, [] {});
                else { compute_half(c_stage, 0); compute_half(c_stage, 1); }
            }
            c_stage = c_stage == STAGES - 1 ? 0 : c_stage + 1;
        }
        // tail iterations: the slab issued belongs to this workgroup's NEXT tile (set-up + first-slab latency overlap the
        // last multiplies and the epilogue of this one)
        for (int kt = hot; kt < nk; ++kt) {
            if (STAGES > 2 && inflight >= 2 && !drain) wait_keep(inflight - 1);
            else VMCNT(0);
            drain = false;
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            inflight += issue_next() - 1;
            if constexpr (X3) compute_split(c_stage, [] {});
            else { compute_half(c_stage, 0); compute_half(c_stage, 1); }
            c_stage = c_stage == STAGES - 1 ? 0 : c_stage + 1;
        }
        if constexpr (DBG >= 3) { stamp[2] = wall_clock64(); cyc = (long long)__builtin_readcyclecounter() - cyc; }

        int co0, pix0;
        tile_coords(c_v, co0, pix0);
        if constexpr (MX) dma_epilogue_mx<BC, BP, WC, WP, FC, FP, (XB == 1024 ? 16 : 64)>(p, acc32, co0, pix0, wc, wp, lane, xpose);
        else dma_epilogue<BC, BP, WC, WP, MF, DBG, FC, FP, X3>(p, acc, acc32, co0, pix0, wc, wp, lane);
        drain = true;

        if constexpr (DBG >= 3) {       // DIAGNOSTIC: overwrite part of the output with this tile's time stamps
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stamp[3] = wall_clock64();
            if (tid == 0) {
                long long* o = DBG == 3 ? reinterpret_cast<long long*>(reinterpret_cast<f16*>(p.y) + (size_t)pix0 * p.cout + co0)
                                        : reinterpret_cast<long long*>(reinterpret_cast<f16*>(p.y) + (size_t)(256 + 32 * (size_t)(pix0 / BP)) * p.cout);
                o[0] = 0x7157a3b5ll; o[1] = stamp[0]; o[2] = stamp[1]; o[3] = stamp[2]; o[4] = stamp[3];
                o[5] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_ID
                o[6] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // XCC_ID
                o[7] = cyc;                                          // shader cycles spent in the k-loop (s_memtime)
            }
        }
    }
    if constexpr (DBG == 6) {           // DIAGNOSTIC: phase sums of this wave over the first bytes of the output (32 bytes per wave)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (lane == 0) {
            unsigned* o = reinterpret_cast<unsigned*>(p.y) + ((size_t)blockIdx.x * NW + wave) * 8;
            o[0] = 0x5157a3b6u; o[1] = ph_slabs; o[2] = ph_sum[0]; o[3] = ph_sum[1]; o[4] = ph_sum[2]; o[5] = ph_sum[3]; o[6] = ph_sum[4];
            o[7] = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_ID (SIMD / CU of this wave)
        }
    }
}

template <int BC, int BP, int WC, int WP, int STAGES, int MF = 16, int DBG = 0, bool X3 = false, bool SPREAD = false, bool PIPE = false, bool MX = false, bool SWP = false, bool SGN = false>
static int launch_dma_cfg(const ConvArgs& a, hipStream_t st) {
    constexpr int LDS = STAGES * (BC + BP) * 128 + WC * WP * dma_mx_xpose_bytes<WC * WP, MX>(STAGES * (BC + BP) * 128);
    auto kern = conv_dma_kernel<BC, BP, WC, WP, STAGES, MF, DBG, X3, SPREAD, PIPE, MX, SWP, SGN>;
    static thread_local DeviceOnce attr_once;      // per instantiation, per thread, per device
    if (!attr_once.done()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return mnet_fail(MNET_E_LAUNCH, "hipFuncSetAttribute(dma): %s", hipGetErrorString(e));
        attr_once.mark();
    }
    ConvArgs b = a;
    auto log2_or_minus1 = [](int v) { return v > 0 && (v & (v - 1)) == 0 ? __builtin_ctz((unsigned)v) : -1; };
    b.howo_shift = log2_or_minus1(a.howo); b.wo_shift = log2_or_minus1(a.wo);
    b.tilesC = (a.cout + BC - 1) / BC;
    const int tilesP = (a.npix + BP - 1) / BP;
    b.ntiles = b.tilesC * tilesP;
    int grid = b.ntiles;
    const int lim = dma_grid_limit();
    static const bool env_one_tile = [] { const char* e = getenv("MNET_DMA_ONE_TILE"); return e && atoi(e) != 0; }();   // A/B knob
    if (grid > lim && !a.one_tile_per_wg && !env_one_tile) grid = lim & ~7;      // multiple of 8: virtual block v keeps blockIdx's XCD (v % 8) on every pass
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WC * WP * 64), LDS, st, b);
    MNET_LAUNCH_CHECK("conv_dma_kernel");
    return MNET_OK;
}

#ifdef MNET_DMA_SWP_GN_TU
// conv_dma_swp_gn.hip: this translation unit holds ONE instantiation — the software-pipelined 256x256 fp16+8 tile with the GroupNorm-sum block
int launch_dma_swp_gn(const ConvArgs& a, hipStream_t st) { return launch_dma_cfg<256, 256, 2, 4, 2, 32, 0, true, false, false, true, true, true>(a, st); }
#else
int launch_dma_swp_gn(const ConvArgs& a, hipStream_t st);      // conv_dma_swp_gn.hip

// tile configurations (BC x BP, waves, LDS stages, MFMA shape); MNET_CONV_ALGO_DMA_CFG0 + id selects one explicitly.
// Production ids 0-6 all use v_mfma_f32_16x16x32_f16 and walk k in the same order (64-channel slice outer, tap inner), so a
// conv gives the same bits whatever tile configuration its launch size selects (batch-size-invariant results; eligibility
// for this kernel never depends on the batch size).  The register-staged kernel walks k tap-outer: same products, fp32
// partial sums associated differently.
// ids 7-9: v_mfma_f32_32x32x16_f16 forms (fp32 sums associate differently).
// ids 11-15: DIAGNOSTIC builds that produce wrong results on purpose (tools/wg_timeline.py, tools/conv_bench.py).
static int launch_dma_id(int id, const ConvArgs& a, hipStream_t st) {
    if (a.split == 2) { // fp16+8 (MNET_F16M) instantiations: the same tile shapes on 32x32 MFMAs
        switch (id) {
            case 0: return launch_dma_cfg<256, 256, 4, 4, 2, 32, 0, true, true, false, true>(a, st);
            case 1: return launch_dma_cfg<256, 128, 4, 2, 3, 32, 0, true, false, false, true>(a, st);
            case 2: return launch_dma_cfg<128, 256, 2, 4, 3, 32, 0, true, false, false, true>(a, st);
            case 3: return launch_dma_cfg<64, 256, 1, 8, 3, 32, 0, true, false, false, true>(a, st);
            case 4: return launch_dma_cfg<128, 512, 2, 8, 2, 32, 0, true, true, false, true>(a, st);
            case 5: return launch_dma_cfg<64, 512, 1, 8, 2, 32, 0, true, true, false, true>(a, st);
            case 6: return launch_dma_cfg<256, 256, 2, 4, 2, 32, 0, true, true, false, true>(a, st);          // 8 waves, 128x64 per wave: AUTO for cout >= 256
            case 7: return launch_dma_cfg<128, 512, 2, 4, 2, 32, 0, true, true, false, true>(a, st);          // 8 waves, 64x128 per wave
            case 8: return launch_dma_cfg<128, 512, 1, 8, 2, 32, 0, true, true, false, true>(a, st);          // 8 waves, 128x64 per wave
            case 10: return launch_dma_cfg<128, 128, 2, 4, 4, 32, 0, true, false, false, true>(a, st);
            case 11: return launch_dma_cfg<256, 256, 2, 4, 2, 32, 0, true, true, true, true>(a, st);          // id 6 + LDS reads placed by scheduling hints
            case 12: return launch_dma_cfg<128, 512, 1, 8, 2, 32, 0, true, true, true, true>(a, st);          // id 8, same
            case 13: return launch_dma_cfg<64, 512, 1, 8, 2, 32, 0, true, true, true, true>(a, st);           // id 5, same
            case 15: return launch_dma_cfg<256, 256, 2, 4, 2, 32, 0, true, false, false, true, true>(a, st);       // id 6 with the slab loop software-pipelined across the barrier (SWP)
            case 9: return launch_dma_cfg<128, 512, 1, 8, 2, 32, 0, true, false, false, true, true>(a, st);        // id 8, same
            case 16:                                       // round 6: the 256x256 tile with ONE wave per SIMD (4 waves x 128x128, accumulators in a[0:255]): conv_dma_w4.hip
                if (a.ktiles > 512) return launch_conv_dma(a, st, 15);   // (its slab table holds 512 k-slabs per tile: cin * taps <= 16384 halves; same bytes from id 15)
                return launch_conv_dma_w4(a, st);
            case 14: {                  // DIAGNOSTIC (wrong results): id 11 with per-phase cycle sums written over the output (tools/slab_phases.py)
                static const bool allow = [] { const char* e = getenv("MNET_ALLOW_DIAGNOSTIC_KERNELS"); return e && atoi(e) != 0; }();
                if (!allow) return mnet_fail(MNET_E_ARG, "conv: fp16+8 LDS-DMA id 14 is a diagnostic build with wrong results (set MNET_ALLOW_DIAGNOSTIC_KERNELS=1 to use it)");
                static const bool swp = [] { const char* e = getenv("MNET_DIAG_SWP"); return e && atoi(e) != 0; }();      // the same stamps in the software-pipelined tile (id 15)
                if (swp) return launch_dma_cfg<256, 256, 2, 4, 2, 32, 6, true, false, false, true, true>(a, st);
                return launch_dma_cfg<256, 256, 2, 4, 2, 32, 6, true, true, true, true>(a, st);
            }
            default: return mnet_fail(MNET_E_ARG, "conv: LDS-DMA tile configuration %d has no fp16+8 form", id);
        }
    }
    if (a.split) {      // split-half (fp16x3) instantiations of the production tile configurations
        switch (id) {
            case 0: return launch_dma_cfg<256, 256, 4, 4, 2, 16, 0, true>(a, st);
            case 1: return launch_dma_cfg<256, 128, 4, 2, 3, 16, 0, true>(a, st);
            case 2: return launch_dma_cfg<128, 256, 2, 4, 3, 16, 0, true>(a, st);
            case 3: return launch_dma_cfg<64, 256, 1, 8, 3, 16, 0, true>(a, st);
            case 4: return launch_dma_cfg<128, 512, 2, 8, 2, 16, 0, true>(a, st);
            case 5: return launch_dma_cfg<64, 512, 1, 8, 2, 16, 0, true>(a, st);
            case 6: return launch_dma_cfg<256, 256, 2, 4, 2, 16, 0, true>(a, st);          // 8 waves, 128x64 per wave: AUTO for cout >= 256
            case 7: return launch_dma_cfg<128, 512, 2, 4, 2, 16, 0, true>(a, st);          // 8 waves, 64x128 per wave: AUTO for cout 128
            case 8: return launch_dma_cfg<256, 256, 2, 4, 2, 16, 0, true, true>(a, st);    // id 6 with the DMA pieces after the first multiply group
            case 9: return launch_dma_cfg<128, 512, 2, 4, 2, 16, 0, true, true>(a, st);    // id 7, same
            case 11: return launch_dma_cfg<256, 256, 2, 4, 2, 16, 0, true, true, true>(a, st);   // id 8 + LDS reads placed by scheduling hints
            case 12: return launch_dma_cfg<128, 512, 2, 4, 2, 16, 0, true, true, true>(a, st);   // id 9, same
            case 10: return launch_dma_cfg<128, 128, 2, 4, 4, 16, 0, true>(a, st);
            case 20: return launch_dma_cfg<256, 256, 2, 4, 2, 32, 0, true, true>(a, st);          // id 8 on v_mfma_f32_32x32x16_f16
            case 21: return launch_dma_cfg<128, 512, 2, 4, 2, 32, 0, true, true>(a, st);          // id 9, same
            default: return mnet_fail(MNET_E_ARG, "conv: LDS-DMA tile configuration %d has no split-half form", id);
        }
    }
    switch (id) {
        case 0: return launch_dma_cfg<256, 256, 4, 4, 2>(a, st);
        case 1: return launch_dma_cfg<256, 128, 4, 2, 3>(a, st);
        case 2: return launch_dma_cfg<128, 256, 2, 4, 3>(a, st);
        case 3: return launch_dma_cfg<64, 256, 1, 8, 3>(a, st);
        case 4: return launch_dma_cfg<128, 512, 2, 8, 2>(a, st);
        case 5: return launch_dma_cfg<64, 512, 1, 8, 2>(a, st);
        case 6: return launch_dma_cfg<256, 256, 2, 4, 2>(a, st);          // 8 waves, 128x64 per wave
        case 7: return launch_dma_cfg<256, 256, 4, 4, 2, 32>(a, st);
        case 8: return launch_dma_cfg<256, 256, 4, 4, 2, 16, 0, false, true>(a, st);       // id 0 with the DMA pieces issued between the two half slabs
        case 9: return launch_dma_cfg<128, 512, 2, 8, 2, 16, 0, false, true>(a, st);       // id 4, same
        case 16: return launch_dma_cfg<256, 256, 2, 4, 2, 16, 0, false, true, true>(a, st);   // 8 waves (128x64 per wave), both half slabs' fragments requested up
                                                                                            // front (the second half's LDS reads run under the first half's MFMAs),
                                                                                            // DMA pieces between the halves: AUTO for cout >= 256
        case 17: return launch_dma_cfg<128, 512, 2, 4, 2, 16, 0, false, true, true>(a, st);   // the same form of the 128x512 tile (64x128 per wave)
        case 10: return launch_dma_cfg<128, 128, 2, 4, 4>(a, st);          // small launches: twice the workgroups of ids 1 / 2, 3 slabs in flight
        case 11: case 12: case 13: case 14: case 15: {
            // MNET_F16 ids 11-15 are DIAGNOSTIC builds that produce WRONG results on purpose (tools/wg_timeline.py, tools/conv_bench.py):
            // refused unless the process opts in, so that a C-ABI host cannot select one by accident (the same ids are production
            // tiles for MNET_F16X2 / MNET_F16M launches, which never reach this table)
            static const bool allow = [] { const char* e = getenv("MNET_ALLOW_DIAGNOSTIC_KERNELS"); return e && atoi(e) != 0; }();
            if (!allow) return mnet_fail(MNET_E_ARG, "conv: MNET_F16 LDS-DMA id %d is a diagnostic build with wrong results (set MNET_ALLOW_DIAGNOSTIC_KERNELS=1 to use it)", id);
            break;
        }
        default: break;
    }
    switch (id) {
        case 11: return launch_dma_cfg<256, 256, 4, 4, 2, 16, 4>(a, st);  // DIAGNOSTIC: all tiles store over tile 0; stamps after tile 0
        case 12: return launch_dma_cfg<256, 256, 4, 4, 2, 16, 5>(a, st);  // DIAGNOSTIC: no output stores; stamps after tile 0
        case 13: return launch_dma_cfg<256, 256, 4, 4, 2, 16, 3>(a, st);  // DIAGNOSTIC: per-tile time stamps written over the output
        case 14: return launch_dma_cfg<256, 256, 4, 4, 2, 16, 1>(a, st);  // DIAGNOSTIC: activations read from a 256 KiB window
        case 15: return launch_dma_cfg<256, 256, 4, 4, 2, 16, 2>(a, st);  // DIAGNOSTIC: no DMA after a tile's first k-slab
        default: return mnet_fail(MNET_E_ARG, "conv: unknown LDS-DMA tile configuration %d", id);
    }
}

int conv_dma_pick(const ConvArgs& a) {
    const bool big = a.npix >= 256 * 256;
    static const int env_big256 = [] { const char* e = getenv("MNET_DMA_CFG_BIG256"); return e ? atoi(e) : 16; }();   // A/B knob
    // a launch that would leave a quarter or more of the CUs without a tile (a strip at a time: 4096-16384 pixels) takes the
    // 128x128 tile instead: twice the workgroups (same k order, same bits)
    const long long t128 = (a.npix + 127) / 128, t256 = (a.npix + 255) / 256;
    // split-half (fp16x3): the 8-wave forms of the two big tiles.  Three products per slab need a third set of operand
    // fragments live: the 16-wave tiles (128 VGPRs per wave) spill (59 / 133 VGPRs) and park 62 % of their wave cycles in
    // s_waitcnt / barriers; with 2 waves per SIMD and 256 VGPRs the same tiles run 19 % faster (measured: 446 vs 372 TFLOP/s
    // algorithmic on the 256x256 tile, B = 64) — the opposite of the f16 kernel, where the 16-wave form wins by 6 %.
    static const int env_x3_16w = [] { const char* e = getenv("MNET_X3_16WAVE"); return e ? atoi(e) : 0; }();                 // A/B knob
    // ids 8 / 9 = ids 6 / 7 with the next slab's DMA pieces issued after the first of the three multiply groups instead of right
    // after the barrier: +5.6 % / +3.7 % (438 vs 414 TFLOP/s on the 256x256 tile, same box, B = 64).  (Two insertion points —
    // weights after the first group, activations after the second — keep the DMA state live across all three groups: 167-275
    // VGPRs spill with scratch reloads inside the k-loop, 227 TFLOP/s.)
    static const int env_x3_128 = [] { const char* e = getenv("MNET_X3_CFG128"); return e ? atoi(e) : 9; }();                 // A/B knobs
    // id 11 = id 8 with the LDS reads placed by scheduling hints (all hi fragments up front, the lo activation fragments under the
    // first group's MFMAs): 472 vs 465 TFLOP/s (+1.5 %); the 128x512 tile does not gain (id 12 stays an A/B knob)
    static const int env_x3_256 = [] { const char* e = getenv("MNET_X3_CFG256"); return e ? atoi(e) : 11; }();
    if (a.split == 2) {
        // A/B knobs.  id 15 (round 4) = id 6 with the slab loop software-pipelined across the barrier: +0.5 ... +3.8 % over id 11 (= id 6 with the LDS reads
        // placed by scheduling hints) on the four shapes that carry the step, +1.2 % end to end, same bytes (profiles/r4g_*)
        // id 16 (round 6) = the same tile with ONE wave per SIMD (conv_dma_w4.hip: 4 waves x 128x128 outputs, accumulators in a[0:255]): +1.3 ... +2.5 % over id 15 on the
        // shapes that carry the step at a 2-5 % higher shader clock for the same package power, same bytes (profiles/r6i_*); it writes GroupNorm sums itself
        static const int env_mx_256 = [] { const char* e = getenv("MNET_MX_CFG256"); return e ? atoi(e) : 16; }();
        static const int env_mx_128 = [] { const char* e = getenv("MNET_MX_CFG128"); return e ? atoi(e) : 8; }();
        // (launches that write GroupNorm partial sums: id 15 runs its SGN build, conv_dma_swp_gn.hip; the software-pipelined 128x512 tile has no such build → its lock-step form 8)
        static const int env_gn_lockstep = [] { const char* e = getenv("MNET_GN_LOCKSTEP"); return e ? atoi(e) : 0; }();     // A/B knob: 1 = round-5's first form (id 15 → 11)
        const auto no_swp = [&](int id) { return a.gn_partial ? (id == 15 && env_gn_lockstep ? 11 : (id == 9 ? 8 : id)) : id; };
        if (a.cout >= 256) return big ? no_swp(env_mx_256) : (t128 * ((a.cout + 255) / 256) < 200 ? 10 : 1);
        if (a.cout >= 128) return big ? no_swp(env_mx_128) : (t256 * ((a.cout + 127) / 128) < 200 ? 10 : 2);
        static const int env_mx_64 = [] { const char* e = getenv("MNET_MX_CFG64"); return e ? atoi(e) : 13; }();       // id 13 = id 5 + hints: 260 vs 251
        return big ? env_mx_64 : 3;
    }
    if (a.split && big && a.cout >= 128) return env_x3_16w ? (a.cout >= 256 ? 0 : 4) : (a.cout >= 256 ? env_x3_256 : env_x3_128);
    // f16 big tiles: ids 8 / 9 = ids 0 / 4 with the next slab's DMA pieces issued between the two half slabs instead of right after
    // the barrier (+2.8 % on the 256x256 tile: 1140 vs 1109 TFLOP/s, B = 64; same MFMA sequence, same bits).  id 16 = the 8-wave
    // 256x256 tile with both half slabs' fragments requested up front as well (the second half's LDS reads run under the first
    // half's MFMAs — it has the registers for it): 1176-1188 vs 1148-1166 TFLOP/s for id 8, 1064 for the plain 8-wave id 6 → AUTO
    // for cout >= 256; the same form of the 128x512 tile (id 17) is slower than id 9 (22.3 vs 18.3 ms per step)
    static const int env_big128 = [] { const char* e = getenv("MNET_DMA_CFG_BIG128"); return e ? atoi(e) : 9; }();   // A/B knob
    if (a.cout >= 256) return big ? env_big256 : (t128 * ((a.cout + 255) / 256) < 200 ? 10 : 1);
    if (a.cout >= 128) return big ? env_big128 : (t256 * ((a.cout + 127) / 128) < 200 ? 10 : 2);
    return big ? 5 : 3;
}

int launch_conv_dma(const ConvArgs& a, hipStream_t st, int cfg) {
    int id = cfg >= 0 ? cfg : conv_dma_pick(a);
    // launches that write GroupNorm partial sums: the software-pipelined tiles of THIS translation unit are built without that block (dma_epilogue_mx<..., GN = false>);
    // id 15 has a build with it in conv_dma_swp_gn.hip, id 9 hands over to its lock-step form 8 (the same tile shape, the same MFMA sequence, the same bytes)
    if (a.gn_partial && a.split == 2) {       // (id 16, the one-wave-per-SIMD tile, writes the sums itself)
        if (id == 15) return launch_dma_swp_gn(a, st);
        if (id == 9) id = 8;
    }
    return launch_dma_id(id, a, st);
}

// eligibility of the LDS-DMA path (see header comment); the caller falls back to the register-staged kernel
bool conv_dma_eligible(const ConvArgs& a, int dtype) {
    if ((dtype != MNET_F16 && dtype != MNET_F16X2 && dtype != MNET_F16M) || a.in_scale || a.act > MNET_ACT_LRELU_SQRT2) return false;
    if (dtype != MNET_F16 && a.cout % 32 != 0) return false;          // (a.c0 / a.cin / a.K are physical here: f16 view, doubled)
    if (a.cin % 64 != 0 || a.c0 % 64 != 0 || a.cout < 64 || a.cout % 8 != 0 || a.kh * a.kw > 32 || a.kh > 8 || a.kw > 8) return false;
    // 31-bit buffer offsets: a pixel tile may touch ceil(256/howo)+1 images
    const long long imgs = 512 / a.howo + 2;   // largest pixel tile is 512
    const long long per_img = (long long)a.h * a.w * (a.c0 > a.c1 ? a.c0 : a.c1) * 2;
    if (per_img * imgs >= 0x7fffffffLL) return false;
    if ((long long)a.h * a.w * imgs >= (1ll << 23) || (a.c0 > a.c1 ? a.c0 : a.c1) * 2 >= (1 << 23)) return false;   // 24-bit signed multiply of (pixel, bytes per pixel)
    if ((long long)256 * a.K * 2 >= 0x40000000LL) return false;
    if ((long long)a.cout * a.K * 2 >= 0x7fffffffLL) return false;
    return true;
}
#endif  // MNET_DMA_SWP_GN_TU
