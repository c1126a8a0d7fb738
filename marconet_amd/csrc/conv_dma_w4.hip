// fp16+8 256x256 tile, ONE WAVE PER SIMD (round 6): 4 waves per workgroup, 128 output channels x 128 pixels per wave, the 16 accumulator blocks of
// v_mfma_*_32x32 (256 registers) in the ACCUMULATOR half of the register file, the operand fragments / DMA state / epilogue in the 256 architectural VGPRs.
//
// Why: the 8-wave software-pipelined tile (conv_igemm_dma.hip, id 15) needs ~3260 cycles per 32-channel slab of which 2048 are
// matrix-pipe work, and it runs at the package power limit with a third of the budget moving operands.  With 128x128 per wave a slab's products need a third fewer
// LDS bytes (112 instead of 176 KiB per slab and CU: every weight fragment meets four pixel fragments instead of two), half the waves meet at the slab barrier, and no
// two waves compete for one SIMD's matrix pipe.
//
// How: hipcc keeps MFMA accumulators of the builtin form in VGPRs and — asked for 128 outputs x 128 outputs per wave — spills ~450 registers (HISTORY §3.1e (b)).  Here every
// MFMA is an `asm volatile` statement whose accumulator operand carries the "a" constraint: the 16 blocks are allocated to a[0:255] by construction, the compiler
// still sees their liveness (epilogue reads are v_accvgpr_read_b32 it generates itself), and the statement order — MFMA, LDS read, DMA piece, conversion — is pinned in
// program order by sched_barrier fences exactly as written below.  What the compiler does NOT do for these statements (cdna_hip_programming.md §5.7): hazard padding —
// the wait states an MFMA result needs before a VALU reads it, and a VALU-written operand needs before the MFMA, are s_nop's inside the strings.
//
// Same LDS image, same DMA pieces, same k order and the same MFMA sequence per output as every fp16+8 tile (f16 k-step 0, f16 k-step 1, scaled fp8 — per slab):
// byte-identical outputs (tests/test_mx_gpu.py::test_one_wave_per_simd_tile_*).  Arithmetic served: models/networks.py:336-405,501-505 (every 3x3 conv with
// cout >= 256 on >= 65536 pixels in the fp16x2 mode).
#include <type_traits>
#include "conv_dma_common.h"

// A/B build switches (tools/build_variant.sh w4x conv_dma_w4 -DW4_...=0), defaults = the production form:
//   W4_PREP_IN_F16      the wave-uniform address parts of the NEXT slab's DMA (and a tile crossing's set-up) are formed behind the first MFMA of the f16 part's second
//                       k-step — 16 back-to-back MFMAs with nothing else to issue — instead of between the slab barrier and the first scaled MFMA (pipe idle)
//   W4_FRONT_INTERLEAVED the LDS reads of the slab's first f16 k-step go one behind each of the first 12 scaled MFMAs instead of in front of the first
#ifndef W4_PREP_IN_F16
#define W4_PREP_IN_F16 1
#endif
#ifndef W4_FRONT_INTERLEAVED
#define W4_FRONT_INTERLEAVED 1
#endif
//   W4_STAMPS (0)       DIAGNOSTIC build (wrong results: tools/slab_phases.py --w4): s_memtime differences per phase of the slab loop, summed per wave over the launch and
//                       written over the first bytes of the output — 0 scaled MFMAs + DMA pieces + front reads, 1 tile-closing epilogue (+ cursor), 2 f16 part up to the
//                       second k-step's first MFMA, 3 the next slab's address set-up (prep; a tile crossing's set-up included), 4 rest of the f16 part, 5 s_waitcnt vmcnt(0),
//                       6 s_barrier; each stamp drains the wave's LDS reads (s_memtime is a scalar memory instruction)
#ifndef W4_STAMPS
#define W4_STAMPS 0
#endif

namespace w4 {
constexpr int BC = 256, BP = 256, WC = 2, WP = 2, NW = 4;
constexpr int FA = 4, FB = 4;                         // 32x32 accumulator blocks per wave: weight fragments x pixel fragments
constexpr int WJ = BC / (8 * NW), XJ = BP / (8 * NW), NDMA = WJ + XJ;     // 8 + 8 DMA pieces (1 KiB each) per wave and slab
constexpr int STAGE = (BC + BP) * 128, STAGES = 2;
constexpr int XB = 4096;                              // epilogue transposition scratch per wave behind the stages (dma_epilogue_mx)
constexpr int LDS = STAGES * STAGE + NW * XB;         // 144 KiB
constexpr unsigned OOB = 0x80000000u;
static_assert(FA * FB == NDMA, "one DMA piece behind each scaled MFMA");
}

// the MFMAs of the fp16+8 slab on an accumulator block that lives in a[...] ("+a": allocated to the accumulator file, tied input / output).  Every statement opens with
// `s_nop 1`: hipcc does not know that the statement is an MFMA, so it pads nothing between a VALU write of one of its operands — a v_accvgpr_mov it inserts itself to
// move a block, the v_mov that assembles an operand tuple — and the MFMA's read (measured: register 0 of every block wrong after hipcc's lazy copy of a zeroed block);
// two wait states cover a VALU-written VGPR / AGPR read as SrcA / SrcB / SrcC (cdna_hip_programming.md §5.7 item 2).
// W4_MFMA_ZERO: a block cleared by the matrix pipe itself (0 * 0 + 0: one 8-pass MFMA instead of 16 v_accvgpr_write, and no compiler-made zero tuple that hipcc would
// copy into the blocks lazily, right in front of their first MFMA).
#define W4_MFMA_F16(ACC, A, B) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(ACC) : "v"(A), "v"(B))
#define W4_MFMA_ZERO(ACC, Z) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %1, 0" : "=a"(ACC) : "v"(Z))
#define W4_MFMA_SC(ACC, A8, B8, SA, SB) asm volatile("s_nop 1\n\tv_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+a"(ACC) : "v"(A8), "v"(B8), "v"(SA), "v"(SB))

__global__ void __launch_bounds__(256, 1) conv_dma_w4_kernel(const ConvArgs p) {
    using namespace w4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* const xpose = smem + STAGES * STAGE + wave * XB;
    const int wc = wave / WP, wp = wave % WP;
    const int rg = lane >> 3, pc = lane & 7;             // DMA geometry: lane fills LDS row (wave + NW j)*8 + rg, 16-byte slot pc
    const int l32 = lane & 31, h = lane >> 5;
    const int G = gridDim.x, ntiles = p.ntiles, nk = p.ktiles;

    // XCD-aware bijective tile map (as conv_dma_kernel): virtual block v (v % 8 = the XCD it runs on) → tile
    const int q8 = ntiles >> 3, r8 = ntiles & 7;
    auto tile_coords = [&](int v, int& co0, int& pix0) __attribute__((always_inline)) {
        const int xcd = v & 7;
        const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (v >> 3);
        co0 = (t % p.tilesC) * BC;
        pix0 = (t / p.tilesC) * BP;
    };

    // ================================================================== DMA issue side: one slab stream over all tiles (the addresses of conv_dma_kernel<..., MX>)
    unsigned long long bW = 0, bX0 = 0, bX1 = 0;
    int nW = 0, nX0 = 0, nX1 = 0;
    auto uni64 = [](unsigned long long v) __attribute__((always_inline)) -> unsigned long long {
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return ((unsigned long long)hi << 32) | lo;
    };
    unsigned woff[WJ];                                   // byte offset of this lane's weight chunk at k-slab 0
    int woff_co0 = -1;
    unsigned xpx[XJ], xinv[XJ];                          // activation rows: input pixel of tap 0 (relative to the tile's first image); INVERTED valid-tap bits
    // row (wave + NW j)*8 + rg → (row >> 1) & 7 = 4*(wave & 1) + (rg >> 1) for every j (NW is even)
    const unsigned lcb = (unsigned)((pc ^ (((wave & 1) << 2) + (rg >> 1))) << 4);
    int cur_c = 0, cur_s = 0, cur_tap = 0, cur_tpx = 0;  // wave-uniform k-slab cursor
    const long long img0 = (long long)p.h * p.w * p.c0 * 2, img1 = (long long)p.h * p.w * p.c1 * 2;

    unsigned rowrep = 0;                                 // bit r * kw for every filter row r
#pragma unroll
    for (int r = 0; r < 8; ++r)
        if (r < p.kh) rowrep |= 1u << (r * p.kw);
    auto setup = [&](int v) __attribute__((always_inline)) {
        int co0, pix0;
        tile_coords(v, co0, pix0);
        const long long wbytes = (long long)(p.cout - co0) * p.K * 2;
        bW = (unsigned long long)(reinterpret_cast<const f16*>(p.wgt) + (size_t)co0 * p.K);
        nW = (int)(wbytes < 0x7fffffffLL ? wbytes : 0x7fffffffLL);
        const bool p2 = p.howo_shift >= 0 && p.wo_shift >= 0;
        const int n_first = p2 ? pix0 >> p.howo_shift : pix0 / p.howo;
        const int n_last = p2 ? (min(pix0 + BP, p.npix) - 1) >> p.howo_shift : (min(pix0 + BP, p.npix) - 1) / p.howo;
        const int nimg = n_last - n_first + 1;
        bX0 = (unsigned long long)(reinterpret_cast<const char*>(p.x0) + (size_t)n_first * img0);
        nX0 = (int)(img0 * nimg);
        bX1 = (unsigned long long)(p.x1 ? reinterpret_cast<const char*>(p.x1) + (size_t)n_first * img1 : reinterpret_cast<const char*>(p.x0));
        nX1 = (int)(p.x1 ? img1 * nimg : 0);
        if (co0 != woff_co0) {                                          // (wave-uniform; one channel tile: computed once per launch)
            woff_co0 = co0;
#pragma unroll
            for (int j = 0; j < WJ; ++j) {
                const int row = (wave + NW * j) * 8 + rg;
                const int ch = dma_weight_channel_mx(row);
                const int lc = pc ^ ((row >> 1) & 7);
                woff[j] = (co0 + ch < p.cout) ? (unsigned)(ch * p.K * 2 + lc * 16) : OOB;
            }
        }
        // activation rows: closed-form mask of the filter taps whose input pixel exists (see conv_dma_kernel::setup)
#pragma unroll
        for (int j = 0; j < XJ; ++j) {
            const int row = (wave + NW * j) * 8 + rg;
            const int pix = pix0 + row;
            const int pixc = min(pix, p.npix - 1);
            int n, rem, oh, ow;
            if (p2) { n = pixc >> p.howo_shift; rem = pixc & (p.howo - 1); oh = rem >> p.wo_shift; ow = rem & (p.wo - 1); }
            else { n = pixc / p.howo; rem = pixc - n * p.howo; oh = rem / p.wo; ow = rem - oh * p.wo; }
            const int xvwj = p.valid_w ? p.valid_w[n] : p.w;
            const int ih0 = oh * p.sh - p.ph, iw0 = ow * p.sw - p.pw;
            const int px = ((n - n_first) * p.h + ih0) * p.w + iw0;
            const int vw = min(xvwj, p.w);
            const int qlo = min(p.kw, max(0, -iw0)), qhi = min(p.kw, vw - iw0), rlo = min(p.kh, max(0, -ih0)), rhi = min(p.kh, p.h - ih0);
            const unsigned cm = ((1u << max(qhi, 0)) - 1u) & ~((1u << qlo) - 1u);
            const int sh_hi = max(rhi, 0) * p.kw, sh_lo = rlo * p.kw;
            const unsigned rr = (sh_hi >= 32 ? rowrep : rowrep & ((1u << sh_hi) - 1u)) & (sh_lo >= 32 ? 0u : ~((1u << sh_lo) - 1u));
            const bool live = pix < p.npix && !((lcb >> 4) == 7u && !p.mx_fetch_pad);            // chunk 7 of an fp16+8 activation block is padding: not fetched
            xpx[j] = pix < p.npix ? (unsigned)px : 0u;
            xinv[j] = ~(live ? cm * rr : 0u);
        }
        cur_c = 0; cur_s = 0; cur_tap = 0; cur_tpx = 0;
    };
    auto advance_cursor = [&]() __attribute__((always_inline)) {
        if (p.x1_center && cur_c >= p.c0) { cur_c += 64; return; }
        ++cur_tap;
        if (++cur_s == p.kw) { cur_s = 0; cur_tpx += p.w - (p.kw - 1); } else { ++cur_tpx; }
        if (cur_tap == p.kh * p.kw) {
            cur_tap = 0; cur_s = 0; cur_tpx = 0; cur_c += 64;
            if (p.x1_center && cur_c >= p.c0) { cur_tap = p.center_tap; cur_tpx = p.center_tpx; }
        }
    };

    int i_v = blockIdx.x, i_kt = 0, i_stage = 0;         // head of the slab stream: tile, slab, LDS stage
    bool i_live = true;
    auto sw_begin = [&]() __attribute__((always_inline)) -> bool {
        if (i_kt == nk) {
            i_kt = 0; i_v += G;
            i_live = i_v < ntiles;
            if (i_live) setup(i_v);
        }
        return i_live;
    };
    // wave-uniform parts of a slab's addresses, formed once per slab; a slab that does not exist (end of the stream) is issued with out-of-range offsets
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
            __builtin_amdgcn_raw_ptr_buffer_load_lds(sl_rW, (lds_void*)(sw_ + (wave + NW * idx) * 1024), 16, vo, 0, 0, 0);
        } else {
            const int j = idx - WJ;
            unsigned char* sx_ = sw_ + BC * 128;
            // branch-free: an invalid tap ORs bit 31 into the offset (the INVERTED mask shifted so that the tap's bit is bit 31), beyond every num_records
            const unsigned inval = (xinv[j] << (31u - sl_tap)) & OOB;
            const unsigned vo = ((unsigned)__mul24((int)xpx[j], (int)sl_cb) + sl_uni) | inval | sl_dead;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(sl_rX, (lds_void*)(sx_ + (wave + NW * j) * 1024), 16, vo, 0, 0, 0);
        }
    };
    auto sw_end = [&]() __attribute__((always_inline)) {
        advance_cursor();
        i_stage ^= 1;
        ++i_kt;
    };

    unsigned ph_sum[7] = {0u, 0u, 0u, 0u, 0u, 0u, 0u}, ph_prev = 0u, ph_slabs = 0u, ph_tiles = 0u;
    auto ph_stamp = [&](int i) __attribute__((always_inline)) {
        if constexpr (W4_STAMPS != 0) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned t = (unsigned)__builtin_readcyclecounter();
            if (i >= 0) ph_sum[i] += t - ph_prev;
            ph_prev = t;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // ================================================================== compute side
    f32x16 acc[FA][FB];                                  // a[0:255]
    int mx_sa[FA];                                       // per weight fragment: E8M0 byte of s_w * 2^-11 for this lane's row
    auto load_scales = [&](int v) __attribute__((always_inline)) {
        int co0, pix0;
        tile_coords(v, co0, pix0);
        const unsigned char* wexp = reinterpret_cast<const unsigned char*>(p.wgt) + (size_t)p.cout * p.K * 2;
#pragma unroll
        for (int f = 0; f < FA; ++f) {
            const int ch = co0 + dma_weight_channel_mx(wc * (BC / WC) + f * 32 + l32);
            mx_sa[f] = ch < p.cout ? (int)wexp[ch] : 0;
        }
#pragma unroll
        for (int f = 0; f < FA; ++f) asm volatile("" : "+v"(mx_sa[f]));          // returned here, in the compiler's book-keeping too (no vmcnt(0) inside the slab loop)
    };
    auto zero_acc = [&]() __attribute__((always_inline)) {
        u32x4 z = {0u, 0u, 0u, 0u};
        asm volatile("" : "+v"(z));                      // one operand tuple for the 16 statements (not rematerialised in front of each)
#pragma unroll
        for (int fa = 0; fa < FA; ++fa)
#pragma unroll
            for (int fb = 0; fb < FB; ++fb) W4_MFMA_ZERO(acc[fa][fb], z);
    };
    auto bar = [&]() __attribute__((always_inline)) {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    // per-lane LDS offsets of the operand fragments (fragment f is f * 32 rows = f * 4096 bytes further; the other chunks of a row are XORs on the address)
    const unsigned pa0 = (unsigned)swz_dma(wc * (BC / WC) + l32, h), pa8 = (unsigned)swz_dma(wc * (BC / WC) + l32, 4 + 2 * h);
    const unsigned pb0 = (unsigned)swz_dma(wp * (BP / WP) + l32, h), pbe = (unsigned)swz_dma(wp * (BP / WP) + l32, 6);

    u32x4 a[2][FA], bh[2][FB];                           // f16 operands of the slab whose f16 MFMAs run in this iteration
    i32x8 a8[FA], b8[FB];                                // fp8-side operands, carried to the NEXT iteration's scaled MFMAs
    int eb[FB], ebn[FB];                                 // E8M0 of s_x: of the carried slab / of the slab being read

    // ---- the parts of an iteration
    auto front_read = [&](unsigned so, int i) __attribute__((always_inline)) {   // LDS read i (0-11) of slab s's first f16 k-step and its scale bytes
        const unsigned aa = pa0 + so, ba = pb0 + so + BC * 128u, bea = pbe + so + BC * 128u;
        if (i < FA) a[0][i] = *reinterpret_cast<const u32x4*>(smem + (aa + i * 4096u));                                    // chunk h
        else if (i < FA + FB) bh[0][i - FA] = *reinterpret_cast<const u32x4*>(smem + (ba + (i - FA) * 4096u));
        else ebn[i - FA - FB] = *(smem + (bea + (i - FA - FB) * 4096u));                                                   // chunk 6, byte 0
    };
    auto front = [&](unsigned so) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < FA + 2 * FB; ++i) front_read(so, i);
    };
    // 16 scaled MFMAs of the carried slab; behind each one DMA piece of the next slab (pieces) and one LDS read of the current slab's first k-step (reads)
    auto scaled_prev = [&](auto with_pieces, auto with_reads, unsigned so) __attribute__((always_inline)) {
        constexpr bool pieces = decltype(with_pieces)::value, reads = decltype(with_reads)::value;
#pragma unroll
        for (int fa = 0; fa < FA; ++fa)
#pragma unroll
            for (int fb = 0; fb < FB; ++fb) {
                const int i = fa * FB + fb;
                W4_MFMA_SC(acc[fa][fb], a8[fa], b8[fb], mx_sa[fa], eb[fb]);
                __builtin_amdgcn_sched_barrier(0);
                if (reads && i < FA + 2 * FB) front_read(so, i);
                if (pieces) sw_piece(i);
                __builtin_amdgcn_sched_barrier(0);
            }
    };
    bool more = true;                                    // the slab whose pieces go out in the next scaled phase exists (wave-uniform)
    auto prep = [&]() __attribute__((always_inline)) { more = sw_begin(); sw_slab(more); };
    auto f16_part = [&](unsigned so) __attribute__((always_inline)) {   // 32 f16 MFMAs of slab s; its fp8-side reads and conversions between them
        const unsigned aa = pa0 + so, a8a = pa8 + so, ba = pb0 + so + BC * 128u;
#pragma unroll
        for (int f = 0; f < FB; ++f) bh[1][f] = *reinterpret_cast<const u32x4*>(smem + ((ba ^ 32u) + f * 4096u));         // chunk 2 + h
#pragma unroll
        for (int f = 0; f < FB; ++f) {
            eb[f] = ebn[f];
            const u32x4 lo8 = *reinterpret_cast<const u32x4*>(smem + ((ba ^ 64u) + f * 4096u));                           // chunk 4 + h
            b8[f][4] = (int)lo8[0]; b8[f][5] = (int)lo8[1]; b8[f][6] = (int)lo8[2]; b8[f][7] = (int)lo8[3];
        }
        __builtin_amdgcn_sched_barrier(0);
        // behind MFMA i:  i < 2 FA: one fp8-side weight read (lo8 / hi8 chunk of fragment i / 2);  i < FA FB, last pixel fragment of a row: the second k-step's
        // weight fragment;  i < 4 FB: one x_hi8 conversion group (2 v_cvt_scalef32_pk_fp8_f16 of 16 channels' halves; first k-step first)
        constexpr int NM = 2 * FA * FB;
        auto step = [&](int i) __attribute__((always_inline)) {
            const int k2 = i / (FA * FB), fa = (i % (FA * FB)) / FB, fb = i % FB;
            W4_MFMA_F16(acc[fa][fb], a[k2][fa], bh[k2][fb]);
            if (i < 2 * FA || i < 4 * FB || (i < FA * FB && fb == FB - 1)) {
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
                    int r32 = bitcast<int>(r);
                    asm volatile("" : "+v"(r32));            // pinned HERE: the value has no reader before the next iteration and LLVM sinks it past the control flow of prep()
                    b8[f][2 * kk + d] = r32;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
#pragma unroll
        for (int i = 0; i <= FA * FB; ++i) step(i);
        ph_stamp(2);
        if (W4_PREP_IN_F16) {               // behind the first MFMA of the second k-step (outside the unrolled loops: the tile crossing's set-up is a large block)
            __builtin_amdgcn_sched_barrier(0);
            prep();
            __builtin_amdgcn_sched_barrier(0);
        }
        ph_stamp(3);
#pragma unroll
        for (int i = FA * FB + 1; i < NM; ++i) step(i);
    };
    // the wait states a 16-pass MFMA's result needs before a VALU (v_accvgpr_read of the epilogue) may read it: hipcc pads nothing behind an asm statement
    auto mfma_drain = [&]() __attribute__((always_inline)) { asm volatile("s_nop 15\n\ts_nop 7" ::: "memory"); };
    // the epilogue takes a block's 16 values out of the accumulator file WHERE IT CONSUMES THEM (asm volatile: not hoisted).  Left to the compiler the 256 reads are
    // scheduled to the top of the epilogue — at one wave per SIMD its scheduler sees no reason to keep the pressure under 256 — and the allocator spills the slab loop.
    struct AccFile {
        f32x16 (&r)[FA][FB];
        __device__ __forceinline__ float get(int fa, int px, int q) const {
            float v;
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(r[fa][px][q]));
            return v;
        }
    };
    auto epilogue = [&](int v) __attribute__((always_inline)) {
        int co0, pix0;
        tile_coords(v, co0, pix0);
        mfma_drain();
        dma_epilogue_mx_acc<BC, BP, WC, WP, 2 * FA, 2 * FB, 64, true>(p, AccFile{acc}, co0, pix0, wc, wp, lane, xpose);
    };

    setup(i_v);
    prep();                                                          // slab 0 (a workgroup always has a tile)
#pragma unroll
    for (int idx = 0; idx < NDMA; ++idx) sw_piece(idx);
    if (more) sw_end();
    int c_v = blockIdx.x, c_kt = 0;                                  // tile / slab index inside it of the slab whose f16 MFMAs run in the iteration
    load_scales(c_v);
    zero_acc();
    const int total = ((ntiles - (int)blockIdx.x + G - 1) / G) * nk; // slabs in this workgroup's stream
    // ---- iteration 0: nothing carried yet
    VMCNT(0);
    bar();
    front(0u);
    __builtin_amdgcn_sched_barrier(0);
    prep();
#pragma unroll
    for (int idx = 0; idx < NDMA; ++idx) sw_piece(idx);
    if (more) sw_end();
    __builtin_amdgcn_sched_barrier(0);
    f16_part(0u);
    c_kt = 1;
    ph_stamp(-1);
    for (int s = 1; s < total; ++s) {
        VMCNT(0);                                                    // this wave's pieces of slab s (and the previous epilogue's stores) have landed ...
        ph_stamp(5);
        bar();                                                       // ... everyone's; nobody reads the stage of slab s-1 any more
        ph_stamp(6);
        ++ph_slabs;
        const unsigned so = (unsigned)(s & 1) * (unsigned)STAGE;
        if (!W4_FRONT_INTERLEAVED) front(so);
        __builtin_amdgcn_sched_barrier(0);
        if (!W4_PREP_IN_F16) prep();
        __builtin_amdgcn_sched_barrier(0);
        scaled_prev(std::true_type{}, std::integral_constant<bool, W4_FRONT_INTERLEAVED != 0>{}, so);
        ph_stamp(0);
        if (more) sw_end();
        if (c_kt == nk) {
            ++ph_tiles;                                            // slab s-1 closed its tile: epilogue, then the next tile's scales and a clean accumulator
            epilogue(c_v);
            c_kt = 0; c_v += G;
            if (p.tilesC > 1) load_scales(c_v);                      // (one channel tile: every tile has the same scales)
            zero_acc();
        }
        ph_stamp(1);
        __builtin_amdgcn_sched_barrier(0);
        f16_part(so);
        ph_stamp(4);
        ++c_kt;
    }
    __builtin_amdgcn_sched_barrier(0);
    scaled_prev(std::false_type{}, std::false_type{}, 0u);           // the last slab's scaled MFMAs
    epilogue(c_v);
    if constexpr (W4_STAMPS != 0) {     // DIAGNOSTIC: phase sums of this wave over the first bytes of the output (64 bytes per wave)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (lane == 0) {
            unsigned* o = reinterpret_cast<unsigned*>(p.y) + ((size_t)blockIdx.x * NW + wave) * 16;
            o[0] = 0x5157a3b7u; o[1] = ph_slabs; o[2] = ph_tiles;
#pragma unroll
            for (int i = 0; i < 7; ++i) o[3 + i] = ph_sum[i];
            o[10] = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_ID
        }
    }
}

int launch_conv_dma_w4(const ConvArgs& a, hipStream_t st) {
    static thread_local DeviceOnce attr_once;
    if (!attr_once.done()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_dma_w4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, w4::LDS);
        if (e != hipSuccess) return mnet_fail(MNET_E_LAUNCH, "hipFuncSetAttribute(dma_w4): %s", hipGetErrorString(e));
        attr_once.mark();
    }
    ConvArgs b = a;
    auto log2_or_minus1 = [](int v) { return v > 0 && (v & (v - 1)) == 0 ? __builtin_ctz((unsigned)v) : -1; };
    b.howo_shift = log2_or_minus1(a.howo); b.wo_shift = log2_or_minus1(a.wo);
    b.tilesC = (a.cout + w4::BC - 1) / w4::BC;
    const int tilesP = (a.npix + w4::BP - 1) / w4::BP;
    b.ntiles = b.tilesC * tilesP;
    int grid = b.ntiles;
    const int lim = dma_grid_limit();
    static const bool env_one_tile = [] { const char* e = getenv("MNET_DMA_ONE_TILE"); return e && atoi(e) != 0; }();
    if (grid > lim && !a.one_tile_per_wg && !env_one_tile) grid = lim & ~7;
    hipLaunchKernelGGL(conv_dma_w4_kernel, dim3((unsigned)grid), dim3(w4::NW * 64), w4::LDS, st, b);
    MNET_LAUNCH_CHECK("conv_dma_w4_kernel");
    return MNET_OK;
}
