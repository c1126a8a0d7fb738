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
//   W4_ACT_FIRST (0)    A/B: the activation pieces (HBM / L2) behind the first 8 scaled MFMAs, the weight pieces (L2-resident) behind the last 8
#ifndef W4_ACT_FIRST
#define W4_ACT_FIRST 0
#endif

namespace w4 {
constexpr int BC = 256, BP = 256, WC = 2, WP = 2, NW = 4;
constexpr int FA = 4, FB = 4;                         // 32x32 accumulator blocks per wave: weight fragments x pixel fragments
constexpr int WJ = BC / (8 * NW), XJ = BP / (8 * NW), NDMA = WJ + XJ;     // 8 + 8 DMA pieces (1 KiB each) per wave and slab
constexpr int STAGE = (BC + BP) * 128, STAGES = 2;
constexpr int XB = 4096;                              // epilogue transposition scratch per wave behind the stages (dma_epilogue_mx)
constexpr int PB = 1536;                              // epilogue parameter area per wave (w4_epilogue): bias 2 x 64 floats + two step buffers of 2 x 64 floats
constexpr int TAB_MAX = 512;                          // k-slabs per tile the slab table holds (16 bytes each); the launcher hands longer k loops to the 8-wave tile
constexpr int LDS = STAGES * STAGE + NW * (XB + PB) + TAB_MAX * 16;   // 158 KiB
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
// the f16 MFMAs of the slab loop: operands straight from the LDS (s_waitcnt, no VALU write — tools/isa_mfma_hazards.py checks the built ISA: the only way one gets in is a
// compiler copy), so no s_nop; the _P form also names a VGPR the statement does not touch: the x_hi8 conversion of the gap before it, which has no reader until the next
// iteration and would otherwise be sunk out of its gap (one statement instead of MFMA + an empty pinning statement with its own boundary pad)
#ifndef W4_F16_NOP
#define W4_F16_NOP 0
#endif
#if W4_F16_NOP
#define W4_MFMA_F16_HOT(ACC, A, B) W4_MFMA_F16(ACC, A, B)
#define W4_MFMA_F16_HOT_P(ACC, A, B, P) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %2, %3, %0" : "+a"(ACC), "+v"(P) : "v"(A), "v"(B))
#else
#define W4_MFMA_F16_HOT(ACC, A, B) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(ACC) : "v"(A), "v"(B))
#define W4_MFMA_F16_HOT_P(ACC, A, B, P) asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0" : "+a"(ACC), "+v"(P) : "v"(A), "v"(B))
#endif
#define W4_MFMA_ZERO(ACC, Z) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %1, 0" : "=a"(ACC) : "v"(Z))
#define W4_MFMA_SC(ACC, A8, B8, SA, SB) asm volatile("s_nop 1\n\tv_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+a"(ACC) : "v"(A8), "v"(B8), "v"(SA), "v"(SB))

// Epilogue of the one-wave-per-SIMD tile: the arithmetic of dma_epilogue_mx (conv_dma_common.h) value for value — accumulator * 2^-8, out_scale, bias, residual, activation,
// post_scale, GroupNorm sums, fp16+8 encode, store through the wave's LDS scratch — re-ordered for ONE wave per SIMD.  With two waves per SIMD the partner hides a step's
// latencies; alone, dma_epilogue_mx's 8 steps (4 pixel fragments x 2 channel blocks per lane) cost 68 000 cycles per tile (tools/w4_phases.py, profiles/r6c_*): every step's
// parameter loads sit behind the previous step's stores on the in-order vmcnt counter, i.e. each step waits for the write-back of the one before.  Here
//   * the per-channel vectors go through the wave's LDS parameter area, ONE float per lane: the 64 lanes of a step need 2 x 32 values of each vector (the 32 pixels of a lane
//     half share their 32 channels) — lane l fetches value l of the step's 64-channel window, writes it to the LDS, and every lane reads back its half's 32 values as 8
//     broadcast ds_read_b128 (lgkmcnt: not ordered behind the stores).  The bias window is fetched once per tile (it does not depend on the pixel fragment), the
//     out_scale / post_scale windows (rows of the fragment's image) one step ahead;
//   * the residual block of step t + 1 (7 loads per lane) is requested BEFORE the stores of step t, so the wait in front of step t + 1's arithmetic leaves those stores in flight.
// The one-float-per-lane form needs the 32 pixels of a fragment in ONE image (ho * wo % 32 == 0) and the arithmetic has the identity / LeakyReLU arms only: the launcher
// hands every other launch to the 8-wave tile.  Straight-line arithmetic (round 6, last pass): a bias the launch does not have is zeros, a scale vector it does not have is
// ones, 2^-8 rides in the fma with the bias (or in out_scale) — every runtime branch in the step split hipcc's scheduling region and copied the 32 values at its join.
// ACC::block(fa, px) takes a block's 16 values out of the accumulator file where they are consumed.  `par`: this wave's parameter area (w4::PB bytes).
// SC / RG: the launch has out_scale or post_scale / a residual or GroupNorm sums.  Four builds of the kernel (launch_conv_dma_w4 picks): a tile without them runs an
// epilogue without their code, branches and kernel-argument reloads — measured on the bias + activation launches: 36 000 -> 29 000 cycles per tile (profiles/r6m_*).
template <bool SC, bool RG, typename ACC, typename STAMP>
__device__ __forceinline__ void w4_epilogue(const ConvArgs& p_, const ACC& acc, int co0, int pix0, int wc, int wp, int lane, unsigned char* xpose, unsigned char* par, STAMP&& stamp) {
#pragma clang fp contract(off)      // every product and sum rounded on its own, as in dma_epilogue_mx — whose runtime arms keep hipcc from fusing across them; in this straight-line
                                    // form it fused scale * acc + bias into an fma
    using namespace w4;
    constexpr int NPX = FB, NB = FA / 2, NS = NPX * NB;
    const int h = lane >> 5;
    constexpr bool GN = RG;
    const ConvArgs& p = p_;
    const float* const p_out_scale = SC ? p.out_scale : nullptr;
    const float* const p_post_scale = SC ? p.post_scale : nullptr;
    const void* const p_res = RG ? p.res : nullptr;
    const int last_pix = p.npix - 1;
    float* gnp = nullptr;
    if constexpr (GN) gnp = kernarg_gn_partial();
    int cob[NB], co[NB], co_l[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        cob[b] = co0 + wc * (BC / WC) + b * 64;                             // first channel of lane-half 0's block
        co[b] = cob[b] + h * 32;                                            // first channel of this lane's block
        co_l[b] = min(co[b], p.cout - 32);                                  // (a lane whose block lies beyond cout works on the last block's parameters; nobody stores its result)
    }
    // LDS parameter area of this wave: [bias: NB x 64 floats][step buffer 0: out_scale 64 | post_scale 64 floats][step buffer 1].  A lane's own value of a 64-channel
    // window sits at float index `lane`; its half's 32 values are the 8 chunks at float index 32 h.  (channels >= cout: any in-range value — those blocks are not stored)
    float* const par_f = reinterpret_cast<float*>(par);
#pragma unroll
    for (int b = 0; b < NB; ++b) par_f[b * 64 + lane] = p.bias ? p.bias[min(cob[b] + lane, p.cout - 1)] : 0.f;      // (no bias: zeros — the add below is unconditional)
    struct Step { float osc1, psc1; u32x4 rh[4], rl[2]; unsigned re8; int pixb, pix, n_img, vw; };
    auto request = [&](int t, Step& S) __attribute__((always_inline)) {     // addresses + parameter loads of step t (no use of the values here)
        const int px = t / NB, b = t % NB;
        S.pixb = pix0 + wp * (BP / WP) + px * 32;                           // first pixel of this wave's 32
        S.pix = S.pixb + (lane & 31);
        const bool p2 = p.howo_shift >= 0;                                  // (wave-uniform) every map of this network: a shift instead of an integer division per lane and step
        S.n_img = p2 ? min(S.pix, last_pix) >> p.howo_shift : min(S.pix, last_pix) / p.howo;
        if constexpr (SC) {                                                 // the fragment's image is wave-uniform (ho * wo % 32 == 0): one float of the 64-channel window per lane
            const int img = p2 ? min(S.pixb, last_pix) >> p.howo_shift : min(S.pixb, last_pix) / p.howo;
            const size_t row = (size_t)img * p.cout + min(cob[b] + lane, p.cout - 1);
            // (a vector the launch does not have: ones — the multiplies below are unconditional.  2^-8 of the weight scale rides in out_scale: a power of two, the
            //  product acc * (2^-8 s) is the once-rounded acc * 2^-8 * s of dma_epilogue_mx bit for bit)
            S.osc1 = MNET_SPLIT_WSCALE_INV; S.psc1 = 1.f;
            if (p_out_scale) S.osc1 = p_out_scale[row] * MNET_SPLIT_WSCALE_INV;
            if (p_post_scale) S.psc1 = p_post_scale[row];
        }
        if (GN && gnp && p.valid_w) S.vw = p.valid_w[S.n_img];             // (the GroupNorm sums' column bound: requested with the rest, not behind the step's stores)
        if (p_res && S.pix < p.npix) {
            const int rpix = p.res_mod > 0 ? S.pix % p.res_mod : S.pix;
            const unsigned char* rb = reinterpret_cast<const unsigned char*>(p_res) + (size_t)rpix * p.cout * 4 + (co_l[b] >> 5) * 128;
#pragma unroll
            for (int c = 0; c < 4; ++c) S.rh[c] = ldg16(rb + c * 16);
            S.rl[0] = ldg16(rb + 64); S.rl[1] = ldg16(rb + 80);           // lo bytes of chunks (0, 2) | (1, 3)
            S.re8 = rb[96];
        }
    };
    // every store of the epilogue is UNCONDITIONAL: a raw buffer store through a descriptor of this tile's window of the output (num_records = its valid pixels), a lane
    // with nothing to write gets an out-of-range offset and the hardware drops it.  Stores behind `if (valid)` branches would make the number of VMEM operations issued
    // after the next step's parameter loads path-dependent, and hipcc's s_waitcnt pass then waits with vmcnt(0) — for the stores — in front of every step (measured: 59
    // vmcnt(0) waits per epilogue, 90 000 cycles per tile)
    const int tile_px = min(BP, p.npix - pix0);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(p.y) + (size_t)pix0 * p.cout * 4, 0,
                                                                         __builtin_amdgcn_readfirstlane(tile_px * p.cout * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(gnp, 0, __builtin_amdgcn_readfirstlane(gnp ? (p.npix >> 5) * (p.cout >> 5) * 8 : 0), 0x00020000);
    Step cur;
    request(0, cur);
    stamp(7);                                                               // (W4_STAMPS == 2: sub-phases of the epilogue, see the kernel)
#pragma unroll
    for (int t = 0; t < NS; ++t) {
        const int px = t / NB, b = t % NB;
        const int pixb = cur.pixb, pix = cur.pix;
        float* const stepbuf = par_f + NB * 64 + (t & 1) * 128;
        if constexpr (SC) { stepbuf[lane] = cur.osc1; stepbuf[64 + lane] = cur.psc1; }      // this step's scale windows → LDS (read back below, per half)
        typedef __attribute__((address_space(3))) const f32x4* lds_f32x4;
        f32x4 bia[8];                                                       // this half's 32 bias values (8 broadcast reads, requested in front of the accumulator reads)
        {
            const lds_f32x4 b4 = (lds_f32x4)(par_f + b * 64 + 32 * h);
#pragma unroll
            for (int q = 0; q < 8; ++q) bia[q] = b4[q];
        }
        float v[32];
        {
            const f32x16 b0 = acc.block(2 * b, px), b1 = acc.block(2 * b + 1, px);      // out of the accumulator file HERE (see AccFile)
            if constexpr (SC) {
                const lds_f32x4 s4 = (lds_f32x4)(stepbuf + 32 * h);
#pragma unroll
                for (int q = 0; q < 16; ++q) { v[q] = b0[q] * s4[q >> 2][q & 3]; v[16 + q] = b1[q] * s4[4 + (q >> 2)][q & 3]; }
#pragma unroll
                for (int q = 0; q < 32; ++q) v[q] += bia[q >> 2][q & 3];
            } else {
                // acc * 2^-8 + bias as ONE fma: the product is exact (a power of two), so the single rounding is the sum's — dma_epilogue_mx's value bit for bit
#pragma unroll
                for (int q = 0; q < 16; ++q) { v[q] = __builtin_fmaf(b0[q], MNET_SPLIT_WSCALE_INV, bia[q >> 2][q & 3]); v[16 + q] = __builtin_fmaf(b1[q], MNET_SPLIT_WSCALE_INV, bia[4 + (q >> 2)][q & 3]); }
            }
        }
        if (p_res && pix < p.npix) {
            const float sl = hm_lo_scale((int)cur.re8);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f16x8 h8 = bitcast<f16x8>(cur.rh[c]);
                float l[8];
                hm_decode_lo(u32x2{cur.rl[c & 1][2 * (c >> 1)], cur.rl[c & 1][2 * (c >> 1) + 1]}, sl, l);      // slot of chunk c = hm_lo_slot(c) = 2 (c & 1) + (c >> 1)
#pragma unroll
                for (int q = 0; q < 8; ++q) v[c * 8 + q] += (float)h8[q] + l[q];
            }
        }
        // (LeakyReLU as max(v, 0.2 v): one multiply + one max instead of compare / multiply / select — the same value for every input: 0.2 v > v exactly when v < 0,
        //  a NaN stays a NaN, -inf stays -inf; ReLU keeps act_apply_vec's form, whose relu(-inf) = NaN is wanted)
        //  the launcher hands every other activation to the 8-wave tile: ONE in-place arm here, no second producer of v — hipcc copied all 32 values at the join of two.
        //  v_max_f32 written out: fmaxf() adds a canonicalising v_max v, v, v per value)
        if (p.act != MNET_ACT_NONE) {
            const float post = p.act == MNET_ACT_LRELU_SQRT2 ? 1.41421356237309515f : 1.f;
#pragma unroll
            for (int q = 0; q < 32; q += 2) {
                typedef float pk2 __attribute__((ext_vector_type(2)));
                const pk2 t = pk2{v[q], v[q + 1]} * 0.2f;                  // (one v_pk_mul_f32 per pair)
                float m0, m1;
                asm("v_max_f32 %0, %1, %2" : "=v"(m0) : "v"(v[q]), "v"(t[0]));
                asm("v_max_f32 %0, %1, %2" : "=v"(m1) : "v"(v[q + 1]), "v"(t[1]));
                v[q] = m0 * post; v[q + 1] = m1 * post;
            }
        }
        if constexpr (SC) {
            const lds_f32x4 s4 = (lds_f32x4)(stepbuf + 64 + 32 * h);
#pragma unroll
            for (int q = 0; q < 32; ++q) { v[q] *= s4[q >> 2][q & 3]; asm("" : "+v"(v[q])); }      // (opaque: hipcc would fold this product into the f16 conversion below — v_fma_mixlo_f16, ONE rounding: other bytes on ties)
        }
        // GroupNorm statistics of this output, part 1 (see dma_epilogue_mx): two fp32 sums per lane over its 32 channels of ONE group
        float gs1 = 0.f, gs2 = 0.f;
        if (GN && gnp) {
#pragma unroll
            for (int q = 0; q < 32; ++q) { gs1 += v[q]; gs2 = fmaf(v[q], v[q], gs2); }
        }
        f16x8 hh[4];
        float m32 = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int q = 0; q < 8; ++q) { hh[c][q] = (f16)v[c * 8 + q]; m32 = fmaxf(m32, fabsf(v[c * 8 + q])); }
        const float m = (float)(f16)m32;
        const int e8 = hm_e8_of(m);
        u32x2 lo[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) lo[c] = hm_encode_lo(v + c * 8, hh[c], e8);      // (the mixed-precision VALU form of the streaming kernels: the same bytes, a quarter fewer instructions — this epilogue IS instruction-bound)
        stamp(8);
        // the NEXT step's parameters are requested here — in front of this step's stores on the vmcnt counter
        Step nxt = cur;
        if (t + 1 < NS) request(t + 1, nxt);
        stamp(9);
        // ---- through the LDS: two rounds of 4 pieces per lane (the hi halves, then lo bytes | lo bytes | scale | padding), see dma_epilogue_mx
        const unsigned L = (unsigned)lane;
        const unsigned wsw = (L >> 1) & 3u, j = L & 3u;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            u32x4 pc[4];
            if (half == 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c) pc[c] = bitcast<u32x4>(hh[c]);
            } else {
                pc[0] = u32x4{lo[0][0], lo[0][1], lo[2][0], lo[2][1]};
                pc[1] = u32x4{lo[1][0], lo[1][1], lo[3][0], lo[3][1]};
                pc[2] = u32x4{(unsigned)e8, 0u, 0u, 0u};
                pc[3] = u32x4{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) *reinterpret_cast<u32x4*>(xpose + ((4u * L + ((unsigned)c ^ wsw)) << 4)) = pc[c];
            u32x4 piece[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned P = (L >> 2) + 16u * k;                      // the lane whose block this lane helps to write
                piece[k] = *reinterpret_cast<const u32x4*>(xpose + ((4u * P + (j ^ ((P >> 1) & 3u))) << 4));
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(piece[k]));   // all four reads in flight
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned P = (L >> 2) + 16u * k;
                const int ppix = pixb + (int)(P & 31u), pco = cob[b] + (int)(P >> 5) * 32;
                // (pixels >= npix lie beyond the descriptor's num_records; a block beyond cout gets bit 31)
                const unsigned off = (unsigned)((ppix - pix0) * p.cout * 4 + (pco >> 5) * 128 + half * 64) + j * 16u;
                __builtin_amdgcn_raw_buffer_store_b128(piece[k], ry, (int)(pco < p.cout ? off : OOB), 0, 0);
            }
            stamp(10 + half);
        }
        if constexpr (GN) {
            // part 2: a fixed xor tree over the 32 pixels of the lane half, one 8-byte store per (32 pixels, group) — the fragment and the tree of every fp16+8 tile
            if (gnp) {
                bool ok = pix < p.npix && co[b] < p.cout;
                if (p.valid_w) {
                    const int ow = p.wo_shift >= 0 ? (pix & (p.wo - 1)) : pix % p.wo;
                    ok = ok && ow < cur.vw;
                }
                gs1 = ok ? gs1 : 0.f; gs2 = ok ? gs2 : 0.f;
#pragma unroll
                for (int o = 1; o < 32; o <<= 1) { gs1 += __shfl_xor(gs1, o, 64); gs2 += __shfl_xor(gs2, o, 64); }
                const bool wr = (lane & 31) == 0 && pixb < p.npix && co[b] < p.cout;
                const unsigned goff = (unsigned)(((pixb >> 5) * (p.cout >> 5) + (co[b] >> 5)) * 8);
                __builtin_amdgcn_raw_buffer_store_b64(u32x2{__builtin_bit_cast(unsigned, gs1), __builtin_bit_cast(unsigned, gs2)}, rg, (int)(wr ? goff : OOB), 0, 0);
            }
        }
        stamp(12);
        cur = nxt;
    }
}

template <bool SC, bool RG>
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
    // buffer descriptors of the tile being streamed (weights; first / second concat source), built ONCE per tile by setup() through readfirstlane (provably in SGPRs: a
    // descriptor hipcc believes divergent gets a waterfall loop around every DMA) — conv_dma_kernel rebuilds them per slab, ~25 scalar instructions this tile's single
    // wave per SIMD has no partner to hide.  After the last slab of the stream they are swapped for `dnull` (num_records = 0): the surplus pieces of the software
    // pipeline then write zeros into a stage nobody reads, without an "| dead" on every offset.
    const __amdgpu_buffer_rsrc_t dnull = __builtin_amdgcn_make_buffer_rsrc((void*)nullptr, 0, 0, 0x00020000);
    __amdgpu_buffer_rsrc_t dW = dnull, dX0 = dnull, dX1 = dnull;
    auto uni64 = [](unsigned long long v) __attribute__((always_inline)) -> unsigned long long {
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return ((unsigned long long)hi << 32) | lo;
    };
    unsigned woff[WJ];                                   // byte offset of this lane's weight chunk at k-slab 0
    int woff_co0 = -1;
    unsigned xpx[XJ], xinv[XJ];                          // activation rows: input pixel of tap 0 (relative to the tile's first image); INVERTED valid-tap bits
    // row (wave + NW j)*8 + rg → (row >> 1) & 7 = 4*(wave & 1) + (rg >> 1) for every j (NW is even)
    const unsigned lcb = (unsigned)((pc ^ (((wave & 1) << 2) + (rg >> 1))) << 4);
    // The k-slab sequence of a tile — 64-channel slice outer, filter tap inner; with p.x1_center the slices of the second source one slab each, at the centre tap — is the
    // same for every tile of the launch: its per-slab address parts are tabulated ONCE, in the LDS, by the whole workgroup (slab_table below).  conv_dma_kernel advances a
    // cursor (tap, column, input-pixel offset, slice) and re-derives the parts per slab: ~60 scalar instructions and a dozen branches that a lone wave per SIMD cannot hide.
    //   entry kt = { weight k byte offset, activation byte offset of (tap, slice) without the lane's chunk, 31 - tap | second source << 16, bytes per pixel of the source }
    unsigned char* const tab = smem + STAGES * STAGE + NW * (XB + PB);
    const long long img0 = (long long)p.h * p.w * p.c0 * 2, img1 = (long long)p.h * p.w * p.c1 * 2;

    unsigned rowrep = 0;                                 // bit r * kw for every filter row r
#pragma unroll
    for (int r = 0; r < 8; ++r)
        if (r < p.kh) rowrep |= 1u << (r * p.kw);
    auto setup = [&](int v) __attribute__((always_inline)) {
        int co0, pix0;
        tile_coords(v, co0, pix0);
        const long long wbytes = (long long)(p.cout - co0) * p.K * 2;
        const unsigned long long bW = (unsigned long long)(reinterpret_cast<const f16*>(p.wgt) + (size_t)co0 * p.K);
        const int nW = (int)(wbytes < 0x7fffffffLL ? wbytes : 0x7fffffffLL);
        const bool p2 = p.howo_shift >= 0 && p.wo_shift >= 0;
        const int n_first = p2 ? pix0 >> p.howo_shift : pix0 / p.howo;
        const int n_last = p2 ? (min(pix0 + BP, p.npix) - 1) >> p.howo_shift : (min(pix0 + BP, p.npix) - 1) / p.howo;
        const int nimg = n_last - n_first + 1;
        const unsigned long long bX0 = (unsigned long long)(reinterpret_cast<const char*>(p.x0) + (size_t)n_first * img0);
        const int nX0 = (int)(img0 * nimg);
        const unsigned long long bX1 = (unsigned long long)(p.x1 ? reinterpret_cast<const char*>(p.x1) + (size_t)n_first * img1 : reinterpret_cast<const char*>(p.x0));
        const int nX1 = (int)(p.x1 ? img1 * nimg : 0);
        dW = __builtin_amdgcn_make_buffer_rsrc((void*)uni64(bW), 0, __builtin_amdgcn_readfirstlane(nW), 0x00020000);
        dX0 = __builtin_amdgcn_make_buffer_rsrc((void*)uni64(bX0), 0, __builtin_amdgcn_readfirstlane(nX0), 0x00020000);
        dX1 = __builtin_amdgcn_make_buffer_rsrc((void*)uni64(bX1), 0, __builtin_amdgcn_readfirstlane(nX1), 0x00020000);
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
    };
    {   // slab_table: entry kt by closed form (thread t takes kt = t, t + 256, ...)
        const int ntaps = p.kh * p.kw, first = p.x1_center ? ntaps * (p.c0 / 64) : nk;      // slabs that walk all taps
        for (int kt = tid; kt < nk; kt += NW * 64) {
            int tap, c;
            if (kt < first) { tap = kt % ntaps; c = (kt / ntaps) * 64; }
            else { tap = p.center_tap; c = p.c0 + (kt - first) * 64; }
            const int tpx = (tap / p.kw) * p.w + tap % p.kw;               // input-pixel offset of the tap relative to tap 0 (centre tap: p.center_tpx)
            const bool second = c >= p.c0;
            const unsigned cb = (unsigned)(second ? p.c1 : p.c0) * 2u;
            u32x4 e;
            e[0] = (unsigned)(tap * p.cin + c) * 2u;
            e[1] = (unsigned)(tpx * (int)cb + (second ? c - p.c0 : c) * 2);
            e[2] = (31u - (unsigned)tap) | (second ? 0x10000u : 0u);
            e[3] = cb;
            *reinterpret_cast<u32x4*>(tab + kt * 16) = e;
        }
        __syncthreads();
    }

    int i_v = blockIdx.x, i_kt = 0, i_stage = 0;         // head of the slab stream: tile, slab, LDS stage
    bool i_live = true;
    bool ph_crossed = false;                             // (W4_STAMPS: this iteration's book-keeping crossed into the next tile)
    auto sw_begin = [&]() __attribute__((always_inline)) -> bool {
        if (i_kt == nk) {
            if constexpr (W4_STAMPS != 0) ph_crossed = true;
            i_kt = 0; i_v += G;
            i_live = i_v < ntiles;
            if (i_live) setup(i_v);
            else { dW = dnull; dX0 = dnull; dX1 = dnull; }           // end of the stream: every further piece is out of range
        }
        return i_live;
    };
    // wave-uniform parts of a slab's addresses, formed once per slab; a slab that does not exist (end of the stream) is issued with out-of-range offsets
    // the address parts of the slab whose pieces go out next: fetched from the table (sw_fetch, one broadcast ds_read_b128) and unpacked (sw_slab) in two places of the f16
    // part, so that the read's latency passes under an MFMA.  They stay in VGPRs (wave-uniform values): only the choice of the source descriptor needs a scalar
    __amdgpu_buffer_rsrc_t sl_rX = dnull;
    u32x4 sl_e = {0u, 0u, 0u, 0u};
    unsigned sl_kb = 0, sl_sh = 0, sl_cb = 0, sl_uni = 0;
    auto sw_fetch = [&]() __attribute__((always_inline)) { sl_e = *reinterpret_cast<const u32x4*>(tab + i_kt * 16); };
    auto sw_slab = [&]() __attribute__((always_inline)) {
        sl_kb = sl_e[0];
        sl_uni = sl_e[1] + lcb;
        sl_sh = sl_e[2] & 0xffffu;
        sl_cb = sl_e[3];
        sl_rX = __builtin_amdgcn_readfirstlane(sl_e[2] >> 16) ? dX1 : dX0;
    };
    auto sw_piece = [&](int idx) __attribute__((always_inline)) {
        unsigned char* sw_ = smem + i_stage * STAGE;
        if (idx < WJ) {
            const unsigned vo = woff[idx] + sl_kb;                   // (an OOB row keeps bit 31 through the addition: sl_kb < 2^31)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(dW, (lds_void*)(sw_ + (wave + NW * idx) * 1024), 16, vo, 0, 0, 0);
        } else {
            const int j = idx - WJ;
            unsigned char* sx_ = sw_ + BC * 128;
            // branch-free: an invalid tap ORs bit 31 into the offset (the INVERTED mask shifted so that the tap's bit is bit 31), beyond every num_records
            const unsigned inval = (xinv[j] << sl_sh) & OOB;
            const unsigned vo = ((unsigned)__mul24((int)xpx[j], (int)sl_cb) + sl_uni) | inval;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(sl_rX, (lds_void*)(sx_ + (wave + NW * j) * 1024), 16, vo, 0, 0, 0);
        }
    };
    auto sw_end = [&]() __attribute__((always_inline)) {
        i_stage ^= 1;
        ++i_kt;
    };

    unsigned ph_sum[13] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}, ph_prev = 0u, ph_slabs = 0u, ph_tiles = 0u;
    bool ph_on = false;                                  // (iteration 0 runs the f16 part before the loop's first stamp: not booked)
    auto ph_stamp = [&](int i) __attribute__((always_inline)) {
        if constexpr (W4_STAMPS != 0) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned t = (unsigned)__builtin_readcyclecounter();
            if (i >= 0 && ph_on) ph_sum[i] += t - ph_prev;
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
                if (pieces) sw_piece(W4_ACT_FIRST ? (i + WJ) % NDMA : i);
                __builtin_amdgcn_sched_barrier(0);
            }
    };
    bool more = true;                                    // the slab whose pieces go out in the next scaled phase exists (wave-uniform)
    auto prep = [&]() __attribute__((always_inline)) { more = sw_begin(); sw_fetch(); sw_slab(); };
    auto f16_part = [&](unsigned so) __attribute__((always_inline)) {   // 32 f16 MFMAs of slab s; its fp8-side reads and conversions between them
        const unsigned aa = pa0 + so, a8a = pa8 + so, ba = pb0 + so + BC * 128u;
#pragma unroll
        for (int f = 0; f < FB; ++f) eb[f] = ebn[f];
        __builtin_amdgcn_sched_barrier(0);
        // behind MFMA i:  i < 2 FA: one fp8-side weight read (lo8 / hi8 chunk of fragment i / 2);  i < FA FB, last pixel fragment of a row: the second k-step's
        // weight fragment;  i < 4 FB: one x_hi8 conversion group (2 v_cvt_scalef32_pk_fp8_f16 of 16 channels' halves; first k-step first)
        constexpr int NM = 2 * FA * FB;
        int cvq[2] = {0, 0};                             // conversion results in flight: the one of gap i is named by MFMA statement i + 2 (not i + 1: hipcc pads a wait state between a
                                                         // VALU write and an asm statement that names the register — 16 s_nop per slab in gaps that are full), then filed into b8
        // the 4 FB conversion groups (group g: k-step g / (2 FB), pixel fragment (g >> 1) % FB, dword g & 1) are dealt over the gaps so that none carries more than the ~5
        // instructions a 32-cycle MFMA hides (MI355X_MICROARCH.md): the first k-step's 2 FB groups behind MFMAs 0 ... 2 FB - 1 (beside the fp8-side weight reads), the
        // second k-step's behind MFMAs CG1 ... CG1 + 2 FB - 1 of the second k-step, which have nothing else to issue
        constexpr int CG1 = FA * FB + 5;
        auto group_of = [](int i) constexpr -> int { return i < 0 ? -1 : (i < 2 * FB ? i : (i >= CG1 && i < CG1 + 2 * FB ? 2 * FB + (i - CG1) : -1)); };
        static_assert(CG1 + 2 * FB + 1 < 2 * FA * FB, "the last conversion group needs an MFMA statement two gaps behind it to name its result");
        auto step = [&](int i) __attribute__((always_inline)) {
            const int k2 = i / (FA * FB), fa = (i % (FA * FB)) / FB, fb = i % FB;
            if (group_of(i - 2) >= 0) {
                W4_MFMA_F16_HOT_P(acc[fa][fb], a[k2][fa], bh[k2][fb], cvq[i & 1]);
                const int g = group_of(i - 2), kk = g / (2 * FB), f = (g >> 1) % FB, d = g & 1;
                b8[f][2 * kk + d] = cvq[i & 1];
            } else if (W4_PREP_IN_F16 && (i == FA * FB + 1 || i == FA * FB + 5)) {
                W4_MFMA_F16(acc[fa][fb], a[k2][fa], bh[k2][fb]);       // behind a book-keeping part (control flow joins in front of it): padded
            } else {
                W4_MFMA_F16_HOT(acc[fa][fb], a[k2][fa], bh[k2][fb]);
            }
            if (i < FA * FB || group_of(i) >= 0) {
                __builtin_amdgcn_sched_barrier(0);
                // (the second k-step's pixel fragments and the activations' lo bytes: behind MFMAs 2 FA ... 2 FA + 2 FB - 1, which carry nothing else — requested in front of
                //  MFMA 0 they were 8 LDS reads with the matrix pipe idle)
                if (i >= 2 * FA && i < 2 * FA + FB) bh[1][i - 2 * FA] = *reinterpret_cast<const u32x4*>(smem + ((ba ^ 32u) + (i - 2 * FA) * 4096u));          // chunk 2 + h
                if (i >= 2 * FA + FB && i < 2 * FA + 2 * FB) {
                    const int f = i - 2 * FA - FB;
                    const u32x4 lo8 = *reinterpret_cast<const u32x4*>(smem + ((ba ^ 64u) + f * 4096u));                       // chunk 4 + h
                    b8[f][4] = (int)lo8[0]; b8[f][5] = (int)lo8[1]; b8[f][6] = (int)lo8[2]; b8[f][7] = (int)lo8[3];
                }
                if (i < 2 * FA) {
                    const int f = i >> 1;
                    const u32x4 q = *reinterpret_cast<const u32x4*>(smem + (((i & 1) ? (a8a ^ 16u) : a8a) + f * 4096u));       // chunk 4 + 2 h / 5 + 2 h
                    const int o = (i & 1) * 4;
                    a8[f][o] = (int)q[0]; a8[f][o + 1] = (int)q[1]; a8[f][o + 2] = (int)q[2]; a8[f][o + 3] = (int)q[3];
                }
                if (i < FA * FB && fb == FB - 1)
                    a[1][fa] = *reinterpret_cast<const u32x4*>(smem + ((aa ^ 32u) + fa * 4096u));                          // chunk 2 + h
                if (group_of(i) >= 0) {
                    const int g = group_of(i), kk = g / (2 * FB), f = (g >> 1) % FB, d = g & 1;
                    const float sc = __builtin_bit_cast(float, (unsigned)ebn[f] << 23);
                    s16x2 r = bitcast<s16x2>(b8[f][2 * kk + d]);      // (the destination's stale bytes as the tied operand: both halves are overwritten, no zeroing move)
                    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, bitcast<f16x2>(bh[kk][f][2 * d]), sc, false);
                    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, bitcast<f16x2>(bh[kk][f][2 * d + 1]), sc, true);
                    cvq[i & 1] = bitcast<int>(r);                          // filed into b8 behind MFMA statement i + 2, which names it (see W4_MFMA_F16_HOT_P)
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
#pragma unroll
        for (int i = 0; i <= FA * FB; ++i) step(i);
        ph_stamp(2);
        // the second k-step is 16 MFMAs with nothing else to issue: the slab stream's scalar book-keeping — cursor advance of the slab whose pieces went out in this
        // iteration, then the next slab's address parts (a tile crossing's set-up included: a large block, once per tile) — goes behind its first MFMAs, one part each
        auto fence = [&](auto&& f) __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); f(); __builtin_amdgcn_sched_barrier(0); };
        if (W4_PREP_IN_F16) fence([&]() __attribute__((always_inline)) { if (more) sw_end(); more = sw_begin(); sw_fetch(); });
        step(FA * FB + 1);
        step(FA * FB + 2);
        step(FA * FB + 3);
        step(FA * FB + 4);
        if (W4_PREP_IN_F16) fence([&]() __attribute__((always_inline)) { sw_slab(); });
        if constexpr (W4_STAMPS == 1) { if (ph_crossed) { ph_stamp(7); ph_crossed = false; } }      // slot 7: MFMAs 16-20 + book-keeping of the iterations that cross tiles
        ph_stamp(3);
#pragma unroll
        for (int i = FA * FB + 5; i < NM; ++i) step(i);
    };
    // the wait states a 16-pass MFMA's result needs before a VALU (v_accvgpr_read of the epilogue) may read it: hipcc pads nothing behind an asm statement
    auto mfma_drain = [&]() __attribute__((always_inline)) { asm volatile("s_nop 15\n\ts_nop 7" ::: "memory"); };
    // the epilogue takes a block's 16 values out of the accumulator file WHERE IT CONSUMES THEM.  Left to the compiler the 256 reads are scheduled to the top of the
    // epilogue — at one wave per SIMD its scheduler sees no reason to keep the pressure under 256 — and the allocator spills the slab loop.
    struct AccFile {
        f32x16 (&r)[FA][FB];
        // an empty asm statement that names the block as read-write: hipcc cannot read the block's registers above it (it "changes" there) nor move it above the MFMAs (both
        // volatile), so the 16 v_accvgpr_read_b32 of the copy below are generated — and scheduled, without the boundary pads 32 one-register asm reads cost — where the step
        // consumes them
        __device__ __forceinline__ f32x16 block(int fa, int px) const {
            asm volatile("" : "+a"(r[fa][px]));
            return r[fa][px];
        }
    };
    auto epilogue = [&](int v) __attribute__((always_inline)) {
        int co0, pix0;
        tile_coords(v, co0, pix0);
        mfma_drain();
        // (W4_STAMPS == 2: slots 7-12 = per tile: first request | per step, summed: arithmetic + encode | next step's requests | LDS round + stores 1 | 2 | GroupNorm fold)
        w4_epilogue<SC, RG>(p, AccFile{acc}, co0, pix0, wc, wp, lane, xpose, smem + STAGES * STAGE + NW * XB + wave * PB,
                          [&](int i) __attribute__((always_inline)) { if constexpr (W4_STAMPS == 2) ph_stamp(i); });
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
    if (!W4_PREP_IN_F16 && more) sw_end();                           // (W4_PREP_IN_F16: inside the f16 part)
    __builtin_amdgcn_sched_barrier(0);
    f16_part(0u);
    c_kt = 1;
    ph_stamp(-1);
    ph_on = true;
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
        if (!W4_PREP_IN_F16 && more) sw_end();
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
            unsigned* o = reinterpret_cast<unsigned*>(p.y) + ((size_t)blockIdx.x * NW + wave) * 16;     // (64 bytes per wave)
            o[0] = 0x5157a3b7u; o[1] = ph_slabs; o[2] = ph_tiles;
#pragma unroll
            for (int i = 0; i < 13; ++i) o[3 + i] = ph_sum[i];
        }
    }
}

template <bool SC, bool RG>
static int launch_w4(const ConvArgs& b, int grid, hipStream_t st) {
    static thread_local DeviceOnce attr_once;      // per instantiation, per thread, per device
    if (!attr_once.done()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_dma_w4_kernel<SC, RG>), hipFuncAttributeMaxDynamicSharedMemorySize, w4::LDS);
        if (e != hipSuccess) return mnet_fail(MNET_E_LAUNCH, "hipFuncSetAttribute(dma_w4): %s", hipGetErrorString(e));
        attr_once.mark();
    }
    hipLaunchKernelGGL((conv_dma_w4_kernel<SC, RG>), dim3((unsigned)grid), dim3(w4::NW * 64), w4::LDS, st, b);
    MNET_LAUNCH_CHECK("conv_dma_w4_kernel");
    return MNET_OK;
}

int launch_conv_dma_w4(const ConvArgs& a, hipStream_t st) {
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
    // the epilogue of this tile takes the per-image scale rows one float per lane (32-pixel fragments inside one image) and has the identity / LeakyReLU arms only:
    // anything else runs on the 8-wave tile, same bytes
    if ((a.howo & 31) != 0 || (a.act != MNET_ACT_NONE && a.act != MNET_ACT_LRELU && a.act != MNET_ACT_LRELU_SQRT2)) return launch_conv_dma(a, st, 15);
    const bool sc = a.out_scale || a.post_scale, rg = a.res || a.gn_partial;
    if (sc) return rg ? launch_w4<true, true>(b, grid, st) : launch_w4<true, false>(b, grid, st);
    return rg ? launch_w4<false, true>(b, grid, st) : launch_w4<false, false>(b, grid, st);
}
