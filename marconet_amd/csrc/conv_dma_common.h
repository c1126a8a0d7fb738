// Pieces shared by the two LDS-DMA convolution kernels (conv_igemm_dma.hip: one k-slab of activations per filter tap;
// conv_strip_dma.hip: one activation strip per filter row, reused by its three taps).
#pragma once
#include <cstddef>
#include <cstdlib>
#include "common.h"
#include "conv_args.h"

typedef __attribute__((address_space(3))) void lds_void;
typedef float f32x16 __attribute__((ext_vector_type(16)));

// LDS image of a [rows][128 B] operand tile: 16-byte chunk c of row r lives at chunk c ^ ((r >> 1) & 7)
__device__ __forceinline__ int swz_dma(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

#define VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

// store 8 consecutive output channels co..co+7 of pixel `pix` (DBG: diagnostic variants, see launch_dma_id)
template <int DBG, bool X3 = false>
__device__ __forceinline__ void store8(const ConvArgs& p, const float* v, int pix, int co) {
    if constexpr (X3) {                 // split-half output: 8 hi halves, and 64 bytes further the 8 lo halves
        unsigned char* q = reinterpret_cast<unsigned char*>(p.y) + (size_t)pix * p.cout * 4 + (co >> 5) * 128 + (co & 31) * 2;
        u32x4 hi, lo;
        split8(v, hi, lo);
        stg16(q, hi);
        stg16(q + 64, lo);
        return;
    }
    if constexpr (DBG == 5) {           // DIAGNOSTIC: no stores (values kept live)
        asm volatile("" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
        return;
    }
    if constexpr (DBG == 4) pix &= 255;  // DIAGNOSTIC: every tile stores over tile 0 (L2-resident writes)
    stg16(reinterpret_cast<f16*>(p.y) + (size_t)pix * p.cout + co, Vec<f16>::pack(v));
}

// weight row permutation applied on the DMA source side: LDS row `row` of the weight tile holds output channel
// (tile base +) dma_weight_channel(row), chosen so that the MFMA fragments of a lane together hold 8 CONSECUTIVE channels
//   MF = 16: LDS row 64b + 16f + i holds channel 64b + 32(f/2) + 8(i/4) + 4(f%2) + i%4
//   MF = 32: LDS row 32b + i (D row i = 8q + 4h + e of lane-half h) holds channel 32b + 16h + 4q + e
template <int MF>
__device__ __forceinline__ int dma_weight_channel(int row) {
    if constexpr (MF == 16) {
        const int f = (row >> 4) & 3, i = row & 15;
        return (row & ~63) + ((f >> 1) << 5) + ((i >> 2) << 3) + ((f & 1) << 2) + (i & 3);
    } else {
        const int i = row & 31;
        return (row & ~31) + (((i >> 2) & 1) << 4) + ((i >> 3) << 2) + (i & 3);
    }
}

// fp16+8 (MNET_F16M) launches, MF = 32: LDS row 64b + 32f + i (fragment parity f, D row i = 8q + 4h + e of lane-half h) holds
// channel 64b + 32h + 16f + 4q + e — the 2 x 16 accumulator values of a lane for one pixel (fragments 2b, 2b+1) are then the 32
// CONSECUTIVE channels of one storage block: its scale exponent needs no cross-lane step and the block is stored as 128 contiguous
// bytes by one lane
__device__ __forceinline__ int dma_weight_channel_mx(int row) {
    const int f = (row >> 5) & 1, i = row & 31;
    return (row & ~63) + (((i >> 2) & 1) << 5) + (f << 4) + ((i >> 3) << 2) + (i & 3);
}

// fp16+8 tiles: what the LDS (160 KiB) has left per wave behind the stages for the epilogue's store transposition
#ifndef MNET_MX_XPOSE
#define MNET_MX_XPOSE 1           // A/B build knob (0: every lane stores its own blocks)
#endif
// `lds`: the kernel's own bytes → scratch bytes per wave (4096, 1024 or 0)
template <int NW, bool MX>
constexpr int dma_mx_xpose_bytes(int lds) { return !(MNET_MX_XPOSE && MX) ? 0 : (lds + NW * 4096 <= 160 * 1024 ? 4096 : (lds + NW * 1024 <= 160 * 1024 ? 1024 : 0)); }

// Epilogue of an fp16+8 tile (same passes, same order of operations as dma_epilogue): every lane owns, per pixel, FC/4 whole
// 32-channel blocks.
// `xpose` (this wave's 4 KiB of LDS behind the stages, or nullptr): a lane's block is 8 x 16 bytes of ONE 128-byte line, and lane l + 1 holds
// the next PIXEL — written directly, every store instruction puts 16 bytes into 64 different lines (8 partial writes per line: 16384 write
// requests per 256x256 tile; measured 32 000 cycles per tile, 13 % of a 72-slab tile's time, tools/slab_phases.py --swp).  Through the LDS the
// pieces change lanes — lane l gets piece l % 4 of the block of lane l / 4 + 16 k — so that a store instruction writes 16 x 64 contiguous bytes.
// Same bytes at the same addresses.  The pieces sit at slot 4 L + (c ^ (L / 2 % 4)) of the scratch: conflict-free for the 8-lane groups of
// ds_write_b128 and the 16-lane groups of ds_read_b128 (MI355X_MICROARCH.md, LDS).  One wave's LDS operations execute in order: no waits.
// XL: lanes whose blocks fit the scratch at once — 64 (4 KiB per wave) or 16 (1 KiB per wave: the 64-lane round becomes four 16-lane rounds, one
// read and one store per lane each).
// ConvArgs::gn_partial read from the kernel-argument segment (ConvArgs is the LDS-DMA / strip kernels' only argument, at offset 0)
__device__ __forceinline__ float* kernarg_gn_partial() {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(4))) const char* karg_t;
    karg_t k = (karg_t)__builtin_amdgcn_kernarg_segment_ptr();
    return *(float* const volatile __attribute__((address_space(4)))*)(k + offsetof(ConvArgs, gn_partial));
#else
    return nullptr;
#endif
}

// GN: this instantiation can write the GroupNorm partial sums (ConvArgs::gn_partial).  The software-pipelined tiles are built without it: the extra
// live values of that block moved hipcc's spill choice into their slab loop (two scratch accesses on the hot path, tools/isa_hot_scratch.py); launches
// that ask for the sums take the lock-step form of the same tile (conv_dma_pick), which stays clean.
// ACC: where the accumulator blocks are — `acc.get(fa, px, q)` = value q of block (weight fragment fa, pixel fragment px).  The 8 / 16-wave tiles hand over their VGPR
// array (AccArray); the one-wave-per-SIMD tile (conv_dma_w4.hip) reads its blocks out of the accumulator file where they are consumed.
template <int FA, int FB>
struct AccArray {
    const f32x16 (&r)[FA][FB];
    __device__ __forceinline__ float get(int fa, int px, int q) const { return r[fa][px][q]; }
};
template <int BC, int BP, int WC, int WP, int FC, int FP, int XL = 64, bool GN = true, typename ACC>
__device__ __forceinline__ void dma_epilogue_mx_acc(const ConvArgs& p, const ACC& acc32, int co0, int pix0, int wc, int wp, int lane,
                                                    unsigned char* xpose = nullptr) {
    static_assert(XL == 64 || XL == 16, "scratch of 4 KiB or 1 KiB per wave");
    constexpr int NPX = FP / 2, NB = FC / 4;
    const int h = lane >> 5;
    const int last_pix = p.npix - 1;
    // ConvArgs::gn_partial, read from the kernel-argument segment once per tile (live inside this epilogue only: two SGPRs held across the slab loop moved
    // a spill slot of the 256-VGPR tiles onto their hot path)
    float* gnp = nullptr;
    if constexpr (GN) gnp = kernarg_gn_partial();
#pragma unroll
    for (int px = 0; px < NPX; ++px) {
        const int pixb = pix0 + wp * (BP / WP) + px * 32;                   // first pixel of this wave's 32
        const int pix = pixb + (lane & 31);
        const int n_img = min(pix, last_pix) / p.howo;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int cob = co0 + wc * (BC / WC) + b * 64;                  // first channel of lane-half 0's block
            const int co = cob + h * 32;                                    // first channel of this lane's block
            float v[32];
#pragma unroll
            for (int q = 0; q < 16; ++q) { v[q] = acc32.get(2 * b, px, q) * MNET_SPLIT_WSCALE_INV; v[16 + q] = acc32.get(2 * b + 1, px, q) * MNET_SPLIT_WSCALE_INV; }
            if (!xpose && co >= p.cout) continue;
            // (through the LDS every lane takes part: a lane whose block lies beyond cout works on the last block's parameters instead — loads in
            //  range, no divergent region around 32 live values — and nobody stores its result)
            const int co_l = xpose ? min(co, p.cout - 32) : co;
            {
                if (p.out_scale) {
                    const float* sp = p.out_scale + (size_t)n_img * p.cout + co_l;
#pragma unroll
                    for (int q = 0; q < 32; q += 4) { const f32x4 s4 = *reinterpret_cast<const f32x4*>(sp + q); v[q] *= s4[0]; v[q + 1] *= s4[1]; v[q + 2] *= s4[2]; v[q + 3] *= s4[3]; }
                }
                if (p.bias) {
#pragma unroll
                    for (int q = 0; q < 32; q += 4) { const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + co_l + q); v[q] += b4[0]; v[q + 1] += b4[1]; v[q + 2] += b4[2]; v[q + 3] += b4[3]; }
                }
                if (p.res && pix < p.npix) {
                    const int rpix = p.res_mod > 0 ? pix % p.res_mod : pix;
                    const unsigned char* rb = reinterpret_cast<const unsigned char*>(p.res) + (size_t)rpix * p.cout * 4 + (co_l >> 5) * 128;
                    // the whole block (7 loads) in flight before the first use: decoded piece by piece, every load is a separate exposed latency
                    // (measured: the residual cost 13.7 us per 256x256 tile)
                    u32x4 rh[4], rl[2];
#pragma unroll
                    for (int c = 0; c < 4; ++c) rh[c] = ldg16(rb + c * 16);
                    rl[0] = ldg16(rb + 64); rl[1] = ldg16(rb + 80);          // lo bytes of chunks (0, 2) | (1, 3)
                    unsigned re8 = rb[96];
#pragma unroll
                    for (int c = 0; c < 4; ++c) asm volatile("" : "+v"(rh[c]));
                    asm volatile("" : "+v"(rl[0]), "+v"(rl[1]), "+v"(re8));
                    const float sl = hm_lo_scale((int)re8);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const f16x8 h8 = bitcast<f16x8>(rh[c]);
                        float l[8];
                        hm_decode_lo(u32x2{rl[c & 1][2 * (c >> 1)], rl[c & 1][2 * (c >> 1) + 1]}, sl, l);      // slot of chunk c = hm_lo_slot(c) = 2 (c & 1) + (c >> 1)
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[c * 8 + q] += (float)h8[q] + l[q];
                    }
                }
                act_apply_vec<32, true>(v, p.act);
                if (p.post_scale) {
                    const float* sp = p.post_scale + (size_t)n_img * p.cout + co_l;
#pragma unroll
                    for (int q = 0; q < 32; q += 4) { const f32x4 s4 = *reinterpret_cast<const f32x4*>(sp + q); v[q] *= s4[0]; v[q + 1] *= s4[1]; v[q + 2] *= s4[2]; v[q + 3] *= s4[3]; }
                }
            }
            // GroupNorm statistics of this output (round 5), part 1 — while the 32 values are live: two fp32 sums per lane (the lane holds the 32 channels of
            // ONE group for one pixel).  The fold over the pixels and the store come at the end of the block, where the fewest values are live.
            float gs1 = 0.f, gs2 = 0.f;
            if (GN && gnp) {
#pragma unroll
                for (int q = 0; q < 32; ++q) { gs1 += v[q]; gs2 = fmaf(v[q], v[q], gs2); }
            }
            auto gn_finish = [&]() __attribute__((always_inline)) {
              if constexpr (GN) {
                // part 2: the 32 lanes of a half hold 32 consecutive pixels of one image (ho*wo % 32 == 0) — a fixed xor tree over the half, one 8-byte store.
                // The tree and the fragment (32 pixels aligned to 32) are the same in every tile configuration: batch-invariant bits.  The consumer folds the
                // fragments in fp64 (gn_finalize_frag_kernel): the separate statistics pass over the map (one read of it) is gone.
                if (gnp) {
                    bool ok = pix < p.npix && co < p.cout;
                    if (p.valid_w) {
                        const int ow = p.wo_shift >= 0 ? (pix & (p.wo - 1)) : pix % p.wo;
                        ok = ok && ow < p.valid_w[n_img];
                    }
                    gs1 = ok ? gs1 : 0.f; gs2 = ok ? gs2 : 0.f;
#pragma unroll
                    for (int o = 1; o < 32; o <<= 1) { gs1 += __shfl_xor(gs1, o, 64); gs2 += __shfl_xor(gs2, o, 64); }
                    if ((lane & 31) == 0 && pixb < p.npix && co < p.cout)
                        *reinterpret_cast<f32x2*>(gnp + ((size_t)(pixb >> 5) * (p.cout >> 5) + (co >> 5)) * 2) = f32x2{gs1, gs2};
                }
              }
            };
            if (!xpose && pix >= p.npix) continue;
            f16x8 hh[4];
            float m32 = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int q = 0; q < 8; ++q) { hh[c][q] = (f16)v[c * 8 + q]; m32 = fmaxf(m32, fabsf(v[c * 8 + q])); }
            // max |f16(v)| = f16(max |v|) (round-to-nearest is monotonic): one conversion of the maximum instead of 32 of the halves — hipcc folds the
            // abs of the other form into a second v_cvt_f32_f16 per value, beside the one the lo residual needs
            const float m = (float)(f16)m32;
            const int e8 = hm_e8_of(m);
            u32x2 lo[4];
            if (!xpose) {
                unsigned char* yb = reinterpret_cast<unsigned char*>(p.y) + (size_t)pix * p.cout * 4 + (co >> 5) * 128;
#pragma unroll
                for (int c = 0; c < 4; ++c) { stg16(yb + c * 16, bitcast<u32x4>(hh[c])); lo[c] = hm_encode_lo_ref(v + c * 8, hh[c], e8); }
                // lo bytes in the order 0-7,16-23 | 8-15,24-31 (slot of chunk c = hm_lo_slot(c))
                stg16(yb + 64, u32x4{lo[0][0], lo[0][1], lo[2][0], lo[2][1]});
                stg16(yb + 80, u32x4{lo[1][0], lo[1][1], lo[3][0], lo[3][1]});
                stg16(yb + 96, u32x4{(unsigned)e8, 0u, 0u, 0u});
                stg16(yb + 112, u32x4{0u, 0u, 0u, 0u});
                gn_finish();
                continue;
            }
            // (hm_encode_lo_ref / act_apply_vec, not the mixed-precision VALU forms of common.h: with them this epilogue executes ~25 % fewer VALU
            //  instructions and the tile runs 0 ... -1 % — same-box A/B, profiles/r5c_hm_fast_ab.txt: the epilogue waits for its parameter loads and its
            //  stores, not for the VALU.  The streaming kernels keep the fast forms: +1 ... +2.5 % there.)
            // ---- through the LDS: two rounds of 4 pieces per lane (the hi halves, then lo bytes | lo bytes | scale | padding)
            unsigned L = (unsigned)lane;
            asm volatile("" : "+v"(L));                                     // the scratch addresses are re-derived here: hoisted out of the tile loop they would
            const unsigned wsw = (L >> 1) & 3u, j = L & 3u;                 // live through the k-loop, which has no register to spare (256 allocated)
#pragma unroll
            for (int c = 0; c < 4; ++c) lo[c] = hm_encode_lo_ref(v + c * 8, hh[c], e8);
#pragma unroll 1
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
                if constexpr (XL == 64) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) *reinterpret_cast<u32x4*>(xpose + ((4u * L + ((unsigned)c ^ wsw)) << 4)) = pc[c];
                    u32x4 piece[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const unsigned P = (L >> 2) + 16u * k;              // the lane whose block this lane helps to write
                        piece[k] = *reinterpret_cast<const u32x4*>(xpose + ((4u * P + (j ^ ((P >> 1) & 3u))) << 4));
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(piece[k]));   // all four reads in flight (otherwise each is sunk into its store's predicated block: read, wait, store, four times)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const unsigned P = (L >> 2) + 16u * k;
                        const int ppix = pixb + (int)(P & 31u), pco = cob + (int)(P >> 5) * 32;
                        if (ppix < p.npix && pco < p.cout)
                            stg16(reinterpret_cast<unsigned char*>(p.y) + (size_t)ppix * p.cout * 4 + (pco >> 5) * 128 + half * 64 + j * 16, piece[k]);
                    }
                } else {
                    const unsigned Lq = L & 15u;
#pragma unroll 1
                    for (unsigned r = 0; r < 4; ++r) {                       // the blocks of lanes 16 r .. 16 r + 15
                        if ((L >> 4) == r) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) *reinterpret_cast<u32x4*>(xpose + ((4u * Lq + ((unsigned)c ^ wsw)) << 4)) = pc[c];
                        }
                        const unsigned Pq = L >> 2, P = 16u * r + Pq;
                        const u32x4 piece = *reinterpret_cast<const u32x4*>(xpose + ((4u * Pq + (j ^ ((Pq >> 1) & 3u))) << 4));
                        const int ppix = pixb + (int)(P & 31u), pco = cob + (int)(P >> 5) * 32;
                        if (ppix < p.npix && pco < p.cout)
                            stg16(reinterpret_cast<unsigned char*>(p.y) + (size_t)ppix * p.cout * 4 + (pco >> 5) * 128 + half * 64 + j * 16, piece);
                    }
                }
            }
            gn_finish();
        }
    }
}

template <int BC, int BP, int WC, int WP, int FC, int FP, int XL = 64, bool GN = true>
__device__ __forceinline__ void dma_epilogue_mx(const ConvArgs& p, const f32x16 (&acc32)[FC / 2][FP / 2], int co0, int pix0, int wc, int wp, int lane,
                                                unsigned char* xpose = nullptr) {
    dma_epilogue_mx_acc<BC, BP, WC, WP, FC, FP, XL, GN>(p, AccArray<FC / 2, FP / 2>{acc32}, co0, pix0, wc, wp, lane, xpose);
}

// Epilogue of one (cout tile co0, pixel tile pix0): identical math to conv_igemm.hip.  Every lane owns NG groups of 8
// consecutive output channels for each of its NPX pixels; whole-register-set passes, each behind ONE wave-uniform branch
// (out_scale / bias / residual / activation / post_scale), then 16-byte stores.
template <int BC, int BP, int WC, int WP, int MF, int DBG, int FC, int FP, bool X3 = false>
__device__ __forceinline__ void dma_epilogue(const ConvArgs& p, const f32x4 (&acc)[FC][FP],
                                             const f32x16 (&acc32)[MF == 32 ? FC / 2 : 1][MF == 32 ? FP / 2 : 1],
                                             int co0, int pix0, int wc, int wp, int lane) {
    const int l16 = lane & 15, g = lane >> 4;
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
                if constexpr (X3) ev[px][gi][q] *= MNET_SPLIT_WSCALE_INV;     // weights hold 256*W: exact power-of-two rescale
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
                if constexpr (X3) {
                    const f16* rq = rs + (size_t)rpix * p.cout * 2 + (eco[gi] >> 5) * 64 + (eco[gi] & 31);
                    const f16x8 h8 = bitcast<f16x8>(ldg16(rq)), l8 = bitcast<f16x8>(ldg16(rq + 32));
#pragma unroll
                    for (int q = 0; q < 8; ++q) ev[px][gi][q] += (float)h8[q] + (float)l8[q];
                } else {
                    const f16x8 r8 = bitcast<f16x8>(ldg16(rs + (size_t)rpix * p.cout + eco[gi]));
#pragma unroll
                    for (int q = 0; q < 8; ++q) ev[px][gi][q] += (float)r8[q];
                }
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
            if (eco[gi] < p.cout) store8<DBG, X3>(p, ev[px][gi], epix[px], eco[gi]);
    }
}

// persistent grid: one workgroup per CU (every configuration needs > 80 KiB of LDS)
static inline int dma_grid_limit() {
    static thread_local int ncu_of[256] = {};            // per device: a process may drive several GPUs
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    int& ncu = ncu_of[dev & 255];
    if (ncu == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        const char* e = getenv("MNET_DMA_GRID");        // experiment knob: persistent workgroups (= CUs used) of the LDS-DMA kernels
        if (e && atoi(e) > 0 && atoi(e) < n) n = atoi(e);
        ncu = n;
    }
    return ncu;
}
