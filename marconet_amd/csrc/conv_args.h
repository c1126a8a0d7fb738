// Kernel-argument block shared by the two implicit-GEMM conv kernels (conv_igemm.hip, conv_igemm_dma.hip).
#pragma once
#include "common.h"

struct ConvArgs {
    const void* x0; const void* x1; const void* wgt; void* y; const void* res;
    const float* in_scale; const float* in_shift; const float* out_scale; const float* bias;
    const float* post_scale; const int* valid_w;
    int c0, c1, cin;
    int n, h, w, ho, wo, cout;
    int kh, kw, sh, sw, ph, pw;
    int K, npix, howo;
    int in_swish, act, res_mod;
    int ktiles, tilesC, ntiles;
    int split;               // MNET_F16X2 launch: c0 / c1 / cin / K are PHYSICAL (f16 view with twice the channels: per 32-channel block 32 hi
                             // halves then 32 lo halves); cout stays logical; the epilogue multiplies the accumulator by 2^-8 and stores hi/lo
    int mx_fetch_pad;        // fp16+8 launches, A/B knob (env MNET_MX_FETCH_PAD=1): also fetch the padding chunk 7 of every activation block (whole 128-byte lines)
    int one_tile_per_wg;     // A/B knob (MNET_CONV_ALGO_FLAG_ONE_TILE): grid = #tiles instead of a persistent grid
    int x1_center;           // MNET_CONV_ALGO_FLAG_X1_CENTER: the second source is walked at the centre tap only (LDS-DMA kernels: ktiles = taps * c0 / 64 + c1 / 64,
    int center_tap, center_tpx;   //   physical channels); its tap index kh/2 * kw + kw/2 and input-pixel offset kh/2 * w + kw/2 relative to tap 0
    int howo_shift, wo_shift;     // log2 of ho * wo / of wo when they are powers of two, else -1 (LDS-DMA kernels: the per-tile set-up divides by them, set by the launcher)
    // (LAST, and read from the kernel-argument segment at its point of use: the 256-VGPR tiles' register allocation is sensitive to the layout of this block —
    //  the same field placed before howo_shift put a scratch reload into the software-pipelined tile's slab loop, tools/isa_hot_scratch.py)
    float* gn_partial;       // mnet_conv_desc.gn_partial: per (32-pixel fragment, 32-channel group) sum / sum of squares of the output, written by dma_epilogue_mx
};

// conv_igemm_dma.hip
bool conv_dma_eligible(const ConvArgs& a, int dtype);
int conv_dma_pick(const ConvArgs& a);                           // tile configuration id AUTO would use
int launch_conv_dma(const ConvArgs& a, hipStream_t st, int cfg);   // cfg < 0: conv_dma_pick

// conv_dma_w4.hip (fp16+8 256x256 tile, one wave per SIMD, accumulators in the accumulator file: fp16+8 LDS-DMA id 16)
int launch_conv_dma_w4(const ConvArgs& a, hipStream_t st);

// conv_strip_dma.hip (3x3 / stride 1: one activation strip per filter row)
int conv_strip_pick(const ConvArgs& a, int dtype, bool explicit_request);   // strip configuration id, or -1 when not eligible / not preferred
int launch_conv_strip(const ConvArgs& a, hipStream_t st, int cfg);

// conv_skinny.hip (fp32 1x1 over <= 512 pixels: the TextViT linears of a small batch; bit-identical to the general kernel)
bool conv_skinny_eligible(const ConvArgs& a, int dtype);
int launch_conv_skinny(const ConvArgs& a, hipStream_t st);
int launch_conv_skinny_splitk(const ConvArgs& a, int dtype, int ksplit, float* ws, hipStream_t st);   // filter == stride, split-K + ordered reduce
