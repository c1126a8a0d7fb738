/*
 * marconet_hip.h — C-ABI of libmarconet_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * MARCONet test_sr.py / test_w.py inference forward.
 *
 * Boundary rules:
 *   - plain C: raw device pointers, ints, a dtype enum and a hipStream_t (passed as void*); no torch types
 *   - every buffer (inputs, outputs, workspaces) is owned and allocated by the caller
 *   - every call enqueues work on the given stream and returns; no call synchronises or allocates
 *   - return value: 0 = ok, negative = MNET_E_* ; mnet_last_error() returns a thread-local message
 *   - re-entrant, no mutable global state
 *
 * The reference has no native layer of its own: its operator layer is ATen + one third-party CUDA op
 * (basicsr.ops.fused_act, models/networks.py:10).  Each entry point below therefore cites the reference
 * *call sites* (file:line under /root/reference) whose arithmetic it replaces.
 *
 * Tensor layout: activations are NHWC ("pixel-major, channel-minor") in MNET_F32 or MNET_F16; conv
 * weights are [Cout][KH][KW][Cin] in the same dtype.  All statistics, softmax, demodulation and
 * epilogue math are fp32 (fp64 for the GroupNorm / AdaIN sums).
 */
#ifndef MARCONET_HIP_H
#define MARCONET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MNET_ABI_VERSION 4

/* MNET_F16X2 ("split half", the storage of the fp16x3 precision mode): every logical element is a pair of halves (hi, lo),
 * value = float(hi) + float(lo), hi = f16(v), lo = f16(v - hi): ~22 significant bits at fp16 MFMA rates (x*w is evaluated as
 * hi*hi + hi*lo + lo*hi, fp32 accumulate) — the throughput mode that meets the 1e-3 parity bar.  NHWC layout [.., C] with
 * C % 32 == 0 and a 128-byte aligned base: per pixel 4*C bytes in blocks of 32 channels, 64 bytes of hi followed by 64 bytes
 * of lo.  Conv weights [cout][kh][kw][cin] use the same blocking along cin and hold hi/lo of 256*W (the conv multiplies its
 * accumulator by 2^-8).  Entry points that accept it say so; sizes/strides are always given in LOGICAL elements. */
/* MNET_F16M ("fp16+8", the storage of the fp16x2 precision mode): the same 4 bytes per logical element and the same blocking
 * (128 bytes per (pixel, 32-channel block), C % 32 == 0, 128-byte aligned base), but only the hi part is a half; the lo part is an
 * OCP e4m3 byte under one E8M0 scale per block, so that x*w = hi*hi on the f16 MFMA + (w_lo8*x_hi8 + w_hi8*x_lo8) as ONE block-scaled
 * fp8 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, twice the f16 rate): 2 MFMA units per product instead of 3, ~16 significant bits.
 *   activations  bytes 0-63 hi = f16(v) of channels 0..31 | 64-95 lo8 = e4m3((v - hi) * 2^11 / s), channel order 0-7,16-23,8-15,24-31 |
 *                byte 96 E: s = 2^(E-127) = 2^(floor(log2 max|hi|) - 7) | bytes 97-127 zero
 *   conv weights bytes 0-63 hi = f16(256 W) | 64-79 lo8 of channels 0-7,16-23 | 80-95 hi8 = e4m3(hi / s) of the same | 96-111 lo8 of
 *                8-15,24-31 | 112-127 hi8 of the same; s per OUTPUT channel; after the cout*kh*kw*cin elements one byte per output
 *                channel (E8M0 of s * 2^-11), i.e. a packed weight tensor is cout*kh*kw*cin*4 + cout bytes (mnet_pack_weights). */
typedef enum { MNET_F32 = 0, MNET_F16 = 1, MNET_F16X2 = 2, MNET_F16M = 3 } mnet_dtype;

typedef enum {
    MNET_ACT_NONE = 0,
    MNET_ACT_RELU = 1,          /* models/resnet.py:16,24,29                                   */
    MNET_ACT_LRELU = 2,         /* nn.LeakyReLU(0.2), models/networks.py:337..404              */
    MNET_ACT_LRELU_SQRT2 = 3,   /* basicsr fused_leaky_relu: sqrt(2)*lrelu_0.2, networks.py:195,241 */
    MNET_ACT_TANH = 4,          /* networks.py:321,375                                         */
    MNET_ACT_GELU = 5,          /* nn.GELU() exact erf, models/textvit_arch.py:49,88           */
    MNET_ACT_SIGMOID = 6        /* textvit_arch.py:51                                          */
} mnet_act;

enum {
    MNET_OK = 0,
    MNET_E_ARG = -1,        /* bad shape / null pointer / unsupported combination */
    MNET_E_ALIGN = -2,      /* pointer or channel count not aligned as required   */
    MNET_E_LAUNCH = -3      /* hipLaunchKernel / hipFuncSetAttribute failed       */
};

const char* mnet_last_error(void);
int mnet_abi_version(void);

/* ---------------------------------------------------------------------------------------------
 * mnet_conv2d_nhwc — implicit-GEMM convolution on the matrix cores (MFMA 16x16x32 f16 / 16x16x4 f32),
 * with fused prologue and epilogue.  One entry point covers:
 *   K1/K2/K3 dense 3x3 / 1x1 / strided convs  models/resnet.py:4-9,21-30 ; models/networks.py:336-405,501-505
 *   K4  modulated conv (activation-side modulation: y = demod[n,o] * conv(W*scale, x * s[n,i]))
 *                                              models/networks.py:281-302 (ModulatedConv2d.forward)
 *   K5  fused bias + LeakyReLU*sqrt2 epilogue  basicsr fused_act, call sites networks.py:195,241-245
 *   K7  every nn.Linear of TextViT / style MLP (1x1 conv over a [1,1,M,K] map)
 *                                              models/textvit_arch.py:34,49-60,84-107 ; networks.py:188-198
 *       the 8x8/stride-8 patch embedding is the same kernel with kh=kw=8 (textvit_arch.py:32-35)
 *   K17 channel concat without materialising it (two sources)   networks.py:415-416
 *   K11 GroupNorm+swish as an input-side affine (scale/shift per (n,c)) + swish   networks.py:508-513
 *
 * y[n,oh,ow,o] = post_scale[n,o] * act( out_scale[n,o] * sum_{r,s,i} W[o,r,s,i] * X'[n, oh*sh-ph+r, ow*sw-pw+s, i]
 *                                       + bias[o] + residual[n,oh,ow,o] )
 *   X' = X                                   if in_scale == NULL
 *   X' = f(X * in_scale[n,i] + in_shift[n,i]),  f = swish if in_swish else identity
 *   X  = channel-concat(x0[..c0], x1[..c1]);  out-of-image taps (and columns >= valid_w[n]) are zero
 *
 * Requirements: (c0+c1) % 8 == 0, c0 % 8 == 0, cout % 4 == 0, all pointers 16-byte aligned.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t dtype;              /* mnet_dtype of x0, x1, wgt, residual, y                         */
    const void* x0; int32_t c0; /* NHWC [n,h,w,c0]                                                */
    const void* x1; int32_t c1; /* NHWC [n,h,w,c1] or NULL/0                                      */
    int32_t n, h, w;
    const void* wgt;            /* [cout][kh][kw][c0+c1]                                          */
    int32_t cout, kh, kw, stride_h, stride_w, pad_h, pad_w;
    int32_t ho, wo;             /* output spatial size (caller computes)                          */
    const float* in_scale;      /* [n][cin] or NULL                                               */
    const float* in_shift;      /* [n][cin] or NULL (treated as 0)                                */
    int32_t in_swish;
    const int32_t* valid_w;     /* [n] or NULL: input columns >= valid_w[n] read as zero          */
    const float* out_scale;     /* [n][cout] or NULL (demodulation)                               */
    const float* bias;          /* [cout] fp32 or NULL                                            */
    const void* residual;       /* NHWC like y, or NULL                                           */
    int32_t res_mod;            /* 0, or residual pixel index = pixel % res_mod (pos-emb add)     */
    int32_t act;                /* mnet_act, applied after bias+residual                          */
    const float* post_scale;    /* [n][cout] or NULL: multiplied AFTER act — lets a StyledConv emit its
                                 * output already modulated by the NEXT layer's style (networks.py:283-284) */
    void* y;                    /* NHWC [n,ho,wo,cout]                                            */
    float* gn_partial;          /* NULL, or (ABI v4) GroupNorm partial sums of the OUTPUT, written by the epilogue that holds the values anyway:
                                 * [n*ho*wo / 32][cout / 32][2] fp32 = (sum, sum of squares) over 32 consecutive output pixels x the 32 channels of a
                                 * group (networks.py:487-493: GroupNorm(C/32 groups)), of the values as they leave the epilogue (after bias /
                                 * activation / post_scale, before the storage rounding); pixels at columns >= valid_w[n] are left out.  Fixed
                                 * reduction tree, independent of the tile configuration: the same bits for every launch size.  Removes the separate
                                 * statistics pass over the map (mnet_groupnorm_affine_from_partial folds them).  Needs an MNET_F16M launch the LDS-DMA /
                                 * strip kernels take, stride 1, cout % 32 == 0 and ho*wo % 32 == 0 — MNET_E_ARG otherwise (nothing is enqueued). */
} mnet_conv_desc;

int mnet_conv2d_nhwc(const mnet_conv_desc* d, void* stream);

/* same with an explicit kernel choice (A/B measurements, tests): AUTO picks the LDS-DMA kernel when the launch is
 * eligible (f16, (c0+c1) % 64 == 0, c0 % 64 == 0, cout >= 64, cout % 8 == 0, no in_scale, act <= LRELU_SQRT2) and the
 * register-staged one otherwise; MNET_CONV_ALGO_DMA_CFG0 + id pins one LDS-DMA tile configuration
 * (cout x pixel tile, waves, LDS stages):
 *   id 0: 256x256 16w 2st   1: 256x128 8w 3st   2: 128x256 8w 3st   3: 64x256 8w 3st   4: 128x512 16w 2st
 *      5: 64x512 8w 2st     6: 256x256 8w 2st (128x64 per wave)      — all on v_mfma_f32_16x16x32_f16 walking k in the
 *      same order (64-channel slice outer, filter tap inner): the same bits for every launch size (batch-invariant);
 *      the register-staged kernel walks k tap-outer (same products, fp32 partial sums associated differently)
 *   id 10: 128x128 8w 4st (3 slabs in flight: a lone small tile per CU exposes the load latency) — AUTO's choice when ids 1 / 2 would give fewer than 200 workgroups (one strip at a time)
 *   id 8 / 9: ids 0 / 4 with the next slab's DMA pieces issued between the two half slabs instead of right after the slab barrier
 *      (AUTO's choice for >= 65536 output pixels: +2.8 % on the 256x256 tile; same MFMA sequence, same bits)
 *   id 7: id 0 on v_mfma_f32_32x32x16_f16 (experimental)
 *   MNET_F16X2 launches: ids 0-6 and 10 as above on the split-half form, id 7 = 128x512 8w (64x128 per wave), ids 8 / 9 = ids 6 / 7
 *      with the DMA pieces issued after the first multiply group, ids 11 / 12 = ids 8 / 9 with the LDS reads placed by scheduling
 *      hints; AUTO takes the 8-wave tiles 11 (cout >= 256) / 9 (cout 128) for >= 65536 output pixels
 *   MNET_F16M launches: ids 0-5, 10 = the same tile shapes on v_mfma_f32_32x32x16_f16 + v_mfma_scale_f32_32x32x64_f8f6f4, id 6 = 256x256
 *      8w (128x64 per wave), id 7 / 8 = 128x512 8w (64x128 / 128x64 per wave), ids 11 / 12 / 13 = ids 6 / 8 / 5 with the LDS reads
 *      placed by scheduling hints; ids 15 / 9 (round 4) = ids 6 / 8 with the slab loop software-pipelined across the slab barrier (the scaled
 *      MFMAs of slab s-1 and one DMA piece of slab s+1 behind each, then the f16 MFMAs of slab s); AUTO takes 15 (cout >= 256) / 8 (cout 128) /
 *      13 (cout 64) for >= 65536 output pixels; every id runs the same MFMA sequence per output (same bytes whatever the launch size selects);
 *      id 14: DIAGNOSTIC build (per-phase cycle sums written over the output, tools/slab_phases.py), refused unless MNET_ALLOW_DIAGNOSTIC_KERNELS=1
 *   MNET_F16 ids 11-15 ONLY: diagnostic builds used by tools/wg_timeline.py and tools/conv_bench.py; they produce WRONG results and
 *      are refused (MNET_E_ARG) unless the process sets MNET_ALLOW_DIAGNOSTIC_KERNELS=1 */
enum { MNET_CONV_ALGO_AUTO = 0, MNET_CONV_ALGO_REG_STAGED = 1, MNET_CONV_ALGO_LDS_DMA = 2,
       MNET_CONV_ALGO_SKINNY = 3 /* fp32 1x1 over <= 512 pixels straight from global memory (the TextViT linears of a small batch;
                                  * 16 x 64 tiles).  AUTO uses it when eligible; bit-identical to REG_STAGED. */,
       MNET_CONV_ALGO_DMA_CFG0 = 16,
       MNET_CONV_ALGO_STRIP_CFG0 = 32 /* + id: the 3x3 "strip" LDS-DMA kernel (one activation strip per filter row shared by its
                                        * three taps; id 0: 256x256 tile, id 1: 64x512 tile).  Eligible: 3x3/stride 1/pad 1, one
                                        * source, cout >= 256 (id 0) or < 128 (id 1), >= 65536 output pixels, whole-row tiles.
                                        * AUTO uses id 1 when eligible (id 0 measured neutral: explicit request only; MNET_F16X2: id 1 only; MNET_F16M: id 1, and id 0 as an
                                        * 8-wave 256x256 tile on explicit request — measured -3 ... -5 %).  Same k order and MFMA as the LDS-DMA kernel → identical bits. */,
       MNET_CONV_ALGO_DMA_CFG16 = 64 /* + (id - 16): LDS-DMA tile configurations 16.. (the ids 0..15 above are full):
                                       *   id 16: 256x256 8w 2st (128x64 per wave) with both half slabs' operand fragments requested up front and the
                                       *          next slab's DMA pieces issued between the halves — AUTO's f16 choice for cout >= 256, >= 65536 pixels
                                       *   id 17: the same form of the 128x512 tile (64x128 per wave)
                                       * same MFMA sequence as ids 0-6: identical bits.
                                       *   MNET_F16M launches, id 16 (round 6): the 256x256 tile with ONE wave per SIMD — 4 waves x 128x128 outputs, the accumulators
                                       *          in the accumulator register file (conv_dma_w4.hip) — AUTO's fp16+8 choice for cout >= 256, >= 65536 pixels; writes
                                       *          mnet_conv_desc.gn_partial itself; same MFMA sequence per output as the fp16+8 ids 0-15: identical bytes.  A launch
                                       *          it is not built for — an activation other than NONE / LRELU / LRELU_SQRT2, ho * wo not a multiple of 32, more than
                                       *          512 k-slabs — runs on id 15 (the 8-wave tile) under this id, same bytes */,
       MNET_CONV_ALGO_FLAG_ONE_TILE = 256 /* OR-ed in: LDS-DMA kernel launched with one workgroup per tile instead of its
                                            * persistent grid (A/B measurements only; same results) */,
       MNET_CONV_ALGO_FLAG_X1_CENTER = 512 /* OR-ed in (round 4): the SECOND source x1 contributes through the filter's CENTRE tap only —
                                             * y = conv_khxkw(x0; W[:, :, :, :c0]) + conv_1x1(x1; W[:, kh/2, kw/2, c0:]) in ONE k-loop: the 1x1
                                             * skip convolution of ResTextBlockV2 (models/networks.py:504-505,514-515: h + conv_out(x)) folded
                                             * into its conv2 as extra K instead of a separate launch + a residual read.  wgt is the ordinary
                                             * [cout][kh][kw][c0+c1] tensor whose x1 part is only read at the centre tap; x1 is NHWC
                                             * [n,h,w,c1] like x0 (stride 1 launches).  LDS-DMA kernel only (MNET_E_ARG when the launch is
                                             * not eligible for it); mnet_conv2d_flops still counts x1 at every tap. */ };
int mnet_conv2d_nhwc_ex(const mnet_conv_desc* d, int32_t algo, void* stream);

/* split-K form of the skinny kernel for a "patchify" conv (filter == stride, no padding; fp32; <= 512 output pixels) — the
 * TextViT patch embedding of a small batch (8x8 / stride 8 over [B,8,512,512]: K = 32768, 64 tokens per strip: one workgroup
 * per 16 channels would stream 2 MiB of weights alone).  K is cut into `ksplit` contiguous slices (a multiple of kh; each a
 * whole number of 16-float steps of one filter row), raw fp32 partial sums go to workspace [ksplit][n*ho*wo][cout] and a
 * second launch folds them in slice order and applies the epilogue — deterministic; differs from mnet_conv2d_nhwc only in
 * the association of the fp32 sum. */
int mnet_conv2d_splitk(const mnet_conv_desc* d, int32_t ksplit, float* workspace, void* stream);

/* which kernel `algo` resolves to for this launch, without launching: MNET_CONV_ALGO_REG_STAGED, MNET_CONV_ALGO_SKINNY,
 * MNET_CONV_ALGO_DMA_CFG0 + id or MNET_CONV_ALGO_STRIP_CFG0 + id; negative MNET_E_* on invalid arguments (bench.py buckets
 * its timings by this) */
int mnet_conv2d_plan(const mnet_conv_desc* d, int32_t algo);

/* 2*MACs of the launch described by d (for roofline accounting in bench.py) */
double mnet_conv2d_flops(const mnet_conv_desc* d);

/* ---------------------------------------------------------------------------------------------
 * layout changes at the module boundary (reference tensors are NCHW fp32, models/networks.py:42,61,411)
 * ------------------------------------------------------------------------------------------- */
/* src fp32 NCHW [n,c,h,w] -> dst NHWC [n,h,w,c_ld] (channels c..c_ld-1 written as zero) */
int mnet_nchw_to_nhwc(const float* src, void* dst, int32_t dst_dtype, int32_t n, int32_t c, int32_t h,
                      int32_t w, int32_t c_ld, void* stream);
/* src NHWC [n,h,w,c_ld] (first c channels used) -> dst fp32 NCHW [n,c,h,w] */
int mnet_nhwc_to_nchw(const void* src, int32_t src_dtype, float* dst, int32_t n, int32_t c, int32_t h,
                      int32_t w, int32_t c_ld, void* stream);

/* K6: bilinear x2, align_corners=False (== polyphase [1/4,3/4] with edge clamp), NHWC, c % 4 == 0 (f32)
 * / c % 8 == 0 (f16).  nn.Upsample / F.interpolate at networks.py:268,318,360,370,415,416 */
int mnet_upsample2x_nhwc(const void* src, void* dst, int32_t dtype, int32_t n, int32_t h, int32_t w,
                         int32_t c, void* stream);
/* same, output multiplied by scale[n][c] (fp32, may be NULL): the style modulation of an up-sampling StyledConv
 * applied once per element here instead of once per filter tap inside the conv (networks.py:283-296) */
int mnet_upsample2x_scale_nhwc(const void* src, void* dst, int32_t dtype, int32_t n, int32_t h, int32_t w,
                               int32_t c, const float* scale, void* stream);
/* the same with the output in another storage type: dst_dtype == dtype, or MNET_F16 for an MNET_F16X2 / MNET_F16M input (the batched
 * driver's image-only generator level, models/networks.py:161-164, reads the 64-px prior in the mode's storage and continues in plain
 * f16 — no separate mnet_convert pass over that map) */
int mnet_upsample2x_convert_nhwc(const void* src, int32_t dtype, void* dst, int32_t dst_dtype, int32_t n, int32_t h, int32_t w,
                                 int32_t c, const float* scale, void* stream);

/* K11 apply: y = f(x * scale[n,c] + shift[n,c]), f = swish if swish else identity (shift may be NULL).
 * GroupNorm-normalise + swish once per element (networks.py:508-509,511-512) — the conv prologue form of the same
 * math costs 9 taps x Cout/128 tiles of redundant expf.  x, y NHWC [n, hw, c]; in-place (y == x) allowed. */
int mnet_affine_act_nhwc(const void* x, void* y, int32_t dtype, int32_t n, int32_t hw, int32_t c,
                         const float* scale, const float* shift, int32_t swish, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K11 GroupNorm statistics (groups of 32 channels, eps, affine), networks.py:487-490 — produces the
 * per-(n,channel) affine consumed by mnet_conv2d_nhwc's prologue:
 *   scale[n,c] = rstd[n,g]*gamma[c] ; shift[n,c] = beta[c] - mean[n,g]*rstd[n,g]*gamma[c]
 * Statistics are over h x valid_w[n] x 32 channels (biased variance), accumulated in fp64.
 * partial: caller workspace of  n * slices * (c/32) * 2  doubles.
 * ------------------------------------------------------------------------------------------- */
int mnet_groupnorm_affine(const void* x, int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c,
                          const int32_t* valid_w, const float* gamma, const float* beta, float eps,
                          double* partial, int32_t slices, float* scale, float* shift, void* stream);
/* the same affine from the partial sums a convolution's epilogue wrote (mnet_conv_desc.gn_partial: [n*h*w/32][c/32][2] fp32): fp64 fold in a fixed
 * order, count = h * min(valid_w[n], w) * 32 per group.  One wave per (image, group): no pass over the map. */
int mnet_groupnorm_affine_from_partial(const float* partial, int32_t n, int32_t h, int32_t w, int32_t c, const int32_t* valid_w,
                                       const float* gamma, const float* beta, float eps, float* scale, float* shift, void* stream);

/* ---------------------------------------------------------------------------------------------
 * per-glyph prior transform of TSPSRNet (networks.py:421-449 and :455-482)
 * Glyph g belongs to image g_img[g]; it covers feature columns [g_x1[g], g_x1[g]+g_w[g]) and prior
 * columns [g_y1[g], g_y1[g]+g_w[g]);  S = prior size (32 or 64) = feature height = max window width.
 * ------------------------------------------------------------------------------------------- */
/* K12+K13 crop + AdaIN + concat:  out [G,S,S,2C]
 *   out[g,y,x,0:C]  = (prior[g,y,y1+x,:] - mean_p)/std_p * std_f + mean_f   (unbiased var + 1e-5, :518-533)
 *   out[g,y,x,C:2C] = feat[img,y,x1+x,:]                                    for x < g_w[g]; 0 beyond */
int mnet_adain_crop_concat(const void* prior, const void* feat, void* out, int32_t dtype, int32_t G,
                           int32_t S, int32_t C, int32_t feat_w, const int32_t* g_img,
                           const int32_t* g_x1, const int32_t* g_y1, const int32_t* g_w, void* stream);

/* same, and additionally the GroupNorm (2C/32 groups of 32 channels, biased variance, eps) affine of the [G,S,S,2C] OUTPUT
 * over each glyph's window — norm1 of conv_32_fuse / conv_64_fuse (networks.py:508) — in closed form from the AdaIN
 * statistics the kernel has anyway (no further pass over the tensor):  scale[g,c] = rstd[g,grp]*gamma[c],
 * shift[g,c] = beta[c] - mean[g,grp]*rstd[g,grp]*gamma[c];  gamma, beta fp32 [2C];  scale, shift fp32 [G][2C] */
int mnet_adain_crop_concat_gn(const void* prior, const void* feat, void* out, int32_t dtype, int32_t G,
                              int32_t S, int32_t C, int32_t feat_w, const int32_t* g_img,
                              const int32_t* g_x1, const int32_t* g_y1, const int32_t* g_w,
                              const float* gamma, const float* beta, float eps, float* scale, float* shift, void* stream);

/* the same result from three launches that spread each glyph over `slices` workgroups — for FEW glyphs (one strip at a time:
 * the one-workgroup-per-glyph kernel above is then a 380 us latency chain).  gamma/beta/scale/shift may all be NULL (no
 * GroupNorm affine).  Scratch owned by the caller: partial fp64 [G][slices][C][4], stat fp32 [G][4][C].  Agrees with the
 * fused kernel up to the association of the fp64 statistic sums. */
int mnet_adain_crop_concat_split(const void* prior, const void* feat, void* out, int32_t dtype, int32_t G,
                                 int32_t S, int32_t C, int32_t feat_w, const int32_t* g_img,
                                 const int32_t* g_x1, const int32_t* g_y1, const int32_t* g_w,
                                 const float* gamma, const float* beta, float eps, float* scale, float* shift,
                                 double* partial, float* stat, int32_t slices, void* stream);

/* K13 ordered scatter:  out[b,y,x,:] = feat + (feat*scale[g,y,x-x1,:] + shift[g,y,x-x1,:]) for the LAST
 * glyph g of image b whose window covers x (later glyph overwrites earlier, :448,481); out = feat where
 * no window covers x.  Glyphs of image b are g_start[b] .. g_start[b+1]-1. */
int mnet_glyph_scatter_affine(const void* feat, const void* scale, const void* shift, void* out,
                              int32_t dtype, int32_t B, int32_t S, int32_t C, int32_t feat_w,
                              const int32_t* g_start, const int32_t* g_x1, const int32_t* g_w,
                              void* stream);

/* ---------------------------------------------------------------------------------------------
 * TextViT pieces (fp32 only), models/textvit_arch.py
 * ------------------------------------------------------------------------------------------- */
/* K9 LayerNorm over the last dim (eps 1e-5, biased var): y[r,:] = (x[r,:]-mean)*rstd*gamma+beta */
int mnet_layernorm(const float* x, const float* gamma, const float* beta, float* y, int32_t rows,
                   int32_t d, float eps, void* stream);
/* K9 token-axis LayerNorm + Linear (linear_seq_maxlen :141-144,155 ; linear_w_maxlen :59-62,72):
 *   y[b,j,d] = bias[j] + sum_t W[j,t] * LN_t(x[b,:,d])[t]      x [B,T,D], y [B,J,D], T <= 64 */
int mnet_token_mix(const float* x, const float* ln_g, const float* ln_b, const float* wgt,
                   const float* bias, float* y, int32_t B, int32_t T, int32_t D, int32_t J, float eps,
                   void* stream);
/* K8 softmax(q k^T * scale) v per (batch, head) on the matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 products — the character
 * indices downstream stay bit-exact); qkv [B,N,3*H*64] packed (q|k|v, each '(h d)'),
 * out [B,N,H*64]; N <= 64, head dim 64   (textvit_arch.py:104-112) */
int mnet_attention(const float* qkv, float* out, int32_t B, int32_t N, int32_t H, float scale,
                   void* stream);

/* ---------------------------------------------------------------------------------------------
 * TSPGAN pieces, models/networks.py
 * ------------------------------------------------------------------------------------------- */
/* K15 PixelNorm: y = x * rsqrt(mean(x^2, dim=1) + 1e-8)   (:170-171), x [N,D] fp32 */
int mnet_pixelnorm(const float* x, float* y, int32_t N, int32_t D, void* stream);
/* K14 SelectText: out NHWC [N,4,4*nc,C] = emb[labels[i,j],:] tiled 4x4, chars side by side (:205-215).
 * labels int64 [N,nc]; returns MNET_E_ARG-free: out-of-range labels must be rejected by the caller. */
int mnet_embed_gather(const float* emb, const int64_t* labels, void* out, int32_t dtype, int32_t N,
                      int32_t nc, int32_t C, int32_t num_classes, void* stream);
/* the same with out[i,y,x,c] multiplied by scale[i,c] (fp32 [N,C], may be NULL) before it is rounded to the storage type: the gathered
 * constant is read by nothing but the first StyledConv (:289-290), whose input modulation x * s (:284, activation side: SURVEY.md §0.5)
 * this applies — that conv then needs no modulation prologue */
int mnet_embed_gather_scaled(const float* emb, const int64_t* labels, const float* scale, void* out, int32_t dtype, int32_t N,
                             int32_t nc, int32_t C, int32_t num_classes, void* stream);
/* demod[n,o] = rsqrt( sum_i style[n,i]^2 * wsq_t[i,o] + 1e-8 ),  wsq_t[i,o] = scale^2 * sum_k W[o,i,k]^2
 * (ModulatedConv2d :284-287 rewritten for activation-side modulation) */
int mnet_demod(const float* style, const float* wsq_t, float* demod, int32_t N, int32_t cin,
               int32_t cout, void* stream);

/* mnet_demod with eps scaled per sample: demod[n,o] = rsqrt( sum_i style[n,i]^2 wsq_t[i,o] + 1e-8 * eps_scale[n] ) — for style rows that
 * mnet_style_rows normalised by 2^-e (eps_scale = 4^-e): the result is then exactly 2^e times mnet_demod of the original row.
 * eps_scale == NULL: mnet_demod. */
int mnet_demod_scaled(const float* style, const float* wsq_t, float* demod, int32_t N, int32_t cin, int32_t cout,
                      const float* eps_scale, void* stream);

/* argmax over the last dim (first maximal index, like torch.max(...,1)[1] in test_w.py:36) */
int mnet_argmax_rows(const float* x, int64_t* idx, int32_t rows, int32_t d, void* stream);

/* flat dtype conversion (count % 4 == 0), used where the fp16 conv stack hands over to the fp32 ViT */
int mnet_convert(const void* src, int32_t src_dtype, void* dst, int32_t dst_dtype, int64_t count, void* stream);

/* K5 standalone: the operator boundary the reference itself has — basicsr.ops.fused_act.fused_leaky_relu
 * (models/networks.py:10,195,241): y = scale * leaky_relu(x + bias[(i/inner) % C], negative_slope) on a
 * contiguous fp32 tensor viewed as [outer, C, inner].  bias may be NULL.  (On the hot path this math is
 * fused into mnet_conv2d_nhwc's epilogue.) */
int mnet_fused_bias_act(const float* x, const float* bias, float* y, int64_t total, int32_t C, int32_t inner,
                        float negative_slope, float scale, void* stream);

/* ToRGB.forward (models/networks.py:313-321; call sites :148,157): modulated 1x1 conv to RGB without demodulation + bias + the
 * bilinearly up-sampled (x2, align_corners=False) RGB image of the level below, then tanh — one streaming pass over x.
 *   out[n,y,x,o] = tanh( scale_b[n] * sum_c wgt[o][c] * (x[n,y,x,c] * style[n][c]) + bias[o] + up2(skip)[n,y,x,o] ),  out[..,3] = 0
 * x NHWC [n,h,w,c] in any storage dtype (c in {64,128,256,512}); wgt fp32 [3][c] (the layer's constant scale folded); style fp32 [n][c];
 * scale_b fp32 [n] or NULL (1); bias fp32 [3]; skip fp32 [n,h/2,w/2,4] or NULL; out fp32 [n,h,w,4]. */
int mnet_torgb(const void* x, int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, const float* wgt, const float* style,
               const float* scale_b, const float* bias, const float* skip, float* out, void* stream);

/* 3x3 / stride 1 / pad 1 convolution of a cin = 64 map to 3 output channels + bias (+ tanh): the last layer of TSPSRNet
 * (conv_final.6 + Tanh, models/networks.py:374-375).  x NHWC [n,h,w,64] (f16 or f32); wgt [3][3][3][64] (cout, kh, kw, cin) in
 * the same dtype; bias fp32 [3]; act MNET_ACT_NONE or MNET_ACT_TANH.  Outputs (either may be NULL, not both): y_nhwc
 * [n,h,w,8] in the input dtype (channels 3..7 zero) and y_nchw fp32 [n,3,h,w] — the tensor the module returns, without a
 * separate layout pass (in f16 mode it holds the same f16-rounded values as y_nhwc). */
int mnet_conv3x3_rgb(const void* x, int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t cin, const void* wgt,
                     const float* bias, int32_t act, void* y_nhwc, float* y_nchw, void* stream);

/* K19, the script's output post-processing (test_sr.py:198-200): sr*0.5+0.5 → HWC → RGB→BGR → clip(0,1)*255.
 * src NHWC [npix][c_ld] (RGB in channels 0..2, f32 or f16); dst [npix][3] BGR as float32 (dst_u8 == 0: exactly what the
 * script hands to cv2.imwrite) or uint8 (dst_u8 != 0: cv2's float→uchar conversion, round half to even) — 4x fewer bytes
 * for the device→host copy and the multi-GPU all-gather (SURVEY.md §8f NEXT-1) */
int mnet_sr_postprocess(const void* src, int32_t src_dtype, void* dst, int32_t dst_u8, int64_t npix, int32_t c_ld, void* stream);

/* Finiteness guard of the half-range precision modes (the role `torch.isfinite(y).all()` would play after test_sr.py:197 — the
 * reference has no such check because its fp32 activations cannot overflow): *flag (int32, device) is set to 0 and then to 1 by any
 * thread that finds an element of x (n elements, MNET_F32 or MNET_F16) that is inf or NaN.  One streaming read of x, no atomics
 * (every writer stores the same value), no synchronisation: the caller reads the flag back when it consumes the result. */
int mnet_nonfinite_flag(const void* x, int32_t dtype, int64_t n, int32_t* flag, void* stream);

/* ---------------------------------------------------------------------------------------------
 * One-time weight packing on the device: from the checkpoint's tensors to the layouts above, without PyTorch
 * or a BLAS on the host side.
 * ------------------------------------------------------------------------------------------- */
/* K18: w_oihw fp32 [cout][cin][kh][kw] (the checkpoint's layout: nn.Conv2d / ModulatedConv2d[0] / nn.Linear with kh = kw = 1)
 *   → packed [cout_pad][kh][kw][cin_pad] in `dtype` (zero padded), every element = (w / sigma) * scale, where
 *   sigma = u^T (W_mat v), W_mat = w viewed as [cout][cin*kh*kw] — the eval-mode old-style torch.nn.utils.spectral_norm fold the
 *   reference recomputes on every forward (models/networks.py:14; weight_orig / weight_u / weight_v of the 33 SN convs of
 *   TSPSRNet); sn_u == sn_v == NULL: no fold (sigma = 1).  `scale`: the layer's constant factor (ModulatedConv2d
 *   1/sqrt(cin*k*k), networks.py:262,284; EqualLinear lr_mul/sqrt(in), :180,192).  MNET_F16X2 additionally stores hi/lo of
 *   256 * value (cin_pad % 32 == 0).  workspace: cout + 1 doubles (only read when sn_u != NULL); sigma is summed in fp64 in a
 *   fixed order (deterministic). */
int mnet_pack_weights(const float* w_oihw, int32_t cout, int32_t cin, int32_t kh, int32_t kw, const float* sn_u,
                      const float* sn_v, float scale, int32_t dtype, int32_t cout_pad, int32_t cin_pad, void* packed,
                      double* workspace, void* stream);
/* demodulation table of a ModulatedConv2d for activation-side modulation (input of mnet_demod):
 *   wsq_t[i][o] = sum_{r,s} (scale * w[o][i][r][s])^2,  fp32 [cin][cout]   (networks.py:284-287) */
int mnet_pack_wsq(const float* w_oihw, int32_t cout, int32_t cin, int32_t khw, float scale, float* wsq_t, void* stream);

/* mnet_gather_rows with every output row divided by 2^e, e = exponent of the row window's largest magnitude (max * 2^-e in [0.5, 1)):
 * the style rows of a modulated conv, normalised so that the modulated activations x * s stay below |x| in the half-precision
 * storage modes (ModulatedConv2d's s can be large with trained weights; models/networks.py:283-287).  Exact (power-of-two factors):
 * eps_scale[r] = 4^-e goes to mnet_demod_scaled, scale_b[r][0..bcast) = 2^e is the out_scale of a conv without demodulation (ToRGB,
 * networks.py:305-321).  eps_scale / scale_b may be NULL. */
int mnet_style_rows(const float* src, int32_t src_rows, int32_t ld, int32_t col0, int32_t ncols, const int64_t* idx,
                    int32_t rows, float* dst, float* eps_scale, float* scale_b, int32_t bcast, void* stream);

/* dst[r][0..ncols) = src[idx ? idx[r] : r][col0 .. col0+ncols)  (fp32; src [src_rows][ld]; idx int64 [rows] or NULL): every glyph
 * takes its image's row of the per-style tensors (test_sr.py:183 gives all glyphs of a strip the same w), every StyledConv its
 * column window of the one batched modulation GEMM (networks.py:141,283).  Indices are not range-checked. */
int mnet_gather_rows(const float* src, int32_t src_rows, int32_t ld, int32_t col0, int32_t ncols, const int64_t* idx,
                     int32_t rows, float* dst, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MARCONET_HIP_H */
