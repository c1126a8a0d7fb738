// "Skinny" fp32 kernel for 1x1 launches over a few hundred pixels — the TextViT linears of a small batch (test_sr.py's own
// loop runs one strip at a time: 64 tokens x 512..2048 features per linear, models/textvit_arch.py via networks.py:62-66).
// The general register-staged kernel (conv_igemm.hip) needs >= 128 pixels per tile and walks K one 32-float slab per
// barrier: at 64 tokens it runs 16-32 workgroups whose k-loop is a chain of exposed global-load latencies (58 us per linear,
// 1.9 ms of a 8.8 ms batch-1 forward).  Here nothing is staged through LDS: every lane loads its MFMA operands straight from
// global memory as 16-byte vectors into a ring of PIPE k-steps kept in flight, and the tile is 16 output channels x 64
// pixels (4 waves, 16 pixels each), so a 512-channel linear over 64 tokens still spreads over 32 workgroups and
// cout = 2048 / 6736 over 128 / 421.
//
// Results are bit-identical to the general kernel's: per output the same v_mfma_f32_16x16x4_f32 sequence over the same
// k grouping (k-steps of 16; lane group g holds k = 4g..4g+3 of the step, MFMA j consumes component j — see Mma<float> in
// conv_igemm.hip), one accumulator, the same epilogue order.  So an image's result does not depend on which of the two
// kernels the batch size selects.
#include <cstdlib>
#include "common.h"
#include "conv_args.h"

namespace {

constexpr int SK_BC = 16, SK_BP = 64, SK_PIPE = 16;

// epilogue of one lane's 4 consecutive channels of one pixel, same order as conv_igemm_kernel
__device__ __forceinline__ void skinny_epilogue(const ConvArgs& p, f32x4 acc, int pix, int co) {
    if (p.out_scale) acc *= *reinterpret_cast<const f32x4*>(p.out_scale + (size_t)(pix / p.howo) * p.cout + co);
    if (p.bias) acc += *reinterpret_cast<const f32x4*>(p.bias + co);
    if (p.res) {
        const int rpix = p.res_mod > 0 ? pix % p.res_mod : pix;
        acc += *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.res) + (size_t)rpix * p.cout + co);
    }
    {
        float ev[4] = {acc[0], acc[1], acc[2], acc[3]};
        act_apply_vec<4>(ev, p.act);
        acc = f32x4{ev[0], ev[1], ev[2], ev[3]};
    }
    if (p.post_scale) acc *= *reinterpret_cast<const f32x4*>(p.post_scale + (size_t)(pix / p.howo) * p.cout + co);
    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.y) + (size_t)pix * p.cout + co) = acc;
}

// Filter == stride ("patchify", which includes 1x1 / stride 1): a token's window row r is kw*cin CONTIGUOUS floats of the NHWC
// map, so K = kh rows of `rowlen` floats.  blockIdx.y = K-slice z (split-K): slice z covers a contiguous 1/(ksplit/kh) of filter
// row z / (ksplit/kh); with a workspace the raw fp32 partial sums go to ws[z][pixel][cout] and skinny_reduce_kernel folds them in
// slice order (deterministic) and applies the epilogue.  ksplit == 1 (1x1 only): epilogue here, no workspace.
__global__ void __launch_bounds__(256) conv_skinny_f32_kernel(const ConvArgs p, float* __restrict__ ws, int ksplit) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l16 = lane & 15, g = lane >> 4;
    const int tc = blockIdx.x % p.tilesC, tp = blockIdx.x / p.tilesC;
    const int co0 = tc * SK_BC, pix0 = tp * SK_BP + wave * 16;
    if (pix0 >= p.npix) return;                                        // wave-uniform; no barriers in this kernel
    const int z = blockIdx.y, spr = ksplit / p.kh;                     // slices per filter row
    const int rowlen = p.kw * p.cin, len = rowlen / spr;
    const int r = z / spr, sub = z - r * spr;
    const int n16 = len >> 4;
    const int tpix = min(pix0 + l16, p.npix - 1);
    const int tn = tpix / p.howo, trem = tpix - tn * p.howo, toh = trem / p.wo, tow = trem - toh * p.wo;
    const float* wrow = reinterpret_cast<const float*>(p.wgt) + (size_t)min(co0 + l16, p.cout - 1) * p.K +
                        (size_t)r * rowlen + (size_t)sub * len + 4 * g;
    const float* xrow = reinterpret_cast<const float*>(p.x0) + ((size_t)(tn * p.h + toh * p.sh + r) * p.w + (size_t)tow * p.sw) * p.cin +
                        (size_t)sub * len + 4 * g;

    // k-steps in flight: a ring of SK_PIPE (A, B) operand pairs; a slot is refilled with step s + SK_PIPE right after its
    // MFMAs were issued, so the loop waits on vmcnt(2*(SK_PIPE-1)) — never on the newest loads.
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    auto mma4 = [&](const f32x4& av, const f32x4& bv) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bv[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bv[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], bv[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], bv[3], acc, 0, 0, 0);
    };
    const int groups = n16 / SK_PIPE;                                  // full ring turns
    int s = 0;
    if (groups > 0) {
        f32x4 a[SK_PIPE], b[SK_PIPE];
#pragma unroll
        for (int i = 0; i < SK_PIPE; ++i) {
            a[i] = *reinterpret_cast<const f32x4*>(wrow + 16 * i);
            b[i] = *reinterpret_cast<const f32x4*>(xrow + 16 * i);
        }
        for (int gi = 0; gi + 1 < groups; ++gi) {
            const float* wn = wrow + 16 * (s + SK_PIPE);
            const float* xn = xrow + 16 * (s + SK_PIPE);
#pragma unroll
            for (int i = 0; i < SK_PIPE; ++i) {
                mma4(a[i], b[i]);
                a[i] = *reinterpret_cast<const f32x4*>(wn + 16 * i);
                b[i] = *reinterpret_cast<const f32x4*>(xn + 16 * i);
            }
            s += SK_PIPE;
        }
#pragma unroll
        for (int i = 0; i < SK_PIPE; ++i) mma4(a[i], b[i]);
        s += SK_PIPE;
    }
    for (; s < n16; ++s) {                                             // K/16 not a multiple of the ring: plain steps
        const f32x4 av = *reinterpret_cast<const f32x4*>(wrow + 16 * s);
        const f32x4 bv = *reinterpret_cast<const f32x4*>(xrow + 16 * s);
        mma4(av, bv);
    }

    // ---- lane owns channels co..co+3 of one pixel
    const int pix = pix0 + l16, co = co0 + 4 * g;
    if (pix >= p.npix || co >= p.cout) return;
    if (ws) *reinterpret_cast<f32x4*>(ws + ((size_t)z * p.npix + pix) * p.cout + co) = acc;
    else skinny_epilogue(p, acc, pix, co);
}

__global__ void __launch_bounds__(256) skinny_reduce_kernel(const ConvArgs p, const float* __restrict__ ws, int ksplit) {
    const int id = blockIdx.x * 256 + threadIdx.x, c4 = p.cout >> 2;
    if (id >= p.npix * c4) return;
    const int pix = id / c4, co = (id - pix * c4) * 4;
    const float* src = ws + (size_t)pix * p.cout + co;
    const size_t zs = (size_t)p.npix * p.cout;
    f32x4 acc = *reinterpret_cast<const f32x4*>(src);
    for (int z = 1; z < ksplit; ++z) acc += *reinterpret_cast<const f32x4*>(src + z * zs);
    skinny_epilogue(p, acc, pix, co);
}

}  // namespace

bool conv_skinny_eligible(const ConvArgs& a, int dtype) {
    static const int max_pix = [] { const char* e = getenv("MNET_SKINNY_MAX_PIX"); return e ? atoi(e) : 512; }();   // A/B knob
    return dtype == MNET_F32 && a.kh == 1 && a.kw == 1 && a.sh == 1 && a.sw == 1 && a.ph == 0 && a.pw == 0 && a.c1 == 0 &&
           !a.in_scale && !a.valid_w && a.cin % 16 == 0 && a.npix <= max_pix;
}

int launch_conv_skinny(const ConvArgs& a, hipStream_t st) {
    ConvArgs b = a;
    b.tilesC = (a.cout + SK_BC - 1) / SK_BC;
    const int tilesP = (a.npix + SK_BP - 1) / SK_BP;
    hipLaunchKernelGGL(conv_skinny_f32_kernel, dim3((unsigned)(b.tilesC * tilesP)), dim3(256), 0, st, b, (float*)nullptr, 1);
    MNET_LAUNCH_CHECK("conv_skinny_f32_kernel");
    return MNET_OK;
}

int launch_conv_skinny_splitk(const ConvArgs& a, int dtype, int ksplit, float* ws, hipStream_t st) {
    MNET_CHECK_ARG(dtype == MNET_F32 && a.c1 == 0 && !a.in_scale && !a.valid_w && a.ph == 0 && a.pw == 0 && a.kh == a.sh && a.kw == a.sw &&
                   a.npix <= 512,
                   "conv_splitk: needs fp32, one source, no input transform, filter == stride, no padding and <= 512 output pixels");
    MNET_CHECK_ARG(ws != nullptr && ksplit >= a.kh && ksplit <= 256 && ksplit % a.kh == 0, "conv_splitk: ksplit=%d must be a multiple of kh=%d (<= 256) and the workspace non-null", ksplit, a.kh);
    const int rowlen = a.kw * a.cin, spr = ksplit / a.kh;
    MNET_CHECK_ARG(rowlen % spr == 0 && (rowlen / spr) % 16 == 0, "conv_splitk: kw*cin=%d does not split into %d slices of whole 16-float steps", rowlen, spr);
    MNET_CHECK_ALIGN(a.cout % 4 == 0 && aligned16(ws), "conv_splitk: unaligned");
    ConvArgs b = a;
    b.tilesC = (a.cout + SK_BC - 1) / SK_BC;
    const int tilesP = (a.npix + SK_BP - 1) / SK_BP;
    hipLaunchKernelGGL(conv_skinny_f32_kernel, dim3((unsigned)(b.tilesC * tilesP), (unsigned)ksplit), dim3(256), 0, st, b, ws, ksplit);
    MNET_LAUNCH_CHECK("conv_skinny_f32_kernel (split-K)");
    hipLaunchKernelGGL(skinny_reduce_kernel, dim3((unsigned)((a.npix * (a.cout / 4) + 255) / 256)), dim3(256), 0, st, b, (const float*)ws, ksplit);
    MNET_LAUNCH_CHECK("skinny_reduce_kernel");
    return MNET_OK;
}
