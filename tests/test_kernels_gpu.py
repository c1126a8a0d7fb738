"""-m gpu: every C-ABI kernel against a plain PyTorch fp32 CPU reference of the same op (the op-level
oracle for floating-point kernels).  Tolerances: fp32 kernels 2e-5 relative to the output scale (summation
order only); f16 kernels are fed fp16-rounded inputs, the reference is computed in fp32 from the same rounded
values, so the only differences are accumulation order and the final fp16 rounding (2e-3 relative)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _ops():
    from marconet_amd import ops
    return ops


def _rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def _P():
    from marconet_amd import packing
    return packing


# the two 4-byte blocked storages (split half: 22 significant bits; fp16+8: ~16 bits relative to the largest value of a 32-channel
# block) go through the HOST packers of marconet_amd/packing.py / mxfmt.py — independent of the device converter
SPLIT, MX = "split", "mx"
ALL_DTYPES = [torch.float32, torch.float16, SPLIT, MX]


def _dt(dtype):
    return {SPLIT: _P().SPLIT_DTYPE, MX: _P().MX_DTYPE}.get(dtype, dtype)


def _q(t, dtype):
    """round an NCHW tensor through the storage dtype (identity for fp32)"""
    if dtype in (SPLIT, MX):
        P = _P()
        return P.to_float(P.from_float(t.permute(0, 2, 3, 1).contiguous(), _dt(dtype))).permute(0, 3, 1, 2).contiguous()
    return t.to(dtype).float()


def _nhwc(t, dtype):   # NCHW cpu fp32 -> NHWC device
    if dtype in (SPLIT, MX):
        return _P().from_float(t.permute(0, 2, 3, 1).contiguous(), _dt(dtype)).to(DEV)
    return t.permute(0, 2, 3, 1).contiguous().to(dtype).to(DEV)


def _nchw(t):          # NHWC device -> NCHW cpu fp32
    if _P().is_split(t.dtype):
        return _P().to_float(t.cpu()).permute(0, 3, 1, 2).contiguous()
    return t.float().cpu().permute(0, 3, 1, 2).contiguous()


def _pack_w(w, dtype):  # OIHW -> [O,KH,KW,I]
    if dtype in (SPLIT, MX):
        return _P().pack_conv_weight(w, _dt(dtype)).to(DEV)
    return w.permute(0, 2, 3, 1).contiguous().to(dtype).to(DEV)


def _tol(dtype):
    return {torch.float32: 2e-5, torch.float16: 2.5e-3, SPLIT: 2e-6, MX: 2e-5}[dtype]


def _check(name, got, ref, dtype, extra=1.0):
    scale = max(ref.abs().max().item(), 1e-6)
    err = (got - ref).abs().max().item()
    print("%-44s max|d|=%.3e  ref max=%.3e  rel=%.3e" % (name, err, scale, err / scale))
    assert err <= _tol(dtype) * extra * scale, "%s: err %.3e > tol (scale %.3e)" % (name, err, scale)


ACT_REF = {
    0: lambda v: v,
    1: F.relu,
    2: lambda v: F.leaky_relu(v, 0.2),
    3: lambda v: F.leaky_relu(v, 0.2) * 2 ** 0.5,
    4: torch.tanh,
    5: F.gelu,
    6: torch.sigmoid,
}

CONV_CASES = [
    # n, h, w, c0, c1, cout, k, stride, pad
    (2, 9, 13, 64, 0, 128, 3, (1, 1), 1),
    (1, 16, 24, 32, 0, 64, 3, (1, 1), 1),
    (3, 7, 11, 16, 0, 32, 3, (2, 1), 1),
    (2, 12, 20, 8, 0, 8, 3, (1, 1), 1),
    (2, 12, 20, 64, 0, 256, 3, (2, 2), 1),
    (2, 8, 8, 128, 0, 136, 1, (1, 1), 0),
    (2, 8, 16, 256, 128, 256, 3, (1, 1), 1),
    (1, 32, 32, 512, 0, 256, 3, (1, 1), 1),
    (4, 6, 10, 24, 8, 48, 1, (2, 1), 0),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_plain(case, dtype):
    ops = _ops()
    n, h, w, c0, c1, cout, k, stride, pad = case
    x = _q(_rnd((n, c0 + c1, h, w), 1), dtype)
    wt = _q(_rnd((cout, c0 + c1, k, k), 2, 1.0 / math.sqrt((c0 + c1) * k * k)), dtype)
    ref = F.conv2d(x, wt, stride=stride, padding=pad)
    x0 = _nhwc(x[:, :c0], dtype)
    x1 = _nhwc(x[:, c0:], dtype) if c1 else None
    y = ops.conv2d(x0, _pack_w(wt, dtype), cout, k, k, stride, (pad, pad), x1=x1)
    torch.cuda.synchronize()
    _check("conv %s %s" % (case, dtype), _nchw(y), ref, dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("act", [0, 1, 2, 3, 4, 5, 6])
def test_conv_epilogue(act, dtype):
    ops = _ops()
    n, h, w, cin, cout = 3, 10, 14, 32, 72
    x = _q(_rnd((n, cin, h, w), 3), dtype)
    wt = _q(_rnd((cout, cin, 3, 3), 4, 1.0 / math.sqrt(cin * 9)), dtype)
    bias = _rnd((cout,), 5, 0.3)
    osc = _rnd((n, cout), 6).abs() + 0.5
    res = _q(_rnd((n, cout, h, w), 7), dtype)
    ref = F.conv2d(x, wt, padding=1) * osc[:, :, None, None] + bias[None, :, None, None] + res
    ref = ACT_REF[act](ref)
    y = ops.conv2d(_nhwc(x, dtype), _pack_w(wt, dtype), cout, 3, 3, (1, 1), (1, 1), out_scale=osc.to(DEV),
                   bias=bias.to(DEV), residual=_nhwc(res, dtype), act=act)
    torch.cuda.synchronize()
    _check("conv epilogue act=%d %s" % (act, dtype), _nchw(y), ref, dtype, extra=2.0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("swish", [False, True])
def test_conv_prologue_affine_validw(swish, dtype):
    """GroupNorm-affine + swish prologue and ragged valid width (per-glyph windows, networks.py:425-448)."""
    ops = _ops()
    n, h, w, cin, cout = 5, 8, 12, 64, 64
    x = _q(_rnd((n, cin, h, w), 8), dtype)
    wt = _q(_rnd((cout, cin, 3, 3), 9, 1.0 / math.sqrt(cin * 9)), dtype)
    sc = _rnd((n, cin), 10).abs() + 0.5
    sh = _rnd((n, cin), 11, 0.3)
    vw = torch.tensor([12, 7, 1, 12, 9], dtype=torch.int32)
    xin = x * sc[:, :, None, None] + sh[:, :, None, None]
    if swish:
        xin = xin * torch.sigmoid(xin)
    xin = _q(xin, dtype)          # kernel rounds the transformed operand to the storage dtype
    refs = []
    for i in range(n):
        v = int(vw[i])
        r = torch.zeros(1, cout, h, w)
        r[..., :v] = F.conv2d(xin[i:i + 1, :, :, :v], wt, padding=1)
        refs.append(r)
    ref = torch.cat(refs)
    y = ops.conv2d(_nhwc(x, dtype), _pack_w(wt, dtype), cout, 3, 3, (1, 1), (1, 1), in_scale=sc.to(DEV),
                   in_shift=sh.to(DEV), in_swish=swish, valid_w=vw.to(DEV))
    torch.cuda.synchronize()
    got = _nchw(y)
    for i in range(n):
        got[i, :, :, int(vw[i]):] = 0       # columns beyond valid_w are unspecified
    _check("conv prologue swish=%s %s" % (swish, dtype), got, ref, dtype, extra=4.0)


def test_conv_patch_embed_fp32():
    """8x8 stride-8 conv over NHWC [B,8,512,512] == Rearrange + Linear(32768,512) (textvit_arch.py:32-35), + pos-emb."""
    ops = _ops()
    B = 2
    feat = _rnd((B, 512, 8, 512), 12)
    W = _rnd((512, 32768), 13, 1.0 / math.sqrt(32768))
    b = _rnd((512,), 14, 0.1)
    pe = _rnd((64, 512), 15)
    tok = feat.reshape(B, 512, 8, 64, 8).permute(0, 3, 2, 4, 1).reshape(B, 64, 32768)
    ref = F.linear(tok, W, b) + pe
    y = ops.conv2d(_nhwc(feat, torch.float32), W.to(DEV), 512, 8, 8, (8, 8), (0, 0), bias=b.to(DEV),
                   residual=pe.to(DEV).reshape(1, 1, 64, 512), res_mod=64)
    torch.cuda.synchronize()
    _check("patch embed", y.cpu().reshape(B, 64, 512), ref, torch.float32, extra=4.0)


def test_linear_cls_fp32():
    ops = _ops()
    x = _rnd((128, 512), 16)
    W = _rnd((6736, 512), 17, 1.0 / math.sqrt(512))
    b = _rnd((6736,), 18, 0.1)
    ref = F.gelu(F.linear(x, W, b))
    y = ops.linear(x.to(DEV), W.to(DEV), 6736, bias=b.to(DEV), act=ops.ACT_GELU)
    torch.cuda.synchronize()
    _check("linear 512->6736 gelu", y.cpu(), ref, torch.float32)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_layout_roundtrip(dtype):
    ops = _ops()
    x = _rnd((3, 3, 32, 50), 19)
    y = ops.nchw_to_nhwc(x.to(DEV), dtype, c_ld=8)
    torch.cuda.synchronize()
    got = y.float().cpu()
    assert (got[..., 3:] == 0).all()
    assert torch.equal(got[..., :3].permute(0, 3, 1, 2), _q(x, dtype))
    back = ops.nhwc_to_nchw(y, c=3).cpu()
    assert torch.equal(back, _q(x, dtype))
    x2 = _rnd((2, 70, 5, 9), 20)
    y2 = ops.nchw_to_nhwc(x2.to(DEV), dtype)
    assert torch.equal(ops.nhwc_to_nchw(y2).cpu(), _q(x2, dtype))


@pytest.mark.parametrize("dtype", ALL_DTYPES)
def test_upsample2x(dtype):
    ops = _ops()
    x = _q(_rnd((2, 32, 5, 7), 21), dtype)
    ref = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
    y = ops.upsample2x(_nhwc(x, dtype))
    torch.cuda.synchronize()
    _check("upsample2x %s" % dtype, _nchw(y), ref, dtype)


@pytest.mark.parametrize("dtype", [SPLIT, MX])
def test_upsample2x_into_f16(dtype):
    """round 5: a split-half / fp16+8 source up-sampled straight into plain f16 (the image-only generator level of the batched driver:
    no separate convert pass) — the interpolation of the source's values, times the fused per-(n, c) scale, rounded to f16 once"""
    ops = _ops()
    x = _q(_rnd((2, 64, 6, 9), 121), dtype)
    sc = _rnd((2, 64), 122).abs() + 0.5
    ref = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False) * sc[:, :, None, None]
    y = ops.upsample2x(_nhwc(x, dtype), scale=sc.to(DEV), out_dtype=torch.float16)
    torch.cuda.synchronize()
    assert y.dtype == torch.float16 and tuple(y.shape) == (2, 12, 18, 64)
    _check("upsample2x %s -> f16" % dtype, _nchw(y), ref, torch.float16)
    with pytest.raises(Exception):
        ops.upsample2x(_nhwc(x, torch.float16), out_dtype=_dt(MX))           # only towards f16


@pytest.mark.parametrize("dtype", ALL_DTYPES)
@pytest.mark.parametrize("c", [64, 256, 512])
def test_groupnorm_affine(c, dtype):
    ops = _ops()
    n, h, w = 3, 16, 40
    x = _q(_rnd((n, c, h, w), 22) * 2 + 0.7, dtype)
    gamma, beta = 1 + 0.1 * _rnd((c,), 23), 0.1 * _rnd((c,), 24)
    vw = torch.tensor([40, 17, 3], dtype=torch.int32)
    sc, sh = ops.groupnorm_affine(_nhwc(x, dtype), gamma.to(DEV), beta.to(DEV), 1e-6, vw.to(DEV))
    torch.cuda.synchronize()
    for i in range(n):
        v = int(vw[i])
        ref = F.group_norm(x[i:i + 1, :, :, :v], c // 32, gamma, beta, 1e-6)
        got = x[i:i + 1, :, :, :v] * sc[i].cpu()[None, :, None, None] + sh[i].cpu()[None, :, None, None]
        _check("groupnorm c=%d img=%d %s" % (c, i, dtype), got, ref, torch.float32, extra=2.0)


@pytest.mark.parametrize("dtype", ALL_DTYPES)
@pytest.mark.parametrize("S", [32, 64])
def test_adain_crop_and_scatter(S, dtype):
    ops = _ops()
    C, B, FW = 256, 2, S * 16
    half = S // 2
    feat = _q(_rnd((B, C, S, FW), 25), dtype)
    windows = [(0, 0, 21 * S // 32), (0, FW // 2 - half, S), (0, FW // 2 - half + 5, S), (1, FW - 17 * S // 32, 17 * S // 32), (1, 3, S)]
    G = len(windows)
    prior = _q(_rnd((G, C, S, S), 26) * 1.5 + 0.2, dtype)
    g_img = torch.tensor([wd[0] for wd in windows], dtype=torch.int32)
    g_x1 = torch.tensor([wd[1] for wd in windows], dtype=torch.int32)
    g_w = torch.tensor([wd[2] for wd in windows], dtype=torch.int32)
    g_y1 = torch.tensor([half - int(wd[2] / 2) for wd in windows], dtype=torch.int32)
    out = ops.adain_crop_concat(_nhwc(prior, dtype), _nhwc(feat, dtype), g_img.to(DEV), g_x1.to(DEV), g_y1.to(DEV), g_w.to(DEV))
    torch.cuda.synchronize()
    got = _nchw(out)

    def ms(f):
        v = f.reshape(f.shape[0], f.shape[1], -1)
        return v.mean(2)[:, :, None, None], (v.var(2) + 1e-5).sqrt()[:, :, None, None]

    # the same launch can also emit the GroupNorm affine of its output (closed form from the AdaIN statistics)
    gamma, beta = _rnd((2 * C,), 29).abs() + 0.5, _rnd((2 * C,), 30) * 0.3
    out2, gsc, gsh = ops.adain_crop_concat_gn(_nhwc(prior, dtype), _nhwc(feat, dtype), g_img.to(DEV), g_x1.to(DEV), g_y1.to(DEV),
                                              g_w.to(DEV), gamma.to(DEV), beta.to(DEV), 1e-6, split=False)
    # ... and the three-launch form used for few glyphs (16 workgroups per glyph): same values up to the association of the fp64
    # statistic sums, i.e. at most an ulp of the storage type apart
    out3, gsc3, gsh3 = ops.adain_crop_concat_gn(_nhwc(prior, dtype), _nhwc(feat, dtype), g_img.to(DEV), g_x1.to(DEV), g_y1.to(DEV),
                                                g_w.to(DEV), gamma.to(DEV), beta.to(DEV), 1e-6, split=True)
    torch.cuda.synchronize()
    assert torch.equal(_nchw(out2), got)
    ulp = {torch.float16: 1e-3, MX: 1e-4}.get(dtype, 2e-7)
    assert torch.allclose(_nchw(out3), got, rtol=ulp, atol=ulp)
    assert torch.allclose(gsc3, gsc, rtol=1e-6, atol=1e-7) and torch.allclose(gsh3, gsh, rtol=1e-6, atol=1e-6)
    print("adain split == fused bit for bit:", bool(torch.equal(_nchw(out3), got) and torch.equal(gsc3, gsc) and torch.equal(gsh3, gsh)))
    for g, (b, x1, gw) in enumerate(windows):
        y1 = int(g_y1[g])
        cp, cl = prior[g:g + 1, :, :, y1:y1 + gw], feat[b:b + 1, :, :, x1:x1 + gw]
        lm, ls = ms(cl)
        pm, ps = ms(cp)
        ref = torch.cat(((cp - pm) / ps * ls + lm, cl), dim=1)
        _check("adain S=%d glyph %d %s" % (S, g, dtype), got[g:g + 1, :, :, :gw], ref, dtype, extra=2.0)
        assert (got[g, :, :, gw:] == 0).all()
        v = ref.reshape(2 * C // 32, -1)                                    # GroupNorm(2C/32 groups) statistics of the exact output
        mean, rstd = v.mean(1), (v.var(1, unbiased=False) + 1e-6).rsqrt()
        ga = gamma * rstd.repeat_interleave(32)
        want_sc, want_sh = ga, beta - mean.repeat_interleave(32) * ga
        assert torch.allclose(gsc[g].cpu(), want_sc, rtol=2e-4, atol=1e-5), "GN scale, glyph %d" % g
        assert torch.allclose(gsh[g].cpu(), want_sh, rtol=2e-3, atol=2e-4), "GN shift, glyph %d" % g
    # ordered scatter
    scale = _q(_rnd((G, C, S, S), 27), dtype)
    shift = _q(_rnd((G, C, S, S), 28), dtype)
    g_start = torch.tensor([0, 3, 5], dtype=torch.int32)
    res = torch.zeros_like(feat)
    for g, (b, x1, gw) in enumerate(windows):
        res[b, :, :, x1:x1 + gw] = feat[b, :, :, x1:x1 + gw] * scale[g, :, :, :gw] + shift[g, :, :, :gw]
    ref = feat + res
    o = ops.glyph_scatter_affine(_nhwc(feat, dtype), _nhwc(scale, dtype), _nhwc(shift, dtype), g_start.to(DEV), g_x1.to(DEV), g_w.to(DEV))
    torch.cuda.synchronize()
    _check("glyph scatter S=%d %s" % (S, dtype), _nchw(o), ref, dtype)


def test_layernorm_tokenmix_attention():
    ops = _ops()
    x = _rnd((130, 512), 29) * 3 + 1
    g, b = 1 + 0.1 * _rnd((512,), 30), 0.1 * _rnd((512,), 31)
    y = ops.layernorm(x.to(DEV), g.to(DEV), b.to(DEV))
    _check("layernorm 512", y.cpu(), F.layer_norm(x, (512,), g, b, 1e-5), torch.float32)
    x64 = _rnd((70, 64), 32)
    g64, b64 = 1 + 0.1 * _rnd((64,), 33), 0.1 * _rnd((64,), 34)
    y = ops.layernorm(x64.to(DEV), g64.to(DEV), b64.to(DEV))
    _check("layernorm 64", y.cpu(), F.layer_norm(x64, (64,), g64, b64, 1e-5), torch.float32)
    # token mix: LN over the token axis then Linear(64 -> J)
    for J in (16, 1):
        xt = _rnd((3, 64, 512), 35)
        W, bb = _rnd((J, 64), 36, 0.125), _rnd((J,), 37, 0.1)
        ref = F.linear(F.layer_norm(xt.permute(0, 2, 1), (64,), g64, b64, 1e-5), W, bb).permute(0, 2, 1)
        y = ops.token_mix(xt.to(DEV), g64.to(DEV), b64.to(DEV), W.to(DEV), bb.to(DEV))
        _check("token_mix J=%d" % J, y.cpu(), ref, torch.float32)
    for N in (64, 16, 37, 5):          # 64 / 16: the TextViT's token counts; 37 / 5: partial last block (masked keys, skipped rows)
        B, H = 3, 8
        qkv = _rnd((B, N, 3 * 512), 38)
        q, k, v = [t.reshape(B, N, H, 64).permute(0, 2, 1, 3) for t in qkv.chunk(3, dim=-1)]
        att = (torch.matmul(q, k.transpose(-1, -2)) * 0.125).softmax(-1)
        ref = torch.matmul(att, v).permute(0, 2, 1, 3).reshape(B * N, 512)
        y = ops.attention(qkv.to(DEV), B, N, H, 0.125)
        _check("attention N=%d" % N, y.cpu(), ref, torch.float32)


def test_gan_small_ops():
    ops = _ops()
    x = _rnd((37, 512), 39)
    y = ops.pixelnorm(x.to(DEV))
    _check("pixelnorm", y.cpu(), x * torch.rsqrt(torch.mean(x ** 2, dim=1, keepdim=True) + 1e-8), torch.float32)
    emb = _rnd((100, 512), 40)
    labels = torch.tensor([[3, 99], [0, 50], [7, 7]], dtype=torch.int64)
    for dtype in (torch.float32, torch.float16):
        o = ops.embed_gather(emb.to(DEV), labels.to(DEV), dtype, 100)
        ref = torch.cat([emb[labels[:, j]][:, :, None, None].expand(3, 512, 4, 4) for j in range(2)], dim=3)
        assert torch.equal(_nchw(o), _q(ref, dtype))
        sc = _rnd((3, 512), 43) + 1.5                                      # per-(sample, channel) factor, applied before the rounding
        o = ops.embed_gather(emb.to(DEV), labels.to(DEV), dtype, 100, scale=sc.to(DEV))
        assert torch.equal(_nchw(o), _q(ref * sc[:, :, None, None], dtype))
    with pytest.raises(ValueError):
        ops.embed_gather(emb.to(DEV), labels.to(DEV), torch.float32, 100, scale=torch.ones(2, 512, device=DEV))
    s = _rnd((9, 256), 41) + 1
    w = _rnd((128, 256, 3, 3), 42)
    scale = 1 / math.sqrt(256 * 9)
    wmod = scale * w[None] * s[:, None, :, None, None]
    ref = torch.rsqrt(wmod.pow(2).sum([2, 3, 4]) + 1e-8)
    wsq_t = ((scale * w) ** 2).sum([2, 3]).t().contiguous()
    d = ops.demod(s.to(DEV), wsq_t.to(DEV))
    _check("demod", d.cpu(), ref, torch.float32, extra=2.0)
    s2, w2 = _rnd((3, 515), 46) + 1, _rnd((100, 515), 47).abs()          # cin, cout off the 64/16 grid; wsq_t given directly
    d2 = ops.demod(s2.to(DEV), w2.t().contiguous().to(DEV))
    _check("demod ragged", d2.cpu(), torch.rsqrt((s2.double() ** 2 @ w2.double().t()) + 1e-8).float(), torch.float32, extra=2.0)
    lg = _rnd((70, 6736), 43)
    lg[5, 100] = lg[5, 200] = 50.0
    assert torch.equal(ops.argmax_rows(lg.to(DEV)).cpu(), lg.argmax(-1))
    xx = _rnd((4, 12, 5, 6), 44)
    bb = _rnd((12,), 45)
    y = ops.fused_bias_act(xx.to(DEV), bb.to(DEV))
    _check("fused_bias_act", y.cpu(), F.leaky_relu(xx + bb.view(1, -1, 1, 1), 0.2) * 2 ** 0.5, torch.float32)
    h = ops.convert(xx.to(DEV), torch.float16)
    assert torch.equal(h.cpu(), xx.half())
    assert torch.equal(ops.convert(h, torch.float32).cpu(), xx.half().float())


@pytest.mark.parametrize("dtype", ALL_DTYPES)
@pytest.mark.parametrize("hw", [(37, 41), (64, 96)])
def test_affine_act_on_larger_maps(dtype, hw):
    """maps of several workgroups per image, ragged (37 x 41: the last workgroup is partly idle) and even (64 x 96)"""
    ops = _ops()
    n, c = 2, 64
    x = _q(_rnd((n, c, hw[0], hw[1]), 150), dtype)
    sc, sh = _rnd((n, c), 151) + 1.0, _rnd((n, c), 152, 0.3)
    t = x * sc[:, :, None, None] + sh[:, :, None, None]
    xd = _nhwc(x, dtype)
    _check("affine_act swish %s %s" % (dtype, hw), _nchw(ops.affine_act(xd, sc.to(DEV), sh.to(DEV), swish=True)), t * torch.sigmoid(t), dtype)
    _check("affine_act affine %s %s" % (dtype, hw), _nchw(ops.affine_act(xd, sc.to(DEV), sh.to(DEV))), t, dtype)
    ops.affine_act(xd, sc.to(DEV), sh.to(DEV), swish=True, out=xd)       # in place
    _check("affine_act in-place %s %s" % (dtype, hw), _nchw(xd), t * torch.sigmoid(t), dtype)


@pytest.mark.parametrize("dtype", ALL_DTYPES)
def test_post_scale_upsample_scale_affine_act(dtype):
    ops = _ops()
    n, h, w, cin, cout = 3, 6, 10, 32, 64 if dtype in (SPLIT, MX) else 40
    x = _q(_rnd((n, cin, h, w), 50), dtype)
    wt = _rnd((cout, cin, 3, 3), 51, 1.0 / math.sqrt(cin * 9))
    wt = wt if dtype in (SPLIT, MX) else _q(wt, dtype)          # (the blocked storages pack 256 W: the reference keeps the true weights)
    bias, osc, post = _rnd((cout,), 52, 0.3), _rnd((n, cout), 53).abs() + 0.5, _rnd((n, cout), 54) + 1.0
    ref = F.leaky_relu(F.conv2d(x, wt, padding=1) * osc[:, :, None, None] + bias[None, :, None, None], 0.2) * 2 ** 0.5
    ref = ref * post[:, :, None, None]
    y = ops.conv2d(_nhwc(x, dtype), _pack_w(wt, dtype), cout, 3, 3, (1, 1), (1, 1), out_scale=osc.to(DEV), bias=bias.to(DEV),
                   act=ops.ACT_LRELU_SQRT2, post_scale=post.to(DEV))
    _check("conv post_scale %s" % dtype, _nchw(y), ref, dtype, extra={SPLIT: 4.0, MX: 2.5}.get(dtype, 2.0))
    sc = _rnd((n, cin), 55) + 1.0
    ref = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False) * sc[:, :, None, None]
    _check("upsample2x*scale %s" % dtype, _nchw(ops.upsample2x(_nhwc(x, dtype), scale=sc.to(DEV))), ref, dtype)
    sh = _rnd((n, cin), 56, 0.3)
    t = x * sc[:, :, None, None] + sh[:, :, None, None]
    xd = _nhwc(x, dtype)
    _check("affine_act swish %s" % dtype, _nchw(ops.affine_act(xd, sc.to(DEV), sh.to(DEV), swish=True)), t * torch.sigmoid(t), dtype)
    _check("affine_act scale only %s" % dtype, _nchw(ops.affine_act(xd, sc.to(DEV))), x * sc[:, :, None, None], dtype)
    ops.affine_act(xd, sc.to(DEV), sh.to(DEV), swish=True, out=xd)       # in place
    _check("affine_act in-place %s" % dtype, _nchw(xd), t * torch.sigmoid(t), dtype)


DMA_CASES = [
    # n, h, w, c0, c1, cout, k, stride, pad
    (2, 9, 13, 64, 0, 256, 3, (1, 1), 1),
    (3, 16, 24, 128, 0, 128, 3, (1, 1), 1),
    (2, 12, 20, 64, 0, 64, 3, (2, 1), 1),
    (2, 12, 20, 128, 0, 320, 3, (2, 2), 1),
    (5, 8, 8, 192, 0, 136, 1, (1, 1), 0),
    (2, 8, 16, 256, 128, 256, 3, (1, 1), 1),
    (2, 8, 16, 256, 64, 256, 3, (1, 1), 1),
    (40, 4, 4, 512, 0, 512, 3, (1, 1), 1),
    (1, 32, 512, 64, 0, 64, 3, (1, 1), 1),
    (5, 120, 130, 64, 0, 320, 3, (1, 1), 1),       # npix >= 65536 and cout >= 256 → 256x256-tile, 16-wave variant
    (70, 31, 33, 64, 64, 256, 3, (1, 1), 1),
]


@pytest.mark.parametrize("case", DMA_CASES)
def test_conv_lds_dma_kernel(case):
    """the LDS-DMA fast path: against F.conv2d and, to f16 output rounding, against the register-staged kernel"""
    ops = _ops()
    dtype = torch.float16
    n, h, w, c0, c1, cout, k, stride, pad = case
    x = _q(_rnd((n, c0 + c1, h, w), 61), dtype)
    wt = _q(_rnd((cout, c0 + c1, k, k), 62, 1.0 / math.sqrt((c0 + c1) * k * k)), dtype)
    bias = _rnd((cout,), 63, 0.3)
    osc, post = _rnd((n, cout), 64).abs() + 0.5, _rnd((n, cout), 65) + 1.0
    ho = (h + 2 * pad - k) // stride[0] + 1
    wo = (w + 2 * pad - k) // stride[1] + 1
    res = _q(_rnd((n, cout, ho, wo), 66), dtype)
    vw = torch.tensor([w - (i % 3) * 2 for i in range(n)], dtype=torch.int32)
    refs = []
    for i in range(n):
        xi = x[i:i + 1].clone()
        xi[..., int(vw[i]):] = 0
        refs.append(F.conv2d(xi, wt, stride=stride, padding=pad))
    ref = F.leaky_relu(torch.cat(refs) * osc[:, :, None, None] + bias[None, :, None, None] + res, 0.2) * post[:, :, None, None]
    x0 = _nhwc(x[:, :c0], dtype)
    x1 = _nhwc(x[:, c0:], dtype) if c1 else None
    kw = dict(x1=x1, valid_w=vw.to(DEV), out_scale=osc.to(DEV), bias=bias.to(DEV), residual=_nhwc(res, dtype),
              act=ops.ACT_LRELU, post_scale=post.to(DEV))
    y_dma = ops.conv2d(x0, _pack_w(wt, dtype), cout, k, k, stride, (pad, pad), algo=2, **kw)
    y_reg = ops.conv2d(x0, _pack_w(wt, dtype), cout, k, k, stride, (pad, pad), algo=1, **kw)
    torch.cuda.synchronize()
    _check("conv LDS-DMA %s" % (case,), _nchw(y_dma), ref, dtype, extra=2.0)
    d = (y_dma.float() - y_reg.float()).abs()     # k walked in a different order: fp32 partial sums associate differently
    assert float(d.max()) <= 2.0 ** -9 * float(y_reg.float().abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 16, 17])
@pytest.mark.parametrize("shape", [(3, 40, 52, 64, 64, 200, 3), (700, 4, 4, 128, 0, 72, 3), (2, 9, 300, 192, 0, 320, 1)])
def test_conv_lds_dma_every_tile_config(cfg, shape):
    """every LDS-DMA tile configuration, pinned explicitly, is bit-identical to the register-staged kernel on shapes with
    cout / pixel tails, two concat sources and ragged valid widths (ids in include/marconet_hip.h)"""
    ops = _ops()
    from marconet_amd._lib import ALGO_DMA_CFG0, ALGO_DMA_CFG16
    dtype = torch.float16
    n, h, w, c0, c1, cout, k = shape
    algo_cfg = ALGO_DMA_CFG0 + cfg if cfg < 16 else ALGO_DMA_CFG16 + (cfg - 16)
    x = _q(_rnd((n, c0 + c1, h, w), 71), dtype)
    wt = _q(_rnd((cout, c0 + c1, k, k), 72, 1.0 / math.sqrt((c0 + c1) * k * k)), dtype)
    bias = _rnd((cout,), 73, 0.3)
    osc = _rnd((n, cout), 74).abs() + 0.5
    res = _nhwc(_q(_rnd((n, cout, h, w), 76), dtype), dtype)
    vw = torch.tensor([w - (i % 4) for i in range(n)], dtype=torch.int32, device=DEV)
    x0 = _nhwc(x[:, :c0], dtype)
    x1 = _nhwc(x[:, c0:], dtype) if c1 else None
    kw = dict(x1=x1, valid_w=vw, out_scale=osc.to(DEV), bias=bias.to(DEV), residual=res, act=ops.ACT_LRELU_SQRT2)
    y_cfg = ops.conv2d(x0, _pack_w(wt, dtype), cout, k, k, (1, 1), (k // 2, k // 2), algo=algo_cfg, **kw)
    y_reg = ops.conv2d(x0, _pack_w(wt, dtype), cout, k, k, (1, 1), (k // 2, k // 2), algo=1, **kw)
    torch.cuda.synchronize()
    # same products as the register-staged kernel, fp32 partial sums associated differently (k order / MFMA shape)
    d = (y_cfg.float() - y_reg.float()).abs()
    assert float(d.max()) <= 2.0 ** -9 * float(y_reg.float().abs().max()), "cfg %d: max diff %g" % (cfg, float(d.max()))
    assert float((d > 0).float().mean()) < 0.2
    if cfg != 7:   # production configurations (7: the 32x32x16 MFMA experiment): bit-identical to each other — results do not depend on the batch size
        y_auto = ops.conv2d(x0, _pack_w(wt, dtype), cout, k, k, (1, 1), (k // 2, k // 2), algo=2, **kw)
        assert torch.equal(y_cfg, y_auto), "tile configuration %d differs from the auto-picked configuration" % cfg


STRIP_CASES = [  # n, h, w, cin, cout  (all >= 65536 output pixels; 3x3 / stride 1 / pad 1)
    (70, 32, 32, 64, 256),        # narrow maps: a 256-pixel tile = 8 image rows, strip 8 x 34
    (256, 16, 16, 128, 320),      # W = 16: one image per tile, strip 16 x 18; cout tail (320 = 256 + 64)
    (2, 64, 512, 64, 256),        # wide maps: a tile = half an image row, strip 258 (left / right neighbours or zero)
    (17, 64, 64, 192, 256),       # 4 rows per tile, 3 channel slices
    (64, 32, 32, 128, 64),        # 64x512 tile: 16 image rows, strip 16 x 34
    (1, 32, 2048, 64, 64),        # 64x512 tile on a wide map
    (5, 128, 128, 64, 96),        # 4 rows of 128 per 512-pixel tile, cout tail
]


@pytest.mark.parametrize("case", STRIP_CASES)
def test_conv_strip_kernel(case):
    """the 3x3 strip kernel (one activation strip per filter row, three taps read it at offsets 0/1/2) is bit-identical to the per-tap LDS-DMA kernel (same k order, same MFMA) with ragged valid
    widths, residual, demodulation scale, bias and activation in the epilogue."""
    ops = _ops()
    from marconet_amd import _lib
    import ctypes
    dtype = torch.float16
    n, h, w, cin, cout = case
    g = torch.Generator().manual_seed(91)
    x = (torch.rand((n, h, w, cin), generator=g) - 0.5).to(dtype).to(DEV)
    wt = ((torch.rand((cout, 3, 3, cin), generator=g) - 0.5) * (2.0 / math.sqrt(9 * cin))).to(dtype).to(DEV)
    bias = (torch.rand((cout,), generator=g) - 0.5).to(DEV)
    osc = (torch.rand((n, cout), generator=g) + 0.5).to(DEV)
    res = (torch.rand((n, h, w, cout), generator=g) - 0.5).to(dtype).to(DEV)
    vw = torch.tensor([w - (i % 5) * 3 for i in range(n)], dtype=torch.int32, device=DEV)
    kw = dict(valid_w=vw, out_scale=osc, bias=bias, residual=res, act=ops.ACT_LRELU)
    strip = _lib.ALGO_STRIP_CFG0 + (0 if cout >= 256 else 1)
    y_auto = ops.conv2d(x, wt, cout, 3, 3, (1, 1), (1, 1), algo=strip, **kw)
    if cout < 128:      # AUTO takes the strip kernel for the 64x512 tile (the 256x256 form is by explicit request only)
        ops.stats.reset(); ops.stats.enabled = ops.stats.timing = True
        y_def = ops.conv2d(x, wt, cout, 3, 3, (1, 1), (1, 1), algo=0, **kw)
        ops.stats.enabled = ops.stats.timing = False
        assert ops.stats.events[-1][4] == strip and torch.equal(y_def, y_auto)
    y_tap = ops.conv2d(x, wt, cout, 3, 3, (1, 1), (1, 1), algo=_lib.ALGO_LDS_DMA, **kw)
    y_one = ops.conv2d(x, wt, cout, 3, 3, (1, 1), (1, 1), algo=strip | _lib.ALGO_FLAG_ONE_TILE, **kw)
    torch.cuda.synchronize()
    assert torch.equal(y_auto, y_tap), "strip kernel differs from the per-tap LDS-DMA kernel"
    assert torch.equal(y_auto, y_one), "persistent and one-tile-per-workgroup launches differ"
    # and against the fp32 reference on a slice of the batch
    k = min(n, 2)
    xs = x[:k].float().cpu().permute(0, 3, 1, 2).clone()
    for i in range(k):
        xs[i, :, :, int(vw[i]):] = 0
    ref = F.conv2d(xs, wt.float().cpu().permute(0, 3, 1, 2), padding=1) * osc[:k].cpu()[:, :, None, None] + bias.cpu()[None, :, None, None] \
        + res[:k].float().cpu().permute(0, 3, 1, 2)
    ref = F.leaky_relu(ref, 0.2)
    _check("conv strip %s" % (case,), y_auto[:k].float().cpu().permute(0, 3, 1, 2), ref, dtype, extra=2.0)


def _fuzz_cases():
    import random
    rng = random.Random(20260926)
    cases = []
    for i in range(24):
        k = rng.choice([1, 3, 3, 3])
        cin = 64 * rng.randint(1, 4)
        c1 = rng.choice([0, 0, 64]) if cin > 64 else 0
        cout = rng.choice([64, 72, 128, 136, 256, 264, 320, 512])
        stride = rng.choice([(1, 1), (1, 1), (1, 1), (2, 1), (2, 2)]) if k == 3 else (1, 1)
        big = i % 3 == 0                                   # every third case: > 256 pixel tiles → persistent grid makes several passes
        h, w = (rng.randint(100, 180), rng.randint(300, 420)) if big else (rng.randint(3, 40), rng.randint(8, 70))
        n = rng.randint(1, 3) if big else rng.randint(1, 9)
        cases.append((n, h, w, cin - c1, c1, cout, k, stride))
    cases += [(4, 64, 64, 64, 0, 256, 3, (1, 1)), (2, 128, 128, 128, 0, 64, 3, (1, 1)), (300, 16, 16, 64, 0, 512, 3, (1, 1))]   # strip-eligible
    return cases


@pytest.mark.parametrize("case", _fuzz_cases())
def test_conv_f16_fuzz_all_paths_agree(case):
    """seeded random shapes (pixel / cout tails, concat, strides, 1x1, ragged widths, multi-pass persistent grids, tilesC > 1):
    whatever AUTO picks (per-tap LDS-DMA, strip, register-staged) == the pinned per-tap kernel bit for bit, and both agree
    with the register-staged kernel to f16 rounding and with F.conv2d on a slice."""
    ops = _ops()
    from marconet_amd import _lib
    dtype = torch.float16
    n, h, w, c0, c1, cout, k, stride = case
    pad = k // 2
    g = torch.Generator().manual_seed(hash(case) & 0xffff)
    x0 = (torch.rand((n, h, w, c0), generator=g) - 0.5).to(dtype).to(DEV)
    x1 = (torch.rand((n, h, w, c1), generator=g) - 0.5).to(dtype).to(DEV) if c1 else None
    wt = ((torch.rand((cout, k, k, c0 + c1), generator=g) - 0.5) * (2.0 / math.sqrt(k * k * (c0 + c1)))).to(dtype).to(DEV)
    bias = (torch.rand((cout,), generator=g) - 0.5).to(DEV)
    vw = torch.tensor([max(1, w - (i % 4) * 2) for i in range(n)], dtype=torch.int32, device=DEV)
    kw = dict(x1=x1, valid_w=vw, bias=bias, act=ops.ACT_LRELU_SQRT2)
    y_auto = ops.conv2d(x0, wt, cout, k, k, stride, (pad, pad), algo=0, **kw)
    y_tap = ops.conv2d(x0, wt, cout, k, k, stride, (pad, pad), algo=_lib.ALGO_LDS_DMA, **kw)
    y_one = ops.conv2d(x0, wt, cout, k, k, stride, (pad, pad), algo=_lib.ALGO_LDS_DMA | _lib.ALGO_FLAG_ONE_TILE, **kw)
    y_reg = ops.conv2d(x0, wt, cout, k, k, stride, (pad, pad), algo=_lib.ALGO_REG_STAGED, **kw)
    torch.cuda.synchronize()
    assert torch.equal(y_auto, y_tap) and torch.equal(y_tap, y_one)
    d = (y_tap.float() - y_reg.float()).abs()
    assert float(d.max()) <= 2.0 ** -9 * float(y_reg.float().abs().max())
    xs = torch.cat([x0[:1], x1[:1]], dim=3) if c1 else x0[:1]
    xs = xs.float().cpu().permute(0, 3, 1, 2).clone()
    xs[0, :, :, int(vw[0]):] = 0
    ref = F.leaky_relu(F.conv2d(xs, wt.float().cpu().permute(0, 3, 1, 2), stride=stride, padding=pad) + bias.cpu()[None, :, None, None], 0.2) * 2 ** 0.5
    _check("conv fuzz %s" % (case,), y_auto[:1].float().cpu().permute(0, 3, 1, 2), ref, dtype, extra=2.0)


def test_conv_lds_dma_eligibility():
    ops = _ops()
    from marconet_amd._lib import MarconetHipError
    x = torch.zeros(1, 4, 4, 32, dtype=torch.float16, device=DEV)
    w = torch.zeros(64, 3, 3, 32, dtype=torch.float16, device=DEV)
    with pytest.raises(MarconetHipError):
        ops.conv2d(x, w, 64, 3, 3, (1, 1), (1, 1), algo=2)        # cin % 64 != 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_sr_postprocess_matches_script(dtype):
    """K19 (test_sr.py:198-200): float output == the script's torch/numpy sequence bit for bit; uint8 == its rint + saturate"""
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    y = ((torch.rand((3, 16, 40, 8), generator=g) - 0.5) * 3.0).to(dtype)          # beyond [-1,1] to exercise the clip
    sr = y[..., :3].float().permute(0, 3, 1, 2)                                     # what modelSR would return (NCHW)
    want = np.clip((sr * 0.5 + 0.5).permute(0, 2, 3, 1).flip(3).numpy(), 0, 1) * 255.0
    got_f = ops.sr_postprocess(y.to(DEV), u8=False).cpu().numpy()
    got_u = ops.sr_postprocess(y.to(DEV), u8=True).cpu().numpy()
    assert np.array_equal(got_f, want.astype(np.float32))
    assert np.array_equal(got_u, np.rint(want).astype(np.uint8))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("shape", [(2, 8, 32), (3, 19, 45), (1, 128, 2048)])
def test_conv3x3_rgb_kernel(dtype, shape):
    """conv_final.6 + tanh through the dedicated 64 → 3 kernel (NHWC and fp32-NCHW outputs) vs F.conv2d, incl. partial tiles"""
    ops = _ops()
    n, h, w = shape
    x = _q(_rnd((n, 64, h, w), 301), dtype)
    wt = _q(_rnd((3, 64, 3, 3), 302, 1.0 / math.sqrt(576)), dtype)
    bias = _rnd((3,), 303, 0.2)
    ref = torch.tanh(F.conv2d(x, wt, bias=bias, padding=1))
    y_nhwc, y_nchw = ops.conv3x3_rgb(_nhwc(x, dtype), _pack_w(wt, dtype), bias.to(DEV), ops.ACT_TANH, nhwc=True, nchw=True)
    torch.cuda.synchronize()
    _check("conv3x3_rgb nhwc %s %s" % (shape, dtype), _nchw(y_nhwc)[:, :3], ref, dtype, extra=2.0)
    assert float(y_nhwc[..., 3:].abs().max()) == 0.0
    assert torch.equal(y_nchw.cpu(), _nchw(y_nhwc)[:, :3].contiguous())


@pytest.mark.parametrize("case", [(64, 512, 512, 5, True), (64, 2048, 512, 0, True), (64, 512, 6736, 0, False), (100, 512, 2048, 5, False),
                                  (512, 48, 20, 1, True), (1, 16, 4, 6, False), (130, 272, 36, 4, True)])
def test_skinny_linear_equals_general_kernel(case):
    """the fp32 skinny kernel (TextViT linears of a small batch) gives the register-staged kernel's bits — an image's result must
    not depend on which of the two the batch size selects — and both match torch within the fp32 tolerance"""
    ops = _ops()
    from marconet_amd import _lib
    m, k, cout, act, with_res = case
    x = _rnd((m, k), 61).to(DEV)
    wt = _rnd((cout, k), 62, 1.0 / math.sqrt(k)).to(DEV)
    bias = _rnd((cout,), 63, 0.3).to(DEV)
    res = _rnd((m, cout), 64).to(DEV) if with_res else None
    kw = dict(bias=bias, act=act, residual=None if res is None else res.reshape(1, 1, m, cout))
    y_s = ops.conv2d(x.reshape(1, 1, m, k), wt, cout, algo=_lib.ALGO_SKINNY, **kw)
    y_g = ops.conv2d(x.reshape(1, 1, m, k), wt, cout, algo=_lib.ALGO_REG_STAGED, **kw)
    y_a = ops.conv2d(x.reshape(1, 1, m, k), wt, cout, **kw)
    torch.cuda.synchronize()
    assert torch.equal(y_s, y_g) and torch.equal(y_a, y_s)
    ref = x.double() @ wt.double().t() + bias.double()
    if res is not None:
        ref = ref + res.double()
    ref = {0: lambda v: v, 1: torch.relu, 4: torch.tanh, 5: lambda v: F.gelu(v), 6: torch.sigmoid}[act](ref)
    assert (y_s.reshape(m, cout).double() - ref).abs().max().item() < 2e-4


@pytest.mark.parametrize("case", [(1, 8, 512, 512, 8, 512, 32), (3, 8, 64, 32, 8, 20, 8), (2, 4, 40, 16, 4, 36, 4), (5, 1, 7, 64, 1, 128, 2)])
def test_patchify_conv_splitk(case):
    """mnet_conv2d_splitk (TextViT patch embedding of a small batch): equals the general kernel up to the association of the fp32
    sum, is deterministic, and matches torch"""
    ops = _ops()
    n, h, w, c, k, cout, ksplit = case
    x = _rnd((n, c, h, w), 71)
    wt = _rnd((cout, c, k, k), 72, 1.0 / math.sqrt(c * k * k))
    bias = _rnd((cout,), 73, 0.3).to(DEV)
    wo = w // k
    res = _rnd((1, 1, wo, cout), 74).to(DEV)
    kw = dict(bias=bias, residual=res, res_mod=wo, act=5)
    x0 = _nhwc(x, torch.float32)
    y_s = ops.conv2d(x0, _pack_w(wt, torch.float32), cout, k, k, (k, k), (0, 0), splitk=ksplit, **kw)
    y_s2 = ops.conv2d(x0, _pack_w(wt, torch.float32), cout, k, k, (k, k), (0, 0), splitk=ksplit, **kw)
    y_g = ops.conv2d(x0, _pack_w(wt, torch.float32), cout, k, k, (k, k), (0, 0), **kw)
    torch.cuda.synchronize()
    assert torch.equal(y_s, y_s2)
    assert torch.allclose(y_s, y_g, rtol=1e-4, atol=2e-5)          # K up to 32768 fp32 terms, associated differently
    ref = F.conv2d(x.double(), wt.double(), stride=k) + bias.cpu().double().view(1, -1, 1, 1)
    ref = ref + res.cpu().double().reshape(wo, cout).t()[None, :, None, :].expand(n, cout, h // k, wo)
    _check("patchify split-K %s" % (case,), _nchw(y_s), F.gelu(ref).float(), torch.float32)
    with pytest.raises(RuntimeError):
        ops.conv2d(x0, _pack_w(wt, torch.float32), cout, k, k, (k, k), (0, 0), splitk=ksplit * 3 + 1, **kw)


@pytest.mark.parametrize("dtype_name", ["fp32", "fp16", "fp16x3"])
@pytest.mark.parametrize("sn", [False, True])
def test_pack_weights_on_device_matches_host_packing(dtype_name, sn):
    """mnet_pack_weights (spectral-norm fold + scale + OIHW → [O][KH][KW][I] repack + padding, in the library) against the torch
    host path of packing.pack_conv_weight on the same tensors"""
    from marconet_amd import packing as P
    ops = _ops()
    dtype = P.torch_dtype(dtype_name)
    w = _rnd((70, 40, 3, 3), 201, 0.05)
    u, v = _rnd((70,), 202), _rnd((360,), 203)
    u, v = u / u.norm(), v / v.norm()
    snv = (u, v) if sn else None
    host = P.pack_conv_weight(w, dtype, scale=0.37, sn=snv)
    dev = P.pack_conv_weight(w.to(DEV), dtype, scale=0.37, sn=None if not sn else (u.to(DEV), v.to(DEV)))
    assert dev.is_cuda and dev.shape == host.shape and dev.dtype == host.dtype
    if dtype == P.SPLIT_DTYPE:
        a, b = P.unsplit_halves(dev.cpu()), P.unsplit_halves(host)
    else:
        a, b = dev.float().cpu(), host.float()
    # sigma: fp64 sums on the device vs torch's fp32 mv/dot on the host → 1e-6 relative; plus one rounding of the storage type
    tol = {"fp32": 3e-6, "fp16": 1.5e-3, "fp16x3": 3e-6}[dtype_name]
    assert (a - b).abs().max().item() <= tol * b.abs().max().item()
    assert a.shape[0] >= 72 and float(a[70:].abs().max()) == 0.0                                        # zero padding (cout 70 → 72 / 96)
    assert a.shape[3] == 40 or float(a[:, :, :, 40:].abs().max()) == 0.0                               # (cin 40 → 64 in split-half)


def test_pack_wsq_linear_and_gather_rows():
    from marconet_amd import packing as P
    ops = _ops()
    w = _rnd((48, 24, 3, 3), 211, 0.1)
    a, b = P.pack_wsq(w.to(DEV), 0.25).cpu(), P.pack_wsq(w, 0.25)
    assert a.shape == (24, 48) and (a - b).abs().max().item() <= 2e-6 * b.abs().max().item()
    lw = _rnd((33, 17), 212)
    assert torch.equal(P.pack_linear_weight(lw.to(DEV), 0.5).cpu(), P.pack_linear_weight(lw, 0.5))
    src = _rnd((9, 40), 213)
    idx = torch.tensor([8, 0, 0, 3, 7, 7, 1], dtype=torch.int64)
    assert torch.equal(ops.gather_rows(src.to(DEV), 5, 20, idx.to(DEV)).cpu(), src[idx][:, 5:25])
    assert torch.equal(ops.gather_rows(src.to(DEV), 8, 32).cpu(), src[:, 8:40])
    assert torch.equal(ops.gather_rows(src.to(DEV), idx=idx.to(DEV)).cpu(), src[idx])


def test_style_rows_and_scaled_demod():
    """mnet_style_rows (row window + gather + power-of-two normalisation) and mnet_demod_scaled: the normalised evaluation equals the
    plain one exactly — demod' = 2^e demod, rows' = 2^-e rows"""
    ops = _ops()
    src = _rnd((7, 96), 221) * torch.tensor([1e-3, 1.0, 37.0, 4e4, 0.0, 0.26, 1.0]).reshape(7, 1)     # row 4: all zeros
    idx = torch.tensor([3, 3, 0, 6, 4, 2, 1, 5], dtype=torch.int64)
    rows, eps, sb = ops.style_rows(src.to(DEV), 16, 64, idx.to(DEV), bcast=8)
    plain = src[idx][:, 16:80]
    m = plain.abs().amax(dim=1)
    e = torch.where(m > 0, torch.frexp(m).exponent.float(), torch.zeros_like(m))
    assert torch.equal(rows.cpu(), plain * torch.exp2(-e)[:, None])
    assert torch.equal(eps.cpu(), torch.exp2(-2 * e)) and torch.equal(sb.cpu(), torch.exp2(e)[:, None].expand(8, 8))
    nz = m > 0
    assert bool(((rows.cpu().abs().amax(dim=1)[nz] >= 0.5) & (rows.cpu().abs().amax(dim=1)[nz] < 1.0)).all())
    wsq = _rnd((64, 40), 222).abs() * 0.01
    d0 = ops.demod(plain.contiguous().to(DEV), wsq.to(DEV))
    d1 = ops.demod(rows, wsq.to(DEV), eps)
    want = d0.cpu() * torch.exp2(e)[:, None]
    assert float(((d1.cpu() - want).abs() / want).max()) <= 2e-7                     # rsqrt of an argument scaled by 4^-e
    r2, eps2, none = ops.style_rows(src.to(DEV), 0, 96)
    assert none is None and r2.shape == (7, 96) and float(r2[4].abs().max()) == 0.0 and float(eps2[4]) == 1.0


@pytest.mark.parametrize("dtype", ALL_DTYPES)
@pytest.mark.parametrize("c,hw", [(128, (6, 10)), (512, (6, 10)), (256, (64, 64)), (128, (36, 50)), (64, (10, 6)), (512, (2, 2))])
def test_torgb_kernel(c, hw, dtype):
    """ToRGB.forward (models/networks.py:313-321): modulated 1x1 conv (no demodulation) + bias + bilinear x2 of the skip + tanh
    (the larger maps: several trips per workgroup, a ragged last trip)"""
    ops = _ops()
    n, (h, w) = 3, hw
    x = _q(_rnd((n, c, h, w), 90), dtype)
    wt = _rnd((3, c), 91, 1.0 / math.sqrt(c))
    style = _rnd((n, c), 92).abs() + 0.5
    sb = _rnd((n, 1), 93).abs() + 0.5
    bias = torch.cat([_rnd((3,), 94, 0.2), torch.zeros(1)])
    skip = torch.tanh(_rnd((n, 3, h // 2, w // 2), 95))
    skip4 = torch.cat([skip, torch.zeros(n, 1, h // 2, w // 2)], dim=1).permute(0, 2, 3, 1).contiguous()
    conv = F.conv2d(x * style[:, :, None, None], wt.reshape(3, c, 1, 1)) * sb.reshape(n, 1, 1, 1) + bias[:3].reshape(1, 3, 1, 1)
    for sk, ref in ((None, torch.tanh(conv)),
                    (skip4, torch.tanh(conv + F.interpolate(skip, scale_factor=2, mode="bilinear", align_corners=False)))):
        y = ops.torgb(_nhwc(x, dtype), wt.to(DEV), style.to(DEV), sb.to(DEV), bias.to(DEV), None if sk is None else sk.to(DEV))
        torch.cuda.synchronize()
        assert y.dtype == torch.float32 and y.shape == (n, h, w, 4) and float(y[..., 3].abs().max()) == 0.0
        _check("torgb c=%d skip=%s %s" % (c, sk is not None, dtype), y[..., :3].cpu().permute(0, 3, 1, 2), ref, torch.float32)
    y1 = ops.torgb(_nhwc(x, dtype), wt.to(DEV), style.to(DEV), None, bias.to(DEV), None)          # no scale_b
    _check("torgb no scale_b %s" % dtype, y1[..., :3].cpu().permute(0, 3, 1, 2),
           torch.tanh(F.conv2d(x * style[:, :, None, None], wt.reshape(3, c, 1, 1)) + bias[:3].reshape(1, 3, 1, 1)), torch.float32)


def test_f16_diagnostic_tile_ids_are_refused():
    """MNET_F16 LDS-DMA ids 11-15 are wrong-on-purpose diagnostic builds: not reachable through the C-ABI unless the process opts in"""
    ops = _ops()
    from marconet_amd._lib import MarconetHipError
    x = _nhwc(_rnd((1, 64, 8, 16), 96), torch.float16)
    wt = _pack_w(_rnd((64, 64, 3, 3), 97, 0.05), torch.float16)
    for id_ in (11, 12, 13, 14, 15):
        with pytest.raises(MarconetHipError, match="diagnostic"):
            ops.conv2d(x, wt, 64, 3, 3, (1, 1), (1, 1), algo=16 + id_)
    ops.conv2d(x, wt, 64, 3, 3, (1, 1), (1, 1), algo=16 + 3)          # a production id still launches
    torch.cuda.synchronize()
