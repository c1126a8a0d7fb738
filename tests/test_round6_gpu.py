"""-m gpu: round-6 additions.
  * the module-level helpers of the reference's models/networks.py (:492-493, :518-533) re-exported by the drop-in package, on the HIP kernels;
  * GroupNorm sums of the conv epilogue on maps whose |mean| >> std; the two fp16+8 lo-byte encoders on blocks whose largest half is a
    subnormal (conv epilogue: v_cvt_pk_fp8_f32 of a pre-scaled residual; streaming kernels: v_cvt_scalef32_pk_fp8_f32) against the host packer."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def test_reference_module_level_helpers():
    from models.networks import adaptive_instance_normalization, calc_mean_std_4D, swish
    x = _rnd((2, 5, 7, 13), 1, 3.0)                       # 91 elements per plane: the padded path
    y = swish(x.to(DEV)).cpu()
    assert y.shape == x.shape and (y - x * torch.sigmoid(x)).abs().max().item() <= 2e-6
    for shape, seed in (((2, 64, 16, 16), 2), ((3, 7, 5, 9), 3)):
        f = _rnd(shape, seed, 2.0) + _rnd(shape[:2] + (1, 1), seed + 10, 4.0)
        p = _rnd(shape, seed + 20, 0.7) - 1.5
        mean, std = calc_mean_std_4D(f.to(DEV))
        b, c = shape[:2]
        rv = f.view(b, c, -1).var(dim=2) + 1e-5               # models/networks.py:518-525
        assert mean.shape == (b, c, 1, 1) and std.shape == (b, c, 1, 1)
        assert (mean.cpu().view(b, c) - f.view(b, c, -1).mean(dim=2)).abs().max().item() <= 2e-5
        assert (std.cpu().view(b, c) / rv.sqrt() - 1).abs().max().item() <= 2e-5
        got = adaptive_instance_normalization(p.to(DEV), f.to(DEV)).cpu()
        pm, ps = p.view(b, c, -1).mean(2).view(b, c, 1, 1), (p.view(b, c, -1).var(2) + 1e-5).sqrt().view(b, c, 1, 1)
        fm, fs = f.view(b, c, -1).mean(2).view(b, c, 1, 1), rv.sqrt().view(b, c, 1, 1)
        ref = (p - pm) / ps * fs + fm                           # models/networks.py:527-533
        assert got.shape == p.shape and (got - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("ratio", [30.0, 100.0])
def test_groupnorm_epilogue_sums_when_the_mean_dwarfs_the_spread(ratio):
    """the epilogue folds 32 values per lane and a 32-lane tree in fp32 before the fp64 fold (the pass it replaced was fp64 throughout);
    var = SS/n - mean^2 then cancels.  A conv output with |mean| / std = 30 ... 100 (a large bias): the affine from the epilogue sums against the
    fp64 statistics of the stored map."""
    from marconet_amd import mxfmt, ops, packing
    n, h, w, cin, cout = 2, 32, 64, 64, 256
    x = _rnd((n, h, w, cin), 11)
    x = mxfmt.unpack_act(mxfmt.pack_act(x), cin)
    wt = _rnd((cout, cin, 3, 3), 12, 1.0 / math.sqrt(cin * 9))
    bias = torch.full((cout,), ratio) * (1.0 + 0.1 * _rnd((cout,), 13))      # conv output has std ~1: |mean| / std ~ ratio
    gamma, beta = _rnd((cout,), 14).abs() + 0.5, _rnd((cout,), 15, 0.2)
    xd = ops.convert(x.to(DEV), packing.MX_DTYPE)
    wp = packing.pack_conv_weight(wt, packing.MX_DTYPE).to(DEV)
    part = ops.gn_partial_buffer(n, h, w, cout, DEV)
    y = ops.conv2d(xd, wp, cout, 3, 3, (1, 1), (1, 1), bias=bias.to(DEV), gn_partial=part)
    sc, sh = ops.groupnorm_affine_from_partial(part, n, h, w, cout, gamma.to(DEV), beta.to(DEV), 1e-6, None)
    yf = ops.convert(y, torch.float32).cpu().double()                          # [n, h, w, cout]
    g = yf.reshape(n, h * w, cout // 32, 32)
    mean, var = g.mean(dim=(1, 3)), g.var(dim=(1, 3), unbiased=False)
    rstd = (1.0 / (var + 1e-6).sqrt()).repeat_interleave(32, dim=1)
    mu = mean.repeat_interleave(32, dim=1)
    ref_sc = gamma.double()[None] * rstd
    ref_sh = beta.double()[None] - mu * ref_sc
    # what matters downstream is the normalised value: y * scale + shift over the map, O(1) by construction
    got = yf * sc.cpu().double()[:, None, None, :] + sh.cpu().double()[:, None, None, :]
    ref = yf * ref_sc[:, None, None, :] + ref_sh[:, None, None, :]
    err = (got - ref).abs().max().item()
    print("ratio %.0f: normalised map max|d| = %.3e (|mean|/std measured %.1f)" % (ratio, err, (mean.abs() / var.sqrt()).mean().item()))
    assert err <= 2e-4


def test_lo_byte_encoders_agree_on_subnormal_blocks():
    """blocks whose largest half is an fp16 subnormal (max |v| in [2^-24, 2^-14)): the residual * 2^11 / s reaches 2^13 ... 2^17, beyond
    e4m3's 448.  The conv epilogue (hm_encode_lo_ref), the streaming kernels (hm_encode_lo, the scaled conversion) and the host packer must write the
    SAME finite bytes — saturated, never NaN — and decode to the same values."""
    from marconet_amd import mxfmt, ops, packing
    cout, cin, n, h, w = 256, 64, 1, 8, 32
    g = torch.Generator().manual_seed(77)
    # per block (8 of them) a maximum in a different binade of the subnormal range, full fp32 mantissas below it
    mags = torch.tensor([2.0 ** e for e in (-24, -23, -22, -20, -18, -16, -15, -14.2)])
    bias = ((torch.rand((8, 32), generator=g) * 2 - 1) * mags[:, None]).reshape(-1).float()
    bias[5] = 0.0
    want = bias[None, None, None, :].expand(n, h, w, cout).contiguous()
    host = mxfmt.pack_act(want)
    assert not torch.isnan(host[..., :].view(n, h, w, cout // 32, 128)[..., 64:96].contiguous().view(torch.float8_e4m3fn).float()).any()
    stream = ops.convert(want.to(DEV), packing.MX_DTYPE)                       # streaming-kernel encoder
    x = torch.zeros((n, h, w, cin))
    xd = ops.convert(x.to(DEV), packing.MX_DTYPE)
    wp = packing.pack_conv_weight(torch.zeros((cout, cin, 3, 3)), packing.MX_DTYPE).to(DEV)
    y = ops.conv2d(xd, wp, cout, 3, 3, (1, 1), (1, 1), bias=bias.to(DEV))      # LDS-DMA epilogue encoder: v = bias exactly
    assert ops.plan_is_lds_dma(ops.conv_plan(xd, cout, 3, 3, (1, 1), (1, 1)))
    torch.cuda.synchronize()
    sb, yb = stream.cpu().view(torch.uint8).reshape(host.shape), y.cpu().view(torch.uint8).reshape(host.shape)
    assert torch.equal(sb, host), "streaming encoder differs from the host packer on subnormal blocks"
    assert torch.equal(yb, host), "conv epilogue encoder differs from the host packer on subnormal blocks"
    dec = ops.convert(y, torch.float32).cpu()
    assert torch.isfinite(dec).all() and torch.equal(dec, mxfmt.unpack_act(host, cout))


# ---------------------------------------------------------------------------------------------------------------------
# round 6, last pass: the streaming kernels' new launch shapes against plain torch on the values the storage holds
def _mx_roundtrip(x_nchw):
    """NCHW fp32 -> (fp16+8 NHWC tensor on the device, the values it holds as NCHW fp32 on the host)"""
    from marconet_amd import ops, packing
    xs = ops.convert(x_nchw.permute(0, 2, 3, 1).contiguous().to(DEV), packing.MX_DTYPE)
    return xs, ops.convert(xs, torch.float32).cpu().permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize("shape", [(8, 64, 9, 12), (16, 32, 4, 4), (3, 64, 13, 5), (8, 512, 6, 6)])
def test_upsample_row_runs_in_xcd_aware_order(shape):
    """up-sample: runs of 4 input rows per thread (a last, shorter run when H % 4 != 0) and the XCD-aware workgroup order, taken when the (image, workgroup) count is a
    multiple of 8 (first, second and last case) and not (third) — bilinear x2 of the stored values, within the storage's rounding"""
    from marconet_amd import ops
    xs, xv = _mx_roundtrip(_rnd(shape, 301, 1.5))
    ref = F.interpolate(xv, scale_factor=2, mode="bilinear", align_corners=False)
    y = ops.convert(ops.upsample2x(xs), torch.float32).cpu().permute(0, 3, 1, 2)
    assert y.shape == ref.shape and (y - ref).abs().max().item() <= 3e-5 * max(1.0, ref.abs().max().item())
    sc = (_rnd((shape[0], shape[1]), 302).abs() + 0.5)
    y16 = ops.upsample2x(xs, scale=sc.to(DEV), out_dtype=torch.float16).float().cpu().permute(0, 3, 1, 2)
    ref16 = ref * sc[:, :, None, None]
    assert (y16 - ref16).abs().max().item() <= 1e-3 * max(1.0, ref16.abs().max().item())


@pytest.mark.parametrize("shape", [(3, 1024, 5, 9), (2, 1024, 32, 32), (5, 512, 7, 9), (2, 64, 33, 17)])
def test_groupnorm_apply_one_trip_and_two_chunks_per_thread(shape):
    """GroupNorm apply + swish: one trip per thread on every map; two chunks per thread where a pixel has >= 128 chunks (C = 1024), with a tail workgroup whose second
    chunk lies beyond the image (first case: 5 760 chunks = 11 workgroups of 512 + a quarter)"""
    from marconet_amd import ops
    n, c, h, w = shape
    xs, xv = _mx_roundtrip(_rnd(shape, 311, 2.0) + 0.3)
    sc, sh = _rnd((n, c), 312).abs() + 0.5, _rnd((n, c), 313)
    for swish in (True, False):
        t = xv * sc[:, :, None, None] + sh[:, :, None, None]
        ref = t * torch.sigmoid(t) if swish else t
        y = ops.convert(ops.affine_act(xs, sc.to(DEV), sh.to(DEV), swish=swish), torch.float32).cpu().permute(0, 3, 1, 2)
        assert (y - ref).abs().max().item() <= 4e-5 * max(1.0, ref.abs().max().item())
    y_in_place = ops.affine_act(xs, sc.to(DEV), sh.to(DEV), swish=True, out=xs)
    assert y_in_place is xs
    t = xv * sc[:, :, None, None] + sh[:, :, None, None]
    assert (ops.convert(xs, torch.float32).cpu().permute(0, 3, 1, 2) - t * torch.sigmoid(t)).abs().max().item() <= 4e-5 * max(1.0, t.abs().max().item())


def test_glyph_scatter_row_runs_with_a_short_last_run():
    """scatter: runs of 8 rows per thread; S = 12 leaves a run of 4 (the model's maps are 32 / 64 rows); columns no glyph owns are copied"""
    from marconet_amd import ops
    B, C, S, FW = 2, 64, 12, 40
    feat_s, feat = _mx_roundtrip(_rnd((B, C, S, FW), 321))
    windows = [(0, 3, 9), (0, 8, 12), (1, 25, 12)]                      # (image, x1, width): the second glyph overwrites columns 8..11 of the first
    G = len(windows)
    sc_s, sc = _mx_roundtrip(_rnd((G, C, S, S), 322) * 0.5)
    sh_s, sh = _mx_roundtrip(_rnd((G, C, S, S), 323) * 0.5)
    g_start = torch.tensor([0, 2, 3], dtype=torch.int32)
    g_x1 = torch.tensor([wd[1] for wd in windows], dtype=torch.int32)
    g_w = torch.tensor([wd[2] for wd in windows], dtype=torch.int32)
    out = ops.convert(ops.glyph_scatter_affine(feat_s, sc_s, sh_s, g_start.to(DEV), g_x1.to(DEV), g_w.to(DEV)), torch.float32).cpu().permute(0, 3, 1, 2)
    ref = feat.clone()
    for g, (b, x1, gw) in enumerate(windows):                            # models/networks.py:446-449 — res = f * scale + shift on the window, last writer wins; out = ori + res
        f = feat[b, :, :, x1:x1 + gw]
        ref[b, :, :, x1:x1 + gw] = f + (f * sc[g, :, :, :gw] + sh[g, :, :, :gw])
    assert (out - ref).abs().max().item() <= 4e-5 * max(1.0, ref.abs().max().item())
    assert torch.equal(out[0, :, :, 20:], feat[0, :, :, 20:]) and torch.equal(out[1, :, :, :25], feat[1, :, :, :25])
