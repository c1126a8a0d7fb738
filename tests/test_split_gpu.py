"""-m gpu: the split-half storage type (MNET_F16X2, precision mode "fp16x3") — every kernel that reads or writes it, through
the C-ABI, against a plain PyTorch fp32/fp64 CPU reference of the same op computed from the SAME split-representable values.

A split-half element holds hi + lo with hi = f16(v), lo = f16(v - hi): 22 significant bits.  Inputs are first rounded
through that representation (``_q``), so the only differences left are the dropped lo*lo product (2^-22 relative per
product), fp32 accumulation order and the final split rounding of the output: tolerance 4e-6 relative to the output scale
for convolutions (vs 2.5e-3 for plain fp16 storage and 2e-5 for the exact-fp32 kernels' summation order)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 4e-6


def _ops():
    from marconet_amd import ops
    return ops


def _P():
    from marconet_amd import packing
    return packing


def _rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def _q(t):
    """round through the split-half representation"""
    hi = t.to(torch.float16)
    lo = (t - hi.float()).to(torch.float16)
    return hi.float() + lo.float()


def _to_split(t_nchw):
    """fp32 NCHW (cpu) → split-half NHWC on the device, through the library's own converter"""
    ops, P = _ops(), _P()
    x = t_nchw.permute(0, 2, 3, 1).contiguous().to(DEV)
    return ops.convert(x, P.SPLIT_DTYPE)


def _from_split(t):
    ops = _ops()
    return ops.convert(t, torch.float32).cpu().permute(0, 3, 1, 2).contiguous()


def _pack_w(w):
    return _P().pack_conv_weight(w, _P().SPLIT_DTYPE).to(DEV)


def _check(name, got, ref, tol=TOL):
    scale = max(ref.abs().max().item(), 1e-6)
    err = (got.double() - ref.double()).abs().max().item()
    print("%-52s max|d|=%.3e  ref max=%.3e  rel=%.3e" % (name, err, scale, err / scale))
    assert err <= tol * scale, "%s: err %.3e > %.1e x scale %.3e" % (name, err, tol, scale)


def test_convert_round_trip_and_layout():
    ops, P = _ops(), _P()
    x = _rnd((3, 5, 7, 64), 1) * torch.logspace(-3, 2, 64)            # six decades of magnitude across the channels
    s = ops.convert(x.to(DEV), P.SPLIT_DTYPE)
    assert s.dtype == P.SPLIT_DTYPE and s.shape == x.shape
    back = ops.convert(s, torch.float32).cpu()
    assert ((back - x).abs() <= x.abs() * 2.0 ** -21 + 2.0 ** -24).all()          # 22 bits, or the subnormal floor of lo
    assert torch.equal(back, _q(x))                                                 # exactly the (f16, f16) pair arithmetic
    assert torch.equal(P.unsplit_halves(s.cpu()), back)                             # device layout == host packing layout
    assert torch.equal(P.split_halves(x).view(torch.float16), s.cpu().view(torch.float16))
    h = ops.convert(ops.convert(s, torch.float16), torch.float32).cpu()             # split → plain half
    assert torch.equal(h, back.to(torch.float16).float())


def test_layout_kernels_nchw():
    ops, P = _ops(), _P()
    x = _rnd((2, 3, 6, 10), 2)
    s = ops.nchw_to_nhwc(x.to(DEV), P.SPLIT_DTYPE, c_ld=32)
    assert s.shape == (2, 6, 10, 32)
    f = ops.convert(s, torch.float32).cpu()
    assert torch.equal(f[..., :3], _q(x).permute(0, 2, 3, 1)) and float(f[..., 3:].abs().max()) == 0.0
    assert torch.equal(ops.nhwc_to_nchw(s, c=3).cpu(), _q(x))
    y = _rnd((2, 96, 4, 5), 3)
    assert torch.equal(ops.nhwc_to_nchw(_to_split(y)).cpu(), _q(y))


CONV_CASES = [
    # n, h, w, c0, c1, cout, k, stride, pad, algo (0 auto: LDS-DMA when eligible, 1 register-staged)
    (2, 9, 13, 64, 0, 128, 3, (1, 1), 1, 0),
    (2, 9, 13, 64, 0, 128, 3, (1, 1), 1, 1),
    (1, 16, 24, 32, 0, 64, 3, (1, 1), 1, 0),
    (3, 7, 11, 32, 0, 32, 3, (2, 1), 1, 0),
    (2, 12, 20, 64, 0, 256, 3, (2, 2), 1, 0),
    (2, 8, 8, 128, 0, 160, 1, (1, 1), 0, 0),
    (2, 8, 16, 256, 128, 256, 3, (1, 1), 1, 0),
    (2, 8, 16, 256, 128, 256, 3, (1, 1), 1, 1),
    (1, 32, 32, 512, 0, 256, 3, (1, 1), 1, 0),
    (4, 6, 10, 32, 32, 64, 1, (2, 1), 0, 0),
    (1, 64, 1024, 256, 0, 256, 3, (1, 1), 1, 0),         # the 256x256 persistent tile (>= 65536 pixels)
    (1, 64, 1024, 128, 0, 128, 3, (1, 1), 1, 0),         # 128x512 tile
    (1, 64, 1024, 64, 0, 64, 3, (1, 1), 1, 0),           # 64x512 tile
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_plain(case):
    ops = _ops()
    n, h, w, c0, c1, cout, k, stride, pad, algo = case
    x = _q(_rnd((n, c0 + c1, h, w), 11))
    wt = _rnd((cout, c0 + c1, k, k), 12, 1.0 / math.sqrt((c0 + c1) * k * k))
    wq = _q(wt * 256.0) / 256.0                                      # the packed weights hold hi/lo of 256*W
    ref = F.conv2d(x.double(), wq.double(), stride=stride, padding=pad)
    x0 = _to_split(x[:, :c0])
    x1 = _to_split(x[:, c0:]) if c1 else None
    y = ops.conv2d(x0, _pack_w(wt), cout, k, k, stride, (pad, pad), x1=x1, algo=algo)
    torch.cuda.synchronize()
    _check("split conv %s" % (case,), _from_split(y), ref)


@pytest.mark.parametrize("id_", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12])
def test_every_lds_dma_tile_configuration(id_):
    """each split-half instantiation of the LDS-DMA kernel, pinned explicitly, with the full epilogue"""
    ops = _ops()
    n, h, w, cin, cout = 2, 16, 40, 128, 288
    x = _q(_rnd((n, cin, h, w), 21))
    wt = _rnd((cout, cin, 3, 3), 22, 1.0 / math.sqrt(cin * 9))
    wq = _q(wt * 256.0) / 256.0
    bias = _rnd((cout,), 23, 0.3)
    osc = _rnd((n, cout), 24).abs() + 0.5
    psc = _rnd((n, cout), 25).abs() + 0.5
    res = _q(_rnd((n, cout, h, w), 26))
    vw = torch.tensor([40, 23], dtype=torch.int32)
    xm = x.clone()
    xm[1, :, :, 23:] = 0
    ref = F.conv2d(xm.double(), wq.double(), padding=1) * osc[:, :, None, None].double() + bias[None, :, None, None].double() + res.double()
    ref = F.leaky_relu(ref, 0.2) * 2 ** 0.5 * psc[:, :, None, None].double()
    y = ops.conv2d(_to_split(x), _pack_w(wt), cout, 3, 3, (1, 1), (1, 1), out_scale=osc.to(DEV), bias=bias.to(DEV),
                   residual=_to_split(res), act=3, post_scale=psc.to(DEV), valid_w=vw.to(DEV), algo=16 + id_)
    torch.cuda.synchronize()
    _check("split LDS-DMA id %d" % id_, _from_split(y), ref, tol=6e-6)


@pytest.mark.parametrize("act", [0, 1, 2, 3, 4, 5, 6])
def test_conv_epilogue_register_staged(act):
    ops = _ops()
    n, h, w, cin, cout = 3, 10, 14, 32, 96
    ACT = {0: lambda v: v, 1: F.relu, 2: lambda v: F.leaky_relu(v, 0.2), 3: lambda v: F.leaky_relu(v, 0.2) * 2 ** 0.5,
           4: torch.tanh, 5: F.gelu, 6: torch.sigmoid}
    x = _q(_rnd((n, cin, h, w), 31))
    wt = _rnd((cout, cin, 3, 3), 32, 1.0 / math.sqrt(cin * 9))
    wq = _q(wt * 256.0) / 256.0
    bias = _rnd((cout,), 33, 0.3)
    osc = _rnd((n, cout), 34).abs() + 0.5
    res = _q(_rnd((n, cout, h, w), 35))
    ref = ACT[act](F.conv2d(x.double(), wq.double(), padding=1) * osc[:, :, None, None].double() + bias[None, :, None, None].double() + res.double())
    y = ops.conv2d(_to_split(x), _pack_w(wt), cout, 3, 3, (1, 1), (1, 1), out_scale=osc.to(DEV), bias=bias.to(DEV),
                   residual=_to_split(res), act=act)
    torch.cuda.synchronize()
    _check("split conv epilogue act=%d" % act, _from_split(y), ref, tol=8e-6)        # tanhf / erff / expf: ~1 ulp each


@pytest.mark.parametrize("swish", [False, True])
def test_conv_input_transform(swish):
    """style modulation (x * s[n,i]) / GroupNorm affine + swish applied to the staged slab: hi + lo → fp32 → split again"""
    ops = _ops()
    n, h, w, cin, cout = 4, 8, 12, 64, 32
    x = _q(_rnd((n, cin, h, w), 41))
    wt = _rnd((cout, cin, 3, 3), 42, 1.0 / math.sqrt(cin * 9))
    wq = _q(wt * 256.0) / 256.0
    sc = _rnd((n, cin), 43).abs() + 0.5
    sh = _rnd((n, cin), 44, 0.2) if swish else None
    t = x * sc[:, :, None, None] + (sh[:, :, None, None] if swish else 0.0)
    if swish:
        t = t * torch.sigmoid(t)
    ref = F.conv2d(_q(t).double(), wq.double(), padding=1)             # the kernel re-splits the transformed value
    y = ops.conv2d(_to_split(x), _pack_w(wt), cout, 3, 3, (1, 1), (1, 1), in_scale=sc.to(DEV),
                   in_shift=None if sh is None else sh.to(DEV), in_swish=swish)
    torch.cuda.synchronize()
    _check("split conv prologue swish=%s" % swish, _from_split(y), ref, tol=2e-5 if swish else 8e-6)


def test_pointwise_kernels_match_the_fp32_kernels():
    """upsample / affine+swish / GroupNorm statistics / embedding gather on split-half against the fp32 kernels on the same
    values: equal up to the split rounding of the stored result (2^-22) and fma contraction"""
    ops, P = _ops(), _P()
    x = _q(_rnd((3, 64, 6, 10), 51))
    xs, xf = _to_split(x), x.permute(0, 2, 3, 1).contiguous().to(DEV)
    sc = (_rnd((3, 64), 52).abs() + 0.5).to(DEV)
    sh = _rnd((3, 64), 53, 0.3).to(DEV)
    for name, a, b in (("upsample2x", ops.upsample2x(xs), ops.upsample2x(xf)),
                       ("upsample2x*scale", ops.upsample2x(xs, scale=sc), ops.upsample2x(xf, scale=sc)),
                       ("affine", ops.affine_act(xs, sc, sh), ops.affine_act(xf, sc, sh)),
                       ("affine+swish", ops.affine_act(xs, sc, sh, swish=True), ops.affine_act(xf, sc, sh, swish=True))):
        got = ops.convert(a, torch.float32).cpu()
        # the same fp32 arithmetic, then one split rounding (the two instantiations may contract mul+add into fma differently)
        _check("split vs fp32 " + name, got, b.cpu(), tol=1e-6)
    g, be = (_rnd((64,), 54).abs() + 0.5).to(DEV), _rnd((64,), 55, 0.2).to(DEV)
    vw = torch.tensor([10, 7, 3], dtype=torch.int32, device=DEV)
    for v in (None, vw):
        s1, h1 = ops.groupnorm_affine(xs, g, be, 1e-6, v)
        s2, h2 = ops.groupnorm_affine(xf, g, be, 1e-6, v)
        assert (s1 - s2).abs().max().item() <= 1e-6 * s2.abs().max().item() and (h1 - h2).abs().max().item() <= 1e-6
    emb = _rnd((50, 64), 56).to(DEV)
    lab = torch.tensor([[3], [49], [0]], device=DEV)
    e1 = ops.convert(ops.embed_gather(emb, lab, P.SPLIT_DTYPE, 50), torch.float32)
    e2 = ops.embed_gather(emb, lab, torch.float32, 50)
    assert torch.equal(e1.cpu(), _q(e2.cpu()))
    sc = (_rnd((3, 64), 57) + 1.5).to(DEV)
    e3 = ops.convert(ops.embed_gather(emb, lab, P.SPLIT_DTYPE, 50, scale=sc), torch.float32)
    e4 = ops.embed_gather(emb, lab, torch.float32, 50, scale=sc)
    assert torch.equal(e4.cpu(), (emb[lab[:, 0]] * sc).cpu().reshape(3, 1, 1, 64).expand(3, 4, 4, 64)) and torch.equal(e3.cpu(), _q(e4.cpu()))


@pytest.mark.parametrize("G", [5, 300])
def test_glyph_kernels_match_the_fp32_kernels(G):
    """AdaIN + crop + concat (+ closed-form GroupNorm affine), both launch forms, and the ordered scatter"""
    ops = _ops()
    import numpy as np
    from marconet_amd.glyphs import GlyphTables
    B, S, C, FW = max(1, G // 10), 16, 64, 256
    counts = [G // B + (1 if b < G % B else 0) for b in range(B)]
    rng = np.random.default_rng(5)
    locs = np.zeros((B, 2 * max(counts)), dtype=np.float32)
    locs[:, 0::2] = rng.random((B, max(counts))).astype(np.float32)
    tab = GlyphTables(locs, counts, FW, S // 2, DEV)
    prior = _q(_rnd((G, C, S, S), 61))
    feat = _q(_rnd((B, C, S, FW), 62))
    g, be = (_rnd((2 * C,), 63).abs() + 0.5).to(DEV), _rnd((2 * C,), 64, 0.2).to(DEV)
    ps, pf = _to_split(prior), prior.permute(0, 2, 3, 1).contiguous().to(DEV)
    fs, ff = _to_split(feat), feat.permute(0, 2, 3, 1).contiguous().to(DEV)
    for split in (False, True):
        o1, s1, h1 = ops.adain_crop_concat_gn(ps, fs, tab.g_img, tab.g_x1, tab.g_y1, tab.g_w, g, be, 1e-6, split=split)
        o2, s2, h2 = ops.adain_crop_concat_gn(pf, ff, tab.g_img, tab.g_x1, tab.g_y1, tab.g_w, g, be, 1e-6, split=split)
        _check("split vs fp32 adain (split=%s)" % split, ops.convert(o1, torch.float32).cpu(), o2.cpu(), tol=1e-6)
        assert (s1 - s2).abs().max().item() <= 1e-6 * s2.abs().max().item() and (h1 - h2).abs().max().item() <= 1e-6 * max(1.0, h2.abs().max().item())
    sc, sh = _q(_rnd((G, C, S, S), 65)), _q(_rnd((G, C, S, S), 66))
    a = ops.glyph_scatter_affine(fs, _to_split(sc), _to_split(sh), tab.g_start, tab.g_x1, tab.g_w)
    b = ops.glyph_scatter_affine(ff, sc.permute(0, 2, 3, 1).contiguous().to(DEV), sh.permute(0, 2, 3, 1).contiguous().to(DEV),
                                 tab.g_start, tab.g_x1, tab.g_w)
    _check("split vs fp32 scatter", ops.convert(a, torch.float32).cpu(), b.cpu(), tol=1e-6)


def test_conv3x3_rgb_split_input():
    ops = _ops()
    x = _q(_rnd((2, 64, 16, 40), 71))
    wt = _rnd((3, 64, 3, 3), 72, 1.0 / 24)
    bias = _rnd((3,), 73, 0.1)
    wr = wt.permute(0, 2, 3, 1).contiguous().to(DEV)
    y1, y2 = ops.conv3x3_rgb(_to_split(x), wr, bias.to(DEV), ops.ACT_TANH, nhwc=True, nchw=True)
    z1, z2 = ops.conv3x3_rgb(x.permute(0, 2, 3, 1).contiguous().to(DEV), wr, bias.to(DEV), ops.ACT_TANH, nhwc=True, nchw=True)
    # the fp32 arithmetic of the fp32 kernel after staging (32 channels at a time: another fp32 summation order)
    assert y1.dtype == torch.float32 and (y1 - z1).abs().max().item() <= 5e-6 and (y2 - z2).abs().max().item() <= 5e-6
    ref = torch.tanh(F.conv2d(x.double(), wt.double(), padding=1) + bias[None, :, None, None].double())
    _check("conv3x3_rgb split", y2.cpu(), ref, tol=2e-6)


def _fuzz_cases():
    import random
    rng = random.Random(20260927)
    cases = []
    for i in range(18):
        k = rng.choice([1, 3, 3, 3])
        cin = 32 * rng.randint(1, 8)
        c1 = rng.choice([0, 0, 32, 64]) if cin > 64 else 0
        cout = rng.choice([32, 64, 96, 128, 160, 256, 288, 512])
        stride = rng.choice([(1, 1), (1, 1), (1, 1), (2, 1), (2, 2)]) if k == 3 else (1, 1)
        big = i % 3 == 0                                   # every third case: > 256 pixel tiles → the persistent grid makes several passes
        h, w = (rng.randint(100, 180), rng.randint(300, 420)) if big else (rng.randint(3, 40), rng.randint(8, 70))
        n = rng.randint(1, 2) if big else rng.randint(1, 9)
        cases.append((n, h, w, cin - c1, c1, cout, k, stride))
    cases += [(4, 64, 64, 64, 0, 256, 3, (1, 1)), (2, 128, 128, 128, 0, 64, 3, (1, 1)), (300, 16, 16, 64, 0, 128, 3, (1, 1)),   # >= 65536 pixels: the big tiles
              (4, 128, 128, 64, 0, 64, 3, (1, 1)), (1, 32, 2048, 96, 0, 64, 3, (1, 1)), (64, 32, 32, 32, 0, 96, 3, (1, 1))]       # strip kernel (cout < 128)
    return cases


@pytest.mark.parametrize("case", _fuzz_cases())
def test_conv_fuzz_all_paths_agree(case):
    """seeded random shapes (pixel / cout tails, concat, strides, 1x1, ragged widths, multi-pass persistent grids): whatever AUTO
    picks == the same launch with one workgroup per tile, bit for bit; the LDS-DMA and the register-staged kernel agree to fp32
    summation order; and the result matches F.conv2d (fp64) on a slice"""
    ops, P = _ops(), _P()
    from marconet_amd import _lib
    n, h, w, c0, c1, cout, k, stride = case
    pad = k // 2
    g = torch.Generator().manual_seed(hash(case) & 0xffff)
    xa = _q(torch.rand((n, c0 + c1, h, w), generator=g) - 0.5)
    wt = (torch.rand((cout, c0 + c1, k, k), generator=g) - 0.5) * (2.0 / math.sqrt(k * k * (c0 + c1)))
    bias = (torch.rand((cout,), generator=g) - 0.5)
    x0 = _to_split(xa[:, :c0])
    x1 = _to_split(xa[:, c0:]) if c1 else None
    vw = torch.tensor([max(1, w - (i % 4) * 2) for i in range(n)], dtype=torch.int32, device=DEV)
    kw = dict(x1=x1, valid_w=vw, bias=bias.to(DEV), act=ops.ACT_LRELU_SQRT2)
    wp = _pack_w(wt)
    y_auto = ops.conv2d(x0, wp, cout, k, k, stride, (pad, pad), algo=0, **kw)
    y_reg = ops.conv2d(x0, wp, cout, k, k, stride, (pad, pad), algo=_lib.ALGO_REG_STAGED, **kw)
    a = ops.convert(y_auto, torch.float32)
    r = ops.convert(y_reg, torch.float32)
    if cout >= 64 and (c0 + c1) % 32 == 0:
        y_one = ops.conv2d(x0, wp, cout, k, k, stride, (pad, pad), algo=_lib.ALGO_LDS_DMA | _lib.ALGO_FLAG_ONE_TILE, **kw)
        assert torch.equal(y_auto.view(torch.float16), y_one.view(torch.float16))
        y_tap = ops.conv2d(x0, wp, cout, k, k, stride, (pad, pad), algo=_lib.ALGO_DMA_CFG0 + 3, **kw)      # a pinned per-tap tile
        assert torch.equal(y_auto.view(torch.float16), y_tap.view(torch.float16))                          # (AUTO may be the strip kernel)
    torch.cuda.synchronize()
    assert float((a - r).abs().max()) <= 4e-6 * float(r.abs().max())
    xs = xa[:1].clone()
    xs[0, :, :, int(vw[0]):] = 0
    wq = _q(wt * 256.0) / 256.0
    ref = F.leaky_relu(F.conv2d(xs.double(), wq.double(), stride=stride, padding=pad) + bias[None, :, None, None].double(), 0.2) * 2 ** 0.5
    _check("split conv fuzz %s" % (case,), a[:1].cpu().permute(0, 3, 1, 2), ref, tol=6e-6)
