"""-m gpu: the "fp16+8" storage (MNET_F16M, precision mode "fp16x2") — every kernel that reads or writes it, through the C-ABI.

An element holds hi = f16(v) and an e4m3 byte for (v - hi) under one E8M0 scale per (pixel, 32-channel block): ~16 significant
bits relative to the block's largest value.  References:
  * storage kernels: the host packers of marconet_amd/mxfmt.py (byte-exact, decoded values exact);
  * LDS-DMA convs (hi*hi on the f16 MFMA + one block-scaled fp8 MFMA for w_lo8*x_hi8 + w_hi8*x_lo8): a plain PyTorch fp32 CPU
    evaluation of exactly those three products (``_emulate``; agreement to fp32 summation order, 3e-6) AND the fp64 convolution
    of the stored values (4e-5: what the fp8 rounding of the correction operands leaves; a missing or mis-scaled correction term
    shows as 3e-4);
  * register-staged convs (the lo bytes decoded to halves, then the split-half three-product sequence): fp64 convolution of the
    decoded operands (6e-6)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL_EMU, TOL_MX, TOL_X3 = 3e-6, 4e-5, 6e-6


def _ops():
    from marconet_amd import ops
    return ops


def _P():
    from marconet_amd import packing
    return packing


def _M():
    from marconet_amd import mxfmt
    return mxfmt


def _rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def _q(t_nchw):
    """round an NCHW fp32 tensor through the fp16+8 representation (blocks of 32 channels per pixel)"""
    M = _M()
    x = t_nchw.permute(0, 2, 3, 1).contiguous()
    return M.unpack_act(M.pack_act(x), x.shape[-1]).permute(0, 3, 1, 2).contiguous()


def _to_mx(t_nchw):
    """fp32 NCHW (cpu) → fp16+8 NHWC on the device, through the library's own converter"""
    return _ops().convert(t_nchw.permute(0, 2, 3, 1).contiguous().to(DEV), _P().MX_DTYPE)


def _from_mx(t):
    return _ops().convert(t, torch.float32).cpu().permute(0, 3, 1, 2).contiguous()


def _pack_w(w):
    return _P().pack_conv_weight(w, _P().MX_DTYPE).to(DEV)


def _wdec(w):
    """the weights a register-staged fp16+8 conv multiplies with: hi + lo8 * s * 2^-11 of 256 W, per output channel scale"""
    M = _M()
    wn = w.permute(0, 2, 3, 1).contiguous() * M.WSCALE
    hi = wn.to(torch.float16).float()
    m = hi.abs().reshape(w.shape[0], -1).amax(-1)
    s = torch.pow(2.0, ((M._floor_log2(m) - 7 + 127).clamp(11, 254) - 127).float()).reshape(-1, 1, 1, 1)
    lo = M._e4m3((wn - hi) * 2048.0 / s).float() * s / 2048.0
    return ((hi + lo) / M.WSCALE).permute(0, 3, 1, 2).contiguous()


def _emulate(x, w, **kw):
    """fp32 evaluation of the three products the LDS-DMA fp16+8 kernel forms (x [N,C,H,W] already representable, w true weights)"""
    M = _M()
    xn = x.permute(0, 2, 3, 1)
    xb = xn.reshape(xn.shape[:-1] + (-1, 32))
    xh = xb.to(torch.float16).float()
    xs = torch.pow(2.0, (M.block_e8(xh) - 127).float())
    xh8 = M._e4m3(xh / xs).float() * xs
    xl8 = M._e4m3((xb - xh) * 2048.0 / xs).float() * xs / 2048.0
    back = lambda t: t.reshape(xn.shape).permute(0, 3, 1, 2)
    wn = w.permute(0, 2, 3, 1) * M.WSCALE
    wh = wn.to(torch.float16).float()
    m = wh.abs().reshape(w.shape[0], -1).amax(-1)
    ws = torch.pow(2.0, ((M._floor_log2(m) - 7 + 127).clamp(11, 254) - 127).float()).reshape(-1, 1, 1, 1)
    wh8 = M._e4m3(wh / ws).float() * ws
    wl8 = M._e4m3((wn - wh) * 2048.0 / ws).float() * ws / 2048.0
    wb = lambda t: t.permute(0, 3, 1, 2)
    y = F.conv2d(back(xh), wb(wh), **kw) + F.conv2d(back(xl8), wb(wh8), **kw) + F.conv2d(back(xh8), wb(wl8), **kw)
    return y / M.WSCALE


def _tile(id_):
    """algo value that pins LDS-DMA tile configuration ``id_`` (ids 0-15: MNET_CONV_ALGO_DMA_CFG0 + id; 16...: MNET_CONV_ALGO_DMA_CFG16 + id - 16)"""
    from marconet_amd import _lib
    return _lib.ALGO_DMA_CFG0 + id_ if id_ < 16 else _lib.ALGO_DMA_CFG16 + (id_ - 16)


def _check(name, got, ref, tol):
    scale = max(ref.abs().max().item(), 1e-6)
    err = (got.double() - ref.double()).abs().max().item()
    print("%-56s max|d|=%.3e  ref max=%.3e  rel=%.3e" % (name, err, scale, err / scale))
    assert err <= tol * scale, "%s: err %.3e > %.1e x scale %.3e" % (name, err, tol, scale)


def test_convert_round_trip_and_layout():
    ops, P, M = _ops(), _P(), _M()
    x = _rnd((3, 5, 7, 64), 1) * torch.logspace(-3, 2, 64)            # five decades of magnitude across the channels
    x[1, 2] *= 30.0
    x[2, :, 3, :32] = 0.0                                             # an all-zero block
    s = ops.convert(x.to(DEV), P.MX_DTYPE)
    assert s.dtype == P.MX_DTYPE and s.shape == x.shape
    host = M.pack_act(x)
    assert torch.equal(s.cpu().view(torch.uint8), host)               # device layout == host packing, byte for byte
    back = ops.convert(s, torch.float32).cpu()
    assert torch.equal(back, M.unpack_act(host, 64))
    blockmax = x.abs().reshape(3, 5, 7, 2, 32).amax(-1, keepdim=True).expand(3, 5, 7, 2, 32).reshape(x.shape)
    assert ((back - x).abs() <= blockmax * 2.0 ** -15 + 1e-30).all()   # hi half + 4-bit lo under the block scale
    h = ops.convert(ops.convert(s, torch.float16), torch.float32).cpu()             # fp16+8 → plain half
    assert torch.equal(h, back.to(torch.float16).float())
    s3 = ops.convert(s, P.SPLIT_DTYPE)                                              # → split half: exact up to the f16 subnormal step of lo
    assert float((ops.convert(s3, torch.float32).cpu() - back).abs().max()) <= 2.0 ** -25


def test_layout_kernels_nchw():
    ops, P = _ops(), _P()
    x = _rnd((2, 3, 6, 10), 2)
    s = ops.nchw_to_nhwc(x.to(DEV), P.MX_DTYPE, c_ld=32)
    assert s.shape == (2, 6, 10, 32)
    xp = torch.zeros((2, 32, 6, 10))
    xp[:, :3] = x
    f = ops.convert(s, torch.float32).cpu()
    assert torch.equal(f.permute(0, 3, 1, 2), _q(xp))
    assert torch.equal(s.cpu().view(torch.uint8), _M().pack_act(xp.permute(0, 2, 3, 1).contiguous()))
    assert torch.equal(ops.nhwc_to_nchw(s, c=3).cpu(), _q(xp)[:, :3])
    y = _rnd((2, 96, 4, 5), 3)
    assert torch.equal(ops.nhwc_to_nchw(_to_mx(y)).cpu(), _q(y))


def test_pack_weights_device_equals_host():
    ops, P, M = _ops(), _P(), _M()
    w = _rnd((96, 64, 3, 3), 4, 0.05)
    w[5] *= 30.0
    w[7] = 0.0
    u, v = _rnd((96,), 5), _rnd((64 * 9,), 6)
    host = P.pack_conv_weight(w, P.MX_DTYPE, sn=(u, v))                 # torch path (CPU tensors)
    dev = P.pack_conv_weight(w.to(DEV), P.MX_DTYPE, sn=(u.to(DEV), v.to(DEV)))
    assert host.shape == dev.shape == (P.mx_weight_rows(96, 3, 3, 64), 3, 3, 64)
    hb, db = host.view(torch.uint8).reshape(-1), dev.cpu().view(torch.uint8).reshape(-1)
    nb = 96 * 9 * 64 * 4
    assert torch.equal(hb[nb:nb + 96], db[nb:nb + 96])                # the per-channel scale bytes
    # sigma is an fp32 dot product on the host and an fp64 one on the device: the halves may differ in the last place
    dh = (hb[:nb].reshape(-1, 128)[:, :64].contiguous().view(torch.float16).float() - db[:nb].reshape(-1, 128)[:, :64].contiguous().view(torch.float16).float()).abs()
    assert float(dh.max()) <= 2.0 ** -10 * float(hb[:nb].reshape(-1, 128)[:, :64].contiguous().view(torch.float16).float().abs().max())
    plain = P.pack_conv_weight(w.to(DEV), P.MX_DTYPE)                  # no spectral norm: byte-exact
    assert torch.equal(plain.cpu().view(torch.uint8), P.pack_conv_weight(w, P.MX_DTYPE).view(torch.uint8))


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
    from marconet_amd import _lib
    n, h, w, c0, c1, cout, k, stride, pad, algo = case
    x = _rnd((n, c0 + c1, h, w), 11)
    x[0, :, : max(1, h // 3)] *= 50.0                                # block scales follow the data
    x = torch.cat([_q(x[:, :c0]), _q(x[:, c0:])], dim=1) if c1 else _q(x)
    wt = _rnd((cout, c0 + c1, k, k), 12, 1.0 / math.sqrt((c0 + c1) * k * k))
    wt[min(3, cout - 1)] *= 20.0
    x0 = _to_mx(x[:, :c0])
    x1 = _to_mx(x[:, c0:]) if c1 else None
    wp = _pack_w(wt)
    plan_dma = cout >= 64 and (c0 + c1) % 32 == 0 and c0 % 32 == 0 and algo != 1
    y = ops.conv2d(x0, wp, cout, k, k, stride, (pad, pad), x1=x1, algo=algo)
    torch.cuda.synchronize()
    got = _from_mx(y)
    if plan_dma:
        # the stored result adds one fp16+8 rounding (2^-16 of the block's largest value) to the arithmetic
        # |stored - exact| <= half a lo step (2^-16 of the block's largest value) on top of the fp32 summation order
        _check("fp16+8 conv vs emulation %s" % (case,), got, _emulate(x, wt, stride=stride, padding=pad), 2e-5)
        _check("fp16+8 conv vs fp64     %s" % (case,), got, F.conv2d(x.double(), wt.double(), stride=stride, padding=pad), TOL_MX)
    else:
        _check("fp16+8 conv (register-staged) %s" % (case,), got, F.conv2d(x.double(), _wdec(wt).double(), stride=stride, padding=pad), 2e-5)


@pytest.mark.parametrize("id_", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15, 16])
def test_every_lds_dma_tile_configuration(id_):
    """each fp16+8 instantiation of the LDS-DMA kernel, pinned explicitly, with the full epilogue"""
    ops = _ops()
    n, h, w, cin, cout = 2, 16, 40, 128, 288
    x = _q(_rnd((n, cin, h, w), 21))
    wt = _rnd((cout, cin, 3, 3), 22, 1.0 / math.sqrt(cin * 9))
    bias = _rnd((cout,), 23, 0.3)
    osc = _rnd((n, cout), 24).abs() + 0.5
    psc = _rnd((n, cout), 25).abs() + 0.5
    res = _q(_rnd((n, cout, h, w), 26))
    vw = torch.tensor([40, 23], dtype=torch.int32)
    xm = x.clone()
    xm[1, :, :, 23:] = 0
    def full(conv):
        r = conv * osc[:, :, None, None] + bias[None, :, None, None] + res
        return F.leaky_relu(r, 0.2) * 2 ** 0.5 * psc[:, :, None, None]
    y = ops.conv2d(_to_mx(x), _pack_w(wt), cout, 3, 3, (1, 1), (1, 1), out_scale=osc.to(DEV), bias=bias.to(DEV),
                   residual=_to_mx(res), act=3, post_scale=psc.to(DEV), valid_w=vw.to(DEV), algo=_tile(id_))
    torch.cuda.synchronize()
    got = _from_mx(y)
    _check("fp16+8 LDS-DMA id %d vs emulation" % id_, got, full(_emulate(xm, wt, padding=1)), 2e-5)
    _check("fp16+8 LDS-DMA id %d vs fp64" % id_, got, full(F.conv2d(xm.double(), wt.double(), padding=1)), TOL_MX)


def test_lds_dma_tiles_agree_bit_for_bit():
    """every fp16+8 tile runs the same MFMA sequence per output: a conv gives the same bytes whatever tile its launch size selects"""
    ops = _ops()
    n, h, w, cin, cout = 1, 24, 64, 96, 256
    x = _to_mx(_rnd((n, cin, h, w), 27))
    wp = _pack_w(_rnd((cout, cin, 3, 3), 28, 0.03))
    outs = [ops.conv2d(x, wp, cout, 3, 3, (1, 1), (1, 1), act=2, algo=_tile(i)).cpu().view(torch.uint8) for i in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15, 16)]
    for o in outs[1:]:
        assert torch.equal(outs[0], o)


@pytest.mark.parametrize("case", [
    # cout, cin, k, (n, h, w), software-pipelined id, lock-step id of the same tile
    (256, 64, 3, (4, 128, 160), 15, 6),       # 320 pixel tiles on <= 256 workgroups: the slab stream crosses tiles, some workgroups run two
    (256, 32, 1, (4, 128, 160), 15, 6),       # one slab per tile: every iteration closes a tile
    (128, 64, 3, (2, 128, 320), 9, 8),        # 160 tiles of 128 x 512 (10 DMA pieces per wave per slab)
    (288, 96, 3, (3, 96, 100), 15, 6),        # cout tail (two channel tiles: the weight scales change per tile), pixel tail, images smaller than a tile row
    (512, 512, 3, (20, 32, 32), 15, 6),       # 144 slabs per tile, 160 tiles, two channel tiles
])
def test_software_pipelined_tiles_equal_the_lock_step_tiles(case):
    """round 4, ids 15 / 9 (the slab loop pipelined across the barrier: scaled MFMAs of slab s-1 + DMA pieces of slab s+1, then the f16 MFMAs of
    slab s): same MFMA sequence per output as ids 6 / 8, hence the same bytes — on multi-pass persistent grids, with one slab per tile, with
    tails, ragged widths, concat-free and the full epilogue (bias, residual, activation)"""
    ops = _ops()
    cout, cin, k, (n, h, w), swp, ref = case
    x = _to_mx(_rnd((n, cin, h, w), 61))
    wp = _pack_w(_rnd((cout, cin, k, k), 62, 1.0 / math.sqrt(cin * k * k)))
    bias = _rnd((cout,), 63, 0.3).to(DEV)
    res = _to_mx(_rnd((n, cout, h, w), 64))
    vw = torch.tensor([w - 7 * (i % 3) for i in range(n)], dtype=torch.int32, device=DEV)
    outs = []
    for i in (ref, swp, swp):
        y = ops.conv2d(x, wp, cout, k, k, (1, 1), (k // 2, k // 2), bias=bias, residual=res, act=3, valid_w=vw, algo=16 + i)
        torch.cuda.synchronize()
        outs.append(y.cpu().view(torch.uint8))
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])


def test_software_pipelined_tile_with_concat_sources():
    """the second concat source (x1) in the pipelined tile: conv_body_32.0's shape class (256 + 64 channels)"""
    ops = _ops()
    n, h, w, c0, c1, cout = 40, 32, 64, 64, 32, 256
    x0, x1 = _to_mx(_rnd((n, c0, h, w), 81)), _to_mx(_rnd((n, c1, h, w), 82))
    wp = _pack_w(_rnd((cout, c0 + c1, 3, 3), 83, 1.0 / math.sqrt((c0 + c1) * 9)))
    bias = _rnd((cout,), 84, 0.3).to(DEV)
    outs = [ops.conv2d(x0, wp, cout, 3, 3, (1, 1), (1, 1), x1=x1, bias=bias, act=2, algo=a_).cpu().view(torch.uint8) for a_ in (16 + 6, 16 + 15)]
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("case", [
    # n, h, w, cin, cout
    (40, 64, 64, 64, 256),        # W < 256: a tile is 4 whole rows; 640 pixel tiles on <= 256 workgroups (multi-pass persistent grid)
    (2, 64, 1024, 96, 256),       # W >= 256: a tile is 256 pixels of one row; left / right halo from the neighbours
    (300, 16, 16, 64, 512),       # W = 16 (the smallest strip row), two channel tiles, images smaller than a tile
    (5, 128, 128, 32, 256),       # one 32-channel slice: 9 slabs per tile, the strip cursor crosses tiles every 3 slabs
])
def test_strip_256x256_tile_equals_the_per_tap_tile(case):
    """round 4: the 8-wave 256x256 strip tile of the fp16+8 mode (one activation strip per filter row instead of one slab per tap:
    -31 % L2->LDS bytes) runs the MFMA sequence of the per-tap tiles per output -> byte-identical results, with ragged widths, bias,
    residual and the full epilogue"""
    ops = _ops()
    from marconet_amd import _lib
    n, h, w, cin, cout = case
    x = _to_mx(_rnd((n, cin, h, w), 71))
    wp = _pack_w(_rnd((cout, cin, 3, 3), 72, 1.0 / math.sqrt(cin * 9)))
    bias = _rnd((cout,), 73, 0.3).to(DEV)
    res = _to_mx(_rnd((n, cout, h, w), 74))
    vw = torch.tensor([w - 5 * (i % 3) for i in range(n)], dtype=torch.int32, device=DEV)
    outs = []
    for algo in (_lib.ALGO_DMA_CFG0 + 6, _lib.ALGO_STRIP_CFG0 + 0, _lib.ALGO_STRIP_CFG0 + 0):
        y = ops.conv2d(x, wp, cout, 3, 3, (1, 1), (1, 1), bias=bias, residual=res, act=3, valid_w=vw, algo=algo)
        torch.cuda.synchronize()
        outs.append(y.cpu().view(torch.uint8))
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])


@pytest.mark.parametrize("act", [0, 1, 2, 3, 4, 5, 6])
def test_conv_epilogue_register_staged(act):
    ops = _ops()
    n, h, w, cin, cout = 3, 10, 14, 32, 96
    ACT = {0: lambda v: v, 1: F.relu, 2: lambda v: F.leaky_relu(v, 0.2), 3: lambda v: F.leaky_relu(v, 0.2) * 2 ** 0.5,
           4: torch.tanh, 5: F.gelu, 6: torch.sigmoid}
    x = _q(_rnd((n, cin, h, w), 31))
    wt = _rnd((cout, cin, 3, 3), 32, 1.0 / math.sqrt(cin * 9))
    bias = _rnd((cout,), 33, 0.3)
    osc = _rnd((n, cout), 34).abs() + 0.5
    res = _q(_rnd((n, cout, h, w), 35))
    ref = ACT[act](F.conv2d(x.double(), _wdec(wt).double(), padding=1) * osc[:, :, None, None].double() + bias[None, :, None, None].double() + res.double())
    y = ops.conv2d(_to_mx(x), _pack_w(wt), cout, 3, 3, (1, 1), (1, 1), out_scale=osc.to(DEV), bias=bias.to(DEV),
                   residual=_to_mx(res), act=act, algo=1)
    torch.cuda.synchronize()
    _check("fp16+8 conv epilogue act=%d" % act, _from_mx(y), ref, 2e-5)


@pytest.mark.parametrize("swish", [False, True])
def test_conv_input_transform(swish):
    ops = _ops()
    n, h, w, cin, cout = 4, 8, 12, 64, 32
    x = _q(_rnd((n, cin, h, w), 41))
    wt = _rnd((cout, cin, 3, 3), 42, 1.0 / math.sqrt(cin * 9))
    sc = _rnd((n, cin), 43).abs() + 0.5
    sh = _rnd((n, cin), 44, 0.2) if swish else None
    t = x * sc[:, :, None, None] + (sh[:, :, None, None] if swish else 0.0)
    if swish:
        t = t * torch.sigmoid(t)
    ref = F.conv2d(t.double(), _wdec(wt).double(), padding=1)          # the transformed value is re-split into two halves (22 bits)
    y = ops.conv2d(_to_mx(x), _pack_w(wt), cout, 3, 3, (1, 1), (1, 1), in_scale=sc.to(DEV),
                   in_shift=None if sh is None else sh.to(DEV), in_swish=swish)
    torch.cuda.synchronize()
    _check("fp16+8 conv prologue swish=%s" % swish, _from_mx(y), ref, 3e-5)


def test_pointwise_kernels_match_torch():
    """upsample / affine+swish / GroupNorm statistics / embedding gather on fp16+8 tensors against PyTorch on the same values"""
    ops, P = _ops(), _P()
    x = _q(_rnd((3, 64, 6, 10), 51))
    xs = _to_mx(x)
    sc = _rnd((3, 64), 52).abs() + 0.5
    sh = _rnd((3, 64), 53, 0.3)
    up = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
    aff = x * sc[:, :, None, None] + sh[:, :, None, None]
    for name, got, ref in (("upsample2x", ops.upsample2x(xs), up),
                           ("upsample2x*scale", ops.upsample2x(xs, scale=sc.to(DEV)), up * sc[:, :, None, None]),
                           ("affine", ops.affine_act(xs, sc.to(DEV), sh.to(DEV)), aff),
                           ("affine+swish", ops.affine_act(xs, sc.to(DEV), sh.to(DEV), swish=True), aff * torch.sigmoid(aff))):
        g = _from_mx(got)
        _check("fp16+8 vs torch " + name, g, ref, 2e-5)
        assert torch.equal(g, _q(g))                                   # what was stored is a valid fp16+8 tensor (idempotent)
    g, be = _rnd((64,), 54).abs() + 0.5, _rnd((64,), 55, 0.2)
    vw = torch.tensor([10, 7, 3], dtype=torch.int32)
    for v in (None, vw):
        s1, h1 = ops.groupnorm_affine(xs, g.to(DEV), be.to(DEV), 1e-6, None if v is None else v.to(DEV))
        xm = x.clone()
        s_ref, h_ref = torch.zeros(3, 64), torch.zeros(3, 64)
        for n_ in range(3):
            wv = 10 if v is None else int(v[n_])
            xv = xm[n_, :, :, :wv].double().reshape(2, 32, -1)
            mean, var = xv.mean(dim=(1, 2)), xv.var(dim=(1, 2), unbiased=False)
            rstd = 1.0 / torch.sqrt(var + 1e-6)
            s_ref[n_] = (g.double() * rstd.repeat_interleave(32)).float()
            h_ref[n_] = (be.double() - mean.repeat_interleave(32) * g.double() * rstd.repeat_interleave(32)).float()
        assert (s1.cpu() - s_ref).abs().max().item() <= 2e-6 * s_ref.abs().max().item() and (h1.cpu() - h_ref).abs().max().item() <= 2e-6
    emb = _rnd((50, 64), 56)
    lab = torch.tensor([[3], [49], [0]], device=DEV)
    e1 = ops.convert(ops.embed_gather(emb.to(DEV), lab, P.MX_DTYPE, 50), torch.float32).cpu()      # [3,4,4,64]
    want = emb[[3, 49, 0]].reshape(3, 1, 1, 64).expand(3, 4, 4, 64)
    assert torch.equal(e1, _q(want.permute(0, 3, 1, 2).contiguous()).permute(0, 2, 3, 1))
    sc = (_rnd((3, 64), 57) + 1.5)
    e2 = ops.convert(ops.embed_gather(emb.to(DEV), lab, P.MX_DTYPE, 50, scale=sc.to(DEV)), torch.float32).cpu()      # ·scale[i, c] before the rounding
    want2 = (emb[[3, 49, 0]] * sc).reshape(3, 1, 1, 64).expand(3, 4, 4, 64)
    assert torch.equal(e2, _q(want2.permute(0, 3, 1, 2).contiguous()).permute(0, 2, 3, 1))


@pytest.mark.parametrize("G", [5, 300])
def test_glyph_kernels_match_the_fp32_kernels(G):
    """AdaIN + crop + concat (+ closed-form GroupNorm affine), both launch forms, and the ordered scatter: the fp16+8 instantiation
    against the fp32 one (itself pinned to PyTorch in tests/test_kernels_gpu.py) on the same values"""
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
    ps, pf = _to_mx(prior), prior.permute(0, 2, 3, 1).contiguous().to(DEV)
    fs, ff = _to_mx(feat), feat.permute(0, 2, 3, 1).contiguous().to(DEV)
    for split in (False, True):
        o1, s1, h1 = ops.adain_crop_concat_gn(ps, fs, tab.g_img, tab.g_x1, tab.g_y1, tab.g_w, g, be, 1e-6, split=split)
        o2, s2, h2 = ops.adain_crop_concat_gn(pf, ff, tab.g_img, tab.g_x1, tab.g_y1, tab.g_w, g, be, 1e-6, split=split)
        _check("fp16+8 vs fp32 adain (split=%s)" % split, ops.convert(o1, torch.float32).cpu(), o2.cpu(), 2e-5)
        assert (s1 - s2).abs().max().item() <= 1e-6 * s2.abs().max().item() and (h1 - h2).abs().max().item() <= 1e-6 * max(1.0, h2.abs().max().item())
    sc, sh = _q(_rnd((G, C, S, S), 65)), _q(_rnd((G, C, S, S), 66))
    a = ops.glyph_scatter_affine(fs, _to_mx(sc), _to_mx(sh), tab.g_start, tab.g_x1, tab.g_w)
    b = ops.glyph_scatter_affine(ff, sc.permute(0, 2, 3, 1).contiguous().to(DEV), sh.permute(0, 2, 3, 1).contiguous().to(DEV),
                                 tab.g_start, tab.g_x1, tab.g_w)
    _check("fp16+8 vs fp32 scatter", ops.convert(a, torch.float32).cpu(), b.cpu(), 2e-5)


def test_conv3x3_rgb_mx_input():
    ops = _ops()
    x = _q(_rnd((2, 64, 16, 40), 71))
    wt = _rnd((3, 64, 3, 3), 72, 1.0 / 24)
    bias = _rnd((3,), 73, 0.1)
    wr = wt.permute(0, 2, 3, 1).contiguous().to(DEV)
    y1, y2 = ops.conv3x3_rgb(_to_mx(x), wr, bias.to(DEV), ops.ACT_TANH, nhwc=True, nchw=True)
    assert y1.dtype == torch.float32
    ref = torch.tanh(F.conv2d(x.double(), wt.double(), padding=1) + bias[None, :, None, None].double())
    _check("conv3x3_rgb fp16+8", y2.cpu(), ref, 2e-6)
    assert torch.equal(y1.cpu()[..., :3].permute(0, 3, 1, 2), y2.cpu())


def _fuzz_cases():
    import random
    rng = random.Random(20260928)
    cases = []
    for i in range(14):
        k = rng.choice([1, 3, 3, 3])
        cin = 32 * rng.randint(1, 8)
        c1 = rng.choice([0, 0, 32, 64]) if cin > 64 else 0
        cout = rng.choice([32, 64, 96, 128, 160, 256, 288, 512])
        stride = rng.choice([(1, 1), (1, 1), (1, 1), (2, 1), (2, 2)]) if k == 3 else (1, 1)
        big = i % 3 == 0
        h, w = (rng.randint(100, 180), rng.randint(300, 420)) if big else (rng.randint(3, 40), rng.randint(8, 70))
        n = rng.randint(1, 2) if big else rng.randint(1, 9)
        cases.append((n, h, w, cin - c1, c1, cout, k, stride))
    cases += [(4, 64, 64, 64, 0, 256, 3, (1, 1)), (2, 128, 128, 128, 0, 64, 3, (1, 1)), (300, 16, 16, 64, 0, 128, 3, (1, 1)),
              (1, 32, 2048, 96, 0, 64, 3, (1, 1)), (4, 128, 128, 64, 0, 64, 3, (1, 1)), (64, 32, 32, 32, 0, 96, 3, (1, 1))]       # strip kernel (cout < 128)
    return cases


@pytest.mark.parametrize("case", _fuzz_cases())
def test_conv_fuzz_all_paths_agree(case):
    """seeded random shapes (pixel / cout tails, concat, strides, 1x1, ragged widths, multi-pass persistent grids): whatever AUTO
    picks == the same launch with one workgroup per tile == a pinned small tile, byte for byte; the LDS-DMA and the register-staged
    kernel agree to the fp8 rounding of the correction operands; and the result matches F.conv2d (fp64) on a slice"""
    ops = _ops()
    from marconet_amd import _lib
    n, h, w, c0, c1, cout, k, stride = case
    pad = k // 2
    g = torch.Generator().manual_seed(hash(case) & 0xffff)
    xa = torch.rand((n, c0 + c1, h, w), generator=g) - 0.5
    xa = torch.cat([_q(xa[:, :c0]), _q(xa[:, c0:])], dim=1) if c1 else _q(xa)
    wt = (torch.rand((cout, c0 + c1, k, k), generator=g) - 0.5) * (2.0 / math.sqrt(k * k * (c0 + c1)))
    bias = (torch.rand((cout,), generator=g) - 0.5)
    x0 = _to_mx(xa[:, :c0])
    x1 = _to_mx(xa[:, c0:]) if c1 else None
    vw = torch.tensor([max(1, w - (i % 4) * 2) for i in range(n)], dtype=torch.int32, device=DEV)
    kw = dict(x1=x1, valid_w=vw, bias=bias.to(DEV), act=ops.ACT_LRELU_SQRT2)
    wp = _pack_w(wt)
    y_auto = ops.conv2d(x0, wp, cout, k, k, stride, (pad, pad), algo=0, **kw)
    y_reg = ops.conv2d(x0, wp, cout, k, k, stride, (pad, pad), algo=_lib.ALGO_REG_STAGED, **kw)
    a = ops.convert(y_auto, torch.float32)
    r = ops.convert(y_reg, torch.float32)
    if cout >= 64 and (c0 + c1) % 32 == 0 and c0 % 32 == 0:
        y_one = ops.conv2d(x0, wp, cout, k, k, stride, (pad, pad), algo=_lib.ALGO_LDS_DMA | _lib.ALGO_FLAG_ONE_TILE, **kw)
        assert torch.equal(y_auto.view(torch.uint8), y_one.view(torch.uint8))
        y_tap = ops.conv2d(x0, wp, cout, k, k, stride, (pad, pad), algo=_lib.ALGO_DMA_CFG0 + 3, **kw)
        assert torch.equal(y_auto.view(torch.uint8), y_tap.view(torch.uint8))
    torch.cuda.synchronize()
    assert float((a - r).abs().max()) <= TOL_MX * float(r.abs().max())
    xs = xa[:1].clone()
    xs[0, :, :, int(vw[0]):] = 0
    ref = F.leaky_relu(F.conv2d(xs.double(), wt.double(), stride=stride, padding=pad) + bias[None, :, None, None].double(), 0.2) * 2 ** 0.5
    _check("fp16+8 conv fuzz %s" % (case,), a[:1].cpu().permute(0, 3, 1, 2), ref, TOL_MX)


@pytest.mark.parametrize("storage", ["f16", "x3", "x2"])
@pytest.mark.parametrize("case", [
    # n, h, w, c (conv2 input), cx (skip input), cout, pinned LDS-DMA id or None
    (40, 64, 64, 64, 128, 256, None),      # AUTO at a big launch (x2: the software-pipelined tile; two k regimes: 9 slabs per slice, then 1)
    (3, 20, 36, 32, 64, 256, None),        # a small launch (the 3-/4-stage tiles), pixel tail
    (5, 32, 48, 64, 64, 96, 3),            # cout tail on the 64x256 tile
    (40, 64, 64, 32, 32, 256, 6),          # the lock-step 8-wave tile
])
def test_skip_conv_folded_as_extra_k(storage, case):
    """MNET_CONV_ALGO_FLAG_X1_CENTER (round 4): y = conv3x3(h) + conv1x1(x) + b in ONE k-loop — ResTextBlockV2's h + conv_out(x)
    (models/networks.py:504-505,514-515) without the separate 1x1 launch and the residual read.  Checked against the fp64 evaluation on the
    stored operands, with ragged widths (the skip pixel of a column >= valid_w is read as zero, like every other tap)"""
    ops, P = _ops(), _P()
    n, h, w, c, cx, cout, pin = case
    dt = {"f16": torch.float16, "x3": P.SPLIT_DTYPE, "x2": P.MX_DTYPE}[storage]
    tol = {"f16": 2e-3, "x3": 1e-5, "x2": 8e-5}[storage]
    if storage == "f16" and (c % 64 or cx % 64):
        pytest.skip("f16 slabs are 64 channels deep")
    hq, xq = _rnd((n, c, h, w), 91), _rnd((n, cx, h, w), 92)
    w2 = _rnd((cout, c, 3, 3), 93, 1.0 / math.sqrt(9 * c))
    wo = _rnd((cout, cx, 1, 1), 94, 1.0 / math.sqrt(cx))
    bias = _rnd((cout,), 95, 0.3)
    wz = torch.zeros((cout, cx, 3, 3))
    wz[:, :, 1, 1] = wo[:, :, 0, 0]
    wp = P.pack_conv_weight(torch.cat([w2, wz], 1).to(DEV), dt)
    hd = ops.convert(hq.permute(0, 2, 3, 1).contiguous().to(DEV), dt)
    xd = ops.convert(xq.permute(0, 2, 3, 1).contiguous().to(DEV), dt)
    vw = torch.tensor([w - 3 * (i % 4) for i in range(n)], dtype=torch.int32, device=DEV)
    y = ops.conv2d(hd, wp, cout, 3, 3, (1, 1), (1, 1), x1=xd, bias=bias.to(DEV), valid_w=vw, x1_center=True, algo=0 if pin is None else 16 + pin)
    got = ops.convert(y, torch.float32).cpu().permute(0, 3, 1, 2)
    # reference on the STORED operands (what the kernel multiplies), fp64
    hs = ops.convert(hd, torch.float32).cpu().permute(0, 3, 1, 2).double()
    xs = ops.convert(xd, torch.float32).cpu().permute(0, 3, 1, 2).double()
    for i in range(n):
        hs[i, :, :, int(vw[i]):] = 0
        xs[i, :, :, int(vw[i]):] = 0
    wq = w2.half().double() if storage == "f16" else w2.double()
    woq = wo.half().double() if storage == "f16" else wo.double()
    ref = F.conv2d(hs, wq, padding=1) + F.conv2d(xs, woq) + bias.double()[None, :, None, None]
    scale = float(ref.abs().max())
    err = float((got.double() - ref).abs().max()) / scale
    print("folded skip conv %s %s: max rel err %.2e" % (storage, case, err))
    assert err <= tol
    # and against the two-launch form it replaces (1x1 conv, then 3x3 with the residual): same up to the storage rounding of the skip tensor
    wp2, wpo = P.pack_conv_weight(w2.to(DEV), dt), P.pack_conv_weight(wo.to(DEV), dt)
    skip = ops.conv2d(xd, wpo, cout, bias=bias.to(DEV))
    y2 = ops.conv2d(hd, wp2, cout, 3, 3, (1, 1), (1, 1), valid_w=vw, residual=skip)
    got2 = ops.convert(y2, torch.float32).cpu().permute(0, 3, 1, 2)
    for i in range(n):                    # columns >= valid_w: the two-launch form adds the skip of the UNMASKED x there (never consumed downstream)
        got2[i, :, :, int(vw[i]):] = got[i, :, :, int(vw[i]):]
    assert float((got - got2).abs().max()) / scale <= (4e-3 if storage == "f16" else 1e-4)


# ---------------------------------------------------------------------------------------------------------------- round 5
@pytest.mark.parametrize("case", [
    # cout, cin, (n, h, w), valid widths or None, tile ids to compare
    (256, 64, (3, 16, 64), None, (0, 1, 2, 6, 8, 10, 11)),
    (64, 64, (2, 32, 64), None, (3, 5, 13)),
    (256, 96, (4, 32, 32), [32, 17, 1, 29], (0, 6, 10, 11)),          # glyph maps: statistics over columns < valid_w only
    (128, 64, (2, 16, 96), None, (2, 4, 7, 8, 10)),
])
def test_groupnorm_partial_sums_from_the_conv_epilogue(case):
    """round 5 (mnet_conv_desc.gn_partial): the fp16+8 LDS-DMA epilogue writes, per 32 consecutive pixels x 32-channel group, the sum and the sum of
    squares of its output; mnet_groupnorm_affine_from_partial folds them into the GroupNorm affine — no statistics pass over the map.
    * the affine equals torch's GroupNorm statistics of the conv's output (and the separate-pass kernel's, up to the storage rounding of the map);
    * the partial sums are the same BYTES for every tile configuration (fixed fragment, fixed tree): batch-invariant;
    * the software-pipelined tiles (built without the block) hand the launch to their lock-step forms: same output bytes."""
    ops = _ops()
    cout, cin, (n, h, w), vws, ids = case
    x = _q(_rnd((n, cin, h, w), 131))
    wt = _rnd((cout, cin, 3, 3), 132, 1.0 / math.sqrt(cin * 9))
    bias = _rnd((cout,), 133, 0.3)
    gamma, beta = _rnd((cout,), 134).abs() + 0.5, _rnd((cout,), 135, 0.2)
    vw = None if vws is None else torch.tensor(vws, dtype=torch.int32).to(DEV)
    xd, wp = _to_mx(x), _pack_w(wt)
    parts, outs = [], []
    for i in ids:
        part = ops.gn_partial_buffer(n, h, w, cout, DEV)
        part.fill_(float("nan"))
        y = ops.conv2d(xd, wp, cout, 3, 3, (1, 1), (1, 1), bias=bias.to(DEV), act=2, valid_w=vw, algo=16 + i, gn_partial=part)
        torch.cuda.synchronize()
        assert torch.isfinite(part).all(), "tile id %d left fragments unwritten" % i
        parts.append(part.cpu())
        outs.append(y.cpu().view(torch.uint8))
    for p_, o_ in zip(parts[1:], outs[1:]):
        assert torch.equal(parts[0], p_) and torch.equal(outs[0], o_)
    y0 = ops.conv2d(xd, wp, cout, 3, 3, (1, 1), (1, 1), bias=bias.to(DEV), act=2, valid_w=vw)                 # AUTO, no statistics
    assert torch.equal(y0.cpu().view(torch.uint8), outs[0])
    if cout >= 128:       # an explicit software-pipelined id with gn_partial runs the lock-step form of the same tile
        part = ops.gn_partial_buffer(n, h, w, cout, DEV)
        ys = ops.conv2d(xd, wp, cout, 3, 3, (1, 1), (1, 1), bias=bias.to(DEV), act=2, valid_w=vw, algo=16 + (15 if cout >= 256 else 9), gn_partial=part)
        assert torch.equal(ys.cpu().view(torch.uint8), outs[0]) and torch.equal(part.cpu(), parts[0])
    sc, sh = ops.groupnorm_affine_from_partial(parts[0].to(DEV), n, h, w, cout, gamma.to(DEV), beta.to(DEV), 1e-6, vw)
    sc2, sh2 = ops.groupnorm_affine(y0, gamma.to(DEV), beta.to(DEV), 1e-6, vw)                                    # the separate pass, over the STORED values
    torch.cuda.synchronize()
    yf = _from_mx(y0)
    for i in range(n):
        v = w if vws is None else vws[i]
        ref = F.group_norm(yf[i:i + 1, :, :, :v], cout // 32, gamma, beta, 1e-6)
        got = yf[i:i + 1, :, :, :v] * sc[i].cpu()[None, :, None, None] + sh[i].cpu()[None, :, None, None]
        _check("GroupNorm from epilogue sums, image %d %s" % (i, case[:2]), got, ref, 3e-4 if v < 4 else 6e-5)
    assert torch.allclose(sc.cpu(), sc2.cpu(), rtol=3e-4, atol=1e-6) and torch.allclose(sh.cpu(), sh2.cpu(), rtol=3e-4, atol=3e-4)


def test_groupnorm_partial_sums_strip_kernel_and_refusals():
    """cout 64 on >= 65536 pixels: the strip kernel's epilogue is the same code; launches that cannot write the sums are refused with MNET_E_ARG"""
    ops = _ops()
    from marconet_amd._lib import MarconetHipError
    n, h, w, c = 2, 128, 256, 64
    x = _q(_rnd((n, c, h, w), 141))
    wt = _rnd((c, c, 3, 3), 142, 1.0 / math.sqrt(c * 9))
    xd, wp = _to_mx(x), _pack_w(wt)
    part = ops.gn_partial_buffer(n, h, w, c, DEV)
    y = ops.conv2d(xd, wp, c, 3, 3, (1, 1), (1, 1), act=2, gn_partial=part)                       # AUTO: strip 64x512
    part2 = ops.gn_partial_buffer(n, h, w, c, DEV)
    y2 = ops.conv2d(xd, wp, c, 3, 3, (1, 1), (1, 1), act=2, algo=16 + 5, gn_partial=part2)        # per-tap 64x512 tile
    torch.cuda.synchronize()
    assert torch.equal(y.cpu().view(torch.uint8), y2.cpu().view(torch.uint8)) and torch.equal(part.cpu(), part2.cpu())
    yf = _from_mx(y).double()
    blk = yf.permute(0, 2, 3, 1).reshape(n * h * w // 32, 32, c // 32, 32)
    s1, s2 = blk.sum(dim=(1, 3)), (blk * blk).sum(dim=(1, 3))
    assert torch.allclose(part.cpu()[..., 0].double(), s1, rtol=2e-4, atol=2e-3) and torch.allclose(part.cpu()[..., 1].double(), s2, rtol=2e-4, atol=2e-3)
    with pytest.raises(MarconetHipError):       # stride 2
        ops.conv2d(xd, wp, c, 3, 3, (2, 2), (1, 1), gn_partial=ops.gn_partial_buffer(n, h // 2, w // 2, c, DEV))
    with pytest.raises(MarconetHipError):       # a launch that needs the register-staged kernel (input transform)
        ops.conv2d(xd, wp, c, 3, 3, (1, 1), (1, 1), in_scale=torch.ones(n, c, device=DEV), gn_partial=part)
    xf = ops.convert(xd, torch.float16)
    with pytest.raises(MarconetHipError):       # plain f16 storage
        ops.conv2d(xf, wt.permute(0, 2, 3, 1).contiguous().half().to(DEV), c, 3, 3, (1, 1), (1, 1), gn_partial=part)


# ---------------------------------------------------------------------------------------------------------------- round 6
W4 = 16     # fp16+8 LDS-DMA id 16: the 256x256 tile with ONE wave per SIMD (conv_dma_w4.hip: 4 waves x 128x128 outputs, accumulators in a[0:255])


@pytest.mark.parametrize("case", [
    # cout, cin, k, (n, h, w), reference id (lock-step 8-wave 256x256 tile)
    (256, 64, 3, (4, 128, 160), 6),       # 320 pixel tiles on <= 256 workgroups: the slab stream crosses tiles, some workgroups run two
    (256, 32, 1, (4, 128, 160), 6),       # one slab per tile: every iteration closes a tile
    (288, 96, 3, (3, 96, 100), 6),        # cout tail (two channel tiles: the weight scales change per tile), pixel tail, images smaller than a tile row
    (512, 512, 3, (20, 32, 32), 6),       # 144 slabs per tile, 160 tiles, two channel tiles
    (256, 256, 3, (2, 64, 1024), 15),     # the trunk shape class (72 slabs per tile, 512 tiles) against the software-pipelined production tile
    (256, 64, 3, (1, 8, 24), 6),          # ONE partial tile: 255 of the 256 workgroups... none: a single workgroup, 192 pixels
    (64, 64, 3, (2, 32, 64), 6),          # cout 64: three quarters of the weight rows are out of range (zeros), nothing stored for them
])
def test_one_wave_per_simd_tile_equals_the_8_wave_tiles(case):
    """round 6, id 16: the same LDS image, k order and MFMA sequence per output as ids 6 / 15 -> the same bytes, with the full epilogue (bias, residual,
    activation), ragged widths, multi-pass persistent grids, one slab per tile, channel / pixel tails; twice (no state left behind)"""
    ops = _ops()
    cout, cin, k, (n, h, w), ref = case
    x = _to_mx(_rnd((n, cin, h, w), 161))
    wp = _pack_w(_rnd((cout, cin, k, k), 162, 1.0 / math.sqrt(cin * k * k)))
    bias = _rnd((cout,), 163, 0.3).to(DEV)
    res = _to_mx(_rnd((n, cout, h, w), 164))
    vw = torch.tensor([w - 7 * (i % 3) for i in range(n)], dtype=torch.int32, device=DEV)
    outs = []
    for i in (ref, W4, W4):
        y = ops.conv2d(x, wp, cout, k, k, (1, 1), (k // 2, k // 2), bias=bias, residual=res, act=3, valid_w=vw, algo=_tile(i))
        torch.cuda.synchronize()
        outs.append(y.cpu().view(torch.uint8))
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])


def test_one_wave_per_simd_tile_out_scale_post_scale_and_no_epilogue_terms():
    """the remaining epilogue passes (per-image out_scale / post_scale: the demodulation and the next layer's style) and the bare accumulator"""
    ops = _ops()
    n, h, w, cin, cout = 6, 64, 64, 64, 256
    x = _to_mx(_rnd((n, cin, h, w), 171))
    wp = _pack_w(_rnd((cout, cin, 3, 3), 172, 1.0 / math.sqrt(cin * 9)))
    osc, psc = (_rnd((n, cout), 173).abs() + 0.5).to(DEV), (_rnd((n, cout), 174).abs() + 0.5).to(DEV)
    for kw in (dict(out_scale=osc, post_scale=psc, act=3, bias=_rnd((cout,), 175).to(DEV)), dict()):
        a = ops.conv2d(x, wp, cout, 3, 3, (1, 1), (1, 1), algo=_tile(6), **kw).cpu().view(torch.uint8)
        b = ops.conv2d(x, wp, cout, 3, 3, (1, 1), (1, 1), algo=_tile(W4), **kw).cpu().view(torch.uint8)
        assert torch.equal(a, b)


def test_one_wave_per_simd_tile_concat_and_folded_skip():
    """second concat source (conv_body_32.0's shape class) and MNET_CONV_ALGO_FLAG_X1_CENTER (the fuse blocks' 1x1 skip conv as extra K)"""
    ops = _ops()
    from marconet_amd import _lib
    n, h, w, c0, c1, cout = 40, 32, 64, 64, 32, 256
    x0, x1 = _to_mx(_rnd((n, c0, h, w), 181)), _to_mx(_rnd((n, c1, h, w), 182))
    wp = _pack_w(_rnd((cout, c0 + c1, 3, 3), 183, 1.0 / math.sqrt((c0 + c1) * 9)))
    bias = _rnd((cout,), 184, 0.3).to(DEV)
    vw = torch.tensor([w - 3 * (i % 4) for i in range(n)], dtype=torch.int32, device=DEV)
    outs = [ops.conv2d(x0, wp, cout, 3, 3, (1, 1), (1, 1), x1=x1, bias=bias, act=2, valid_w=vw, algo=_tile(i)).cpu().view(torch.uint8) for i in (6, W4)]
    assert torch.equal(outs[0], outs[1])
    # folded skip: [cout][3][3][c0 + c1] weights whose second part is read at the centre tap only
    outs = [ops.conv2d(x0, wp, cout, 3, 3, (1, 1), (1, 1), x1=x1, bias=bias, valid_w=vw, x1_center=True, algo=_tile(i)).cpu().view(torch.uint8) for i in (6, 15, W4)]
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("case", [(256, 64, (3, 16, 64), None), (256, 96, (4, 32, 32), [32, 17, 1, 29]), (512, 64, (40, 32, 64), None)])
def test_one_wave_per_simd_tile_groupnorm_sums(case):
    """mnet_conv_desc.gn_partial written by id 16's own epilogue: the same fragment bytes and output bytes as the lock-step tile (id 11 / 6)"""
    ops = _ops()
    cout, cin, (n, h, w), vws = case
    x = _to_mx(_rnd((n, cin, h, w), 191))
    wp = _pack_w(_rnd((cout, cin, 3, 3), 192, 1.0 / math.sqrt(cin * 9)))
    bias = _rnd((cout,), 193, 0.3).to(DEV)
    vw = None if vws is None else torch.tensor(vws, dtype=torch.int32).to(DEV)
    got = []
    for i in (6, W4):
        part = ops.gn_partial_buffer(n, h, w, cout, DEV)
        part.fill_(float("nan"))
        y = ops.conv2d(x, wp, cout, 3, 3, (1, 1), (1, 1), bias=bias, act=2, valid_w=vw, algo=_tile(i), gn_partial=part)
        torch.cuda.synchronize()
        got.append((y.cpu().view(torch.uint8), part.cpu()))
    assert torch.isfinite(got[1][1]).all()
    assert torch.equal(got[0][0], got[1][0]) and torch.equal(got[0][1], got[1][1])


def test_one_wave_per_simd_tile_scale_rows_of_maps_that_are_not_multiples_of_32_pixels():
    """w4_epilogue stages out_scale / post_scale through the LDS one float per lane, which needs a 32-pixel fragment inside ONE image (ho * wo % 32 == 0); a request for
    id 16 on any other map is handed to the 8-wave tile by the launcher (as is an activation other than identity / LeakyReLU) — same bytes as the lock-step tile either
    way (fragments straddling images, a pixel tail, ragged widths; the last shape runs on id 16 itself)"""
    ops = _ops()
    for (n, h, w) in ((20, 10, 10), (7, 12, 20), (9, 16, 16)):          # 100 / 240 / 256 pixels per image
        cin, cout = 64, 256
        x = _to_mx(_rnd((n, cin, h, w), 201))
        wp = _pack_w(_rnd((cout, cin, 3, 3), 202, 1.0 / math.sqrt(cin * 9)))
        osc, psc = (_rnd((n, cout), 203).abs() + 0.5).to(DEV), (_rnd((n, cout), 204).abs() + 0.5).to(DEV)
        bias = _rnd((cout,), 205, 0.3).to(DEV)
        vw = torch.tensor([w - (i % 3) for i in range(n)], dtype=torch.int32, device=DEV)
        outs = [ops.conv2d(x, wp, cout, 3, 3, (1, 1), (1, 1), out_scale=osc, post_scale=psc, bias=bias, act=3, valid_w=vw, algo=_tile(i)).cpu().view(torch.uint8) for i in (6, W4)]
        assert torch.equal(outs[0], outs[1]), (n, h, w)
