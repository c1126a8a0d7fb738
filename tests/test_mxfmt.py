"""CPU: the host-side definition of the fp16+8 storage (marconet_amd/mxfmt.py) — layout, round-trip accuracy, the weight packer
and the precision the decomposition buys (hi*hi + block-scaled fp8 corrections vs plain fp16 operands)."""
import torch
import torch.nn.functional as F

from marconet_amd import mxfmt, packing


def _rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def test_activation_layout_and_round_trip():
    x = _rnd((2, 3, 5, 64), 1) * torch.logspace(-2, 2, 64)
    b = mxfmt.pack_act(x)
    assert b.dtype == torch.uint8 and b.shape == (2, 3, 5, 256)
    blk = b.reshape(2, 3, 5, 2, 128)
    hi = x.reshape(2, 3, 5, 2, 32).to(torch.float16)
    assert torch.equal(blk[..., :64].contiguous().view(torch.float16), hi)                  # bytes 0-63: the hi halves in channel order
    e8 = blk[..., 96].int()
    want = torch.floor(torch.log2(hi.float().abs().amax(-1))).int() - 7 + 127
    assert torch.equal(e8, want) and int(blk[..., 97:].max()) == 0                          # byte 96: E8M0 of the block scale; then zeros
    back = mxfmt.unpack_act(b, 64)
    bm = x.abs().reshape(2, 3, 5, 2, 32).amax(-1, keepdim=True).expand(2, 3, 5, 2, 32).reshape(x.shape)
    assert ((back - x).abs() <= bm * 2.0 ** -15).all()
    assert torch.equal(mxfmt.unpack_act(mxfmt.pack_act(back), 64), back)                    # idempotent
    t = packing.from_float(x, packing.MX_DTYPE)
    assert t.dtype == packing.MX_DTYPE and t.shape == x.shape and torch.equal(packing.to_float(t), back)
    # lo bytes are stored in the order the 32x32x64 fp8 MFMA consumes them: channels 0-7, 16-23 | 8-15, 24-31
    lo = (x.reshape(2, 3, 5, 2, 32) - hi.float()) * 2048.0 / torch.pow(2.0, (e8 - 127).float()).unsqueeze(-1)
    assert torch.equal(blk[..., 64:96].contiguous().view(torch.float8_e4m3fn).float(), lo.to(torch.float8_e4m3fn).float()[..., mxfmt.PERM])


def test_weight_packer_layout():
    w = _rnd((8, 3, 3, 64), 2, 0.05)
    w[3] *= 25.0
    flat = mxfmt.pack_weight(w)
    nb = 8 * 9 * 64 * 4
    assert flat.numel() == nb + 16
    rows = flat[:nb].reshape(8, 3, 3, 2, 128)
    wn = (w * 256.0).reshape(8, 3, 3, 2, 32)
    hi = wn.to(torch.float16)
    assert torch.equal(rows[..., :64].contiguous().view(torch.float16), hi)
    e8 = torch.floor(torch.log2(hi.float().abs().reshape(8, -1).amax(-1))).int() - 7 + 127
    assert torch.equal(flat[nb:nb + 8].int(), e8 - 11)                                      # E8M0 of s * 2^-11 per output channel
    s = torch.pow(2.0, (e8 - 127).float()).reshape(8, 1, 1, 1, 1)
    hi8 = (hi.float() / s).to(torch.float8_e4m3fn).float()
    assert torch.equal(rows[..., 80:96].contiguous().view(torch.float8_e4m3fn).float(), hi8[..., mxfmt.PERM[:16]])
    assert torch.equal(rows[..., 112:128].contiguous().view(torch.float8_e4m3fn).float(), hi8[..., mxfmt.PERM[16:]])
    t = packing.pack_conv_weight(w.permute(0, 3, 1, 2).contiguous(), packing.MX_DTYPE)
    assert t.dtype == packing.MX_DTYPE and t.shape == (packing.mx_weight_rows(32, 3, 3, 64), 3, 3, 64)      # cout padded to 32, + scale rows
    assert packing.padded_cout(8, packing.MX_DTYPE) == 32 and torch.equal(t.view(torch.uint8).reshape(-1)[:8 * 9 * 64 * 4], flat[:nb])


def test_decomposition_accuracy():
    """hi*hi + w_hi8*x_lo8 + w_lo8*x_hi8 is ~25x closer to the exact product than fp16 operands, on one 3x3 layer"""
    x = _rnd((1, 64, 12, 12), 3)
    w = _rnd((32, 64, 3, 3), 4, 0.04)
    ref = F.conv2d(x.double(), w.double(), padding=1)
    xn = x.permute(0, 2, 3, 1)
    xb = xn.reshape(1, 12, 12, 2, 32)
    xh = xb.to(torch.float16).float()
    xs = torch.pow(2.0, (mxfmt.block_e8(xh) - 127).float())
    q = lambda v: v.to(torch.float8_e4m3fn).float()
    nchw = lambda t: t.reshape(1, 12, 12, 64).permute(0, 3, 1, 2)
    wn = w.permute(0, 2, 3, 1)
    wh = wn.to(torch.float16).float()
    ws = torch.pow(2.0, torch.floor(torch.log2(wh.abs().reshape(32, -1).amax(-1))) - 7).reshape(32, 1, 1, 1)
    oihw = lambda t: t.permute(0, 3, 1, 2)
    y = (F.conv2d(nchw(xh), oihw(wh), padding=1) + F.conv2d(nchw(q((xb - xh) * 2048 / xs) * xs / 2048), oihw(q(wh / ws) * ws), padding=1)
         + F.conv2d(nchw(q(xh / xs) * xs), oihw(q((wn - wh) * 2048 / ws) * ws / 2048), padding=1))
    e_mx = (y.double() - ref).abs().max().item() / ref.abs().max().item()
    e_16 = (F.conv2d(x.half().float(), w.half().float(), padding=1).double() - ref).abs().max().item() / ref.abs().max().item()
    assert e_mx < 3e-5 and e_16 > 8 * e_mx


def test_blocked_storage_tensors_refuse_arithmetic():
    """the 4-byte tags (complex32 / uint32) only carry bytes for the HIP kernels — any torch arithmetic on them raises,
    storage plumbing (views, slices, cat over outer dimensions, clone, device moves) keeps working and keeps the tag"""
    import pytest
    x = _rnd((2, 3, 4, 64), 9)
    for dt in (packing.SPLIT_DTYPE, packing.MX_DTYPE):
        s = packing.from_float(x, dt)
        assert isinstance(s, packing.BlockedTensor) and s.dtype == dt and s.shape == x.shape
        for bad in (lambda: s + s, lambda: s * 2.0, lambda: s.sum(), lambda: s.float(), lambda: torch.isfinite(s), lambda: torch.add(s, 1), lambda: s.abs(),
                    lambda: s.to(torch.float32), lambda: s.to("cpu", torch.float16)):
            with pytest.raises(TypeError):
                bad()
        assert isinstance(s.to("cpu"), packing.BlockedTensor) and s.to("cpu").dtype == dt          # device moves stay allowed
        part = s[1:]
        assert isinstance(part, packing.BlockedTensor) and isinstance(s.clone(), packing.BlockedTensor) and isinstance(s.contiguous(), packing.BlockedTensor)
        assert part.data_ptr() == s.data_ptr() + 3 * 4 * 64 * 4 and s.is_contiguous() and s.numel() == x.numel() and s.element_size() == 4
        assert torch.equal(packing.to_float(part), packing.to_float(s)[1:])
        from marconet_amd import ops
        both = ops.cat_rows([s, s])
        assert isinstance(both, packing.BlockedTensor) and both.shape == (4, 3, 4, 64) and torch.equal(packing.to_float(both)[2:], packing.to_float(s))
        sel = ops.take_rows(s, torch.tensor([1, 0]))
        assert isinstance(sel, packing.BlockedTensor) and torch.equal(packing.to_float(sel)[0], packing.to_float(s)[1])
