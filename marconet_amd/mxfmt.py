"""Host-side (torch) definition of the "fp16+8" storage (MNET_F16M, precision mode "fp16x2"): the reference packers the device
kernels are tested against, and the offline weight packer.  Never on the per-forward path.

One logical element = 4 bytes, blocked per 32 channels exactly like the split-half storage (128 bytes per (pixel, block)):

  activations   bytes 0-63    hi = f16(v), channels 0..31
                bytes 64-95   lo8 = e4m3(lo * 2^11 / s), lo = v - hi, channel order PERM (0-7, 16-23 | 8-15, 24-31)
                byte  96      E = E8M0 exponent of the block scale s = 2^(E - 127) = 2^(max(floor(log2 max|hi|), -15) - 7); 0 for an all-zero block
                bytes 97-127  zero
  conv weights  bytes 0-63    hi = f16(256 W)
                bytes 64-79 lo8 of channels 0-7,16-23 | 80-95 hi8 of the same | 96-111 lo8 of 8-15,24-31 | 112-127 hi8 of the same
                (hi8 = e4m3(hi / s), lo8 = e4m3(lo * 2^11 / s), s per OUTPUT CHANNEL), followed after the cout * K rows by one byte per
                output channel: E8M0 of s * 2^-11

x*w = hi*hi on the f16 MFMA + (w_lo8*x_hi8 + w_hi8*x_lo8) on the block-scaled fp8 MFMA (x_hi8 is derived from hi in the kernel).
"""
import torch

PERM = list(range(0, 8)) + list(range(16, 24)) + list(range(8, 16)) + list(range(24, 32))
INV_PERM = [PERM.index(i) for i in range(32)]
WSCALE = 256.0


def _floor_log2(m):
    """floor(log2 m) for m > 0 (exact: frexp), -127 for m == 0"""
    mant, ex = torch.frexp(m)
    return torch.where(m > 0, ex - 1, torch.full_like(ex, -127))


def block_e8(hi):
    """hi [..., 32] fp32 (values of halves) -> int32 [..., 1]: E8M0 byte of s = 2^(floor(log2 max|hi|) - 7)"""
    m = hi.abs().amax(-1, keepdim=True)
    # a non-zero block's exponent is floored at 127 - 15 - 7 = 105: below that the hi halves are fp16 subnormals and the scaled residual would leave e4m3's range
    # (the device conversions return NaN there, not the saturated byte — csrc/common.h::hm_e8_of); an all-zero block keeps 0
    e = _floor_log2(m) - 7 + 127
    return torch.where(m > 0, e.clamp(105, 254), torch.zeros_like(e)).to(torch.int32)


def _e4m3(v):
    return v.clamp(-448.0, 448.0).to(torch.float8_e4m3fn)


def pack_act(x):
    """fp32 [..., C] (C % 32 == 0) -> uint8 [..., C * 4] in the activation layout above"""
    C = x.shape[-1]
    assert C % 32 == 0
    xb = x.detach().float().reshape(x.shape[:-1] + (C // 32, 32))
    hi = xb.to(torch.float16)
    lo = xb - hi.float()
    e8 = block_e8(hi.float())
    s = torch.pow(2.0, (e8 - 127).float())
    lo8 = _e4m3(lo * 2048.0 / s)
    out = torch.zeros(xb.shape[:-1] + (128,), dtype=torch.uint8, device=x.device)
    out[..., 0:64] = hi.contiguous().view(torch.uint8)
    out[..., 64:96] = lo8[..., PERM].contiguous().view(torch.uint8)
    out[..., 96] = e8[..., 0].to(torch.uint8)
    return out.reshape(x.shape[:-1] + (C * 4,))


def unpack_act(b, C):
    """inverse of pack_act (what the tail kernels decode): uint8 [..., C * 4] -> fp32 [..., C]"""
    bb = b.reshape(b.shape[:-1] + (C // 32, 128))
    hi = bb[..., 0:64].contiguous().view(torch.float16).float()
    lo8 = bb[..., 64:96].contiguous().view(torch.float8_e4m3fn).float()[..., INV_PERM]
    s = torch.pow(2.0, bb[..., 96:97].float() - 127.0)
    return (hi + lo8 * s * (2.0 ** -11)).reshape(b.shape[:-1] + (C,))


def pack_weight(w):
    """fp32 [O, KH, KW, I] (I % 32 == 0; the TRUE weights, not yet scaled) -> uint8 [O * KH * KW * I * 4 + pad16(O)]"""
    O, KH, KW, I = w.shape
    assert I % 32 == 0
    wb = (w.detach().float() * WSCALE).reshape(O, KH, KW, I // 32, 32)
    hi = wb.to(torch.float16)
    if not bool(torch.isfinite(hi).all()):          # the same contract as the split-half host packer and the device packer (mnet_pack_weights)
        raise OverflowError("a conv weight does not fit the half range of the fp16x2 mode (|256 W| >= 65504)")
    lo = wb - hi.float()
    m = hi.float().abs().reshape(O, -1).amax(-1)
    e8 = (_floor_log2(m) - 7 + 127).clamp(11, 254).to(torch.int32)          # per output channel
    s = torch.pow(2.0, (e8 - 127).float()).reshape(O, 1, 1, 1, 1)
    hi8 = _e4m3(hi.float() / s)
    lo8 = _e4m3(lo * 2048.0 / s)
    rows = torch.zeros((O, KH, KW, I // 32, 128), dtype=torch.uint8, device=w.device)
    rows[..., 0:64] = hi.contiguous().view(torch.uint8)
    p0, p1 = PERM[:16], PERM[16:]
    rows[..., 64:80] = lo8[..., p0].contiguous().view(torch.uint8)
    rows[..., 80:96] = hi8[..., p0].contiguous().view(torch.uint8)
    rows[..., 96:112] = lo8[..., p1].contiguous().view(torch.uint8)
    rows[..., 112:128] = hi8[..., p1].contiguous().view(torch.uint8)
    tail = torch.zeros(((O + 15) // 16 * 16,), dtype=torch.uint8, device=w.device)
    tail[:O] = (e8 - 11).to(torch.uint8)
    return torch.cat([rows.reshape(-1), tail])
