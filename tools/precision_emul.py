#!/usr/bin/env python
"""CPU emulation of candidate conv arithmetics, end to end through the oracle chain (test infrastructure; needs no GPU).

Every F.conv2d of the oracle (ResNet-45, TSPGAN, TSPSRNet; the TextViT linears stay fp32 like in the product) is replaced
by an fp32 evaluation of what a given MFMA decomposition would compute, so that the SR deviation of a scheme is MEASURED
before a kernel is written for it ("measure, don't estimate").  Products inside one MFMA are exact and
the accumulation is fp32 on the hardware, so conv(q(w), q(x)) in fp32 on the CPU reproduces the operand-rounding error,
which is the only error these schemes differ in.

  f16      w -> f16(w), x -> f16(x)                                     1 MFMA unit   (the fp16 mode)
  x3       hi.hi + hi.lo + lo.hi, hi/lo = split halves                   3 units       (the fp16x3 mode)
  mx8      hi.hi (f16) + [w_hi8.x_lo8 + w_lo8.x_hi8] on the MX-scaled fp8 MFMA (e4m3, one E8M0 exponent per 32 channels,
           lo scale = hi scale - 11)                                     2 units
  i8       hi.hi (f16) + the two corrections on v_mfma_i32_16x16x64_i8, per-tensor scale for x, per-cout-row for w
                                                                         2 units
  mx8x     like mx8 but only w_hi8.x_lo8 (activation correction), weights rounded to f16          1.5 units
  mx6      corrections in fp6 e2m3 (4x rate)                             1.5 units

python tools/precision_emul.py [B] [n] [schemes...]"""
import math
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

_conv2d = F.conv2d


def split(v):
    hi = v.to(torch.float16).to(torch.float32)
    return hi, v - hi


def _blocks(v, dim):
    """view v with the channel axis `dim` (=1) split into blocks of 32 (zero padded)."""
    C = v.shape[dim]
    pad = (-C) % 32
    if pad:
        shp = list(v.shape)
        shp[dim] = pad
        v = torch.cat([v, v.new_zeros(shp)], dim)
    shp = list(v.shape)
    return v.reshape(shp[:dim] + [shp[dim] // 32, 32] + shp[dim + 1:]), C


def _fp8(v):
    return v.to(torch.float8_e4m3fn).to(torch.float32)


def _fp6_e2m3(v):
    # e2m3: 1 sign, 2 exponent (bias 1), 3 mantissa; max 7.5, min normal 1.0, subnormal step 0.125
    a = v.abs().clamp(max=7.5)
    e = torch.floor(torch.log2(a.clamp(min=1e-30))).clamp(min=0, max=2)
    step = torch.pow(2.0, e - 3)
    return torch.sign(v) * torch.round(a / step) * step


def _fp6_e3m2(v):
    # e3m2: 1 sign, 3 exponent (bias 3), 2 mantissa; max 28, min normal 0.25, subnormal step 0.0625
    a = v.abs().clamp(max=28.0)
    e = torch.floor(torch.log2(a.clamp(min=1e-30))).clamp(min=-2, max=4)
    step = torch.pow(2.0, e - 2)
    return torch.sign(v) * torch.round(a / step) * step


def mx_pair(v, dim, q=_fp8, top=7):
    """(hi8, lo8) of v with one shared exponent per 32-channel block; the lo scale is the hi scale 2^-11."""
    hi, lo = split(v)
    hb, C = _blocks(hi, dim)
    lb, _ = _blocks(lo, dim)
    m = hb.abs().amax(dim + 1, keepdim=True).clamp(min=2.0 ** -60)
    E = torch.floor(torch.log2(m))
    s_hi = torch.pow(2.0, E - top)
    s_lo = s_hi * 2.0 ** -11
    h8 = (q(hb / s_hi) * s_hi).reshape(_unblock(hb.shape, dim))
    l8 = (q(lb / s_lo) * s_lo).reshape(_unblock(lb.shape, dim))
    return hi, h8.narrow(dim, 0, C), l8.narrow(dim, 0, C)


def _unblock(shp, dim):
    shp = list(shp)
    return shp[:dim] + [shp[dim] * shp[dim + 1]] + shp[dim + 2:]


def i8_pair_x(v):
    hi, lo = split(v)
    s = hi.abs().amax().clamp(min=1e-30) / 127.0
    s = torch.pow(2.0, torch.ceil(torch.log2(s)))
    h8 = torch.round(hi / s).clamp(-127, 127) * s
    sl = s * 2.0 ** -11
    l8 = torch.round(lo / sl).clamp(-127, 127) * sl
    return hi, h8, l8


def i8_pair_w(w):
    hi, lo = split(w)
    s = hi.abs().amax(dim=(1, 2, 3), keepdim=True).clamp(min=1e-30) / 127.0
    s = torch.pow(2.0, torch.ceil(torch.log2(s)))
    h8 = torch.round(hi / s).clamp(-127, 127) * s
    sl = s * 2.0 ** -11
    l8 = torch.round(lo / sl).clamp(-127, 127) * sl
    return hi, h8, l8


def make_conv(scheme):
    def conv(x, w, bias=None, stride=1, padding=0, dilation=1, groups=1):
        kw = dict(stride=stride, padding=padding, dilation=dilation, groups=groups)
        if scheme in ("f16", "f16s"):
            y = _conv2d(split(x)[0], split(w)[0], None, **kw)
        elif scheme == "x3":
            xh, xl = split(x)
            wh, wl = split(w)
            xl, wl = split(xl)[0], split(wl)[0]
            y = _conv2d(xh, wh, None, **kw) + _conv2d(xl, wh, None, **kw) + _conv2d(xh, wl, None, **kw)
        elif scheme in ("mx8", "mx8x", "mx6", "bf6"):
            q, top = (_fp6_e2m3, 2) if scheme == "mx6" else (_fp6_e3m2, 4) if scheme == "bf6" else (_fp8, 7)
            xh, xh8, xl8 = mx_pair(x, 1, q, top)
            wh, wh8, wl8 = mx_pair(w, 1, q, top)
            y = _conv2d(xh, wh, None, **kw) + _conv2d(xl8, wh8, None, **kw)
            if scheme != "mx8x":
                y = y + _conv2d(xh8, wl8, None, **kw)
        elif scheme in ("bf8", "bf8w4"):
            # no block scales: hi8 = bf8(hi16) (same exponent range as f16), lo8 = bf8(lo * 2^11) * 2^-11
            bq = lambda v: v.to(torch.float8_e5m2).to(torch.float32)
            xh, xl = split(x)
            wh, wl = split(w)
            xh8, xl8 = bq(xh), bq(xl * 2048.0) / 2048.0
            if scheme == "bf8":
                wh8, wl8 = bq(wh), bq(wl * 2048.0) / 2048.0
            else:
                _, wh8, wl8 = mx_pair(w, 1, _fp8, 7)
            y = _conv2d(xh, wh, None, **kw) + _conv2d(xl8, wh8, None, **kw) + _conv2d(xh8, wl8, None, **kw)
        elif scheme == "i8":
            xh, xh8, xl8 = i8_pair_x(x)
            wh, wh8, wl8 = i8_pair_w(w)
            y = _conv2d(xh, wh, None, **kw) + _conv2d(xl8, wh8, None, **kw) + _conv2d(xh8, wl8, None, **kw)
        else:
            raise ValueError(scheme)
        if bias is not None:
            y = y + bias.reshape(1, -1, 1, 1)
        if scheme == "f16s":            # plain-f16 STORAGE of the layer's output as well (the product's f16 kernels write halves)
            y = split(y)[0]
        return y
    return conv


def main():
    from marconet_amd import synthetic
    from oracle import marconet_oracle as O
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    schemes = sys.argv[3:] or ["f16", "x3", "mx8", "i8", "mx8x", "mx6"]
    variant = os.environ.get("EMUL_VARIANT", "")
    kw = {"variant": variant} if variant else {}
    sde, sdg, sds = (synthetic.make_encoder_state_dict(**kw), synthetic.make_gan_state_dict(**kw),
                     synthetic.make_sr_state_dict(**kw))
    lq = synthetic.make_lq(1234, B, [512] * B)
    labels = [synthetic.make_labels(1234 + b, n) for b in range(B)]
    locs = synthetic.make_locs([n] * B, [512] * B)
    t = time.time()
    ref = O.end_to_end(sde, sdg, sds, lq, labels, locs)
    print("reference chain: %.1f s; |sr| max %.3f" % (time.time() - t, ref["sr"].abs().max().item()), flush=True)
    print("%-6s  %-10s %-10s %-10s %-10s %-10s argmax" % ("scheme", "sr max", "sr mean", "w max", "p64 max", "logits max"))
    for s in schemes:
        F.conv2d = make_conv(s)
        try:
            out = O.end_to_end(sde, sdg, sds, lq, labels, locs)
        finally:
            F.conv2d = _conv2d
        d = (out["sr"] - ref["sr"]).abs()
        dw = (out["w"] - ref["w"]).abs().max().item()
        dp = max((a - b).abs().max().item() for a, b in zip(out["p64"], ref["p64"]))
        dl = (out["logits"] - ref["logits"]).abs().max().item()
        am = (out["logits"].argmax(-1) == ref["logits"].argmax(-1)).float().mean().item()
        print("%-6s  %.3e  %.3e  %.3e  %.3e  %.3e  %.4f" % (s, d.max().item(), d.mean().item(), dw, dp, dl, am), flush=True)


if __name__ == "__main__":
    main()
