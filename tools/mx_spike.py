#!/usr/bin/env python
"""Spike: rate and error of the fp16+8 conv arithmetic (hi*hi on the f16 MFMA + both correction products as one
block-scaled fp8 MFMA, 2 MFMA units per product) against the fp16x3 kernels (3 units) on the model's dominant layer shapes.
python tools/mx_spike.py [--rounds 7] [--n 64]"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def emulate_conv(x, w, **kw):
    """what the fp16+8 conv computes, in fp32 (x [N,C,H,W], w [O,I,KH,KW] TRUE weights): the expected value of a device result"""
    from marconet_amd import mxfmt
    xn = x.permute(0, 2, 3, 1)
    xb = xn.reshape(xn.shape[:-1] + (-1, 32))
    xh = xb.to(torch.float16).float()
    xs = torch.pow(2.0, (mxfmt.block_e8(xh) - 127).float())
    xh8 = mxfmt._e4m3(xh / xs).float() * xs
    xl8 = mxfmt._e4m3((xb - xh) * 2048.0 / xs).float() * xs / 2048.0
    back = lambda t: t.reshape(xn.shape).permute(0, 3, 1, 2)
    wn = w.permute(0, 2, 3, 1) * mxfmt.WSCALE
    wh = wn.to(torch.float16).float()
    m = wh.abs().reshape(w.shape[0], -1).amax(-1)
    ws = torch.pow(2.0, ((mxfmt._floor_log2(m) - 7 + 127).clamp(11, 254) - 127).float()).reshape(-1, 1, 1, 1)
    wh8 = mxfmt._e4m3(wh / ws).float() * ws
    wl8 = mxfmt._e4m3((wn - wh) * 2048.0 / ws).float() * ws / 2048.0
    wb = lambda t: t.permute(0, 3, 1, 2)
    y = F.conv2d(back(xh), wb(wh), **kw) + F.conv2d(back(xl8), wb(wh8), **kw) + F.conv2d(back(xh8), wb(wl8), **kw)
    return y / mxfmt.WSCALE


def algo_of(i):
    return 2 if i < 0 else (16 + i if i < 16 else 64 + i - 16)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--n", type=int, default=64)
    ap.add_argument("--check-only", action="store_true")
    a = ap.parse_args()
    from marconet_amd import mxfmt, ops, packing
    dev = "cuda"
    torch.manual_seed(0)
    # ---------------- correctness: 256 -> 256 3x3 on 2 x 64 x 256 pixels, every tile family
    n, h, w, cin = 2, 64, 256, 256
    x = torch.randn((n, h, w, cin), device=dev) * 1.5
    x[0, :4] *= 37.0            # block scales must follow the data
    x[1, 5:9] *= 1e-3
    for cout, ids_x3, ids_mx in ((256, (11, 20), (24,)), (128, (9, 21), (25,)), (64, (5,), (26,))):
        wt = torch.randn((cout, 3, 3, cin), device=dev) * 0.02
        wt[3] *= 40.0
        bias = torch.randn(cout, device=dev)
        ref = F.conv2d(x.permute(0, 3, 1, 2).double(), wt.permute(0, 3, 1, 2).double(), bias.double(), padding=1).permute(0, 2, 3, 1)
        den = ref.abs().max().item()
        emu = (emulate_conv(x.permute(0, 3, 1, 2), wt.permute(0, 3, 1, 2), padding=1) + bias.reshape(1, -1, 1, 1)).permute(0, 2, 3, 1)
        print("cout %3d: emulated fp16+8 rel err %.3e" % (cout, (emu.double() - ref).abs().max().item() / den))
        xs, ws = packing.split_halves(x), packing.split_halves(wt * packing.SPLIT_WSCALE)
        for i in ids_x3:
            y = packing.unsplit_halves(ops.conv2d(xs, ws, cout, 3, 3, (1, 1), (1, 1), bias=bias, algo=algo_of(i)))
            print("   x3  id %2d: rel err vs fp64 %.3e" % (i, (y.double() - ref).abs().max().item() / den))
        xm = mxfmt.pack_act(x).view(packing.SPLIT_DTYPE)
        wm = mxfmt.pack_weight(wt).view(packing.SPLIT_DTYPE)
        for i in ids_mx:
            y = packing.unsplit_halves(ops.conv2d(xm, wm, cout, 3, 3, (1, 1), (1, 1), bias=bias, algo=algo_of(i)))
            print("   mx  id %2d (MNET_MX_CVT=%s): rel err vs fp64 %.3e, vs emulation %.3e" %
                  (i, os.environ.get("MNET_MX_CVT", "default"), (y.double() - ref).abs().max().item() / den,
                   (y - emu).abs().max().item() / den), flush=True)
    if a.check_only:
        return
    # ---------------- rate
    shapes = [("sr_trunk_64x1024_256to256", a.n, 64, 1024, 256, 256, ((11, "x3"), (20, "x3"), (24, "mx"))),
              ("resnet_8x512_512to512", a.n * 4, 8, 512, 512, 512, ((11, "x3"), (20, "x3"), (24, "mx"))),
              ("sr_final_256to128", a.n // 2, 64, 1024, 256, 128, ((9, "x3"), (21, "x3"), (25, "mx"))),
              ("gan_128_128", a.n * 4, 128, 128, 128, 128, ((9, "x3"), (21, "x3"), (25, "mx"))),
              ("sr_final_64", a.n // 2, 128, 2048, 64, 64, ((-1, "x3"), (5, "x3"), (26, "mx")))]
    for name, n, h, w, cin, cout, algos in shapes:
        x = torch.randn((n, h, w, cin), device=dev)
        wt = torch.randn((cout, 3, 3, cin), device=dev) * 0.02
        bias = torch.zeros(cout, device=dev)
        data = {"x3": (packing.split_halves(x), packing.split_halves(wt * packing.SPLIT_WSCALE)),
                "mx": (mxfmt.pack_act(x).view(packing.SPLIT_DTYPE), mxfmt.pack_weight(wt).view(packing.SPLIT_DTYPE))}
        del x
        out = torch.empty((n, h, w, cout), dtype=packing.SPLIT_DTYPE, device=dev)
        flops = 2.0 * n * h * w * cout * 9 * cin
        t = {i: [] for i, _ in algos}
        for r in range(a.rounds + 1):
            for i, fmt in algos:
                xs, ws = data[fmt]
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                ops.conv2d(xs, ws, cout, 3, 3, (1, 1), (1, 1), bias=bias, act=ops.ACT_LRELU, out=out, algo=algo_of(i))
                e.record()
                torch.cuda.synchronize()
                if r:
                    t[i].append(s.elapsed_time(e))
        med = lambda v: sorted(v)[len(v) // 2]
        print("%-28s " % name + "  ".join("%s id %2d: %6.1f TF/s (%.2f ms)" % (fmt, i, flops / med(t[i]) / 1e9, med(t[i])) for i, fmt in algos), flush=True)
        del data, out


if __name__ == "__main__":
    main()
