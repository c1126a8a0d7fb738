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


def algo_of(i):
    return 2 if i < 0 else (16 + i if i < 16 else 64 + i - 16)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--n", type=int, default=64)
    a = ap.parse_args()
    from marconet_amd import ops, packing
    dev = "cuda"
    torch.manual_seed(0)
    SP, MX = packing.SPLIT_DTYPE, packing.MX_DTYPE
    # ---------------- error of one layer (256 -> 256 3x3 on 2 x 64 x 256 pixels) in each arithmetic
    n, h, w, cin, cout = 2, 64, 256, 256, 256
    x = torch.randn((n, h, w, cin), device=dev) * 1.5
    x[0, :4] *= 37.0
    wt = torch.randn((cout, cin, 3, 3), device=dev) * 0.02
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), wt.double(), padding=1).permute(0, 2, 3, 1)
    den = ref.abs().max().item()
    for name, dt in (("fp16", torch.float16), ("fp16x3", SP), ("fp16x2", MX)):
        y = ops.convert(ops.conv2d(ops.convert(x, dt), packing.pack_conv_weight(wt, dt), cout, 3, 3, (1, 1), (1, 1)), torch.float32)
        print("one 3x3 layer, %-7s: max rel err vs fp64 %.3e" % (name, (y.double() - ref).abs().max().item() / den), flush=True)
    # ---------------- rate
    shapes = [("sr_trunk_64x1024_256to256", a.n, 64, 1024, 256, 256, ((11, SP), (20, SP), (6, MX), (11, MX))),
              ("resnet_8x512_512to512", a.n * 4, 8, 512, 512, 512, ((11, SP), (6, MX), (11, MX))),
              ("sr_final_256to128", a.n // 2, 64, 1024, 256, 128, ((9, SP), (21, SP), (7, MX), (8, MX), (12, MX))),
              ("gan_128_128", a.n * 4, 128, 128, 128, 128, ((9, SP), (7, MX), (8, MX), (12, MX))),
              ("sr_final_64", a.n // 2, 128, 2048, 64, 64, ((-1, SP), (5, SP), (5, MX), (13, MX), (-1, MX)))]
    for name, n, h, w, cin, cout, algos in shapes:
        x = torch.randn((n, h, w, cin), device=dev)
        wt = torch.randn((cout, cin, 3, 3), device=dev) * 0.02
        bias = torch.zeros(cout, device=dev)
        data = {dt: (ops.convert(x, dt), packing.pack_conv_weight(wt, dt)) for dt in (SP, MX)}
        del x
        outs = {dt: torch.empty((n, h, w, cout), dtype=dt, device=dev) for dt in (SP, MX)}
        flops = 2.0 * n * h * w * cout * 9 * cin
        t = {k: [] for k in algos}
        for r in range(a.rounds + 1):
            for k in algos:
                i, dt = k
                xs, ws = data[dt]
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                ops.conv2d(xs, ws, cout, 3, 3, (1, 1), (1, 1), bias=bias, act=ops.ACT_LRELU, out=outs[dt], algo=algo_of(i))
                e.record()
                torch.cuda.synchronize()
                if r:
                    t[k].append(s.elapsed_time(e))
        med = lambda v: sorted(v)[len(v) // 2]
        print("%-28s " % name + "  ".join("%s id %2d: %6.1f TF/s (%.2f ms)" % ("x3" if k[1] == SP else "x2", k[0], flops / med(t[k]) / 1e9, med(t[k])) for k in algos), flush=True)
        del data, outs
    # ---------------- register-staged path: ToRGB of the 128-px generator level (1x1, 128 -> 3 (32), style modulation prologue, tanh)
    n, hw, cin = 512, 128, 128
    x = torch.randn((n, hw, hw, cin), device=dev)
    wt = torch.randn((3, cin, 1, 1), device=dev) * 0.1
    sc = torch.rand((n, cin), device=dev) + 0.5
    bias = torch.zeros(32, device=dev)
    for name, dt in (("x3", SP), ("x2", MX)):
        xs, ws = ops.convert(x, dt), packing.pack_conv_weight(wt, dt, cout_mult=32)
        out = torch.empty((n, hw, hw, 32), dtype=dt, device=dev)
        ts = []
        for r in range(a.rounds + 1):
            s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_.record()
            ops.conv2d(xs, ws, 32, in_scale=sc, bias=bias, act=ops.ACT_TANH, out=out)
            e_.record()
            torch.cuda.synchronize()
            if r:
                ts.append(s_.elapsed_time(e_))
        ms = sorted(ts)[len(ts) // 2]
        print("torgb_128 (register-staged) %s: %.2f ms, %.2f TB/s read" % (name, ms, xs.numel() * 4 / ms / 1e9), flush=True)


if __name__ == "__main__":
    main()
