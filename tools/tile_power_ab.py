#!/usr/bin/env python
"""Sustained A/B of pinned LDS-DMA tile configurations on the dominant layer shape (256 -> 256 3x3 @ 64x1024): algorithmic TFLOP/s
from HIP events over a few seconds of back-to-back launches, with rocm-smi package power and shader clock sampled meanwhile
(the 32x32x16 split-half tile at the power wall; DVFS hides cycle savings, so rate, clock and W go together).
    python tools/tile_power_ab.py [--seconds 6] [--n 64] [--only x3:11,x3:20,x2:6,x2:11]
With --launches K it runs K launches per arm and exits (for `rocprofv3 --pmc`, which wants few dispatches)."""
import argparse
import os
import re
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def algo_of(i):
    """LDS-DMA tile id -> algo code; "s<cfg>" = strip configuration <cfg> (MNET_CONV_ALGO_STRIP_CFG0 + cfg)"""
    if isinstance(i, str) and i.startswith("s"):
        return 32 + int(i[1:])
    i = int(i)
    return 16 + i if i < 16 else 64 + i - 16


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.stop, self.w, self.clk = False, [], []

    def run(self):
        while not self.stop:
            try:
                out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
            except Exception:      # noqa: BLE001
                return
            m = re.search(r"Package Power \(W\): ([\d.]+)", out)
            c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
            if m:
                self.w.append(float(m.group(1)))
            if c:
                self.clk.append(float(c.group(1)))
            time.sleep(0.5)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=6.0)
    ap.add_argument("--n", type=int, default=64)
    ap.add_argument("--launches", type=int, default=0)
    ap.add_argument("--only", default="x3:11,x3:20,x2:6,x2:11")
    ap.add_argument("--shape", default="", help="n,h,w,cin,cout of the 3x3 layer (default: --n,64,1024,256,256)")
    ap.add_argument("--ragged", action="store_true", help="valid_w[n] = w - (n % 5) * 3 (glyph windows of different widths)")
    ap.add_argument("--zeros", action="store_true", help="zero-filled operands: the same instruction stream at a fraction of the switching power "
                                                         "(what the schedule does when the package power cap does not hold the clock down)")
    a = ap.parse_args()
    from marconet_amd import ops, packing
    dev = "cuda"
    torch.manual_seed(0)
    n, h, w, cin, cout = a.n, 64, 1024, 256, 256
    if a.shape:
        n, h, w, cin, cout = (int(v) for v in a.shape.split(","))
    vw = torch.tensor([w - (i % 5) * 3 for i in range(n)], dtype=torch.int32, device=dev) if a.ragged else None
    x = torch.randn((n, h, w, cin), device=dev)
    wt = torch.randn((cout, cin, 3, 3), device=dev) * 0.02
    if a.zeros:
        x, wt = torch.zeros_like(x), torch.zeros_like(wt)
    bias = torch.zeros(cout, device=dev)
    dts = {"x3": packing.SPLIT_DTYPE, "x2": packing.MX_DTYPE, "f16": torch.float16}
    data = {k: (ops.convert(x, dt), packing.pack_conv_weight(wt, dt), torch.empty((n, h, w, cout), dtype=dt, device=dev)) for k, dt in dts.items()
            if any(arm.startswith(k + ":") for arm in a.only.split(","))}
    del x
    flops = 2.0 * n * h * w * cout * 9 * cin
    for arm in a.only.split(","):
        mode, i = arm.split(":")
        xs, ws, out = data[mode]
        run = lambda: ops.conv2d(xs, ws, cout, 3, 3, (1, 1), (1, 1), bias=bias, act=ops.ACT_LRELU, out=out, valid_w=vw, algo=algo_of(i))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        if a.launches:
            for _ in range(a.launches):
                run()
            torch.cuda.synchronize()
            print("%s id %s: %d launches" % (mode, i, a.launches), flush=True)
            continue
        smp = Sampler()
        smp.start()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        k, t0 = 0, time.perf_counter()
        s.record()
        while time.perf_counter() - t0 < a.seconds:
            for _ in range(10):
                run()
            k += 10
            torch.cuda.synchronize()
        e.record()
        torch.cuda.synchronize()
        smp.stop = True
        smp.join(timeout=15)
        ms = s.elapsed_time(e) / k
        avg = lambda v: sum(v) / len(v) if v else float("nan")
        print("%s id %3s: %6.1f TFLOP/s algorithmic (%.3f ms/launch over %d launches) | package %.0f W (max %.0f) | sclk %.0f MHz (%d samples)"
              % (mode, i, flops / ms / 1e9, ms, k, avg(smp.w), max(smp.w) if smp.w else float("nan"), avg(smp.clk), len(smp.w)), flush=True)


if __name__ == "__main__":
    main()
