#!/usr/bin/env python
"""A/B micro-benchmark of the two implicit-GEMM conv kernels on representative MARCONet layer shapes
(interleaved rounds in one process, median of HIP-event timings).  python tools/conv_bench.py [--rounds 7]"""
import argparse
import os

os.environ.setdefault("MNET_ALLOW_DIAGNOSTIC_KERNELS", "1")     # this tool pins the diagnostic f16 tile ids 11-15 on purpose
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = [  # name, n, h, w, c0, c1, cout, k
    ("sr_trunk_64x1024_256", 16, 64, 1024, 256, 0, 256, 3),
    ("sr_glyph64_512to256", 256, 64, 64, 512, 0, 256, 3),
    ("sr_glyph64_256", 256, 64, 64, 256, 0, 256, 3),
    ("gan_32_512", 256, 32, 32, 512, 0, 512, 3),
    ("gan_128_256to128", 256, 128, 128, 256, 0, 128, 3),
    ("gan_128_128", 256, 128, 128, 128, 0, 128, 3),
    ("sr_final_256to128", 32, 64, 1024, 256, 0, 128, 3),
    ("resnet_l5_512", 64, 8, 512, 512, 0, 512, 3),
    ("resnet_l5_1x1", 64, 8, 512, 512, 0, 512, 1),
    ("sr_body32_cat", 64, 32, 512, 256, 64, 256, 3),
    ("sr_final_128to64", 32, 128, 2048, 128, 0, 64, 3),
    ("sr_final_64", 32, 128, 2048, 64, 0, 64, 3),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--algos", default="1,2", help="1 = register-staged, 2 = LDS-DMA auto pick, 16+id = pinned LDS-DMA tile config")
    ap.add_argument("--only", default="", help="comma-separated substrings of shape names")
    a = ap.parse_args()
    from marconet_amd import ops
    from marconet_amd._lib import MarconetHipError
    dev = "cuda"
    algos = [int(v) for v in a.algos.split(",")]
    print("%-26s " % "shape (TF/s per algo)" + " ".join("%9s" % ("algo%d" % g) for g in algos))
    for name, n, h, w, c0, c1, cout, k in SHAPES:
        if a.only and not any(o in name for o in a.only.split(",")):
            continue
        g = torch.Generator(device=dev).manual_seed(1)
        x0 = (torch.rand((n, h, w, c0), device=dev, generator=g) - 0.5).half()
        x1 = (torch.rand((n, h, w, c1), device=dev, generator=g) - 0.5).half() if c1 else None
        wt = ((torch.rand((cout, k, k, c0 + c1), device=dev, generator=g) - 0.5) * 0.05).half()
        bias = torch.zeros(cout, device=dev)
        flops = 2.0 * n * h * w * cout * k * k * (c0 + c1)
        t = {g_: [] for g_ in algos}
        for r in range(a.rounds + 1):
            for algo in algos:
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                try:
                    s.record()
                    ops.conv2d(x0, wt, cout, k, k, (1, 1), (k // 2, k // 2), x1=x1, bias=bias, act=ops.ACT_LRELU, algo=algo)
                    e.record()
                    torch.cuda.synchronize()
                except MarconetHipError:
                    continue
                if r:
                    t[algo].append(s.elapsed_time(e))
        med = lambda v: sorted(v)[len(v) // 2] if v else float("nan")
        print("%-26s " % name + " ".join("%9.1f" % (flops / med(t[g_]) / 1e9) for g_ in algos), flush=True)


if __name__ == "__main__":
    main()
