#!/usr/bin/env python
"""Where a wave's cycles go in the slab loop of the one-wave-per-SIMD fp16+8 tile (conv_dma_w4.hip, id 16): run with a library built with -DW4_STAMPS=1
(tools/build_variant.sh w4_stamps conv_dma_w4 -DW4_STAMPS=1; MARCONET_HIP_LIB=tools/_build/w4_stamps/libmarconet_hip.so) — wrong results, phase sums over the output.
    python tools/w4_phases.py [--shape n,h,w,cin,cout] [--zeros]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NAMES = ["16 scaled MFMAs + 16 DMA pieces + 12 front reads", "cursor + tile-closing epilogue (per slab)", "f16 part up to MFMA 16 (reads, conversions)",
         "prep: next slab's addresses (+ set-up at a tile crossing)", "f16 part, MFMAs 17-31", "s_waitcnt vmcnt(0)", "s_barrier"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="64,64,1024,256,256")
    ap.add_argument("--zeros", action="store_true")
    ap.add_argument("--epilogue", default="plain", choices=["plain", "gn", "residual", "scales", "all"], help="which build of the tile: extra epilogue terms of the launch")
    a = ap.parse_args()
    from marconet_amd import _lib, ops, packing
    n, h, w, cin, cout = (int(v) for v in a.shape.split(","))
    dev = "cuda"
    torch.manual_seed(0)
    x = torch.randn((n, h, w, cin), device=dev)
    wt = torch.randn((cout, cin, 3, 3), device=dev) * 0.02
    if a.zeros:
        x, wt = torch.zeros_like(x), torch.zeros_like(wt)
    xs, ws = ops.convert(x, packing.MX_DTYPE), packing.pack_conv_weight(wt, packing.MX_DTYPE)
    del x
    out = torch.empty((n, h, w, cout), dtype=packing.MX_DTYPE, device=dev)
    bias = torch.zeros(cout, device=dev)
    algo = _lib.ALGO_DMA_CFG16 + 0
    kw = {}
    if a.epilogue in ("gn", "all"): kw["gn_partial"] = ops.gn_partial_buffer(n, h, w, cout, dev)
    if a.epilogue in ("residual", "all"): kw["residual"] = ops.convert(torch.randn((n, h, w, cout), device=dev), packing.MX_DTYPE)
    if a.epilogue in ("scales", "all"): kw["out_scale"] = torch.rand((n, cout), device=dev) + 0.5; kw["post_scale"] = torch.rand((n, cout), device=dev) + 0.5
    for _ in range(2):
        ops.conv2d(xs, ws, cout, 3, 3, (1, 1), (1, 1), bias=bias, act=ops.ACT_LRELU, out=out, algo=algo, **kw)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        ops.conv2d(xs, ws, cout, 3, 3, (1, 1), (1, 1), bias=bias, act=ops.ACT_LRELU, out=out, algo=algo, **kw)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    print("id 16 (this library): %.3f ms per launch, %.1f TFLOP/s algorithmic%s" % (ms, 2.0 * n * h * w * cout * 9 * cin / ms / 1e9, " (zero operands)" if a.zeros else ""))
    raw = packing.untag(out).view(torch.int32).reshape(-1)[: 256 * 4 * 16].cpu().reshape(-1, 16)
    ok = raw[:, 0] == 0x5157a3b7
    if not ok.any():
        print("no stamp records: this library was not built with -DW4_STAMPS=1")
        return
    r = raw[ok].double()
    slabs, tiles = r[:, 1], r[:, 2]
    tot = r[:, 3:16].sum(1) / slabs
    nk = 9 * 2 * cin // 64
    print("%d wave records; hot slabs per wave %.0f, tiles closed per wave %.1f; cycles per slab (mean over waves) %.0f [min %.0f max %.0f]; 2048 of them are matrix-pipe work"
          % (int(ok.sum()), slabs.mean(), tiles.mean(), tot.mean(), tot.min(), tot.max()))
    wave_id = torch.arange(raw.shape[0])[ok] % 4
    for k, nm in enumerate(NAMES):
        v = r[:, 3 + k] / slabs
        print("  %-62s %7.0f cycles (%4.1f %%)   by wave: %s" % (nm, v.mean(), 100 * v.mean() / tot.mean(), " ".join("%6.0f" % v[wave_id == j].mean() for j in range(4))))
    if r[:, 10].sum() > 0 and r[:, 11:16].sum() == 0:      # -DW4_STAMPS=1: slot 7 = the prep phase of the iterations that cross into the next tile (set-up included)
        print("    prep phase of a tile-crossing iteration (5 MFMAs + the next tile's set-up): %.0f cycles per tile; of an ordinary iteration: %.0f cycles per slab"
              % ((r[:, 10] / tiles.clamp(min=1)).mean(), (r[:, 6] / (slabs - tiles)).mean()))
    if r[:, 11:16].sum() > 0:      # -DW4_STAMPS=2: sub-phases of the epilogue (phase 1 then holds only what is left outside them)
        sub = ["first step's parameter requests (per tile)", "accumulator reads + parameters + arithmetic + encode (8 steps)", "next step's requests (8 steps)",
               "LDS round 1 + stores (8 steps)", "LDS round 2 + stores (8 steps)", "GroupNorm fold + store (8 steps)"]
        for k, nm in enumerate(sub):
            print("    epilogue: %-66s %8.0f cycles per tile" % (nm, (r[:, 10 + k] / tiles.clamp(min=1)).mean()))
    print("slabs per tile %d: phase 1 per TILE %.0f cycles (epilogue + zeroing), phase 3 per tile beyond %d x the ordinary prep: see the per-slab figure"
          % (nk, (r[:, 4] / tiles.clamp(min=1)).mean(), nk))


if __name__ == "__main__":
    main()
