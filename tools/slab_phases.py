#!/usr/bin/env python
"""Where a wave's cycles go inside one hot k-slab of the dominant fp16+8 tile (256x256, 8 waves): the DIAGNOSTIC build id 14
(conv_dma_kernel<..., DBG = 6>, wrong results) sums s_memtime differences per phase over the launch and writes them over the output.
    MNET_ALLOW_DIAGNOSTIC_KERNELS=1 python tools/slab_phases.py [--shape n,h,w,cin,cout]"""
import argparse
import os
import sys

os.environ.setdefault("MNET_ALLOW_DIAGNOSTIC_KERNELS", "1")
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NAMES = ["LDS reads + 16 f16 MFMAs", "DMA issue (8 pieces)", "fp8 cvt + 8 scaled MFMAs", "s_waitcnt vmcnt(0)", "s_barrier"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="64,64,1024,256,256")
    ap.add_argument("--zeros", action="store_true")
    ap.add_argument("--swp", action="store_true", help="the software-pipelined tile (id 15) instead of the lock-step tile (id 11)")
    a = ap.parse_args()
    if a.swp:
        os.environ["MNET_DIAG_SWP"] = "1"
        NAMES[:] = ["tile-closing epilogue (per slab: x slabs per tile = per tile)", "front LDS reads + 8 scaled MFMAs + 8 DMA pieces", "16 f16 MFMAs + fp8-side reads + cvt",
                    "s_waitcnt vmcnt(0)", "s_barrier", "s_waitcnt vmcnt(0) of the slab after an epilogue (per slab)"]
    from marconet_amd import ops, packing
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
    for algo, tag in ((16 + (15 if a.swp else 11), "production id %d" % (15 if a.swp else 11)), (16 + 14, "diagnostic id 14")):
        for _ in range(2):
            ops.conv2d(xs, ws, cout, 3, 3, (1, 1), (1, 1), bias=bias, act=ops.ACT_LRELU, out=out, algo=algo)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            ops.conv2d(xs, ws, cout, 3, 3, (1, 1), (1, 1), bias=bias, act=ops.ACT_LRELU, out=out, algo=algo)
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 5
        print("%s: %.3f ms per launch, %.1f TFLOP/s algorithmic" % (tag, ms, 2.0 * n * h * w * cout * 9 * cin / ms / 1e9))
    raw = packing.untag(out).view(torch.int32).reshape(-1)[: 256 * 8 * 8].cpu().reshape(-1, 8)
    ok = raw[:, 0] == 0x5157a3b6
    r = raw[ok].double()
    print("%d wave records" % int(ok.sum()))
    slabs = r[:, 1]
    tot = (r[:, 2:8] if a.swp else r[:, 2:7]).sum(1) / slabs
    hw = raw[ok][:, 7]
    wave_id = torch.arange(raw.shape[0])[ok] % 8
    print("hot slabs per wave %.0f; cycles per hot slab (mean over waves) %.0f  [min %.0f max %.0f]" % (slabs.mean(), tot.mean(), tot.min(), tot.max()))
    for k, nm in enumerate(NAMES):
        v = r[:, 2 + k] / slabs
        print("  %-50s %7.0f cycles (%4.1f %%)   waves 0-3: %6.0f   waves 4-7: %6.0f" % (nm, v.mean(), 100 * v.mean() / tot.mean(), v[wave_id < 4].mean(), v[wave_id >= 4].mean()))
    if a.swp:
        nk = 9 * 2 * cin // 64                     # physical channels: two per logical channel in the blocked storages
        print("slabs per tile %d: epilogue %.0f cycles per tile, vmcnt(0) after it %.0f cycles per tile, i.e. %.1f slab times per tile"
              % (nk, (r[:, 2] / slabs).mean() * nk, (r[:, 7] / slabs).mean() * nk, ((r[:, 2] + r[:, 7]) / slabs).mean() * nk / tot.mean()))
    else:
        simd = (hw >> 4) & 3
        print("SIMD of waves 0..7 in workgroup 0:", [int(v) for v in simd[:8]])


if __name__ == "__main__":
    main()
