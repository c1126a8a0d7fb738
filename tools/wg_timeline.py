#!/usr/bin/env python
"""Per-workgroup timeline of the 256x256 LDS-DMA conv kernel.  The DIAGNOSTIC tile configurations 11/12/13 write
wall-clock stamps (entry / k-loop start / k-loop end / stores acknowledged) over the output tensor: this prints the
prologue / k-loop / epilogue durations and the idle gap between consecutive workgroups on one CU.
   13: normal kernel      11: every tile stores over tile 0 (L2-resident writes)      12: no output stores"""
import os

os.environ.setdefault("MNET_ALLOW_DIAGNOSTIC_KERNELS", "1")     # this tool pins the diagnostic f16 tile ids 11-15 on purpose
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def report(name, st):
    ok = st[:, 0] == 0x7157a3b5
    st = st[ok]
    t0, t1, t2, t3, hw, xcc, cyc = (st[:, i].astype(np.int64) for i in range(1, 8))
    cu = ((hw >> 8) & 0xF) | (((hw >> 13) & 0x7) << 4) | (((hw >> 12) & 1) << 7) | ((xcc & 0xF) << 8)
    us = 0.01     # wall_clock64 ticks at 100 MHz
    print("%s: %d of %d workgroups stamped, %d distinct CU ids" % (name, ok.sum(), len(ok), len(np.unique(cu))))
    for lbl, d in (("prologue", t1 - t0), ("k-loop", t2 - t1), ("epilogue", t3 - t2)):
        print("  %-9s mean %7.2f us  p10 %7.2f  p90 %7.2f" % (lbl, d.mean() * us, np.percentile(d, 10) * us, np.percentile(d, 90) * us))
    gaps = []
    for c_ in np.unique(cu):
        m = cu == c_
        o = np.argsort(t0[m])
        gaps.extend((t0[m][o][1:] - t3[m][o][:-1]).tolist())
    gaps = np.array(gaps)
    print("  gap between consecutive workgroups on a CU: mean %.2f us  p10 %.2f  p90 %.2f" % (gaps.mean() * us, np.percentile(gaps, 10) * us, np.percentile(gaps, 90) * us))
    print("  kernel span %.1f us;  k-loop shader clock (s_memtime cycles / wall time): %.2f GHz" % ((t3.max() - t0.min()) * us, np.median(cyc / np.maximum(t2 - t1, 1)) / 10.0))


def main():
    from marconet_amd import ops
    dev = "cuda"
    for name, n, h, w, c, cout in (("sr_trunk 256->256 K=2304", 16, 64, 1024, 256, 256), ("glyph 512->256 K=4608", 256, 64, 64, 512, 256)):
        g = torch.Generator(device=dev).manual_seed(1)
        x = (torch.rand((n, h, w, c), device=dev, generator=g) - 0.5).half()
        wt = ((torch.rand((cout, 3, 3, c), device=dev, generator=g) - 0.5) * 0.05).half()
        npix = n * h * w
        ntile = npix // 256
        for cfg, label in ((13, "normal"), (13, "normal, ZERO-FILLED operands"), (11, "stores aliased onto tile 0"), (12, "no stores")):
            if "ZERO" in label:
                x, wt = torch.zeros_like(x), torch.zeros_like(wt)
            for _ in range(2):
                y = ops.conv2d(x, wt, cout, 3, 3, (1, 1), (1, 1), act=ops.ACT_LRELU, algo=16 + cfg)
            torch.cuda.synchronize()
            if cfg == 13:     # stamps at the head of each tile (cout == 256: a tile is 256 consecutive 512-byte pixel rows)
                st = y.reshape(npix, cout)[::256, :32]
            else:             # stamps of workgroup wg at pixel row 256 + 32*wg
                st = y.reshape(npix, cout)[256:256 + 32 * ntile:32, :32]
            report("%s [%s]" % (name, label), st.contiguous().cpu().view(torch.int64).reshape(-1, 8).numpy())


if __name__ == "__main__":
    main()
