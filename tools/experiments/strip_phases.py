"""Where a wave's time goes in the 64x512 strip tile (diagnostic build: tools/experiments/strip_stamps.patch, -DMNET_STRIP_STAMPS=1, wrong results): s_memrealtime-style
stamps (__builtin_readcyclecounter: a constant-rate counter, 100 MHz) summed per phase and wave."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from marconet_amd import _lib, ops, packing
n, h, w, cin, cout = 64, 128, 2048, 64, 64
torch.manual_seed(0)
x = torch.randn((n, h, w, cin), device="cuda"); wt = torch.randn((cout, cin, 3, 3), device="cuda") * 0.05
xs, ws = ops.convert(x, packing.MX_DTYPE), packing.pack_conv_weight(wt, packing.MX_DTYPE); del x
out = torch.empty((n, h, w, cout), dtype=packing.MX_DTYPE, device="cuda"); bias = torch.zeros(cout, device="cuda")
algo = _lib.ALGO_STRIP_CFG0 + 1
for _ in range(3): ops.conv2d(xs, ws, cout, 3, 3, (1, 1), (1, 1), bias=bias, act=ops.ACT_LRELU, out=out, algo=algo)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(5): ops.conv2d(xs, ws, cout, 3, 3, (1, 1), (1, 1), bias=bias, act=ops.ACT_LRELU, out=out, algo=algo)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 5
raw = packing.untag(out).view(torch.int32).reshape(-1)[: 256 * 8 * 16].cpu().reshape(-1, 16)
r = raw[raw[:, 0] == 0x5157a3b8].double()
print("%.3f ms per launch; %d wave records; slabs per wave %.0f, tiles per wave %.1f" % (ms, r.shape[0], r[:, 1].mean(), r[:, 2].mean()))
tot = r[:, 3:9].sum(1)
names = ["s_waitcnt vmcnt(0)", "s_barrier", "DMA issue (weights; strip every third slab)", "compute (LDS reads, 8 + 4 MFMAs, conversions)", "epilogue (+ its barrier)", "tile prologue (clear, scales)"]
for k, nm in enumerate(names):
    v = r[:, 3 + k]
    print("  %-48s %5.1f %% of the wave's stamped time   (%.0f counter ticks x16 per tile)" % (nm, 100 * (v / tot).mean(), (v / r[:, 2]).mean()))
