import os, sys, torch
sys.path.insert(0, "/root/repo")
from marconet_amd import _lib, ops, packing
n, h, w, cin, cout = 64, 64, 1024, 256, 256
torch.manual_seed(0)
x = torch.randn((n, h, w, cin), device="cuda"); wt = torch.randn((cout, cin, 3, 3), device="cuda") * 0.02
xs, ws = ops.convert(x, packing.MX_DTYPE), packing.pack_conv_weight(wt, packing.MX_DTYPE); del x
out = torch.empty((n, h, w, cout), dtype=packing.MX_DTYPE, device="cuda"); bias = torch.zeros(cout, device="cuda")
for _ in range(3): ops.conv2d(xs, ws, cout, 3, 3, (1, 1), (1, 1), bias=bias, act=ops.ACT_LRELU, out=out, algo=_lib.ALGO_DMA_CFG16 + 0)
torch.cuda.synchronize()
raw = packing.untag(out).view(torch.int32).reshape(-1)[: 256 * 4 * 16].cpu().reshape(-1, 16)
r = raw[raw[:, 0] == 0x5157a3b7].double()
slabs, tiles = r[:, 1].mean(), r[:, 2].mean()
nk = 72
print("slabs/wave %.0f tiles %.1f" % (slabs, tiles))
names = {5: "other taps (6 of 9 slabs)", 7: "tap 0 of a slice, not the tile's first slab (7 of 72)", 8: "tap 3 (8 of 72)", 9: "tap 6 (8 of 72)", 10: "first slab of a tile (1 of 72)"}
cnt = {5: slabs * 48 / 72, 7: slabs * 7 / 72, 8: slabs * 8 / 72, 9: slabs * 8 / 72, 10: tiles}
for k in (10, 7, 8, 9, 5):
    v = r[:, 3 + k].mean()
    print("  vmcnt wait, %-55s total %9.0f cycles/wave = %6.0f per such slab" % (names[k], v, v / cnt[k]))
