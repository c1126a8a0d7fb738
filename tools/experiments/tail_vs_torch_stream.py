"""The HBM-bound tail kernels against torch's own elementwise kernels on the SAME box and the same byte counts: is there bandwidth left above the 5.1-5.4 TB/s of
GroupNorm apply and the 4.0 TB/s of the up-sample?  Shapes of the bench step (batch 256 x 16 glyphs)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from marconet_amd import ops, packing
dev = "cuda"
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
def torch_ref(nbytes_in, ratio):
    a = torch.empty(nbytes_in // 4, device=dev).normal_()
    if ratio == 1:
        b = torch.empty_like(a)
        t = timeit(lambda: torch.add(a, 1.0, out=b))
    else:
        b = torch.empty((ratio, a.numel()), device=dev)
        t = timeit(lambda: torch.add(a.expand(ratio, a.numel()), 1.0, out=b))
    return t, (1 + ratio) * nbytes_in / t / 1e6
for (name, shape) in (("glyph level 4096 x 32x32 x 512", (4096, 32, 32, 512)), ("SR 256 x 128x2048 x 64", (256, 128, 2048, 64)), ("SR 256 x 64x1024 x 256", (256, 64, 1024, 256))):
    n, h, w, c = shape
    x = ops.convert(torch.randn(shape, device=dev) , packing.MX_DTYPE)
    sc, sh = torch.rand((n, c), device=dev) + 0.5, torch.randn((n, c), device=dev)
    y = ops.affine_act(x, sc, sh, swish=True)
    nb = n * h * w * c * 4
    t = timeit(lambda: ops.affine_act(x, sc, sh, swish=True, out=y))
    tt, tg = torch_ref(nb, 1)
    print("%-34s GroupNorm apply + swish (fp16+8): %7.3f ms %6.0f GB/s | torch add of the same bytes: %7.3f ms %6.0f GB/s" % (name, t, 2 * nb / t / 1e6, tt, tg))
    del y
for (name, shape) in (("up-sample 4096 x 32x32 x 512", (4096, 32, 32, 512)), ("up-sample 256 x 64x1024 x 64", (256, 64, 1024, 64)), ("up-sample 4096 x 16x16 x 512", (4096, 16, 16, 512))):
    n, h, w, c = shape
    x = ops.convert(torch.randn(shape, device=dev), packing.MX_DTYPE)
    nb = n * h * w * c * 4
    t = timeit(lambda: ops.upsample2x(x))
    tt, tg = torch_ref(nb, 4)
    print("%-34s bilinear x2 (fp16+8): %7.3f ms %6.0f GB/s | torch broadcast add 1 -> 4 of the same bytes: %7.3f ms %6.0f GB/s" % (name, t, 5 * nb / t / 1e6, tt, tg))
