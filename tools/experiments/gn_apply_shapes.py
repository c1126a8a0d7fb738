import os, sys, torch
sys.path.insert(0, "/root/repo")
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
for shape in ((4096, 32, 32, 512), (256, 128, 128, 512), (16384, 16, 16, 512), (4096, 32, 32, 1024), (4096, 64, 64, 256), (4096, 64, 64, 512), (256, 128, 2048, 64), (4096, 32, 32, 64), (1024, 64, 64, 512)):
    n, h, w, c = shape
    x = ops.convert(torch.randn(shape, device=dev), packing.MX_DTYPE)
    sc, sh = torch.rand((n, c), device=dev) + 0.5, torch.randn((n, c), device=dev)
    y = ops.affine_act(x, sc, sh, swish=True)
    nb = n * h * w * c * 4
    t = timeit(lambda: ops.affine_act(x, sc, sh, swish=True, out=y))
    t2 = timeit(lambda: ops.affine_act(x, sc, sh, swish=False, out=y))
    print("%-24s swish %7.3f ms %6.0f GB/s | affine only %7.3f ms %6.0f GB/s" % (shape, t, 2 * nb / t / 1e6, t2, 2 * nb / t2 / 1e6))
    del x, y
