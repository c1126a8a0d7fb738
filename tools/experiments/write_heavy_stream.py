"""What is the chip's rate for a write-heavy stream (the up-sample pass reads N and writes 4 N bytes)?  torch's own elementwise kernels: copy (1 : 1), fill (0 : 1),
broadcast copy (1 : 4)."""
import torch
dev = "cuda"
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
N = 1 << 28                                  # 1 GiB of fp32
x = torch.randn(N, device=dev)
y = torch.empty(N, device=dev)
y4 = torch.empty((4, N), device=dev)
t = timeit(lambda: y.copy_(x));            print("copy  1:1   %7.3f ms  %6.0f GB/s (read + write)" % (t, 2 * N * 4 / t / 1e6))
t = timeit(lambda: y4.fill_(1.0));         print("fill  0:1   %7.3f ms  %6.0f GB/s (write)" % (t, 4 * N * 4 / t / 1e6))
t = timeit(lambda: y4.copy_(x.expand(4, N))); print("bcast 1:4   %7.3f ms  %6.0f GB/s (read + write, algorithmic 5 N)" % (t, 5 * N * 4 / t / 1e6))
t = timeit(lambda: torch.add(x, 1.0, out=y)); print("add   1:1   %7.3f ms  %6.0f GB/s" % (t, 2 * N * 4 / t / 1e6))
