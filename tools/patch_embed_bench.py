import sys, torch
sys.path.insert(0, ".")
from marconet_amd import ops
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
for B in (4, 64):
    feat = torch.rand((B, 8, 512, 512), device=dev, generator=g) - 0.5
    w = (torch.rand((512, 8, 8, 512), device=dev, generator=g) - 0.5) * 0.01
    b = torch.zeros(512, device=dev)
    ts = []
    for r in range(6):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); ops.conv2d(feat, w, 512, 8, 8, (8, 8), (0, 0), bias=b); e.record(); torch.cuda.synchronize()
        if r: ts.append(s.elapsed_time(e))
    ms = sorted(ts)[len(ts) // 2]
    print("patch-embed B=%d: %.3f ms  %.1f TF/s" % (B, ms, 2.0 * B * 64 * 512 * 32768 / ms / 1e9))
