"""Batch-1 latency of the script-shaped call (test_sr.py:77 loop, one strip at a time): eager launches vs one HIP-graph replay.
    python tools/graph_latency.py [glyphs=16] [iters=50] [precision=fp16]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from marconet_amd import networks                                     # noqa: E402
from marconet_amd.pipeline import GraphedForward, MarconetPipeline    # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
prec = sys.argv[3] if len(sys.argv) > 3 else "fp16"
torch.manual_seed(0)
from marconet_amd import synthetic                                    # noqa: E402
nets = [networks.TextContextEncoderV2(), networks.TSPGAN(), networks.TSPSRNet()]
for m, sd in zip(nets, (synthetic.make_encoder_state_dict(), synthetic.make_gan_state_dict(), synthetic.make_sr_state_dict())):
    m.load_state_dict(sd, strict=True)              # seeded checkpoints with O(1) activations (default-initialised spectral-norm vectors blow up)
nets = [m.eval().cuda() for m in nets]
pipe = MarconetPipeline(*nets, precision=prec, check_finite=False)
lq = torch.rand(1, 3, 32, 512, device="cuda")
labels = [torch.randint(0, 6736, (n,))]
locs = torch.zeros(1, 32)
locs[0, 0::2] = (torch.arange(16) + 0.5) / 16
locs[0, 1::2] = 1.0 / 32


def clock(f):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


eager = clock(lambda: pipe.forward_batch(lq, labels, locs))
gf = GraphedForward(pipe, 1, [n])
graphed = clock(lambda: gf(lq, labels, locs))
same = torch.equal(gf(lq, labels, locs), pipe.forward_batch(lq, labels, locs))
print("batch 1, %d glyphs, %s: eager %.2f ms/image, HIP graph %.2f ms/image (%.2fx), identical=%s" % (n, prec, eager, graphed, eager / graphed, same))
