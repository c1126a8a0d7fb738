"""worker of tests/test_ab_forms_gpu.py: digests of a few kernels' outputs on fixed seeded inputs, printed as JSON — run once per setting of the
A/B environment knobs (they are read once per process by the library)"""
import hashlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def digest(t):
    from marconet_amd import packing
    t = packing.untag(t).detach().contiguous().cpu()
    return hashlib.sha256(t.reshape(-1).view(torch.uint8).numpy().tobytes()).hexdigest()[:24]


def main():
    from marconet_amd import ops, packing
    dev = "cuda"
    g = torch.Generator().manual_seed(7)
    out = {}
    for name, dt in (("f32", torch.float32), ("f16", torch.float16), ("split", packing.SPLIT_DTYPE), ("mx", packing.MX_DTYPE)):
        S, C, B = 32, 256, 3
        FW = S * 8
        feat = packing.from_float(torch.randn((B, S, FW, C), generator=g), dt).to(dev)
        G = 7
        prior = packing.from_float(torch.randn((G, S, S, C), generator=g) * 1.5 + 0.2, dt).to(dev)
        g_img = torch.tensor([0, 0, 1, 1, 2, 2, 2], dtype=torch.int32, device=dev)
        g_w = torch.tensor([S, 21, S, 17, S, 5, S - 1], dtype=torch.int32, device=dev)
        g_x1 = torch.tensor([0, 40, 100, FW - 17, 3, 77, 120], dtype=torch.int32, device=dev)
        g_y1 = (S // 2 - g_w // 2).to(torch.int32)
        gamma, beta = torch.rand((2 * C,), generator=g).to(dev) + 0.5, (torch.randn((2 * C,), generator=g) * 0.3).to(dev)
        o, sc, sh = ops.adain_crop_concat_gn(prior, feat, g_img, g_x1, g_y1, g_w, gamma, beta, 1e-6, split=False)
        out["adain.%s" % name] = [digest(o), digest(sc), digest(sh)]
        x = packing.from_float(torch.randn((3, 36, 50, 128), generator=g), dt).to(dev)
        wt = (torch.randn((3, 128), generator=g) / 11.0).to(dev)
        st = (torch.rand((3, 128), generator=g) + 0.5).to(dev)
        bias = torch.tensor([0.1, -0.2, 0.05, 0.0], device=dev)
        skip = torch.tanh(torch.randn((3, 18, 25, 4), generator=g)).to(dev)
        out["torgb.%s" % name] = digest(ops.torgb(x, wt, st, None, bias, skip))
    if "--chain" in sys.argv:
        from marconet_amd import checkpoints, synthetic
        from marconet_amd.pipeline import MarconetPipeline
        sde, sdg, sds, _ = checkpoints.load_state_dicts(path="")
        pipe = MarconetPipeline(*checkpoints.build_networks(sde, sdg, sds, dev), precision="fp16x2")
        lq = synthetic.make_lq(5, 2, [512, 300]).to(dev)
        labels = [synthetic.make_labels(6, 5), synthetic.make_labels(7, 3)]
        locs = synthetic.make_locs([5, 3], [512, 300])
        out["chain.sr.fp16x2"] = digest(pipe.forward_batch(lq, labels, locs))
    torch.cuda.synchronize()
    print("AB_DIGESTS " + json.dumps(out, sort_keys=True))


if __name__ == "__main__":
    main()
