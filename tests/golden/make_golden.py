"""Generates tests/golden/golden_v1.npz from the REAL reference modules (/root/reference, imported unmodified
with the basicsr.ops.fused_act stub of oracle/ref_loader.py) on the seeded synthetic checkpoints of
oracle/synth.py.  Run in the build container only (the reference tree does not exist on the GPU box):

    python tests/golden/make_golden.py

The reference ships no golden tensors; these reference-run-here outputs are the pin for
oracle/marconet_oracle.py (tests/test_oracle.py) and, through it, for the HIP path (tests/test_modules_gpu.py).
Full tensors would be ~100 MB, so spatially strided samples plus whole-tensor moments are stored.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import synth  # noqa: E402
from oracle.ref_loader import load_reference_networks  # noqa: E402
from tests.golden import cases  # noqa: E402


def moments(t):
    t = t.double()
    return np.array([t.mean().item(), t.abs().mean().item(), t.pow(2).mean().sqrt().item(), t.abs().max().item()])


def main():
    torch.manual_seed(0)
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    nw = load_reference_networks()
    sde, sdg, sds = synth.make_encoder_state_dict(), synth.make_gan_state_dict(), synth.make_sr_state_dict()
    enc, gan, sr = nw.TextContextEncoderV2().eval(), nw.TSPGAN().eval(), nw.TSPSRNet().eval()
    enc.load_state_dict(sde, strict=True)
    gan.load_state_dict(sdg, strict=True)
    sr.load_state_dict(sds, strict=True)
    out = {}
    with torch.no_grad():
        # ---- encoder
        lq = cases.encoder_input()
        logits, locs, w = enc(lq)
        out["enc.argmax"] = logits.argmax(-1).numpy()
        out["enc.logits_s"] = cases.sample_logits(logits).numpy()
        out["enc.locs"] = locs.numpy()
        out["enc.w"] = w.numpy()
        top2 = logits.topk(2, dim=-1).values
        out["enc.min_top2_gap"] = np.array([(top2[..., 0] - top2[..., 1]).min().item()])
        # ---- GAN alone (random styles)
        styles, labels = cases.gan_input()
        img, p64, p32 = gan(styles=styles, labels=labels, noise=None)
        for k, t in (("img", img), ("p64", p64), ("p32", p32)):
            out["gan.%s_s" % k] = cases.sample_map(t, k).numpy()
            out["gan.%s_m" % k] = moments(t)
        # ---- SR chains
        for name in cases.SR_CASES:
            lq, locs, labels_per_img = cases.sr_input(name)
            _, _, w = enc(lq)
            p64s, p32s = [], []
            for b, lab in enumerate(labels_per_img):
                _, a, c = gan(styles=w[b:b + 1].repeat(lab.shape[0], 1), labels=lab, noise=None)
                p64s.append(a)
                p32s.append(c)
            y = sr(lq, p64s, p32s, locs)
            out["sr.%s.out_s" % name] = cases.sample_map(y, "sr").numpy()
            out["sr.%s.out_m" % name] = moments(y)
    # ---- checkpoint fingerprints (bit-reproducibility of oracle/synth.py across machines)
    for tag, sd, keys in (("enc", sde, cases.FINGERPRINT_KEYS["enc"]), ("gan", sdg, cases.FINGERPRINT_KEYS["gan"]),
                          ("sr", sds, cases.FINGERPRINT_KEYS["sr"])):
        for k in keys:
            out["fp.%s.%s" % (tag, k)] = np.frombuffer(cases.fingerprint(sd[k]), dtype=np.uint8)
    path = os.path.join(ROOT, "tests", "golden", "golden_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB;", len(out), "arrays")
    print("min top-2 logit gap:", out["enc.min_top2_gap"])


if __name__ == "__main__":
    main()
