#!/usr/bin/env python
"""Writes the three checkpoint files of the reference's release (checkpoints/download_github.py:4-6) from a seeded synthetic regime, in the reference's
format ({'params': state_dict}, test_sr.py:43-51) — a stand-in to exercise the MARCONET_CKPT_DIR hook end to end where the real files cannot be fetched.
    python tools/make_fake_checkpoints.py <dir> [regime=trained]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from marconet_amd import checkpoints, synthetic
    out = sys.argv[1]
    regime = sys.argv[2] if len(sys.argv) > 2 else "trained"
    os.makedirs(out, exist_ok=True)
    sds = {"encoder": synthetic.make_encoder_state_dict(regime=regime), "gan": synthetic.make_gan_state_dict(regime=regime),
           "sr": synthetic.make_sr_state_dict(regime=regime)}
    for role, name in checkpoints.CKPT_FILES.items():
        torch.save({"params": sds[role]}, os.path.join(out, name))
        print("wrote %s (%d tensors)" % (os.path.join(out, name), len(sds[role])))


if __name__ == "__main__":
    main()
