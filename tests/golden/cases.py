"""Seeded parity cases shared by the golden generator, the oracle tests and the GPU parity tests.
Inputs are regenerated bit-identically everywhere from oracle/synth.py (integer PRNG), so only outputs
need to be stored in golden_v1.npz."""
import hashlib

import numpy as np
import torch

from oracle import synth


def encoder_input():
    return synth.make_lq(7, 2, [512, 300])


def gan_input():
    return synth.make_styles(11, 3), synth.make_labels(12, 3)


# name -> (lq seed, content widths, glyph counts, explicit loc centres or None)
SR_CASES = {
    # regular grid; the 300-px image has overlapping 32-px windows (spacing 20 px)
    "grid": (21, [512, 300], [5, 15], None),
    # clipped at both borders, heavy overlap (later glyph wins), a single-glyph image
    "edges": (22, [512, 200], [6, 1], [[5 / 512, 16 / 512, 40 / 512, 0.5, 500 / 512, 511 / 512], [0.31]]),
    # the configuration the headline metric is quoted on, per image: a full-width strip with the maximum of 16 glyphs
    "full16": (23, [512], [16], None),
    # narrowest strip of BASELINE configs[4] (128 px of content in the 512-px canvas), glyph windows 32 px apart
    "narrow128": (24, [128], [4], None),
}


def sr_input(name):
    seed, widths, counts, centres = SR_CASES[name]
    lq = synth.make_lq(seed, len(widths), widths)
    if centres is None:
        locs = synth.make_locs(counts, widths)
    else:
        m = max(counts)
        a = np.zeros((len(widths), 2 * m), dtype=np.float32)
        for b, cs in enumerate(centres):
            for c, v in enumerate(cs):
                a[b, 2 * c] = np.float32(v)
                a[b, 2 * c + 1] = np.float32(16 / 512)
        locs = torch.from_numpy(a)
    labels = [synth.make_labels(seed * 100 + b, n) for b, n in enumerate(counts)]
    return lq, locs, labels


def sample_logits(logits):
    return logits[:, :, ::37].contiguous()


_STRIDES = {"img": (1, 4, 4), "p64": (8, 4, 4), "p32": (8, 2, 2), "sr": (1, 4, 8)}


def sample_map(t, kind):
    sc, sh, sw = _STRIDES[kind]
    return t[:, ::sc, ::sh, ::sw].contiguous()


FINGERPRINT_KEYS = {
    "enc": ["resnet.layer5.2.conv2.weight", "transformer.linear_cls.1.weight", "transformer.linear_w_maxlen.0.bias"],
    "gan": ["TextGenerator.convs.6.conv.weight", "TextGenerator.style_mlp.3.bias"],
    "sr": ["conv_up.3.conv1.weight_orig", "conv_final.6.bias"],
}


def fingerprint(t):
    return hashlib.sha256(t.contiguous().numpy().tobytes()).digest()
