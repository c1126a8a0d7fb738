"""Where the three state_dicts of the path come from.

The reference loads ``torch.load('./checkpoints/<file>')['params']`` with ``strict=True`` (test_sr.py:43-51); the files are GitHub
release assets fetched by checkpoints/download_github.py:4-9, which this build cannot reach (no network), so everything here runs on
the seeded synthetic state_dicts of ``marconet_amd.synthetic`` — UNLESS the real files are present:

    MARCONET_CKPT_DIR=/path/to/checkpoints   (net_transformer_encoder.pth, net_prior_generation.pth, net_sr.pth)

is picked up by ``bench.py``, ``__graft_entry__.smoke()`` and the parity harness (``tests/conftest.py::harness_weights``,
``tests/test_regimes_gpu.py``) without any other change: the same oracle-vs-HIP comparison then runs on the trained weights.
"""
import os

import torch

# role -> file name (checkpoints/download_github.py:4-6; test_sr.py:43-51 pairs them with the three classes)
CKPT_FILES = {"encoder": "net_transformer_encoder.pth", "gan": "net_prior_generation.pth", "sr": "net_sr.pth"}
ENV = "MARCONET_CKPT_DIR"


def checkpoint_dir(path=None):
    """the directory holding ALL three reference checkpoints, or None (``path``: explicit directory, default $MARCONET_CKPT_DIR)"""
    d = path if path is not None else os.environ.get(ENV, "")
    if not d:
        return None
    missing = [f for f in CKPT_FILES.values() if not os.path.isfile(os.path.join(d, f))]
    if missing:         # a configured directory that is incomplete is an error, not a silent fall-back to synthetic weights
        raise FileNotFoundError("%s=%s lacks %s (expected the files of checkpoints/download_github.py:4-6)" % (ENV, d, ", ".join(missing)))
    return d


def load_reference_checkpoint(path):
    """one reference checkpoint file → OrderedDict[str, fp32 CPU tensor] (the ``['params']`` entry, test_sr.py:44)"""
    blob = torch.load(path, map_location="cpu", weights_only=True)
    if not isinstance(blob, dict) or "params" not in blob:
        raise KeyError("%s: no 'params' entry (the reference stores its state_dict under that key, test_sr.py:44)" % path)
    return {k: v.detach().float().contiguous() if v.is_floating_point() else v.detach().contiguous() for k, v in blob["params"].items()}


def load_state_dicts(path=None, regime="tame", seed=1234):
    """→ (encoder sd, TSPGAN sd, TSPSRNet sd, source string).  Real checkpoints when a checkpoint directory is configured (see the
    module docstring), otherwise the seeded synthetic ones of the given regime."""
    d = checkpoint_dir(path)
    if d is not None:
        sds = [load_reference_checkpoint(os.path.join(d, CKPT_FILES[r])) for r in ("encoder", "gan", "sr")]
        return sds[0], sds[1], sds[2], "checkpoints:%s" % d
    from . import synthetic
    return (synthetic.make_encoder_state_dict(seed, regime=regime), synthetic.make_gan_state_dict(seed, regime=regime),
            synthetic.make_sr_state_dict(seed, regime=regime), "synthetic:%s" % regime)


def build_networks(sde, sdg, sds, device=None):
    """the three HIP modules with these weights loaded ``strict=True`` (test_sr.py:43-52), in eval mode, on ``device``"""
    from . import networks
    enc, gan, sr = networks.TextContextEncoderV2(), networks.TSPGAN(), networks.TSPSRNet()
    enc.load_state_dict(sde, strict=True)
    gan.load_state_dict(sdg, strict=True)
    sr.load_state_dict(sds, strict=True)
    mods = [m.eval() for m in (enc, gan, sr)]
    return tuple(m.to(device) for m in mods) if device is not None else tuple(mods)
