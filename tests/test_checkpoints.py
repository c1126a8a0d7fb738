"""CPU: the checkpoint hook (marconet_amd/checkpoints.py) and the trained-like synthetic regime (marconet_amd/synthetic.py)."""
import json
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tiny_sd(keys=("a.weight", "b.bias")):
    return {k: torch.arange(6, dtype=torch.float32).reshape(2, 3) + i for i, k in enumerate(keys)}


def test_real_checkpoints_are_picked_up_in_the_reference_format(tmp_path, monkeypatch):
    """files named as checkpoints/download_github.py:4-6, dict key 'params' as test_sr.py:43-51 → the loader returns them; without the
    directory it returns the seeded synthetic ones; an incomplete directory is an error, not a silent fall-back"""
    from marconet_amd import checkpoints
    monkeypatch.delenv(checkpoints.ENV, raising=False)
    assert checkpoints.checkpoint_dir() is None
    sds = {r: _tiny_sd(("%s.weight" % r, "%s.bias" % r)) for r in checkpoints.CKPT_FILES}
    for r, f in checkpoints.CKPT_FILES.items():
        torch.save({"params": sds[r], "iter": 1}, tmp_path / f)
    monkeypatch.setenv(checkpoints.ENV, str(tmp_path))
    e, g, s, src = checkpoints.load_state_dicts()
    assert src == "checkpoints:%s" % tmp_path
    for got, r in ((e, "encoder"), (g, "gan"), (s, "sr")):
        assert list(got) == list(sds[r]) and all(torch.equal(got[k], sds[r][k]) for k in got)
    # path="" forces the synthetic weights even with the variable set (bench.py's second regime)
    assert checkpoints.checkpoint_dir("") is None
    os.remove(tmp_path / checkpoints.CKPT_FILES["sr"])
    with pytest.raises(FileNotFoundError):
        checkpoints.load_state_dicts()
    torch.save({"state_dict": {}}, tmp_path / checkpoints.CKPT_FILES["sr"])
    with pytest.raises(KeyError):
        checkpoints.load_state_dicts()


def test_trained_like_regime_keeps_the_reference_schema_and_is_deterministic():
    from marconet_amd import synthetic
    schema = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_schema.json")))
    made = {"TextContextEncoderV2": synthetic.make_encoder_state_dict(regime="trained"), "TSPGAN": synthetic.make_gan_state_dict(regime="trained"),
            "TSPSRNet": synthetic.make_sr_state_dict(regime="trained")}
    for cls, sd in made.items():
        want = {k: tuple(v) for k, v in schema[cls]["state_dict"].items()}
        assert {k: tuple(v.shape) for k, v in sd.items()} == want
        assert all(torch.isfinite(v).all() for v in sd.values())
    again = synthetic.make_sr_state_dict(regime="trained")
    assert torch.equal(again["conv_up.3.conv1.weight_orig"], made["TSPSRNet"]["conv_up.3.conv1.weight_orig"])
    # the default regime is untouched by the new code path (the golden fixtures depend on it)
    tame = synthetic.make_sr_state_dict()
    assert not torch.equal(tame["conv_up.3.conv1.weight_orig"], again["conv_up.3.conv1.weight_orig"])
    with pytest.raises(ValueError):
        synthetic.make_gan_state_dict(regime="wild")


def test_trained_like_regime_has_the_advertised_shape():
    """heavy tails, a modulation spread >= 10^3 per layer, sigma(SN) spread over [0.1, 10]"""
    from marconet_amd import synthetic
    g = synthetic.make_gan_state_dict(regime="trained")
    mb = g["TextGenerator.convs.3.conv.modulation.bias"]
    assert float(mb.max() / mb.min()) >= 3e2 and float(mb.min()) > 0
    w = g["TextGenerator.convs.3.conv.weight"].double().flatten()
    kurt = float(((w - w.mean()) ** 4).mean() / w.var() ** 2)
    assert kurt > 8.0, kurt                      # a Gaussian has 3
    s = synthetic.make_sr_state_dict(regime="trained")
    sig = []
    for k in s:
        if k.endswith(".weight_orig"):
            p = k[:-len(".weight_orig")]
            wm = s[k].reshape(s[k].shape[0], -1).double()
            sig.append(float(s[p + ".weight_u"].double() @ (wm @ s[p + ".weight_v"].double())))
    assert min(sig) < 0.3 and max(sig) > 3.0 and min(sig) > 0.05 and max(sig) < 20.0, (min(sig), max(sig))


def test_oracle_on_the_trained_like_regime_is_a_real_test():
    """the regime is only useful if the reference arithmetic itself stays sane on it: finite, tanh not saturated, ordinary output range"""
    from marconet_amd import synthetic
    from oracle import marconet_oracle as O
    sde, sdg, sds = (synthetic.make_encoder_state_dict(regime="trained"), synthetic.make_gan_state_dict(regime="trained"),
                     synthetic.make_sr_state_dict(regime="trained"))
    lq = synthetic.make_lq(5, 1, [200])
    labels = [synthetic.make_labels(6, 3)]
    locs = synthetic.make_locs([3], [200])
    sr = O.end_to_end(sde, sdg, sds, lq, labels, locs)["sr"]
    assert torch.isfinite(sr).all()
    assert 0.3 < float(sr.abs().max()) <= 1.0 and float((sr.abs() > 0.999).float().mean()) < 0.05 and float(sr.std()) > 0.15
