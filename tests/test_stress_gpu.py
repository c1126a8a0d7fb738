"""-m gpu: the half-range throughput modes (fp16x3, fp16x2, fp16) outside the friendly synthetic regime.

Real checkpoints cannot be fetched (checkpoints/download_github.py), so the regimes they may reach are built from seeded
variants of the synthetic checkpoints (marconet_amd/synthetic.py) and compared with the CPU oracle on the SAME weights:
  * the reference's own ResNet initialisation (models/resnet.py:45-48): features of std ~240, |max| ~3000;
  * activations pushed towards the fp16 limit (|max| ~2.4e4: must still meet the bar) and past it (|max| ~1.9e5: must raise
    FloatingPointError, not return garbage);
  * activations pushed down to ~1e-3 and below (the lo halves become fp16 subnormals);
  * near-tie logits (top-2 gaps of 1e-5 .. 1e-4 found by scanning seeds with linear_cls at gain 1.0)."""
import functools

import pytest
import torch

from oracle import marconet_oracle as O
from oracle import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-3
MODES = ["fp32", "fp16x3", "fp16x2"]


@functools.lru_cache(maxsize=None)
def _encoder_sd(**kw):
    """one state_dict object per variant, shared by the precision cases (4 s to build; its identity keys conftest's oracle memo)"""
    from marconet_amd import synthetic
    return synthetic.make_encoder_state_dict(**kw)


def _encoder(sd, prec):
    from marconet_amd import networks
    enc = networks.TextContextEncoderV2()
    enc.load_state_dict(sd, strict=True)
    return enc.eval().to(DEV).set_precision(prec)


def _cmp(name, got, ref):
    e = (got.detach().float().cpu() - ref).abs().max().item()
    print("%-60s max|d| = %.3e" % (name, e))
    return e


@pytest.mark.parametrize("prec", MODES)
@pytest.mark.parametrize("input_gain", [1.0, 8.0, 2.0 ** -20])
def test_encoder_at_reference_init_and_scaled_ranges(prec, input_gain):
    """ResNet at the reference's own init gain (|feature| up to ~3e3), x8 (~2.4e4, a factor 2.7 under the fp16 limit) and x2^-20
    (~3e-3: most values below the fp16 normal range, every lo half subnormal): logits / w / locs <= 1e-3, indices exact"""
    sd = _encoder_sd(resnet_gain=1.0, input_gain=input_gain)
    lq = synth.make_lq(7, 2, [512, 400])
    with torch.no_grad():
        r_logits, r_locs, r_w = O.encoder_forward(sd, lq)
    logits, locs, w = _encoder(sd, prec)(lq.to(DEV))
    tag = "enc refinit x%g %s" % (input_gain, prec)
    assert torch.isfinite(logits).all() and torch.isfinite(w).all()
    assert _cmp(tag + " logits", logits, r_logits) <= TOL and _cmp(tag + " w", w, r_w) <= TOL and _cmp(tag + " locs", locs, r_locs) <= TOL
    assert torch.equal(logits.argmax(-1).cpu(), r_logits.argmax(-1))


@pytest.mark.parametrize("prec", ["fp16", "fp16x3", "fp16x2"])
def test_overflow_raises_instead_of_returning_garbage(prec, ckpts):
    """activations past 65504 in a half-range mode: the pipeline raises FloatingPointError (the fp32 mode handles the same weights)"""
    from marconet_amd import networks
    from marconet_amd.pipeline import MarconetPipeline
    sde = _encoder_sd(resnet_gain=1.0, input_gain=64.0)          # ResNet features up to ~1.9e5
    sdg, sds = ckpts[1], ckpts[2]
    enc, gan, sr = networks.TextContextEncoderV2(), networks.TSPGAN(), networks.TSPSRNet()
    enc.load_state_dict(sde); gan.load_state_dict(sdg); sr.load_state_dict(sds)
    pipe = MarconetPipeline(enc.eval().to(DEV), gan.eval().to(DEV), sr.eval().to(DEV), precision=prec)
    lq = synth.make_lq(7, 1, [512])
    labels, locs = [synth.make_labels(8, 3)], synth.make_locs([3], [512])
    with pytest.raises(FloatingPointError):
        pipe.forward_batch(lq.to(DEV), labels, locs)
    pipe.set_precision("fp32")
    y = pipe.forward_batch(lq.to(DEV), labels, locs)
    ref = O.end_to_end(sde, sdg, sds, lq, labels, locs)["sr"]
    assert torch.isfinite(y).all() and _cmp("overflow weights, fp32 mode", y, ref) <= TOL


NEAR_TIES = [(3021, 6), (3021, 3), (3022, 6), (3022, 2), (3027, 3), (3030, 1)]       # (make_lq seed, image): min top-2 gap 1.3e-5 .. 4.4e-5


@pytest.mark.parametrize("prec", MODES)
def test_near_tie_logits(prec):
    """linear_cls at gain 1.0 (narrow top-2 gaps).  Claim checked: the logits deviate by <= 3e-5 in every parity mode (fp16x2 runs the
    encoder's ResNet in fp16x3), hence every position whose top-2 gap exceeds 1e-4 gets the reference's index; below that the
    chosen index is one of the reference's top two (fp32 summation order of either implementation decides)"""
    sd = _encoder_sd(cls_gain=1.0)
    lq = torch.cat([synth.make_lq(seed, 8, [512] * 8)[b:b + 1] for seed, b in NEAR_TIES])
    with torch.no_grad():
        r_logits = O.encoder_forward(sd, lq)[0]
    top = r_logits.topk(2, -1)
    gap = top.values[..., 0] - top.values[..., 1]
    assert float(gap.min()) <= 2e-5 and int((gap <= 1e-4).sum()) >= 6          # the fixture really contains near ties
    logits = _encoder(sd, prec)(lq.to(DEV))[0].cpu()
    err = (logits - r_logits).abs().max().item()
    idx = logits.argmax(-1)
    safe = gap > 1e-4
    flips = int((idx != top.indices[..., 0]).sum())
    print("near ties %s: logits max|d| %.3e, min gap %.2e, positions with gap <= 1e-4: %d, index differences: %d" %
          (prec, err, float(gap.min()), int((~safe).sum()), flips))
    assert err <= 3e-5
    assert torch.equal(idx[safe], top.indices[..., 0][safe])
    assert ((idx == top.indices[..., 0]) | (idx == top.indices[..., 1])).all()


# ---- outlier channels inside a 32-channel storage block -------------------------------------------------------
# The fp16+8 format shares ONE E8M0 scale per (pixel, 32-channel block): an outlier channel takes the block's scale and the other 31
# channels' lo bytes lose resolution (e4m3 keeps 4 significant bits down to 2^-6 of the block scale's range, i.e. over a factor of
# ~3e4 between the largest and the smallest channel).  Trained StyleGAN-type weights produce such channels; the synthetic
# checkpoints do not, so they are planted here: a GroupNorm gain (its output feeds a conv directly), a row of a spectral-normalised
# conv (its output channel dominates the next conv's input block) and a modulation bias of the generator (one modulated input channel).
OUTLIERS = {
    # name: (generator outliers, SR-net outliers, bar for fp16x2 / None = report only)
    "x100": ({"TextGenerator.convs.5.conv.modulation.bias": (9, 100.0)},
             {"conv_up.3.norm1.weight": (37, 100.0), "conv_64_fuse.0.norm2.weight": (200, 100.0),
              "conv_body_32.0.weight_orig": (17, 100.0), "conv_64_scale.0.weight_orig": (5, 100.0)}, TOL),
    "x1000_trunk_gn": ({}, {"conv_up.3.norm1.weight": (37, 1000.0)}, TOL),
    # CPU emulation of the arithmetic (tools/precision_plan.py machinery): fp16x2 1.4e-3, fp16x3 2.7e-4 — the network itself is 10x
    # worse conditioned with this gain (fp16x3 is normally at 2.7e-5); fp16x2 is reported, fp16x3 must hold the bar
    "x1000_fuse64_gn": ({}, {"conv_64_fuse.0.norm2.weight": (200, 1000.0)}, None),
}


@functools.lru_cache(maxsize=None)
def _outlier_case(name):
    from marconet_amd import synthetic
    g, s, _ = OUTLIERS[name]
    sdg = synthetic.make_gan_state_dict(outliers=g)
    sds = synthetic.make_sr_state_dict(outliers=s)
    return sdg, sds


@pytest.mark.parametrize("name", list(OUTLIERS))
def test_outlier_channel_inside_a_storage_block(name, ckpts):
    """one channel 1e2 / 1e3 times larger than its 31 block neighbours, planted in mid-network activations: SR deviation of the
    fp16x2 and fp16x3 modes against the CPU oracle on the same weights"""
    from marconet_amd import networks
    from marconet_amd.pipeline import MarconetPipeline
    sdg, sds = _outlier_case(name)
    bar = OUTLIERS[name][2]
    enc, gan, sr = networks.TextContextEncoderV2(), networks.TSPGAN(), networks.TSPSRNet()
    enc.load_state_dict(ckpts[0]); gan.load_state_dict(sdg); sr.load_state_dict(sds)
    n = 8
    lq = synth.make_lq(77, 1, [512])
    labels, locs = [synth.make_labels(78, n)], synth.make_locs([n], [512])
    ref = O.end_to_end(ckpts[0], sdg, sds, lq, labels, locs)["sr"]
    pipe = MarconetPipeline(enc.eval().to(DEV), gan.eval().to(DEV), sr.eval().to(DEV), precision="fp32")
    errs = {}
    for prec in ("fp32", "fp16x3", "fp16x2"):
        pipe.set_precision(prec)
        y = pipe.forward_batch(lq.to(DEV), labels, locs)
        assert torch.isfinite(y).all()
        errs[prec] = _cmp("outlier %s %s" % (name, prec), y, ref)
    assert errs["fp32"] <= TOL and errs["fp16x3"] <= TOL
    if bar is not None:
        assert errs["fp16x2"] <= bar
    else:
        # report only: the deviation must still be the format's resolution on a badly conditioned net (a small multiple of fp16x3's),
        # not a breakdown
        assert errs["fp16x2"] <= 1e-2 and errs["fp16x2"] <= 12 * max(errs["fp16x3"], 1e-5)
