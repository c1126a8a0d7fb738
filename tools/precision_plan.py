#!/usr/bin/env python
"""Per-layer precision plan by CPU emulation (test infrastructure like tools/precision_emul.py; needs no GPU).

The fp16x2 mode spends 2 MFMA units per product on every TSPGAN / TSPSRNet conv.  This tool asks, layer by layer, what the SR output
loses when ONE layer (or a named set) is demoted to plain fp16 operands (1 unit, f16 storage of its input) while everything else keeps
the product's arithmetic (ResNet: three f16 products; TSPGAN / TSPSRNet: hi.hi + block-scaled e4m3 corrections; TextViT fp32), and then
builds the largest set that keeps the SR deviation under a budget.

    python tools/precision_plan.py sweep  [n_glyphs=16] [seed=1234]      one layer at a time → table
    python tools/precision_plan.py plan   name1,name2=f16s,...  [n] [seed]    a given set → SR max-abs (scheme per name: f16 = plain-f16 operands [default],
                                                                          f16s = plain-f16 operands AND output storage, x3, mx8, fp32)
Layer names are the reference's module paths (models/networks.py): conv_final.3, conv_64_fuse.0.conv1, TextGenerator.convs.7, ...; a name
matches every conv whose path STARTS with it."""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import precision_emul as E  # noqa: E402
from oracle import marconet_oracle as O  # noqa: E402

DEFAULT_SCHEME = os.environ.get("MNET_EMUL_DEFAULT", "mx8")      # arithmetic of the TSPGAN / TSPSRNet convs that the plan leaves alone
_CUR = {"name": None, "default": None, "plan": {}, "seen": []}
_CONVS = {}


def _scheme_for(name):
    for k, v in _CUR["plan"].items():
        if name is not None and name.startswith(k):
            return v
    return _CUR["default"]


def _dispatch(x, w, bias=None, stride=1, padding=0, dilation=1, groups=1):
    name = _CUR["name"]
    if name is None and w.shape[-1] == 1 and w.shape[1] == 512 and w.shape[0] == 256:
        name = _CUR.get("res_key", "?") + ".conv_out"           # ResTextBlockV2's 1x1 skip conv is called through F.conv2d directly
    s = _scheme_for(name)
    if name not in _CUR["seen"]:
        _CUR["seen"].append(name)
    if s == "fp32":
        return E._conv2d(x, w, bias, stride, padding, dilation, groups)
    if s not in _CONVS:
        _CONVS[s] = E.make_conv(s)
    return _CONVS[s](x, w, bias, stride=stride, padding=padding, dilation=dilation, groups=groups)


def _named(fn, name_of):
    def wrapper(*a, **kw):
        prev = _CUR["name"]
        _CUR["name"] = name_of(*a, **kw)
        try:
            return fn(*a, **kw)
        finally:
            _CUR["name"] = prev
    return wrapper


class Emulated:
    """context: the oracle's convs run through the per-layer dispatcher"""

    def __init__(self, default, plan=None):
        self.default, self.plan = default, dict(plan or {})

    def __enter__(self):
        self.saved = (F.conv2d, O._snconv, O._styled_conv, O._to_rgb, O.res_text_block)
        _CUR.update(default=self.default, plan=self.plan, name=None)
        F.conv2d = _dispatch
        O._snconv = _named(self.saved[1], lambda sd, key, x, stride=1: key)
        O._styled_conv = _named(self.saved[2], lambda sd, p, x, latent, upsample: p)
        O._to_rgb = _named(self.saved[3], lambda sd, p, x, latent, skip: p)
        rt = self.saved[4]

        def res_block(sd, key, x):
            _CUR["res_key"] = key
            return rt(sd, key, x)
        O.res_text_block = res_block
        return self

    def __exit__(self, *exc):
        F.conv2d, O._snconv, O._styled_conv, O._to_rgb, O.res_text_block = self.saved
        _CUR["name"] = None


GAN_LAYERS = ["TextGenerator.conv1"] + ["TextGenerator.convs.%d" % i for i in range(8)]          # convs.8 / .9 (128 px) feed only `image`
SR_LAYERS = ["conv_first_32.0", "conv_first_16.0", "conv_first_8.0", "conv_first_8.2", "conv_body_16.0", "conv_body_16.2",
             "conv_body_32.0", "conv_body_32.2", "conv_32_to256.0", "conv_32_to256.2",
             "conv_32_fuse.0.conv1", "conv_32_fuse.0.conv2", "conv_32_fuse.0.conv_out", "conv_32_scale.0", "conv_32_scale.2",
             "conv_32_shift.0", "conv_32_shift.2", "conv_up.1", "conv_up.3.conv1", "conv_up.3.conv2", "conv_up.4",
             "conv_64_fuse.0.conv1", "conv_64_fuse.0.conv2", "conv_64_fuse.0.conv_out", "conv_64_scale.0", "conv_64_scale.2",
             "conv_64_shift.0", "conv_64_shift.2", "conv_final.0", "conv_final.3", "conv_final.5.conv1", "conv_final.5.conv2", "conv_final.6"]


def run(sde, sdg, sds, lq, labels, locs, plan, cache):
    """the product's split of arithmetics with ``plan`` (name prefix → scheme) on top → SR tensor.  ``cache``: encoder / generator
    results of the un-demoted chain, reused when the plan touches neither"""
    n = labels[0].shape[0]
    with torch.no_grad():
        if "enc" not in cache:
            with Emulated("x3"):
                cache["enc"] = O.encoder_forward(sde, lq)
        _, _, w = cache["enc"]
        gan_touched = any(k.startswith("TextGenerator") for k in plan)
        if gan_touched or "gan" not in cache:
            with Emulated(DEFAULT_SCHEME, plan):
                g = O.tspgan_forward(sdg, w[:1].repeat(n, 1), labels[0])
            if not gan_touched:
                cache["gan"] = g
        else:
            g = cache["gan"]
        with Emulated(DEFAULT_SCHEME, plan):
            return O.tspsr_forward(sds, lq, [g[1]], [g[2]], locs)


def main():
    from marconet_amd import synthetic
    mode = sys.argv[1] if len(sys.argv) > 1 else "sweep"
    args = sys.argv[2:]
    names = []
    if mode == "plan":
        names, args = [v for v in args[0].split(",") if v], args[1:]
    n = int(args[0]) if len(args) > 0 else 16
    seed = int(args[1]) if len(args) > 1 else 1234
    sde, sdg, sds = synthetic.make_encoder_state_dict(), synthetic.make_gan_state_dict(), synthetic.make_sr_state_dict()
    lq = synthetic.make_lq(seed, 1, [512])
    labels = [synthetic.make_labels(seed, n)]
    locs = synthetic.make_locs([n], [512])
    t = time.time()
    ref = O.end_to_end(sde, sdg, sds, lq, labels, locs)["sr"]
    cache = {}
    base = run(sde, sdg, sds, lq, labels, locs, {}, cache)
    e0 = (base - ref).abs().max().item()
    print("strip seed %d, %d glyphs; reference + baseline chain: %.0f s" % (seed, n, time.time() - t), flush=True)
    print("baseline (ResNet x3, TSPGAN / TSPSRNet fp16+8): SR max-abs %.3e" % e0, flush=True)
    if mode == "plan":
        y = run(sde, sdg, sds, lq, labels, locs, {k.split("=")[0]: (k.split("=")[1] if "=" in k else "f16") for k in names}, cache)
        print("plan f16 on {%s}: SR max-abs %.3e (mean %.3e)" % (", ".join(names), (y - ref).abs().max().item(), (y - ref).abs().mean().item()))
        return
    rows = []
    print("%-28s %-12s %-12s %s" % ("layer demoted to plain fp16", "SR max-abs", "added (rss)", "seconds"), flush=True)
    for name in SR_LAYERS + GAN_LAYERS:
        t = time.time()
        y = run(sde, sdg, sds, lq, labels, locs, {name: "f16"}, cache)
        e = (y - ref).abs().max().item()
        add = max(e * e - e0 * e0, 0.0) ** 0.5
        rows.append((name, e, add))
        print("%-28s %.3e    %.3e    %.0f" % (name, e, add, time.time() - t), flush=True)
    print("\nby added deviation:")
    for name, e, add in sorted(rows, key=lambda r: r[2]):
        print("  %-28s %.3e  (+%.3e)" % (name, e, add))


if __name__ == "__main__":
    main()
