#!/usr/bin/env python
"""Per-launch table of the implicit-GEMM conv kernel for one bench step (HIP events on the launch stream):
geometry, ms, TFLOP/s.  Usage: python tools/profile_layers.py [--batch 64] [--glyphs 16] [--out gpurun_out/layers.txt]"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--glyphs", type=int, default=16)
    ap.add_argument("--precision", default="fp16")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "layers.txt"))
    a = ap.parse_args()
    from marconet_amd import _lib, networks, ops, synthetic
    from marconet_amd.pipeline import MarconetPipeline
    dev = torch.device("cuda:0")
    sde, sdg, sds = synthetic.make_encoder_state_dict(), synthetic.make_gan_state_dict(), synthetic.make_sr_state_dict()
    enc, gan, sr = networks.TextContextEncoderV2(), networks.TSPGAN(), networks.TSPSRNet()
    enc.load_state_dict(sde); gan.load_state_dict(sdg); sr.load_state_dict(sds)
    pipe = MarconetPipeline(enc.eval().to(dev), gan.eval().to(dev), sr.eval().to(dev), precision=a.precision)
    B, n = a.batch, a.glyphs
    lq = synthetic.make_lq(1234, B, [512] * B).to(dev)
    labels = [synthetic.make_labels(1234 + b, n).to(dev) for b in range(B)]
    locs = synthetic.make_locs([n] * B, [512] * B).to(dev)
    pipe.forward_batch(lq, labels, locs)
    torch.cuda.synchronize()

    # wrap the C call to capture descriptors
    lib = _lib.load()
    recs = []
    orig = lib.mnet_conv2d_nhwc_ex

    class Wrap:
        def __call__(self, dref, algo, stream):
            d = dref._obj
            recs.append(dict(dt=d.dtype, n=d.n, h=d.h, w=d.w, c0=d.c0, c1=d.c1, cout=d.cout, k=d.kh, s=(d.stride_h, d.stride_w),
                             ho=d.ho, wo=d.wo, pro=bool(d.in_scale), sw=d.in_swish, vw=bool(d.valid_w), osc=bool(d.out_scale),
                             res=bool(d.residual), act=d.act))
            return orig(dref, algo, stream)
    lib.mnet_conv2d_nhwc_ex = Wrap()
    ops.stats.reset(); ops.stats.enabled = ops.stats.timing = True
    s0 = torch.cuda.Event(enable_timing=True); s1 = torch.cuda.Event(enable_timing=True)
    s0.record()
    pipe.forward_batch(lq, labels, locs)
    s1.record()
    torch.cuda.synchronize()
    lib.mnet_conv2d_nhwc_ex = orig
    total = s0.elapsed_time(s1)
    lines = []
    tot_conv = 0.0
    for r, (s, e, fl, d, kid) in zip(recs, ops.stats.events):
        ms = s.elapsed_time(e)
        tot_conv += ms
        lines.append("%-4s n=%-5d %4dx%-4d c=%3d+%-3d -> %4d k%d s%s out %4dx%-4d %s%s%s%s%s act%d k%-2d %8.3f ms %8.1f TF/s %5.1f%%"
                     % ({0: "f32", 1: "f16", 2: "x3 ", 3: "x2 "}[r["dt"]], r["n"], r["h"], r["w"], r["c0"], r["c1"], r["cout"], r["k"], r["s"], r["ho"], r["wo"],
                        "P" if r["pro"] else "-", "S" if r["sw"] else "-", "V" if r["vw"] else "-", "D" if r["osc"] else "-",
                        "R" if r["res"] else "-", r["act"], kid, ms, fl / ms / 1e9, 100 * ms / total))
    # per (kernel id, dtype): launches, time, algorithmic FLOPs and algorithmic bytes (input(s) + output + residual + weights, each once)
    per = {}
    for r, (s, e, fl, d, kid) in zip(recs, ops.stats.events):
        esz = {0: 4, 1: 2, 2: 4, 3: 4}[r["dt"]]
        by = esz * (r["n"] * r["h"] * r["w"] * (r["c0"] + r["c1"]) + r["n"] * r["ho"] * r["wo"] * r["cout"] * (2 if r["res"] else 1)
                    + r["cout"] * r["k"] * r["k"] * (r["c0"] + r["c1"]))
        a_ = per.setdefault((kid, r["dt"]), [0, 0.0, 0.0, 0.0])
        a_[0] += 1; a_[1] += s.elapsed_time(e); a_[2] += fl; a_[3] += by
    lines.append("per kernel id (mnet_conv2d_plan) and dtype: launches, ms, algorithmic TFLOP per launch, algorithmic GB per launch, TFLOP/s")
    for (kid, dt), (cnt, ms, fl, by) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        lines.append("  id %-3d %s  %3d launches  %9.3f ms  %7.3f TFLOP/launch  %7.3f GB/launch  %7.1f TFLOP/s"
                     % (kid, {0: "f32", 1: "f16", 2: "x3 ", 3: "x2 "}[dt], cnt, ms, fl / cnt / 1e12, by / cnt / 1e9, fl / ms / 1e9))
    lines.append("step total %.2f ms, conv launches %.2f ms (%d), other %.2f ms" % (total, tot_conv, len(recs), total - tot_conv))
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    open(a.out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[-(len(per) + 2):]))


if __name__ == "__main__":
    main()
