#!/usr/bin/env python
"""HBM-bound kernels of the path at the bench shapes (batch 256 x 16 glyphs, generator chunks of 1024 glyphs), timed in isolation with HIP
events, for every A/B form that sits behind an environment knob (the knobs are read once per process: one worker process per setting).

    python tools/tail_ab.py [--reps 5] [--only adain,torgb,upsample,convert,gn,scatter]

Prints ms per launch and algorithmic GB/s (every input and output tensor once, in its storage type).  Same-box A/B: the forms run back to back."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VARIANTS = [
    ("default (round 5 forms)", {}),
    ("round-4 forms: MNET_TORGB_TRIPS=1", {"MNET_TORGB_TRIPS": "1"}),
]


def worker(reps, only):
    import torch
    from marconet_amd import ops, packing
    dev = "cuda"
    MX, F16 = packing.MX_DTYPE, torch.float16

    def rnd(shape, dt):
        """random tensor in a storage dtype, generated on the device in slabs (fp16+8 through mnet_convert)"""
        import math
        out = packing.new_tensor(shape, dt, dev)
        raw = packing.untag(out).view(torch.float16) if packing.is_split(dt) else out       # plain halves for the slab copies
        n0 = max(1, (1 << 28) // max(1, math.prod(shape[1:])))
        for s in range(0, shape[0], n0):
            e = min(shape[0], s + n0)
            src = torch.randn((e - s,) + tuple(shape[1:]), device=dev, dtype=torch.float16)
            if dt == F16:
                raw[s:e].copy_(src)
            else:
                raw[s:e].copy_(packing.untag(ops.convert(src, dt)).view(torch.float16))
        return out

    def nbytes(*ts):
        return float(sum(t.numel() * t.element_size() for t in ts if t is not None))

    def timeit(fn):
        fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / reps

    res = {}

    def run(name, fn, *tensors):
        ms = timeit(fn)
        res[name] = (round(ms, 4), round(nbytes(*tensors) / ms / 1e6, 1))

    B, n = 256, 16
    G = B * n
    if "adain" in only:
        for S, tag in ((64, "64"), (32, "32")):
            prior, feat = rnd((G, S, S, 256), MX), rnd((B, S, 16 * S, 256), MX)
            g_img = torch.arange(G, device=dev, dtype=torch.int32) // n
            g_x1 = (torch.arange(G, device=dev, dtype=torch.int32) % n) * S
            g_y1 = torch.zeros(G, device=dev, dtype=torch.int32)
            g_w = torch.full((G,), S, device=dev, dtype=torch.int32)
            gamma, beta = torch.rand(512, device=dev) + 0.5, torch.randn(512, device=dev) * 0.1
            out = []
            def f():
                out[:] = [ops.adain_crop_concat_gn(prior, feat, g_img, g_x1, g_y1, g_w, gamma, beta, 1e-6, split=False)[0]]
            f()
            run("adain_crop_concat_gn S=%s [4096 glyphs, fp16+8]" % tag, f, prior, prior, out[0])
            del prior, feat, out
            torch.cuda.empty_cache()
    if "torgb" in only:
        for (hw, c, dt, tag) in ((128, 128, F16, "f16"), (64, 256, MX, "fp16+8"), (32, 512, MX, "fp16+8")):
            x = rnd((1024, hw, hw, c), dt)
            wt, st = torch.randn(3, c, device=dev) / 16, torch.rand(1024, c, device=dev) + 0.5
            bias = torch.zeros(4, device=dev)
            skip = torch.tanh(torch.randn(1024, hw // 2, hw // 2, 4, device=dev))
            o = ops.torgb(x, wt, st, None, bias, skip)
            run("torgb %dx%d c=%d %s [1024 glyphs]" % (hw, hw, c, tag), lambda: ops.torgb(x, wt, st, None, bias, skip), x, skip, o)
            del x, o
            torch.cuda.empty_cache()
    if "upsample" in only:
        x = rnd((1024, 64, 64, 256), MX)
        sc = torch.rand(1024, 256, device=dev) + 0.5
        o = ops.upsample2x(x, scale=sc)
        run("upsample2x 64->128 c=256 fp16+8 -> fp16+8 [1024 glyphs]", lambda: ops.upsample2x(x, scale=sc), x, o)
        del o
        o = ops.upsample2x(x, scale=sc, out_dtype=F16)
        run("upsample2x 64->128 c=256 fp16+8 -> f16 (convert fused) [1024 glyphs]", lambda: ops.upsample2x(x, scale=sc, out_dtype=F16), x, o)
        del o
        xf = ops.convert(x, F16)
        run("convert fp16+8 -> f16 of that map", lambda: ops.convert(x, F16), x, xf)
        o = ops.upsample2x(xf, scale=sc)
        run("upsample2x 64->128 c=256 f16 -> f16 [1024 glyphs]", lambda: ops.upsample2x(xf, scale=sc), xf, o)
        del x, xf, o
        torch.cuda.empty_cache()
        x = rnd((64, 64, 1024, 128), MX)
        o = ops.upsample2x(x)
        run("upsample2x 64x1024 -> 128x2048 c=128 fp16+8 [64 strips]", lambda: ops.upsample2x(x), x, o)
        del x, o
        torch.cuda.empty_cache()
    if "gn" in only:
        x = rnd((64, 64, 1024, 256), MX)
        gamma, beta = torch.rand(256, device=dev) + 0.5, torch.randn(256, device=dev) * 0.1
        sc, sh = ops.groupnorm_affine(x, gamma, beta)
        run("groupnorm statistics 64x1024 c=256 fp16+8 [64 strips]", lambda: ops.groupnorm_affine(x, gamma, beta), x)
        y = ops.affine_act(x, sc, sh, swish=True)
        run("groupnorm apply + swish, same map", lambda: ops.affine_act(x, sc, sh, swish=True, out=y), x, y)
        del x, y
        torch.cuda.empty_cache()
    print("TAIL_AB " + json.dumps(res))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", default="adain,torgb,upsample,gn")
    ap.add_argument("--worker", action="store_true")
    a = ap.parse_args()
    if a.worker:
        return worker(a.reps, a.only.split(","))
    rows = {}
    for name, env in VARIANTS:
        e = dict(os.environ)
        e.update(env)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", "--reps", str(a.reps), "--only", a.only], env=e, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("TAIL_AB ")]
        if not line:
            print("[tail_ab] %s failed:\n%s" % (name, r.stderr[-1500:]))
            continue
        rows[name] = json.loads(line[-1][len("TAIL_AB "):])
    names = list(rows)
    keys = list(rows[names[0]]) if names else []
    print("%-72s %s" % ("kernel (ms per launch, algorithmic GB/s)", " | ".join("%-40s" % n_[:40] for n_ in names)))
    for k in keys:
        print("%-72s %s" % (k, " | ".join("%9.3f ms %8.1f GB/s %12s" % (tuple(rows[n_][k]) + ("",)) if k in rows[n_] else "-" for n_ in names)))


if __name__ == "__main__":
    main()
