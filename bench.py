#!/usr/bin/env python
"""bench.py — SR images/sec of the MARCONet hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole hot path (ResNet-45 + TextViT encoder → TSPGAN over all glyphs → TSPSRNet)
over one batch of synthetic 32x512 LR strips per GPU (BASELINE.json configs[1]: batch 64, fp16 storage with fp32
accumulation/statistics, 16 glyphs per image), inputs resident in HBM, random-init seeded checkpoints of the
reference's exact architecture.  N>1: weak scaling, every rank processes its own batch and the SR outputs are
all-gathered over RCCL (the one collective of the path).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F16_TFLOPS = 2500.0       # MI355X dense fp16 MFMA (MI355X_MICROARCH.md)
GF_F16_FIXED = 108.01 + 484.12  # ResNet + SR trunk, GFLOP / image
GF_F16_PER_GLYPH = 41.78 + 47.25
GF_FP32_VIT = 3.69


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU (configs[1]: 64)")
    ap.add_argument("--glyphs", type=int, default=16, help="glyphs per image (SURVEY.md §8d: n=16)")
    ap.add_argument("--precision", default="fp16", choices=["fp16", "fp32"])
    ap.add_argument("--cpu-images", type=int, default=1, help="images timed on the host CPU oracle (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=32, help="cap on host threads for the CPU baseline")
    ap.add_argument("--no-gather", action="store_true")
    return ap.parse_args()


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (a.gpus, a.gpus))
    import torch.distributed as dist
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    from marconet_amd import networks, ops, synthetic
    from marconet_amd.pipeline import MarconetPipeline, all_gather_outputs

    sde, sdg, sds = synthetic.make_encoder_state_dict(), synthetic.make_gan_state_dict(), synthetic.make_sr_state_dict()
    enc, gan, sr = networks.TextContextEncoderV2(), networks.TSPGAN(), networks.TSPSRNet()
    enc.load_state_dict(sde, strict=True)
    gan.load_state_dict(sdg, strict=True)
    sr.load_state_dict(sds, strict=True)
    pipe = MarconetPipeline(enc.eval().to(dev), gan.eval().to(dev), sr.eval().to(dev), precision=a.precision)

    B, n = a.batch, a.glyphs
    widths = [512] * B
    lq = synthetic.make_lq(1234 + rank, B, widths).to(dev)
    labels = [synthetic.make_labels(1234 + 1000 * rank + b, n).to(dev) for b in range(B)]
    locs = synthetic.make_locs([n] * B, widths).to(dev)

    def step():
        y = pipe.forward_batch(lq, labels, locs)
        if world > 1 and not a.no_gather:
            y = all_gather_outputs(y, B * world)
        return y

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    ops.stats.reset()
    ops.stats.enabled = ops.stats.timing = True          # HIP events around every conv launch (same stream)
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        y = step()
    fence()
    dt = time.perf_counter() - t0
    ops.stats.enabled = ops.stats.timing = False
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert torch.isfinite(y).all()

    # ---- roofline of the dominant kernel (f16 implicit-GEMM conv), from the live HIP events of the timed steps
    f16_ms = sum(s.elapsed_time(e) for s, e, _, d in ops.stats.events if d == 1)
    f16_launches = sum(1 for ev in ops.stats.events if ev[3] == 1)
    f16_flops_launched = sum(fl for _, _, fl, d in ops.stats.events if d == 1)
    f32_ms = sum(s.elapsed_time(e) for s, e, _, d in ops.stats.events if d == 0)
    alg_gf_step = B * (GF_F16_FIXED + GF_F16_PER_GLYPH * n)           # algorithmic GFLOP of the f16 convs per step
    prec16 = a.precision == "fp16"
    dom_ms = (f16_ms if prec16 else f16_ms + f32_ms) / max(a.steps, 1)
    achieved = alg_gf_step / max(dom_ms, 1e-9)                         # GFLOP/ms == TFLOP/s
    peak = PEAK_F16_TFLOPS if prec16 else 157.3
    roofline = {
        "bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
        "frac": round(achieved / peak, 4), "traffic": None,
        "kernel": "conv_igemm_kernel<f16> (all tile configs)" if prec16 else "conv_igemm_kernel<float>",
        "launches_per_step": f16_launches // max(a.steps, 1),
        "avg_launch_ms": round(dom_ms / max(f16_launches // max(a.steps, 1), 1), 4),
        "kernel_ms_per_step": round(dom_ms, 3),
        "algorithmic_gflop_per_step": round(alg_gf_step, 1),
        "launched_gflop_per_step": round(f16_flops_launched / max(a.steps, 1) / 1e9, 1),
        "fp32_vit_gemm_ms_per_step": round(f32_ms / max(a.steps, 1), 3),
    }

    out = {
        "metric": "SR images/sec (32x512 LR -> 128x2048 SR)", "value": round(B * world * a.steps / dt, 3),
        "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f16" if prec16 else "f32", "data": "synthetic",
        "config": {"workload": "BASELINE.json configs[1]: batch %d synthetic 32x512 LR strips per GPU, %d glyphs/image, "
                               "%s storage + fp32 accumulate, encoder+TSPGAN+TSPSRNet, seeded random-init checkpoints"
                               % (B, n, a.precision),
                   "per_gpu_batch": B, "global_batch": B * world, "glyphs_per_image": n,
                   "parallelism": "dp%d" % world, "collective": "all_gather(SR outputs)" if world > 1 and not a.no_gather else "none"},
        "roofline": roofline,
    }

    # ---- CPU baseline (the oracle = port of the reference's CPU forward) + parity, rank 0 at N=1 only
    if rank == 0 and world == 1 and a.cpu_images > 0:
        from oracle import marconet_oracle as O
        k = a.cpu_images
        # the threads actually used: the affinity mask, capped (oneDNN convs at batch 1 stop scaling well before
        # that, and 256 oversubscribed threads on a cgroup-limited box measured 100x slower)
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        threads = max(1, min(avail, a.cpu_threads))
        torch.set_num_threads(threads)
        lq_c, locs_c = lq[:k].cpu(), locs[:k].cpu()
        lab_c = [l.cpu() for l in labels[:k]]
        O.end_to_end(sde, sdg, sds, lq_c[:1], [lab_c[0][:2]], locs_c[:1])          # warm-up (small)
        t0 = time.perf_counter()
        refs = [O.end_to_end(sde, sdg, sds, lq_c[i:i + 1], lab_c[i:i + 1], locs_c[i:i + 1]) for i in range(k)]  # batch 1, like test_sr.py:77
        cdt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(k / cdt, 4), "unit": "images/s", "cores": threads, "kind": "port",
                               "sample": "%d images (batch 1 each, %d glyphs) of the same workload through oracle/marconet_oracle.py, "
                                         "torch %s CPU fp32, %d threads" % (k, n, torch.__version__, threads)}
        ref_sr = torch.cat([r["sr"] for r in refs])
        par = {}
        for prec in ("fp16", "fp32"):
            pipe.set_precision(prec)
            yk = pipe.forward_batch(lq[:k], labels[:k], locs[:k])
            lg = pipe.encoder(lq[:k])[0]
            par["sr_max_abs_%s" % prec] = round((yk.cpu() - ref_sr).abs().max().item(), 6)
            par["argmax_match_%s" % prec] = round(float((lg.argmax(-1).cpu() == torch.cat([r["logits"] for r in refs]).argmax(-1)).float().mean()), 4)
        pipe.set_precision(a.precision)
        out["parity"] = par

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
