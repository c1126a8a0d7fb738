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
    ap.add_argument("--cpu-images", type=int, default=4, help="images timed on the host CPU oracle (0 = skip); ~3.5 s each on 32 threads")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary (no structure-image) throughput measurement")
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

    if world > 1:       # N ranks build the same seeded checkpoints on the host at the same time: do not oversubscribe its cores
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        torch.set_num_threads(max(1, min(32, avail // world)))

    from marconet_amd import networks, ops, synthetic
    from marconet_amd.pipeline import MarconetPipeline, OverlappedGather

    sde, sdg, sds = synthetic.make_encoder_state_dict(), synthetic.make_gan_state_dict(), synthetic.make_sr_state_dict()
    enc, gan, sr = networks.TextContextEncoderV2(), networks.TSPGAN(), networks.TSPSRNet()
    enc.load_state_dict(sde, strict=True)
    gan.load_state_dict(sdg, strict=True)
    sr.load_state_dict(sds, strict=True)
    pipe = MarconetPipeline(enc.eval().to(dev), gan.eval().to(dev), sr.eval().to(dev), precision=a.precision)

    B, n = a.batch, a.glyphs
    widths = [512] * B
    lq = synthetic.make_lq(1234 + rank, B, widths).to(dev)
    # labels and glyph locations stay on the HOST, where the OCR / detector front-end leaves them (test_sr.py:121-149):
    # the forward then needs no device→host synchronisation at all
    labels = [synthetic.make_labels(1234 + 1000 * rank + b, n) for b in range(B)]
    locs = synthetic.make_locs([n] * B, widths)

    gather = OverlappedGather() if world > 1 and not a.no_gather else None

    def step():
        # N > 1: the all-gather of this step's SR outputs (the one collective of the path) is enqueued asynchronously and
        # overlaps the next step's compute; the fence below waits for the last one, so every gather is inside the timed region
        y = pipe.forward_batch(lq, labels, locs)
        if gather is not None:
            gather.submit(y)
        return y

    def fence():
        if gather is not None:
            gather.flush()
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

    # ---- secondary figure (reported separately, never the headline): the same step without the generator's 128-px
    # structure image, which only feeds test_sr.py's saved visualisation (SURVEY.md §7 "hard parts", last item)
    secondary = None
    prec16 = a.precision == "fp16"
    if not a.no_secondary:
        pipe.need_prior_image = False
        step()
        fence()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        fence()
        dt2 = time.perf_counter() - t0
        pipe.need_prior_image = True
        if world > 1:
            t = torch.tensor([dt2], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt2 = float(t.item())
        secondary = {"images_per_s_without_prior_image": round(B * world * a.steps / dt2, 3), "ms_per_step": round(dt2 / a.steps * 1e3, 3),
                     "note": "opt-in MarconetPipeline(need_prior_image=False): TSPGAN stops at the 64-px level; SR output identical"}
        if world == 1 and prec16:
            # the fp32 parity mode (<= 1e-3 vs the reference, bit-exact indices: see "parity" below) on a 16-image slice
            kb = min(B, 16)
            pipe.set_precision("fp32")
            pipe.forward_batch(lq[:kb], labels[:kb], locs[:kb])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pipe.forward_batch(lq[:kb], labels[:kb], locs[:kb])
            torch.cuda.synchronize()
            secondary["fp32_parity_mode_images_per_s"] = round(kb / (time.perf_counter() - t0), 3)
            secondary["fp32_parity_mode_batch"] = kb
            pipe.set_precision(a.precision)

    # ---- roofline of the dominant kernel, from the live HIP events of the timed steps (events are recorded on the
    # launch stream around every conv launch; the kernel each launch resolved to comes from mnet_conv2d_plan)
    KNAME = {1: "conv_igemm_kernel (register-staged)", 3: "conv_skinny_f32_kernel", 16: "conv_dma_kernel<256,256,4,4,2,16>", 17: "conv_dma_kernel<256,128,4,2,3,16>",
             18: "conv_dma_kernel<128,256,2,4,3,16>", 19: "conv_dma_kernel<64,256,1,8,3,16>", 20: "conv_dma_kernel<128,512,2,8,2,16>",
             21: "conv_dma_kernel<64,512,1,8,2,16>", 22: "conv_dma_kernel<256,256,2,4,2,16>", 26: "conv_dma_kernel<128,128,2,4,4,16>",
             32: "conv_strip_kernel<256,256,4,4>", 33: "conv_strip_kernel<64,512,1,8>", 34: "conv_strip_kernel<128,256,2,4>"}
    per = {}
    for s_, e_, fl, dt_, kid in ops.stats.events:
        key = (kid, dt_)
        ms = s_.elapsed_time(e_)
        r = per.setdefault(key, [0.0, 0.0, 0])
        r[0] += ms; r[1] += fl; r[2] += 1
    prec16 = a.precision == "fp16"
    cand = {k: v for k, v in per.items() if k[1] == (1 if prec16 else 0)}
    dom = max(cand, key=lambda k: cand[k][0])
    dom_ms, dom_fl, dom_n = cand[dom]
    f16_ms = sum(v[0] for k, v in per.items() if k[1] == 1)
    f32_ms = sum(v[0] for k, v in per.items() if k[1] == 0)
    f16_n = sum(v[2] for k, v in per.items() if k[1] == 1)
    f16_fl = sum(v[1] for k, v in per.items() if k[1] == 1)
    alg_gf_step = B * (GF_F16_FIXED + GF_F16_PER_GLYPH * n)           # algorithmic GFLOP of the f16 convs per step
    peak = PEAK_F16_TFLOPS if prec16 else 157.3
    achieved = dom_fl / max(dom_ms, 1e-9) / 1e9                        # FLOP/ms/1e9 == TFLOP/s
    conv_ms = (f16_ms if prec16 else f16_ms + f32_ms) / max(a.steps, 1)
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")        # measured separately (rocprofv3 --pmc), see DESIGN.md
    if os.path.isfile(tpath):
        try:
            traffic = json.load(open(tpath)).get(KNAME.get(dom[0], ""), {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {
        "bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
        "frac": round(achieved / peak, 4), "traffic": traffic,
        "kernel": KNAME.get(dom[0], str(dom[0])) + (" f16" if dom[1] == 1 else " f32"),
        "launches_per_step": dom_n // max(a.steps, 1),
        "avg_launch_ms": round(dom_ms / max(dom_n, 1), 4),
        "flops_per_launch_avg": round(dom_fl / max(dom_n, 1), 1),
        "kernel_ms_per_step": round(dom_ms / max(a.steps, 1), 3),
        "all_conv_kernels": {
            "achieved": round(alg_gf_step / max(conv_ms, 1e-9), 2), "frac": round(alg_gf_step / max(conv_ms, 1e-9) / peak, 4),
            "launches_per_step": f16_n // max(a.steps, 1), "ms_per_step": round(conv_ms, 3),
            "algorithmic_gflop_per_step": round(alg_gf_step, 1),
            "launched_gflop_per_step": round(f16_fl / max(a.steps, 1) / 1e9, 1),
            "by_kernel_ms_per_step": {KNAME.get(k[0], str(k[0])) + (" f16" if k[1] else " f32"): round(v[0] / max(a.steps, 1), 3)
                                      for k, v in sorted(per.items())},
        },
        "end_to_end_frac_of_peak": None,
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
                   "parallelism": "dp%d" % world, "collective": "all_gather(SR outputs), asynchronous, overlapped with the next step" if world > 1 and not a.no_gather else "none"},
        "roofline": roofline,
        "secondary": secondary,
    }
    roofline["end_to_end_frac_of_peak"] = round(out["value"] / world * (GF_F16_FIXED + GF_F16_PER_GLYPH * n + GF_FP32_VIT) / 1e3 / peak, 4)

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
        lq_c, locs_c = lq[:k].cpu(), locs[:k]
        lab_c = labels[:k]
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
