"""Worker of tests/test_multigpu_gpu.py: one process per GPU (backend nccl = RCCL), launched with RANK / WORLD_SIZE /
MASTER_* in the environment.  Every rank runs MarconetPipeline.forward_sharded over the same seeded global batch; rank 0
additionally runs the whole batch alone and writes whether the gathered result equals it bit for bit."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    out_path, total = sys.argv[1], int(sys.argv[2])
    force = len(sys.argv) > 3 and sys.argv[3] == "force"        # WORLD_SIZE=1: the collectives still run (RCCL on a 1-GPU box)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", device_id=dev)
    try:
        from marconet_amd import networks, synthetic
        from marconet_amd.pipeline import MarconetPipeline, OverlappedGather
        torch.set_num_threads(max(1, (os.cpu_count() or 2) // world))
        enc, gan, sr = networks.TextContextEncoderV2(), networks.TSPGAN(), networks.TSPSRNet()
        enc.load_state_dict(synthetic.make_encoder_state_dict(), strict=True)
        gan.load_state_dict(synthetic.make_gan_state_dict(), strict=True)
        sr.load_state_dict(synthetic.make_sr_state_dict(), strict=True)
        res = {"world": dist.get_world_size(), "backend": dist.get_backend()}
        counts = [(3 * i) % 5 for i in range(total)]                     # includes strips without glyphs
        widths = [512 - 37 * (i % 4) for i in range(total)]
        lq = synthetic.make_lq(91, total, widths)
        labels = [synthetic.make_labels(900 + i, c) for i, c in enumerate(counts)]
        locs = synthetic.make_locs(counts, widths, max_glyphs=5)
        for prec in ("fp32", "fp16x2", "fp16x3", "fp16"):
            pipe = MarconetPipeline(enc.eval().to(dev), gan.eval().to(dev), sr.eval().to(dev), precision=prec)
            for output in ("u8_bgr", "nchw_f32"):
                full = pipe.forward_sharded(lq, labels, locs, output=output, force_collective=force)
                if rank == 0:
                    alone = pipe.forward_batch(lq.to(dev), labels, locs, output=output)
                    res["%s.%s" % (prec, output)] = bool(torch.equal(full, alone)) and tuple(full.shape) == tuple(alone.shape)
            if total % world == 0:
                # bench.py's overlapped form: two steps in flight — the gather of step 0 runs on the collective's stream while step 1
                # computes on the compute stream; the results must be the two steps' own outputs (stream ordering)
                from marconet_amd.pipeline import shard_range
                a, b = shard_range(total, rank, world)
                og = OverlappedGather()
                y0 = pipe.forward_batch(lq[a:b].to(dev), labels[a:b], locs[a:b], output="u8_bgr")
                first = og.submit(y0)
                y1 = pipe.forward_batch(lq[a:b].flip(0).contiguous().to(dev), labels[a:b][::-1], locs[a:b].flip(0).contiguous(), output="u8_bgr")
                g0 = og.submit(y1)
                g0 = g0.clone()
                g1 = og.flush()
                if rank == 0:
                    w0 = pipe.forward_batch(lq.to(dev), labels, locs, output="u8_bgr")
                    res["%s.overlapped" % prec] = (first is None and bool(torch.equal(g0, w0)) and tuple(g1.shape) == tuple(w0.shape)
                                                   and bool(torch.equal(g1[a:b], y1)))
        if rank == 0:
            json.dump(res, open(out_path, "w"))
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
