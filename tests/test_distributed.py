"""N>1 path of the data-parallel driver on CPU: world_size-2 ``gloo`` processes exercise the
image sharding and the one collective of the path (all-gather of SR outputs), including uneven shards.  On the
GPU node the same code runs with backend 'nccl' (= RCCL over xGMI); only the process-group backend differs."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_sr(idx):
    """stand-in for the per-image SR output: a deterministic function of the GLOBAL image index only"""
    g = torch.Generator().manual_seed(1000 + int(idx))
    return torch.rand((3, 8, 32), generator=g)


def _worker(rank, world, port, total, out_q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from marconet_amd.pipeline import all_gather_outputs, shard_range
        a, b = shard_range(total, rank, world)
        local = torch.stack([_fake_sr(i) for i in range(a, b)]) if b > a else torch.zeros((0, 3, 8, 32))
        full = all_gather_outputs(local, total)
        want = torch.stack([_fake_sr(i) for i in range(total)])
        ok = bool(torch.equal(full, want))
        if total % world == 0:        # the overlapped (asynchronous, double-buffered) form used by bench.py --gpus N
            from marconet_amd.pipeline import OverlappedGather
            og = OverlappedGather()
            assert og.submit(local) is None
            prev = og.submit(local * 2)                       # returns step 0's result while step 1's gather is in flight
            ok = ok and bool(torch.equal(prev, want)) and bool(torch.equal(og.flush(), want * 2)) and og.flush() is None
        out_q.put((rank, tuple(full.shape), ok, (a, b)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [8, 7, 3])
def test_all_gather_outputs_world2_gloo(total):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ranges = sorted(r[3] for r in res)
    assert ranges[0][0] == 0 and ranges[-1][1] == total and ranges[0][1] == ranges[1][0]      # contiguous cover
    for rank, shape, equal, _ in res:
        assert shape == (total, 3, 8, 32), (rank, shape)
        assert equal, "rank %d: gathered outputs differ from the 1-process result" % rank


class _HostPipe:
    """MarconetPipeline.forward_sharded over a stand-in forward_batch (a deterministic function of each image alone), so that the
    real sharding / empty-shard / gather logic of the product driver runs on CPU"""

    def __init__(self):
        from marconet_amd.pipeline import MarconetPipeline
        self.sr = torch.nn.Linear(1, 1)                       # forward_sharded only asks it for a device
        self.forward_sharded = MarconetPipeline.forward_sharded.__get__(self)

    def forward_batch(self, lq, labels, locs, output="u8_bgr"):
        y = (lq.sum(dim=(1, 2, 3)) + torch.tensor([float(l.sum()) for l in labels]) + locs.sum(dim=1)).reshape(-1, 1, 1, 1)
        shape = (lq.shape[0], 128, 4 * lq.shape[3], 3) if output == "u8_bgr" else (lq.shape[0], 3, 128, 4 * lq.shape[3])
        y = y.expand(shape)
        return (y.abs() % 255).to(torch.uint8).contiguous() if output == "u8_bgr" else y.float().contiguous()


def _sharded_worker(rank, world, port, total, out_q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(7)
        lq = torch.rand((total, 3, 4, 16), generator=g)
        labels = [torch.arange(i % 3).reshape(-1, 1) for i in range(total)]
        locs = torch.rand((total, 4), generator=g)
        pipe = _HostPipe()
        ok = True
        for output in ("u8_bgr", "nchw_f32"):
            full = pipe.forward_sharded(lq, labels, locs, output=output)
            ok = ok and bool(torch.equal(full, pipe.forward_batch(lq, labels, locs, output=output)))
        out_q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [6, 5, 1])
def test_forward_sharded_world2_gloo(total):
    """the product's data-parallel driver (MarconetPipeline.forward_sharded): contiguous shards, an empty shard when there
    are more ranks than images, uint8 / fp32 gather — every rank ends with the single-process result"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)


def test_shard_range_partitions_exactly():
    from marconet_amd.pipeline import shard_range
    for total in (0, 1, 5, 64, 1024, 1027):
        for world in (1, 2, 4, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_balance_shards_for_mixed_widths():
    """configs[4]: every image lands on exactly one rank and the per-rank algorithmic work is balanced to within one image"""
    import random
    from marconet_amd.pipeline import balance_shards
    rng = random.Random(3)
    widths = [rng.choice([128, 192, 256, 320, 384, 448, 512]) - rng.randint(0, 40) for _ in range(203)]
    counts = [rng.randint(0, 16) for _ in widths]
    f = lambda b: 108.0 + 3.69 + 484.1 * min(512, (widths[b] + 63) // 64 * 64) / 512.0 + 89.03 * counts[b]
    for world in (1, 2, 4, 8):
        parts = balance_shards(widths, counts, world)
        assert sorted(b for p_ in parts for b in p_) == list(range(len(widths)))
        loads = [sum(f(b) for b in p_) for p_ in parts]
        assert max(loads) - min(loads) <= max(f(b) for b in range(len(widths))) + 1e-9


def test_host_side_label_and_loc_helpers():
    """collapse_indices == the oracle's clear_labels on the same logits; (left,right) → (centre, half-width) conversion"""
    import random
    import torch
    from marconet_amd.pipeline import collapse_indices, locs_from_left_right
    from oracle import marconet_oracle as O
    rng = random.Random(5)
    for _ in range(20):
        idx = [rng.choice([3, 3, 7, 6735, 6735, 12, 6734]) for _ in range(64)]
        logits = torch.full((64, 6736), -1.0)
        logits[torch.arange(64), torch.tensor(idx)] = 1.0
        assert collapse_indices(idx) == O.clear_labels(logits)
    lr = torch.rand(3, 32)
    out = locs_from_left_right(lr)
    assert torch.allclose(out[:, 0::2], (lr[:, 1::2] + lr[:, 0::2]) / 2) and torch.allclose(out[:, 1::2], (lr[:, 1::2] - lr[:, 0::2]) / 2)


def _world1_worker(port, out_q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        from marconet_amd.pipeline import OverlappedGather, all_gather_outputs
        local = torch.stack([_fake_sr(i) for i in range(3)])
        same = all_gather_outputs(local, 3)                       # no collective in a world of one ...
        forced = all_gather_outputs(local, 3, force=True)         # ... unless forced: a new tensor, equal content
        og = OverlappedGather()
        first = og.submit(local)
        out_q.put((same is local, forced is not local and bool(torch.equal(forced, local)), first is None and bool(torch.equal(og.flush(), local))))
    finally:
        dist.destroy_process_group()


def test_forced_collective_in_a_world_of_one_gloo():
    """the switch the 1-GPU RCCL test uses (tests/test_multigpu_gpu.py::test_rccl_path_runs_in_a_world_of_one)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_world1_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=120)
    p.join(timeout=60)
    assert res == (True, True, True)


# ------------------------------------------------------------------------------------------------ configs[4] on 8 ranks
class _MixedStandIn:
    """MarconetPipeline._forward_mixed_widths over stand-in nets (CPU): every image's output is a deterministic function of that
    image's strip, bucket width, labels, priors and glyph windows only — what the real nets guarantee (batch-invariant kernels)"""

    class _Enc(torch.nn.Module):
        def forward(self, lq):
            w = lq.mean(dim=(1, 2, 3)).reshape(-1, 1).expand(-1, 512).contiguous()
            return None, None, w

    class _TG:
        precision, class_num = "fp32", 6736

        def forward_nhwc(self, styles, labels, need_image=True, style_index=None, p64_out=None, p32_out=None, image_precision=None):
            v = styles[:, :1] + labels.float() * 1e-3
            return None, v.reshape(-1, 1, 1, 1).expand(-1, 2, 2, 4).contiguous(), (2 * v).reshape(-1, 1, 1, 1).expand(-1, 1, 1, 4).contiguous()

    class _SR(torch.nn.Module):
        precision = "fp32"

        def forward_packed(self, lq, p64, p32, c64, c32, locs, nchw_out=False, tables=None):
            out, g0 = [], 0
            for b, c in enumerate(c64):
                v = lq[b].sum() + float(lq.shape[3])
                if c:
                    t64, t32 = tables[1], tables[0]
                    v = v + p64[g0:g0 + c].sum() + p32[g0:g0 + c].sum() + float(t64.g_x1[g0:g0 + c].sum() + t32.g_w[g0:g0 + c].sum())
                g0 += c
                out.append(v.reshape(1, 1, 1).expand(3, 128, 4 * lq.shape[3]))
            return torch.stack(out).contiguous()

    def __init__(self):
        from marconet_amd.pipeline import MarconetPipeline
        self.encoder, self.sr = self._Enc(), self._SR()
        self.gan = type("G", (), {"TextGenerator": self._TG()})()
        self.precision, self.need_prior_image, self.check_finite, self._finite = "fp32", True, False, None
        self._checks = lambda: False
        self._image_precision = lambda: None
        self._raise_if_not_finite = lambda: None
        self.run = MarconetPipeline._forward_mixed_widths.__get__(self)


def _mixed_problem(total):
    import random
    rng = random.Random(11)
    widths = [rng.choice([128, 192, 256, 320, 384, 448, 512]) - rng.randint(0, 30) for _ in range(total)]
    counts = [min(16, max(0, w // 32 - rng.randint(0, 3))) for w in widths]
    g = torch.Generator().manual_seed(12)
    lq = torch.rand((total, 3, 32, 512), generator=g)
    labels = [torch.randint(0, 6000, (c, 1), generator=g) for c in counts]
    locs = torch.zeros((total, 32))
    for b, (w, c) in enumerate(zip(widths, counts)):
        for j in range(c):
            locs[b, 2 * j] = (j + 0.5) * (w / max(c, 1)) / 512.0
            locs[b, 2 * j + 1] = 0.4 * (w / max(c, 1)) / 512.0
    return lq, widths, counts, labels, locs


def _mixed_worker(rank, world, port, total, out_q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from marconet_amd.pipeline import balance_shards
        lq, widths, counts, labels, locs = _mixed_problem(total)
        mine = balance_shards(widths, counts, world)[rank]
        pipe = _MixedStandIn()
        outs = pipe.run(lq[mine], [widths[i] for i in mine], [labels[i] for i in mine], locs[mine], 64) if mine else []
        local = [(i, tuple(o.shape), float(o.double().sum())) for i, o in zip(mine, outs)]
        gathered = [None] * world
        dist.all_gather_object(gathered, local)
        if rank == 0:
            whole = pipe.run(lq, widths, labels, locs, 64)
            want = {i: (tuple(o.shape), float(o.double().sum())) for i, o in enumerate(whole)}
            got = {i: (s, v) for part in gathered for i, s, v in part}
            out_q.put((sorted(got) == list(range(total)), all(got[i] == want[i] for i in range(total)), [len(p_) for p_ in gathered]))
    finally:
        dist.destroy_process_group()


def test_config5_mixed_widths_on_8_simulated_ranks():
    """BASELINE configs[4] (mixed widths, length bucketing, 8 GPUs) with the hardware replaced by 8 gloo ranks and stand-in nets:
    balance_shards + forward_mixed_widths per rank reproduce the single-rank outputs image for image (test_sr.py:105-110)"""
    world, total = 8, 61
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mixed_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    covered, equal, sizes = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert covered and equal and sum(sizes) == total and max(sizes) - min(sizes) <= 3
