"""Batched driver for the hot path (SURVEY.md §8f NEXT-1): replaces the reference's batch-1 ``for img_name``
loop (test_sr.py:77-197) by one pass over a whole batch —

    encoder(LQ[B]) → w[B] → TSPGAN over all ΣN glyphs of the batch at once (each glyph uses its image's single
    style vector, test_sr.py:183) → TSPSRNet with the NHWC priors handed over in place

— and shards images data-parallel across the GPUs of a node (one process per GPU, torch.distributed on RCCL)
with one all-gather of the SR outputs.  Images are independent, so there is no other
collective and no weight traffic.
"""
import torch

from . import ops
from .packing import torch_dtype


class MarconetPipeline:
    def __init__(self, encoder, gan, sr, precision="fp16", glyph_chunk=1024):
        self.encoder, self.gan, self.sr = encoder, gan, sr
        self.glyph_chunk = glyph_chunk
        self.set_precision(precision)

    def set_precision(self, precision):
        for m in (self.encoder, self.gan, self.sr):
            m.set_precision(precision)
        self.precision = precision
        return self

    @torch.no_grad()
    def forward_batch(self, lq, labels, locs, return_nhwc=False):
        """lq [B,3,32,512] fp32 (device); labels: list of B int64 [n_b,1] tensors; locs [B, ≥2·max n_b] fp32.
        → SR [B,3,128,2048] fp32 NCHW (or NHWC [B,128,2048,8] in the compute dtype if return_nhwc)."""
        dev = lq.device
        B = lq.shape[0]
        counts = [int(l.shape[0]) for l in labels]
        _, _, w = self.encoder(lq)                                    # test_sr.py:146
        tg = self.gan.TextGenerator
        tg.precision = self.precision
        if sum(counts):
            lab = torch.cat([l.reshape(-1, 1) for l in labels if l.shape[0]], dim=0).to(dev).long().contiguous()
            if int(lab.min()) < 0 or int(lab.max()) >= tg.class_num:
                raise RuntimeError("label index out of range [0,%d)" % tg.class_num)
            img_of = torch.repeat_interleave(torch.arange(B, device=dev), torch.tensor(counts, device=dev))
            styles = w.index_select(0, img_of).contiguous()           # w0.repeat(n,1) per image (test_sr.py:183)
            p64s, p32s = [], []
            for s in range(0, lab.shape[0], self.glyph_chunk):        # bounded working set for huge batches
                _, a, c = tg.forward_nhwc(styles[s:s + self.glyph_chunk].contiguous(), lab[s:s + self.glyph_chunk].contiguous())
                p64s.append(a)
                p32s.append(c)
            p64 = p64s[0] if len(p64s) == 1 else torch.cat(p64s)
            p32 = p32s[0] if len(p32s) == 1 else torch.cat(p32s)
        else:
            p64 = p32 = None
        y = self.sr.forward_packed(lq, p64, p32, counts, counts, locs)   # test_sr.py:197
        return y if return_nhwc else ops.nhwc_to_nchw(y, c=3)


def shard_range(total, rank, world):
    """contiguous, balanced split of ``total`` images over ``world`` ranks → (start, stop) of ``rank``"""
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def all_gather_outputs(local, total, group=None):
    """all-gather of per-rank SR outputs [b_r,...] → [total,...] on every rank (RCCL over xGMI when the
    backend is 'nccl'; gloo in the CPU tests).  Uneven shards are padded to the largest shard."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if world == 1:
        return local
    sizes = [shard_range(total, r, world) for r in range(world)]
    mx = max(b - a for a, b in sizes)
    if local.shape[0] < mx:
        pad = torch.zeros((mx - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad])
    out = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    if all(b - a == mx for a, b in sizes):
        return out
    return torch.cat([out[r * mx: r * mx + (b - a)] for r, (a, b) in enumerate(sizes)])
