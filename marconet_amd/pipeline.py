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
from .glyphs import GlyphTables
from .networks import returned_image_precision
from .packing import default_precision, new_tensor, torch_dtype


import os as _os
_NO_STYLE_DEDUPE = bool(int(_os.environ.get("MNET_NO_STYLE_DEDUPE", "0")))      # A/B knob


class MarconetPipeline:
    def __init__(self, encoder, gan, sr, precision=None, glyph_chunk=1024, need_prior_image=True, check_finite=None,
                 prior_image_precision="auto"):
        """``precision``: None → MARCONET_PRECISION or "fp32" (the parity mode, ≤1e-3 vs the CPU reference).  The fp16
        storage modes are opt-in: plain "fp16" stores activations as one half (max 65504, 11 significant bits) and has only
        been validated on O(1)-activation synthetic checkpoints — with trained StyleGAN-type weights a modulated activation
        can overflow.  ``check_finite`` (default: on for every precision but fp32) makes forward_batch verify that the SR
        result is finite (one flag read back per batch) and raise FloatingPointError instead of returning inf/NaN silently."""
        self.encoder, self.gan, self.sr = encoder, gan, sr
        self.glyph_chunk = glyph_chunk
        self.check_finite = check_finite
        self._finite = None
        precision = default_precision() if precision is None else precision
        # True (default): the generator also produces its 128-px structure image, as test_sr.py:183 does (it is only ever
        # used for the saved visualisation, test_sr.py:203-232).  False is an opt-in for throughput serving.
        self.need_prior_image = need_prior_image
        # precision mode of the generator levels that feed only that image (while forward_batch drops it; ``return_prior=True`` returns it, in the mode's arithmetic): "auto" = plain fp16 in
        # the fp16x2 / fp16x3 modes, the mode's own arithmetic otherwise; None = always the mode's own.  The priors and the SR output
        # do not depend on it (tests/test_modules_gpu.py::test_prior_image_precision_leaves_sr_bits_unchanged)
        self.prior_image_precision = prior_image_precision
        self.set_precision(precision)

    def _image_precision(self):
        if self.prior_image_precision == "auto":
            return "fp16" if self.precision in ("fp16x2", "fp16x3") else None
        return self.prior_image_precision

    def _checks(self):
        return self.check_finite if self.check_finite is not None else self.precision != "fp32"

    def set_precision(self, precision):
        for m in (self.encoder, self.gan, self.sr):
            m.set_precision(precision)
        self.precision = precision
        return self

    @torch.no_grad()
    def forward_batch(self, lq, labels, locs, return_nhwc=False, output="nchw_f32", return_prior=False):
        """lq [B,3,32,512] fp32 (device); labels: list of B int64 [n_b,1] tensors; locs [B, ≥2·max n_b] fp32 — labels and
        locs on the device or (preferably: no synchronisation then) on the host.
        → SR [B,3,128,2048] fp32 NCHW (the reference's return), or NHWC [B,128,2048,8] in the compute dtype if return_nhwc,
        or — output="u8_bgr" — the script's post-processed image [B,128,2048,3] uint8 BGR (test_sr.py:198-200; 4x fewer bytes
        to copy to the host or to all-gather).
        ``return_prior=True`` → ``(that, prior_cha)``: the generator's structure images of all glyphs, fp32 NCHW [ΣN,3,128,128] in
        strip order — what test_sr.py:183 names ``prior_cha`` and :207-212 turn into the panel's last row.  The levels behind the two
        prior levels then run in the MODE's arithmetic (``prior_image_precision="auto"`` demotes them to plain fp16 only while the
        image is dropped), so the returned image holds the same bar as ``TSPGAN.forward``'s; the SR output is the same bits either way."""
        dev = lq.device
        counts = [int(l.shape[0]) for l in labels]
        if return_prior and not self.need_prior_image:
            raise ValueError("return_prior=True needs MarconetPipeline(need_prior_image=True): the 128-px level is what produces the image")
        with ops.on_device(lq):
            lab, img_of = self._host_prep(labels, counts, dev)
            prior = torch.empty((sum(counts), 128, 128, 4), dtype=torch.float32, device=dev) if return_prior else None
            y = self._core(lq, lab, img_of, counts, locs, None, return_nhwc, output, prior_images=prior)
            self._raise_if_not_finite()
            return (y, ops.nhwc_to_nchw(prior, c=3) if sum(counts) else prior.new_zeros((0, 3, 128, 128))) if return_prior else y

    def _raise_if_not_finite(self):
        """half-range modes (fp16 / fp16x3 / fp16x2 store |activation| < 65504): an overflow turns into NaN on its way through the
        GroupNorm statistics of the following layers (and the last layer writes an infinite pre-activation as NaN instead of
        tanh's +-1), so the SR result carries it; one flag is read back per batch."""
        # _finite: int32 flags written by mnet_nonfinite_flag (1 = inf / NaN found) — ONE device→host copy, tested on the host
        if self._checks() and self._finite is not None and any(self._finite.cpu().tolist()):
            raise FloatingPointError("marconet_amd: non-finite SR output in %s mode (activations beyond the fp16 range 65504?) — use "
                                     "precision='fp32' for these weights" % self.precision)

    def _host_prep(self, labels, counts, dev):
        """labels (host or device) → validated device tensors (labels [ΣN,1], glyph→image index [ΣN])"""
        tg = self.gan.TextGenerator
        if not sum(counts):
            return None, None
        lab = torch.cat([l.reshape(-1, 1) for l in labels if l.shape[0]], dim=0).long()
        # labels / locs held on the HOST (where the OCR / detector front-end leaves them, test_sr.py:121-149) cost no
        # device→host synchronisation: the whole forward is then enqueued without the host ever waiting for the GPU
        if int(lab.min()) < 0 or int(lab.max()) >= tg.class_num:
            raise RuntimeError("label index out of range [0,%d)" % tg.class_num)
        return lab.to(dev).contiguous(), torch.repeat_interleave(torch.arange(len(counts)), torch.tensor(counts)).to(dev)

    @torch.no_grad()
    def _core(self, lq, lab, img_of, counts, locs, tables, return_nhwc=False, output="nchw_f32", prior_images=None):
        """the device-side part of forward_batch: nothing here touches host data (it can be captured in a HIP graph).
        ``prior_images`` (fp32 [ΣN,128,128,4] or None): filled with the generator's structure images, chunk by chunk"""
        _, _, w = self.encoder(lq)                                    # test_sr.py:146
        tg = self.gan.TextGenerator
        tg.precision = self.precision
        if sum(counts):
            # w0.repeat(n,1) per image (test_sr.py:183): the generator gets the B distinct styles + the glyph→image index
            # the generator runs in chunks of glyphs (bounded working set for huge batches) and writes its two prior levels
            # straight into the all-glyph buffers TSPSRNet reads (no concatenation pass: 34 GB of copies per step at batch 256)
            G, gdt = lab.shape[0], torch_dtype(tg.precision)
            p64 = new_tensor((G, 64, 64, 256), gdt, lq.device)
            p32 = new_tensor((G, 32, 32, 512), gdt, lq.device)
            # a returned image is computed in the mode's arithmetic; a dropped one under ``prior_image_precision`` (see forward_batch)
            # (a RETURNED image in the fp16x2 mode: its image-only level in the three-product arithmetic, networks.returned_image_precision)
            img_prec = self._image_precision() if prior_images is None or self.prior_image_precision != "auto" else returned_image_precision(self.precision)
            for s in range(0, G, self.glyph_chunk):
                e = min(G, s + self.glyph_chunk)
                if _NO_STYLE_DEDUPE:
                    img = tg.forward_nhwc(w.index_select(0, img_of[s:e]).contiguous(), lab[s:e].contiguous(),
                                          need_image=self.need_prior_image, p64_out=p64[s:e], p32_out=p32[s:e], image_precision=img_prec)[0]
                else:
                    img = tg.forward_nhwc(w, lab[s:e].contiguous(), need_image=self.need_prior_image, style_index=img_of[s:e].contiguous(),
                                          p64_out=p64[s:e], p32_out=p32[s:e], image_precision=img_prec)[0]
                if prior_images is not None:
                    prior_images[s:e].copy_(img)
            sr_dtype = torch_dtype(self.sr.precision)                 # the three nets may run in different precision modes
            p64, p32 = ops.convert(p64, sr_dtype), ops.convert(p32, sr_dtype)
        else:
            p64 = p32 = None
        if output != "u8_bgr" and not return_nhwc:
            y = self.sr.forward_packed(lq, p64, p32, counts, counts, locs, nchw_out=True, tables=tables)   # test_sr.py:197
            self._finite = ops.nonfinite_flag(y) if self._checks() else None    # device-side flag (mnet_nonfinite_flag): no synchronisation here
            return y
        y = self.sr.forward_packed(lq, p64, p32, counts, counts, locs, tables=tables)
        self._finite = ops.nonfinite_flag(y) if self._checks() else None
        if output == "u8_bgr":                                           # test_sr.py:198-200 fused: [B,128,2048,3] uint8
            return ops.sr_postprocess(y, u8=True)
        return y if return_nhwc else ops.nhwc_to_nchw(y, c=3)


    def forward_mixed_widths(self, lq, content_widths, labels, locs, bucket=64):
        with torch.no_grad(), ops.on_device(lq):
            return self._forward_mixed_widths(lq, content_widths, labels, locs, bucket)

    def _forward_mixed_widths(self, lq, content_widths, labels, locs, bucket):
        """BASELINE configs[4]: a batch of strips of different content widths (each zero-padded to 512, i.e. -1 after
        Normalize, test_sr.py:100-115).  The encoder needs the 512-wide strip (its token-axis LayerNorm(64)/Linear(64,.)
        pin 64 tokens, models/textvit_arch.py:59-62,141-144); TSPSRNet is width-agnostic, so images are bucketed by
        ``ceil(w_b / bucket) * bucket`` and each bucket runs the SR net at its own width W'.  Glyph centres are the INTEGER centres of
        the 512-padded run the locs were normalised for — trunc(loc * 512) at the 32-row scale, trunc(loc * 1024) at the 64-row scale
        (models/networks.py:426) — with the windows clipped to W' (2 W'); re-normalising loc to W' in fp32 could move a centre by one
        pixel.  SURVEY.md §8d: the oracle of this mode is the reference TSPSRNet called at the same W' with locs that reproduce those
        centres — it is NOT equal to the 512-padded run (GroupNorm reduces over the whole map; a window that crosses the bucket edge is
        clipped there).
        → list (input order) of SR tensors [3, 128, 4·W'_b] fp32."""
        dev = lq.device
        B = lq.shape[0]
        if lq.shape[-1] != 512 or len(content_widths) != B or len(labels) != B:
            raise ValueError("forward_mixed_widths: lq must be [B,3,32,512] with one content width and one label list per image")
        counts = [int(l.shape[0]) for l in labels]
        widths = []
        for w_ in content_widths:
            w_ = int(w_)
            if w_ <= 0 or w_ > 512:                      # the reference skips wider strips (test_sr.py:108-110)
                raise ValueError("content width %d outside (0, 512]: split the strip into <=512-px segments" % w_)
            widths.append(min(512, (w_ + bucket - 1) // bucket * bucket))
        _, _, w = self.encoder(lq)
        tg = self.gan.TextGenerator
        tg.precision = self.precision
        starts = [0]
        for c in counts:
            starts.append(starts[-1] + c)
        p64 = p32 = None
        if starts[-1]:
            lab = torch.cat([l.reshape(-1, 1) for l in labels if l.shape[0]], dim=0).to(dev).long().contiguous()
            if int(lab.min()) < 0 or int(lab.max()) >= tg.class_num:
                raise RuntimeError("label index out of range [0,%d)" % tg.class_num)
            img_of = torch.repeat_interleave(torch.arange(B, device=dev), torch.tensor(counts, device=dev))
            _, p64, p32 = tg.forward_nhwc(w.index_select(0, img_of).contiguous(), lab, need_image=self.need_prior_image,
                                          image_precision=self._image_precision())
        out = [None] * B
        buckets = sorted(set(widths))
        flags = torch.empty((len(buckets),), dtype=torch.int32, device=dev) if self._checks() else None     # one finiteness flag per bucket
        for kb, wb in enumerate(buckets):
            idx = [b for b in range(B) if widths[b] == wb]
            gsel = [g for b in idx for g in range(starts[b], starts[b + 1])]
            cb = [counts[b] for b in idx]
            it = torch.tensor(idx, device=dev)
            lq_b = lq.index_select(0, it)[:, :, :, :wb].contiguous()
            # glyph centres as in the 512-padded run the locs were normalised for (trunc(loc * 512) / trunc(loc * 1024): integer tables
            # from the ORIGINAL locs — re-normalising to the bucket width in fp32 can move a centre by one pixel), windows clipped to
            # the bucket's width
            lh = locs.detach().float().cpu().numpy()[idx]
            tables = (GlyphTables(lh, cb, wb, 16, dev, centre_w=512), GlyphTables(lh, cb, 2 * wb, 32, dev, centre_w=1024)) if gsel else None
            if gsel:
                gt = torch.tensor(gsel, device=dev)
                a, c = ops.take_rows(p64, gt), ops.take_rows(p32, gt)
            else:
                a = c = None
            y = self.sr.forward_packed(lq_b, a, c, cb, cb, None, nchw_out=True, tables=tables)
            if flags is not None:
                ops.nonfinite_flag(y, out=flags[kb:kb + 1])
            for k, b in enumerate(idx):
                out[b] = y[k]
        self._finite = flags
        self._raise_if_not_finite()
        return out


    @torch.no_grad()
    def forward_sharded(self, lq, labels, locs, output="u8_bgr", group=None, force_collective=False):
        """Data-parallel form of forward_batch: every rank of the process group (one process per GPU; backend
        'nccl' = RCCL over xGMI) is handed the SAME global batch description (lq [B,3,32,512] on the host or on the rank's
        device, labels list, locs [B,2m]), processes its contiguous shard ``shard_range(B, rank, world)`` and all-gathers the
        outputs — the one collective of the path; by default the post-processed uint8 BGR image (0.75 MiB per image instead of
        the 3 MiB fp32 tensor).  → [B,128,2048,3] uint8 (or [B,3,128,2048] fp32 for output="nchw_f32") on every rank, equal
        bit for bit to a single-GPU forward_batch of the whole batch (kernels are batch-invariant).
        ``force_collective``: run the all-gather even in a world of one (an initialised process group is required) — the
        RCCL path exercised on a single-GPU box."""
        import torch.distributed as dist
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        B = lq.shape[0]
        a, b = shard_range(B, rank, world)
        dev = next(self.sr.parameters()).device
        if b > a:
            y = self.forward_batch(lq[a:b].to(dev), labels[a:b], locs[a:b], output=output)
        else:                         # more ranks than images: an empty shard still takes part in the collective
            shape = (0, 128, 4 * lq.shape[3], 3) if output == "u8_bgr" else (0, 3, 128, 4 * lq.shape[3])
            y = torch.empty(shape, dtype=torch.uint8 if output == "u8_bgr" else torch.float32, device=dev)
        if world == 1 and not force_collective:
            return y
        if not dist.is_initialized():
            raise RuntimeError("forward_sharded(force_collective=True) needs an initialised process group")
        return all_gather_outputs(y, B, group, force=force_collective)

    @torch.no_grad()
    def restore_strips(self, strips, with_prior=False):
        """The body of test_sr.py's ``for img_name`` loop (:77-201) for a list of strips prepared by ``lq_io.strip_from_png``
        (dicts with lq [1,3,32,512], labels int64 [n,1], locs [1,2n], show_w) — as ONE batch.  → list of uint8 BGR arrays
        [128, show_w, 3] on the host (``ShowSR``, test_sr.py:198-201: the post-processed SR result cropped to the content
        width); ``None`` for a strip the script would skip (a character outside the alphabet → label −1 → the generator
        raises, test_sr.py:181-190; no character at all, :168-170).
        ``with_prior=True`` → list of ``(ShowSR, prior128)``: ``prior128`` = the strip's structure images side by side, float32 RGB
        [128, 128·n, 3] = ``np.hstack`` of ``prior_cha * 0.5 + 0.5`` (test_sr.py:208-211) — the array the script hands to ``cv2.resize``
        for the panel's last row (:212; cv2 is not part of this build, the resize stays with the caller)."""
        dev = next(self.sr.parameters()).device
        n_cls = self.gan.TextGenerator.class_num
        keep = [i for i, s in enumerate(strips)
                if s["labels"].numel() > 0 and int(s["labels"].min()) >= 0 and int(s["labels"].max()) < n_cls]
        out = [None] * len(strips)
        if not keep:
            return out
        m = max(int(strips[i]["labels"].shape[0]) for i in keep)
        locs = torch.zeros((len(keep), 2 * m), dtype=torch.float32)
        for k, i in enumerate(keep):
            locs[k, :strips[i]["locs"].shape[1]] = strips[i]["locs"][0]
        lq = torch.cat([strips[i]["lq"] for i in keep]).to(dev)
        y = self.forward_batch(lq, [strips[i]["labels"] for i in keep], locs, output="u8_bgr", return_prior=with_prior)
        if with_prior:
            y, prior = y
            prior = (prior * 0.5 + 0.5).permute(0, 2, 3, 1).cpu().numpy()            # test_sr.py:208
        y = y.cpu().numpy()
        g = 0
        for k, i in enumerate(keep):
            out[i] = y[k, :, :strips[i]["show_w"], :]
            if with_prior:
                n = int(strips[i]["labels"].shape[0])
                out[i] = (out[i], prior[g:g + n].transpose(1, 0, 2, 3).reshape(128, 128 * n, 3))   # hstack of the n images (:209-211)
                g += n
        return out

    @torch.no_grad()
    def forward_blind(self, lq, max_glyphs=16):
        """Self-contained end-to-end pass (SURVEY.md §8f NEXT-2): labels and glyph locations come from the encoder itself
        instead of the YOLO + OCR front-end — labels = ``clear_labels(logits)`` (test_w.py:34-40), locs = the encoder's
        (left, right) pairs converted to (centre, half-width) (Train/tspgan/models/tspgan_model.py:331-336).
        → (SR [B,3,128,2048], labels list, locs [B,32])."""
        logits, locs_lr, _ = self.encoder(lq)
        labels = clear_labels_batch(logits)
        labels = [l[:max_glyphs] for l in labels]
        locs = locs_from_left_right(locs_lr)
        return self.forward_batch(lq, [l.to(lq.device) for l in labels], locs), labels, locs


ALPHABET_SIZE = 6735          # utils/alphabets.py: 6735 characters; class 6735 = blank / padding


def clear_labels_batch(logits):
    """test_w.py:34-40 for a whole batch: argmax over the 6736 classes (HIP kernel, first maximal index like torch.max),
    ONE device→host copy of the [B,64] indices, then the CTC-style collapse on the host (drop repeats, drop blanks).
    → list of B int64 tensors [n_b, 1] (CPU)."""
    B, T, C = logits.shape
    with ops.on_device(logits):
        idx = ops.argmax_rows(logits.reshape(B * T, C).contiguous().float()).reshape(B, T).cpu().tolist()
    return [torch.tensor(collapse_indices(row), dtype=torch.int64).reshape(-1, 1) for row in idx]


def collapse_indices(row, alphabet_size=ALPHABET_SIZE):
    """the CTC-style collapse of test_w.py:37-39 on one row of argmax indices: drop repeats, drop blanks (>= alphabet size)"""
    return [v for i, v in enumerate(row) if not (i > 0 and row[i - 1] == v) and v < alphabet_size]


def locs_from_left_right(locs_lr):
    """(left, right) pairs → (centre, half-width) pairs, Train/tspgan/models/tspgan_model.py:331-336 (elementwise; tiny)."""
    l, r = locs_lr[:, 0::2], locs_lr[:, 1::2]
    out = torch.empty_like(locs_lr)
    out[:, 0::2] = (r + l) / 2.0
    out[:, 1::2] = (r - l) / 2.0
    return out


@torch.no_grad()
def w_interpolation(gan, w1, w2, labels, steps=11):
    """test_w.py:104-108: structure images for styles s·w1 + (1-s)·w2, s = i/(steps-1) — all interpolation steps in ONE
    generator call (each glyph is independent).  w1, w2 [1,512]; labels int64 [n,1] → images [steps, n, 3, 128, 128]."""
    n = labels.shape[0]
    ws = torch.cat([(w1 * (i / (steps - 1)) + w2 * (1 - i / (steps - 1))).repeat(n, 1) for i in range(steps)], dim=0)
    img, _, _ = gan(styles=ws.contiguous(), labels=labels.to(w1.device).repeat(steps, 1), noise=None)
    return img.reshape(steps, n, *img.shape[1:])


class GraphedForward:
    """The whole forward for one (batch size, glyphs per image) signature captured in a HIP graph: the script's batch-1 loop
    (test_sr.py:77) is launch-bound on the host (≈400 kernel launches, 9.5 ms per image against ≈6 ms of GPU work), a graph
    replay is one submission.  Inputs live in static device buffers refreshed before each replay (LQ, labels, glyph→image
    index, the glyph window tables); the output is a static buffer too — consume or clone it before the next call."""

    def __init__(self, pipe, batch, counts, output="nchw_f32", device="cuda"):
        from .glyphs import GlyphTables
        import numpy as np
        self.pipe, self.counts, self.output = pipe, [int(c) for c in counts], output
        if len(self.counts) != batch:
            raise ValueError("one glyph count per image")
        dev = torch.device(device)
        G = sum(self.counts)
        self.lq = torch.zeros((batch, 3, 32, 512), dtype=torch.float32, device=dev)
        self.lab = torch.zeros((G, 1), dtype=torch.int64, device=dev) if G else None
        self.img_of = torch.repeat_interleave(torch.arange(batch), torch.tensor(self.counts)).to(dev) if G else None
        dummy = np.full((batch, 2 * max(self.counts + [1])), 0.5, dtype=np.float32)
        self.widths = (self.lq.shape[3], 2 * self.lq.shape[3])           # feature widths at the 32- and 64-row scales
        self.tables = (GlyphTables(dummy, self.counts, self.widths[0], 16, dev),
                       GlyphTables(dummy, self.counts, self.widths[1], 32, dev)) if G else None
        with ops.on_device(self.lq):
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):                              # warm-up: packs weights, raises kernel attributes
                for _ in range(2):
                    pipe._core(self.lq, self.lab, self.img_of, self.counts, None, self.tables, output=output)
            torch.cuda.current_stream(dev).wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.out = pipe._core(self.lq, self.lab, self.img_of, self.counts, None, self.tables, output=output)
            # the finiteness flag THIS graph refreshes on every replay (pipe._finite is reassigned by every other forward of the
            # pipe — an eager forward_batch, another GraphedForward — and must not be read here)
            self._flag = pipe._finite
            self._checks = pipe._checks()
            self._quiet_without_flag = pipe.precision == "fp32" and pipe.check_finite is None     # as captured

    @property
    def has_flag(self):
        """True when this graph refreshes a finiteness flag on every replay (captured with the check on)"""
        return self._flag is not None

    def raise_if_not_finite(self):
        """reads this graph's own flag (one device→host synchronisation) — for callers that replay with check=False and test later"""
        if self._flag is None:
            if self._quiet_without_flag:
                return       # fp32 has no half-range overflow to report: code shared with the half-range modes may call this unconditionally
            raise RuntimeError("GraphedForward: this graph was captured with the finiteness check off (check_finite=False): "
                               "there is no flag to read — capture with MarconetPipeline(check_finite=True); `has_flag` tells")
        if any(self._flag.cpu().tolist()):
            raise FloatingPointError("marconet_amd: non-finite SR output in %s mode (activations beyond the fp16 range 65504?) — use "
                                     "precision='fp32' for these weights" % self.pipe.precision)

    @torch.no_grad()
    def __call__(self, lq, labels, locs, check=None):
        """``check``: None = as the pipe was configured at capture (on for the half-range modes: one flag read back, i.e. one
        synchronisation per replay); False = no read-back on the latency path — call raise_if_not_finite() when the output is consumed;
        True on a graph captured WITHOUT the flag raises RuntimeError (a check that cannot happen must not pass silently)"""
        from .glyphs import GlyphTables
        counts = [int(l.shape[0]) for l in labels]
        if counts != self.counts or tuple(lq.shape) != tuple(self.lq.shape):
            raise ValueError("this graph was captured for batch %d with glyph counts %s" % (self.lq.shape[0], self.counts))
        self.lq.copy_(lq, non_blocking=True)
        if self.lab is not None:
            lab, _ = self.pipe._host_prep(labels, counts, self.lq.device)
            self.lab.copy_(lab, non_blocking=True)
            lh = locs.detach().float().cpu().numpy()
            GlyphTables(lh, counts, self.widths[0], 16, "cpu").copy_into(self.tables[0])
            GlyphTables(lh, counts, self.widths[1], 32, "cpu").copy_into(self.tables[1])
        self.graph.replay()
        if self._checks if check is None else check:
            self.raise_if_not_finite()
        return self.out


def balance_shards(content_widths, glyph_counts, world, bucket=64):
    """configs[4] on N GPUs: assign images to ranks so that the algorithmic work per rank is balanced
    (longest-processing-time greedy on F_b = 108.0 + 3.69 + 484.1·W'_b/512 + 89.03·n_b GFLOP).
    → list of ``world`` index lists (each sorted by bucket width so a rank runs few distinct widths)."""
    cost = []
    for b, (w_, n) in enumerate(zip(content_widths, glyph_counts)):
        wb = min(512, (int(w_) + bucket - 1) // bucket * bucket)
        cost.append((108.0 + 3.69 + 484.1 * wb / 512.0 + 89.03 * int(n), wb, b))
    load = [0.0] * world
    parts = [[] for _ in range(world)]
    for c, wb, b in sorted(cost, reverse=True):
        r = min(range(world), key=lambda k: (load[k], k))
        load[r] += c
        parts[r].append((wb, b))
    return [[b for _, b in sorted(p_)] for p_ in parts]


def shard_range(total, rank, world):
    """contiguous, balanced split of ``total`` images over ``world`` ranks → (start, stop) of ``rank``"""
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def all_gather_outputs(local, total, group=None, force=False):
    """all-gather of per-rank SR outputs [b_r,...] → [total,...] on every rank (RCCL over xGMI when the
    backend is 'nccl'; gloo in the CPU tests).  Uneven shards are padded to the largest shard.  A world of one returns
    ``local`` without a collective unless ``force`` (the collective then really runs: one rank gathering from itself)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if world == 1 and not force:
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


class OverlappedGather:
    """The all-gather of step i runs (on the collective backend's own stream) while step i+1 computes: ``submit`` enqueues
    it asynchronously into one of two rotating output buffers and returns the gathered result of the PREVIOUS submit
    (None the first time); ``flush`` waits for the last one.  Even shards only (the data-parallel bench)."""

    def __init__(self, group=None):
        self.group = group
        self._bufs = [None, None]
        self._pending = None          # (work handle, buffer, source tensor kept alive)
        self._i = 0

    def submit(self, local):
        import torch.distributed as dist
        world = dist.get_world_size(self.group)
        done = self.flush()
        shape = (world * local.shape[0],) + tuple(local.shape[1:])
        buf = self._bufs[self._i]
        if buf is None or buf.shape != shape or buf.dtype != local.dtype or buf.device != local.device:
            buf = self._bufs[self._i] = torch.empty(shape, dtype=local.dtype, device=local.device)
        src = local.contiguous()
        work = dist.all_gather_into_tensor(buf, src, group=self.group, async_op=True)
        self._pending = (work, buf, src)
        self._i ^= 1
        return done

    def flush(self):
        if self._pending is None:
            return None
        work, buf, _ = self._pending
        work.wait()
        self._pending = None
        return buf
