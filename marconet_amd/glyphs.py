"""Host-side glyph window arithmetic of TSPSRNet's per-character loops (models/networks.py:425-441 and
:459-474) — computed once per forward from a single device→host copy of ``locs`` instead of
~6 implicit syncs per glyph per scale, then uploaded as small int32 tables for the batched kernels.

    center = trunc_toward_zero( fp32(locs[b,2c]) * fp32(W) )        (.int() on a 0-d fp32 tensor)
    x1 = 0 if center < half else center - half
    x2 = W if center + half > W else center + half
    y1 = half - trunc((x2-x1)/2) ;  y2 = y1 + (x2-x1)
The width entry locs[b,2c+1] is read by the reference but overwritten by a constant (:427-428) — ignored.
"""
import numpy as np
import torch


def window(loc_center, feat_w, half):
    center = int(np.float32(loc_center) * np.float32(feat_w))
    x1 = 0 if center < half else center - half
    x2 = feat_w if center + half > feat_w else center + half
    gw = x2 - x1
    y1 = half - int(gw / 2)
    return x1, gw, y1


class GlyphTables:
    """int32 device tables for one scale: g_img[G], g_x1[G], g_y1[G], g_w[G], g_start[B+1].
    Vectorised (numpy) over all glyphs of the batch — at 64 images x 16 glyphs the per-glyph Python loop cost ~60 ms of
    host time per forward; ``window`` above stays the scalar statement of the same arithmetic (and is what the tests pin)."""

    def __init__(self, locs_host, counts, feat_w, half, device, centre_w=None):
        """``centre_w`` (mixed-width bucketing): the feature width the centres are computed at — the width of the 512-padded run the
        locs were normalised for — while windows are clipped to this map's ``feat_w``; default: feat_w (the reference's arithmetic)"""
        counts = np.asarray(counts, dtype=np.int64)
        B = counts.shape[0]
        if B and int(counts.max(initial=0)) * 2 > locs_host.shape[1]:
            b = int(np.argmax(counts))
            raise IndexError("locs has %d entries for image %d but %d glyph priors were given" % (locs_host.shape[1], b, int(counts[b])))
        g_img = np.repeat(np.arange(B, dtype=np.int64), counts)
        g_start = np.concatenate([[0], np.cumsum(counts)])
        c_idx = np.arange(g_img.shape[0], dtype=np.int64) - g_start[g_img]                    # glyph index inside its image
        loc = np.asarray(locs_host, dtype=np.float32)[g_img, 2 * c_idx] if g_img.size else np.zeros((0,), np.float32)
        center = (loc * np.float32(feat_w if centre_w is None else centre_w)).astype(np.int32)   # fp32 product, truncation toward zero
        x1 = np.where(center < half, 0, center - half)
        x2 = np.where(center + half > feat_w, feat_w, center + half)
        gw = x2 - x1
        y1 = half - (gw.astype(np.float64) / 2).astype(np.int64)                              # int(gw / 2): truncation toward zero
        bad = (gw <= 0) | (gw > 2 * half) | (y1 < 0)
        if bad.any():
            g = int(np.argmax(bad))
            # the reference fails here too (empty / negative slice → error inside var()/conv2d)
            raise ValueError("glyph %d of image %d: window [%d,%d) is empty or outside the %d-wide feature map"
                             % (int(c_idx[g]), int(g_img[g]), int(x1[g]), int(x2[g]), feat_w))
        self.G = int(g_img.shape[0])
        t = lambda v: torch.from_numpy(np.ascontiguousarray(v, dtype=np.int32)).to(device)
        self.g_img, self.g_x1, self.g_y1, self.g_w, self.g_start = t(g_img), t(x1), t(y1), t(gw), t(g_start)

    def copy_into(self, static):
        """overwrite the (same-shaped) device tables of ``static`` with these — the HIP-graph path keeps its tables at fixed
        addresses and refreshes their contents before every replay"""
        if static.G != self.G or static.g_start.shape != self.g_start.shape:
            raise ValueError("glyph tables of a different shape (%d vs %d glyphs)" % (self.G, static.G))
        for name in ("g_img", "g_x1", "g_y1", "g_w", "g_start"):
            getattr(static, name).copy_(getattr(self, name), non_blocking=True)
