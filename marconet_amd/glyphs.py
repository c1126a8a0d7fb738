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
    """int32 device tables for one scale: g_img[G], g_x1[G], g_y1[G], g_w[G], g_start[B+1]."""

    def __init__(self, locs_host, counts, feat_w, half, device):
        g_img, g_x1, g_y1, g_w, g_start = [], [], [], [], [0]
        for b, n in enumerate(counts):
            if 2 * n > locs_host.shape[1]:
                raise IndexError("locs has %d entries for image %d but %d glyph priors were given"
                                 % (locs_host.shape[1], b, n))
            for c in range(n):
                x1, gw, y1 = window(locs_host[b, 2 * c], feat_w, half)
                if gw <= 0 or gw > 2 * half or y1 < 0:
                    # the reference fails here too (empty / negative slice → error inside var()/conv2d)
                    raise ValueError("glyph %d of image %d: window [%d,%d) is empty or outside the %d-wide feature map"
                                     % (c, b, x1, x1 + gw, feat_w))
                g_img.append(b); g_x1.append(x1); g_y1.append(y1); g_w.append(gw)
            g_start.append(len(g_img))
        self.G = len(g_img)
        t = lambda v: torch.tensor(v, dtype=torch.int32, device=device)
        self.g_img, self.g_x1, self.g_y1, self.g_w, self.g_start = t(g_img), t(g_x1), t(g_y1), t(g_w), t(g_start)
