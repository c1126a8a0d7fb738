"""One-time weight folding / repacking (host plumbing, runs once per (device, precision) after a
``load_state_dict`` / ``.to()``): spectral-norm sigma fold, EqualLinear / ModulatedConv scale fold,
OIHW fp32 → [O][KH][KW][I] in the compute dtype with channel padding, demodulation table.

Nothing here is on the per-forward hot path; it replaces work the reference redoes *every* forward
(spectral-norm ``W/σ`` — 211 ``addmv_`` + 275 ``div`` per SR forward, SURVEY.md §2a K18 — and the per-sample
weight modulation, models/networks.py:284-290).
"""
import math
import os

import torch


def default_precision():
    p = os.environ.get("MARCONET_PRECISION", "fp32").lower()
    if p not in ("fp32", "fp16"):
        raise ValueError("MARCONET_PRECISION must be fp32 or fp16, got %r" % p)
    return p


def torch_dtype(precision):
    return torch.float32 if precision == "fp32" else torch.float16


def _round_up(v, m):
    return (v + m - 1) // m * m


def pack_conv_weight(w, dtype, cin_mult=8, cout_mult=4):
    """w [O,I,KH,KW] fp32 → contiguous [O_pad,KH,KW,I_pad] in ``dtype`` (zero padded)."""
    o, i, kh, kw = w.shape
    op, ip = _round_up(o, cout_mult), _round_up(i, cin_mult)
    if op != o or ip != i:
        wp = torch.zeros((op, ip, kh, kw), dtype=w.dtype, device=w.device)
        wp[:o, :i] = w
        w = wp
    return w.permute(0, 2, 3, 1).contiguous().to(dtype)


def pack_vec(b, n_pad=None):
    b = b.detach().reshape(-1).float()
    if n_pad is not None and n_pad != b.numel():
        bp = torch.zeros((n_pad,), dtype=torch.float32, device=b.device)
        bp[: b.numel()] = b
        b = bp
    return b.contiguous()


def sn_fold(weight_orig, u, v):
    """eval-mode old-style spectral norm (models/networks.py:14): W_orig / (uᵀ (W_mat v)); no power iteration."""
    wm = weight_orig.detach().reshape(weight_orig.shape[0], -1)
    sigma = torch.dot(u.detach(), torch.mv(wm, v.detach()))
    return weight_orig.detach() / sigma


class PackCache:
    """Caches packed tensors per (precision); invalidated when any parameter/buffer storage or version changes."""

    def __init__(self):
        self._sig = None
        self._store = {}

    @staticmethod
    def signature(module):
        return tuple((t.data_ptr(), t._version, t.device) for t in list(module.parameters()) + list(module.buffers()))

    def get(self, module, precision, builder):
        sig = self.signature(module)
        if sig != self._sig:
            self._store = {}
            self._sig = sig
        if precision not in self._store:
            with torch.no_grad():
                self._store[precision] = builder(torch_dtype(precision))
        return self._store[precision]


def posemb_sincos_1x64(device):
    """K16: models/textvit_arch.py:170-181 for the fixed 1x64 token grid, evaluated once on the host in fp32
    with the same op sequence as the reference, then uploaded."""
    dim, temperature = 512, 10000
    y, x = torch.meshgrid(torch.arange(1), torch.arange(64), indexing="ij")
    omega = torch.arange(dim // 4) / (dim // 4 - 1)
    omega = 1.0 / (temperature ** omega)
    y = y.flatten()[:, None] * omega[None, :]
    x = x.flatten()[:, None] * omega[None, :]
    pe = torch.cat((x.sin(), x.cos(), y.sin(), y.cos()), dim=1).type(torch.float32)
    return pe.contiguous().to(device)


def equal_linear_scale(in_channels, lr_mul):
    return (1 / math.sqrt(in_channels)) * lr_mul
