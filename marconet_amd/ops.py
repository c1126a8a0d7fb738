"""Tensor-level wrappers over the C-ABI.  PyTorch is plumbing only: it owns the device memory
(caching allocator) and the current HIP stream; every arithmetic op below is a hand-written gfx950 kernel.

Activations are NHWC torch tensors of shape [N,H,W,C] and dtype float32, float16 or — the split-half storage of the fp16x3
mode (MNET_F16X2) — packing.SPLIT_DTYPE (a 4-byte tag: a (hi, lo) pair of halves per logical element, C % 32 == 0).
"""
import ctypes
import functools
import os

import torch

from . import _lib
from ._lib import (ACT_GELU, ACT_LRELU, ACT_LRELU_SQRT2, ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH, MNET_F16,
                   MNET_F16M, MNET_F16X2, MNET_F32, ConvDesc)
from .packing import MX_DTYPE, SPLIT_DTYPE, is_split, mx_weight_rows, new_tensor, tag, untag

__all__ = ["conv2d", "linear", "nchw_to_nhwc", "nhwc_to_nchw", "upsample2x", "affine_act", "groupnorm_affine",
           "adain_crop_concat", "adain_crop_concat_gn", "glyph_scatter_affine", "layernorm", "token_mix", "attention", "pixelnorm",
           "embed_gather", "demod", "argmax_rows", "convert", "fused_bias_act", "sr_postprocess", "conv3x3_rgb", "torgb", "stats",
           "pack_weights", "pack_wsq", "gather_rows", "style_rows", "nonfinite_flag", "gn_partial_buffer", "can_emit_gn_partial", "groupnorm_affine_from_partial",
           "ACT_NONE", "ACT_RELU", "ACT_LRELU", "ACT_LRELU_SQRT2", "ACT_TANH", "ACT_GELU", "ACT_SIGMOID"]


def _plumbing(fn):
    """kernel wrappers read shapes / dtypes / pointers of blocked-storage tensors many times per launch: inside a wrapper the
    BlockedTensor guard (packing.BlockedTensor.__torch_function__, ~2.5 us per access) is switched off; outputs are tagged again"""
    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        with torch._C.DisableTorchFunctionSubclass():
            return fn(*args, **kwargs)
    return wrapper


def _dt(t):
    if t.dtype == torch.float32:
        return MNET_F32
    if t.dtype == torch.float16:
        return MNET_F16
    if t.dtype == SPLIT_DTYPE:          # split half (fp16x3 mode): (hi, lo) per logical element, tagged as complex32
        return MNET_F16X2
    if t.dtype == MX_DTYPE:             # fp16+8 (fp16x2 mode): hi half + e4m3 lo byte + block scale, tagged as uint32
        return MNET_F16M
    raise TypeError("marconet_amd: unsupported dtype %s" % t.dtype)


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_cuda(*ts):
    """every tensor of a launch must be contiguous and live on the CURRENT HIP device: the C side launches on the current
    device, on that device's current stream (the module forwards enter ``torch.cuda.device(input.device)`` first)"""
    cur = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("marconet_amd: tensors must live on a HIP device (got %s); there is no CPU path" % t.device)
        if cur is None:
            cur = torch.cuda.current_device()
        if t.device.index != cur:
            raise RuntimeError("marconet_amd: tensor on %s but the current device is cuda:%d — enter torch.cuda.device(...) "
                               "(the nn.Module forwards and MarconetPipeline do)" % (t.device, cur))
        if not t.is_contiguous():
            raise RuntimeError("marconet_amd: non-contiguous tensor passed to a kernel wrapper")


def _raw(t):
    """split-half tensors are moved by PyTorch as plain halves (twice the channels): cat / index_select over the outer dimension
    never depend on the (experimental) complex32 support of an operator"""
    return untag(t).view(torch.float16) if is_split(t.dtype) else t


@_plumbing
def cat_rows(parts):
    """concatenation of NHWC tensors along dim 0 (any storage dtype)"""
    if len(parts) == 1:
        return parts[0]
    out = torch.cat([_raw(p) for p in parts], dim=0)
    return tag(out.view(parts[0].dtype)) if is_split(parts[0].dtype) else out


@_plumbing
def take_rows(t, idx):
    """t[idx] along dim 0 (any storage dtype)"""
    out = _raw(t).index_select(0, idx)
    return tag(out.view(t.dtype)) if is_split(t.dtype) else out


def on_device(t):
    """context manager: the HIP device of ``t`` becomes the current device (launches and the stream lookup follow the current
    device); a CPU tensor raises — there is no CPU path"""
    if not t.is_cuda:
        raise RuntimeError("marconet_amd: tensors must live on a HIP device (got %s); there is no CPU path" % t.device)
    return torch.cuda.device(t.device)


class _Stats:
    """Optional accounting for bench.py: algorithmic FLOPs and per-launch HIP-event timing of the conv kernel."""

    def __init__(self):
        self.enabled = False
        self.timing = False
        self.reset()

    def reset(self):
        self.conv_flops = 0.0
        self.conv_launches = 0
        self.events = []       # (start, end, flops, dtype, kernel id from mnet_conv2d_plan)
        self.tail_events = []  # HBM-bound kernels: (start, end, algorithmic bytes — every tensor of the launch once —, name)

    def conv_time_ms(self):
        return sum(ev[0].elapsed_time(ev[1]) for ev in self.events)

    def tail(self, name, tensors, launch):
        """run ``launch()``; in an instrumented pass (bench.py's roofline pass) bracket it with HIP events on the launch stream and
        book the storage bytes of ``tensors`` (inputs and outputs, each once) under ``name``"""
        if not (self.enabled and self.timing):
            return launch()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = launch()
        e.record()
        self.tail_events.append((s, e, float(sum(t.numel() * t.element_size() for t in tensors if t is not None)), name))
        return r


stats = _Stats()


@_plumbing
def conv2d(x0, wgt, cout, kh=1, kw=1, stride=(1, 1), pad=(0, 0), x1=None, in_scale=None, in_shift=None,
           in_swish=False, valid_w=None, out_scale=None, bias=None, residual=None, res_mod=0, act=ACT_NONE,
           post_scale=None, out=None, algo=0, splitk=0, x1_center=False, gn_partial=None):
    """mnet_conv2d_nhwc(_ex).  x0 [N,H,W,C0] (+ optional x1 [N,H,W,C1]); wgt packed [cout,kh,kw,C0+C1] same dtype.
    ``splitk`` > 0: mnet_conv2d_splitk with that many K-slices (fp32 filter == stride convs over <= 512 output pixels).
    ``x1_center``: x1 enters through the filter's centre tap only (MNET_CONV_ALGO_FLAG_X1_CENTER: a 1x1 skip conv as extra K).
    ``gn_partial``: fp32 [n*ho*wo/32, cout/32, 2] buffer (``gn_partial_buffer``) the epilogue fills with the GroupNorm partial sums of the output
    (fp16+8 launches on the LDS-DMA / strip kernels only: MarconetHipError otherwise, nothing enqueued)."""
    lib = _lib.load()
    _need_cuda(x0, x1, wgt, in_scale, in_shift, valid_w, out_scale, bias, residual, post_scale, out)
    n, h, w, c0 = x0.shape
    c1 = 0 if x1 is None else x1.shape[3]
    need = cout * kh * kw * (c0 + c1)
    if x0.dtype == MX_DTYPE:            # + the per-output-channel scale bytes behind the weight rows
        need = mx_weight_rows(cout, kh, kw, c0 + c1) * kh * kw * (c0 + c1)
    if wgt.dtype != x0.dtype or wgt.numel() != need:
        raise RuntimeError("conv2d: weight dtype/shape mismatch (%s %s vs cout=%d k=%dx%d cin=%d)"
                           % (wgt.dtype, tuple(wgt.shape), cout, kh, kw, c0 + c1))
    ho = (h + 2 * pad[0] - kh) // stride[0] + 1
    wo = (w + 2 * pad[1] - kw) // stride[1] + 1
    if out is None:
        out = new_tensor((n, ho, wo, cout), x0.dtype, x0.device)
    elif tuple(out.shape) != (n, ho, wo, cout) or out.dtype != x0.dtype:
        raise RuntimeError("conv2d: out is %s %s, expected %s %s" % (tuple(out.shape), out.dtype, (n, ho, wo, cout), x0.dtype))
    d = ConvDesc()
    d.dtype = _dt(x0)
    d.x0, d.c0 = x0.data_ptr(), c0
    d.x1, d.c1 = (None if x1 is None else x1.data_ptr()), c1
    d.n, d.h, d.w = n, h, w
    d.wgt = wgt.data_ptr()
    d.cout, d.kh, d.kw = cout, kh, kw
    d.stride_h, d.stride_w, d.pad_h, d.pad_w = stride[0], stride[1], pad[0], pad[1]
    d.ho, d.wo = ho, wo
    d.in_scale = None if in_scale is None else in_scale.data_ptr()
    d.in_shift = None if in_shift is None else in_shift.data_ptr()
    d.in_swish = 1 if in_swish else 0
    d.valid_w = None if valid_w is None else valid_w.data_ptr()
    d.out_scale = None if out_scale is None else out_scale.data_ptr()
    d.bias = None if bias is None else bias.data_ptr()
    d.residual = None if residual is None else residual.data_ptr()
    d.res_mod = res_mod
    d.act = act
    d.post_scale = None if post_scale is None else post_scale.data_ptr()
    d.y = out.data_ptr()
    d.gn_partial = None
    if gn_partial is not None:
        _need_cuda(gn_partial)
        if gn_partial.dtype != torch.float32 or gn_partial.numel() != (n * ho * wo // 32) * (cout // 32) * 2:
            raise RuntimeError("conv2d: gn_partial must be fp32 [n*ho*wo/32, cout/32, 2]")
        d.gn_partial = gn_partial.data_ptr()
    for t, nm in ((in_scale, "in_scale"), (in_shift, "in_shift"), (out_scale, "out_scale"), (bias, "bias"),
                  (post_scale, "post_scale")):
        if t is not None and t.dtype != torch.float32:
            raise TypeError("conv2d: %s must be float32" % nm)
    if valid_w is not None and valid_w.dtype != torch.int32:
        raise TypeError("conv2d: valid_w must be int32")
    if residual is not None and residual.dtype != x0.dtype:
        raise TypeError("conv2d: residual dtype mismatch")
    if x1_center:
        if x1 is None:
            raise RuntimeError("conv2d: x1_center needs x1")
        algo |= _lib.ALGO_FLAG_X1_CENTER
    if splitk:
        ws = torch.empty((splitk * n * ho * wo * cout,), dtype=torch.float32, device=x0.device)
        if stats.enabled:
            stats.conv_flops += lib.mnet_conv2d_flops(ctypes.byref(d))
            stats.conv_launches += 1
        _lib.check(lib.mnet_conv2d_splitk(ctypes.byref(d), splitk, _p(ws), _stream()), "mnet_conv2d_splitk")
        return out
    if stats.enabled:
        fl = lib.mnet_conv2d_flops(ctypes.byref(d))
        if x1_center:                       # the second source is multiplied at one tap, not kh * kw
            fl -= 2.0 * n * ho * wo * cout * (kh * kw - 1) * c1
        stats.conv_flops += fl
        stats.conv_launches += 1
        if stats.timing:
            s = torch.cuda.Event(enable_timing=True)
            e = torch.cuda.Event(enable_timing=True)
            s.record()
            _lib.check(lib.mnet_conv2d_nhwc_ex(ctypes.byref(d), algo, _stream()), "mnet_conv2d_nhwc")
            e.record()
            stats.events.append((s, e, fl, d.dtype, lib.mnet_conv2d_plan(ctypes.byref(d), algo)))
            return out
    _lib.check(lib.mnet_conv2d_nhwc_ex(ctypes.byref(d), algo, _stream()), "mnet_conv2d_nhwc")
    return out


def linear(x, wgt, out_features, bias=None, act=ACT_NONE, residual=None, res_mod=0):
    """nn.Linear as a 1x1 conv over a [1,1,M,K] map: x [M,K] → [M,out_features]."""
    m, k = x.shape
    r = None if residual is None else residual.reshape(1, 1, -1, out_features)
    y = conv2d(x.reshape(1, 1, m, k), wgt, out_features, bias=bias, act=act, residual=r, res_mod=res_mod)
    return y.reshape(m, out_features)


@_plumbing
def nchw_to_nhwc(src, dtype, c_ld=None):
    lib = _lib.load()
    _need_cuda(src)
    if src.dtype != torch.float32:
        raise TypeError("nchw_to_nhwc: fp32 NCHW input expected (the reference's tensors are fp32)")
    n, c, h, w = src.shape
    c_ld = c_ld or c
    dst = new_tensor((n, h, w, c_ld), dtype, src.device)
    stats.tail("layout", (src, dst), lambda: _lib.check(lib.mnet_nchw_to_nhwc(_p(src), _p(dst), _dt(dst), n, c, h, w, c_ld, _stream()), "mnet_nchw_to_nhwc"))
    return dst


@_plumbing
def nhwc_to_nchw(src, c=None):
    lib = _lib.load()
    _need_cuda(src)
    n, h, w, c_ld = src.shape
    c = c or c_ld
    dst = torch.empty((n, c, h, w), dtype=torch.float32, device=src.device)
    stats.tail("layout", (src, dst), lambda: _lib.check(lib.mnet_nhwc_to_nchw(_p(src), _dt(src), _p(dst), n, c, h, w, c_ld, _stream()), "mnet_nhwc_to_nchw"))
    return dst


@_plumbing
def upsample2x(src, scale=None, out_dtype=None):
    """bilinear x2 (align_corners=False); optional per-(n,c) fp32 multiplier fused into the store; ``out_dtype``: torch.float16 for a
    split-half / fp16+8 source (the conversion rides in the store: no separate convert pass), default the source's storage type"""
    lib = _lib.load()
    _need_cuda(src, scale)
    n, h, w, c = src.shape
    dst = new_tensor((n, 2 * h, 2 * w, c), out_dtype or src.dtype, src.device)
    stats.tail("upsample2x", (src, dst), lambda: _lib.check(lib.mnet_upsample2x_convert_nhwc(_p(src), _dt(src), _p(dst), _dt(dst), n, h, w, c, _p(scale), _stream()),
                                                            "mnet_upsample2x_convert_nhwc"))
    return dst


@_plumbing
def affine_act(x, scale, shift=None, swish=False, out=None):
    """y = f(x*scale[n,c] + shift[n,c]) elementwise over NHWC x [N,H,W,C] (GroupNorm apply + swish)."""
    lib = _lib.load()
    _need_cuda(x, scale, shift, out)
    n, h, w, c = x.shape
    y = tag(torch.empty_like(x)) if out is None else out
    stats.tail("groupnorm_apply", (x, y), lambda: _lib.check(lib.mnet_affine_act_nhwc(_p(x), _p(y), _dt(x), n, h * w, c, _p(scale), _p(shift), 1 if swish else 0,
                                                                                   _stream()), "mnet_affine_act_nhwc"))
    return y


@_plumbing
def groupnorm_affine(x, gamma, beta, eps=1e-6, valid_w=None):
    """→ (scale [N,C], shift [N,C]) fp32 for conv2d(in_scale=, in_shift=, in_swish=True)."""
    lib = _lib.load()
    _need_cuda(x, gamma, beta, valid_w)
    n, h, w, c = x.shape
    slices = max(1, min(128, (h * w) // 512))      # a function of the map size only: the fp64 fold order never depends on the batch
    partial = torch.empty((n * slices * (c // 32) * 2,), dtype=torch.float64, device=x.device)
    scale = torch.empty((n, c), dtype=torch.float32, device=x.device)
    shift = torch.empty((n, c), dtype=torch.float32, device=x.device)
    stats.tail("groupnorm_stats", (x,), lambda: _lib.check(lib.mnet_groupnorm_affine(_p(x), _dt(x), n, h, w, c, _p(valid_w), _p(gamma), _p(beta), eps,
                                                                                  _p(partial), slices, _p(scale), _p(shift), _stream()), "mnet_groupnorm_affine"))
    return scale, shift


def gn_partial_buffer(n, h, w, c, device):
    """buffer for conv2d(gn_partial=...): per (32-pixel fragment, 32-channel group) sum and sum of squares of the conv's output"""
    return torch.empty((n * h * w // 32, c // 32, 2), dtype=torch.float32, device=device)


_PLAN_CACHE = {}


def conv_plan(x0, cout, kh=1, kw=1, stride=(1, 1), pad=(0, 0), x1=None, algo=0):
    """mnet_conv2d_plan for the launch geometry of ``conv2d(x0, …, x1=…)``: the kernel ``algo`` resolves to (_lib.ALGO_REG_STAGED, ALGO_SKINNY,
    ALGO_DMA_CFG0 + id, ALGO_STRIP_CFG0 + id, ALGO_DMA_CFG16 + id), or a negative MNET_E_* when the planner refuses it.  Nothing is launched; the
    answer depends on geometry and storage type only and is cached per (dtype, shape, filter, algo) — the callers that choose between two forms
    of a layer ask this instead of launching and parsing an error message."""
    n, h, w, c0 = x0.shape
    c1 = 0 if x1 is None else x1.shape[3]
    key = (_dt(x0), n, h, w, c0, c1, cout, kh, kw, tuple(stride), tuple(pad), algo)
    k = _PLAN_CACHE.get(key)
    if k is None:
        lib = _lib.load()
        d = ConvDesc()
        d.dtype = key[0]
        d.x0, d.c0 = x0.data_ptr(), c0
        d.x1, d.c1 = (None if x1 is None else x1.data_ptr()), c1
        d.n, d.h, d.w = n, h, w
        d.wgt = d.y = x0.data_ptr()            # (the planner checks presence and alignment of the pointers, nothing behind them)
        d.cout, d.kh, d.kw = cout, kh, kw
        d.stride_h, d.stride_w, d.pad_h, d.pad_w = stride[0], stride[1], pad[0], pad[1]
        d.ho, d.wo = (h + 2 * pad[0] - kh) // stride[0] + 1, (w + 2 * pad[1] - kw) // stride[1] + 1
        k = _PLAN_CACHE[key] = int(lib.mnet_conv2d_plan(ctypes.byref(d), algo))
    return k


def plan_is_lds_dma(k):
    """a planner answer that names one of the LDS-DMA tile configurations (not the strip kernel, not the register-staged / skinny ones)"""
    return _lib.ALGO_DMA_CFG0 <= k < _lib.ALGO_STRIP_CFG0 or k >= _lib.ALGO_DMA_CFG16


def can_emit_gn_partial(x0, x1, cout, stride, ho, wo):
    """the launches whose epilogue can write GroupNorm partial sums (mnet_conv_desc.gn_partial): fp16+8 storage, and a launch the planner gives to the
    LDS-DMA / strip kernels (3x3 'same' geometry of this network's GroupNorm producers)"""
    c = x0.shape[3] + (0 if x1 is None else x1.shape[3])
    if not (x0.dtype == MX_DTYPE and tuple(stride) == (1, 1) and cout >= 64 and cout % 32 == 0 and c % 32 == 0 and x0.shape[3] % 32 == 0
            and (ho * wo) % 32 == 0 and not _NO_EPILOGUE_GN):
        return False
    return conv_plan(x0, cout, 3, 3, stride, (1, 1), x1=x1) >= _lib.ALGO_DMA_CFG0


def groupnorm_affine_from_partial(partial, n, h, w, c, gamma, beta, eps=1e-6, valid_w=None):
    """→ (scale [N,C], shift [N,C]) fp32 from the partial sums a conv epilogue wrote (no pass over the map)"""
    lib = _lib.load()
    _need_cuda(partial, gamma, beta, valid_w)
    scale = torch.empty((n, c), dtype=torch.float32, device=partial.device)
    shift = torch.empty((n, c), dtype=torch.float32, device=partial.device)
    _lib.check(lib.mnet_groupnorm_affine_from_partial(_p(partial), n, h, w, c, _p(valid_w), _p(gamma), _p(beta), eps, _p(scale), _p(shift), _stream()),
               "mnet_groupnorm_affine_from_partial")
    return scale, shift


@_plumbing
def adain_crop_concat(prior, feat, g_img, g_x1, g_y1, g_w):
    lib = _lib.load()
    _need_cuda(prior, feat, g_img, g_x1, g_y1, g_w)
    G, S, S2, C = prior.shape
    B, FH, FW, FC = feat.shape
    if S != S2 or FH != S or FC != C or prior.dtype != feat.dtype:
        raise RuntimeError("adain_crop_concat: shape mismatch prior %s feat %s" % (tuple(prior.shape), tuple(feat.shape)))
    out = new_tensor((G, S, S, 2 * C), prior.dtype, prior.device)
    _lib.check(lib.mnet_adain_crop_concat(_p(prior), _p(feat), _p(out), _dt(prior), G, S, C, FW, _p(g_img), _p(g_x1),
                                          _p(g_y1), _p(g_w), _stream()), "mnet_adain_crop_concat")
    return out


_NO_EPILOGUE_GN = os.environ.get("MNET_NO_EPILOGUE_GN", "0") == "1"      # A/B knob: GroupNorm statistics by their own pass over the map (round 4)
ADAIN_SPLIT_BELOW = 256      # glyphs per launch below which the three-launch (16 workgroups per glyph) form is used
_ADAIN_SPLIT = {"0": False, "1": True}.get(os.environ.get("MNET_ADAIN_SPLIT", ""))     # A/B knob


@_plumbing
def adain_crop_concat_gn(prior, feat, g_img, g_x1, g_y1, g_w, gamma, beta, eps=1e-6, split=None):
    """adain_crop_concat + the GroupNorm affine of its output (closed form from the AdaIN statistics) → (out, scale, shift).
    ``split``: None = by glyph count; the two forms agree up to the association of the fp64 statistic sums."""
    lib = _lib.load()
    _need_cuda(prior, feat, g_img, g_x1, g_y1, g_w, gamma, beta)
    G, S, S2, C = prior.shape
    B, FH, FW, FC = feat.shape
    if S != S2 or FH != S or FC != C or prior.dtype != feat.dtype or gamma.numel() != 2 * C:
        raise RuntimeError("adain_crop_concat_gn: shape mismatch prior %s feat %s" % (tuple(prior.shape), tuple(feat.shape)))
    out = new_tensor((G, S, S, 2 * C), prior.dtype, prior.device)
    scale = torch.empty((G, 2 * C), dtype=torch.float32, device=prior.device)
    shift = torch.empty((G, 2 * C), dtype=torch.float32, device=prior.device)
    if split is None:
        split = _ADAIN_SPLIT if _ADAIN_SPLIT is not None else G < ADAIN_SPLIT_BELOW
    if split:       # few glyphs (a strip at a time): spread each glyph over 16 workgroups instead of one
        slices = 16
        partial = torch.empty((G * slices * C * 4,), dtype=torch.float64, device=prior.device)
        stat = torch.empty((G, 4, C), dtype=torch.float32, device=prior.device)
        stats.tail("adain", (prior, prior, out), lambda: _lib.check(lib.mnet_adain_crop_concat_split(
            _p(prior), _p(feat), _p(out), _dt(prior), G, S, C, FW, _p(g_img), _p(g_x1), _p(g_y1), _p(g_w), _p(gamma), _p(beta), eps, _p(scale), _p(shift),
            _p(partial), _p(stat), slices, _stream()), "mnet_adain_crop_concat_split"))
        return out, scale, shift
    # bytes: the prior, a window of feat of the prior's size, the concatenated output
    stats.tail("adain", (prior, prior, out), lambda: _lib.check(lib.mnet_adain_crop_concat_gn(
        _p(prior), _p(feat), _p(out), _dt(prior), G, S, C, FW, _p(g_img), _p(g_x1), _p(g_y1), _p(g_w), _p(gamma), _p(beta), eps, _p(scale), _p(shift), _stream()),
        "mnet_adain_crop_concat_gn"))
    return out, scale, shift


@_plumbing
def glyph_scatter_affine(feat, scale, shift, g_start, g_x1, g_w):
    lib = _lib.load()
    _need_cuda(feat, scale, shift, g_start, g_x1, g_w)
    B, S, FW, C = feat.shape
    out = tag(torch.empty_like(feat))
    stats.tail("glyph_scatter", (feat, scale, shift, out), lambda: _lib.check(lib.mnet_glyph_scatter_affine(
        _p(feat), _p(scale), _p(shift), _p(out), _dt(feat), B, S, C, FW, _p(g_start), _p(g_x1), _p(g_w), _stream()), "mnet_glyph_scatter_affine"))
    return out


def layernorm(x, gamma, beta, eps=1e-5):
    lib = _lib.load()
    _need_cuda(x, gamma, beta)
    rows, d = x.shape
    y = tag(torch.empty_like(x))
    _lib.check(lib.mnet_layernorm(_p(x), _p(gamma), _p(beta), _p(y), rows, d, eps, _stream()), "mnet_layernorm")
    return y


def token_mix(x, ln_g, ln_b, wgt, bias, eps=1e-5):
    lib = _lib.load()
    _need_cuda(x, ln_g, ln_b, wgt, bias)
    B, T, D = x.shape
    J = wgt.shape[0]
    y = torch.empty((B, J, D), dtype=torch.float32, device=x.device)
    _lib.check(lib.mnet_token_mix(_p(x), _p(ln_g), _p(ln_b), _p(wgt), _p(bias), _p(y), B, T, D, J, eps, _stream()),
               "mnet_token_mix")
    return y


def attention(qkv, B, N, H, scale):
    lib = _lib.load()
    _need_cuda(qkv)
    out = torch.empty((B * N, H * 64), dtype=torch.float32, device=qkv.device)
    _lib.check(lib.mnet_attention(_p(qkv), _p(out), B, N, H, scale, _stream()), "mnet_attention")
    return out


def pixelnorm(x):
    lib = _lib.load()
    _need_cuda(x)
    y = tag(torch.empty_like(x))
    _lib.check(lib.mnet_pixelnorm(_p(x), _p(y), x.shape[0], x.shape[1], _stream()), "mnet_pixelnorm")
    return y


@_plumbing
def embed_gather(emb, labels, dtype, num_classes, scale=None):
    """SelectText gather; ``scale`` (fp32 [N, C], optional): per-(sample, channel) factor applied before the storage rounding"""
    lib = _lib.load()
    _need_cuda(emb, labels, scale)
    N, nc = labels.shape
    C = emb.shape[1]
    if scale is not None and (tuple(scale.shape) != (N, C) or scale.dtype != torch.float32 or not scale.is_contiguous()):
        raise ValueError("embed_gather: scale must be a contiguous fp32 [N, C] tensor")
    out = new_tensor((N, 4, 4 * nc, C), dtype, emb.device)
    _lib.check(lib.mnet_embed_gather_scaled(_p(emb), _p(labels), _p(scale), _p(out), _dt(out), N, nc, C, num_classes, _stream()),
               "mnet_embed_gather_scaled")
    return out


def demod(style, wsq_t, eps_scale=None):
    """rsqrt(Σ_i style² wsq_t + 1e-8·eps_scale[n]) — eps_scale (fp32 [N], 4^-e) for rows normalised by ``style_rows``"""
    lib = _lib.load()
    _need_cuda(style, wsq_t, eps_scale)
    N, cin = style.shape
    cout = wsq_t.shape[1]
    out = torch.empty((N, cout), dtype=torch.float32, device=style.device)
    _lib.check(lib.mnet_demod_scaled(_p(style), _p(wsq_t), _p(out), N, cin, cout, _p(eps_scale), _stream()), "mnet_demod_scaled")
    return out


def style_rows(src, col0, ncols, idx=None, bcast=0):
    """mnet_style_rows: window + row gather of the modulation GEMM's output with every row divided by 2^e (largest magnitude in
    [0.5, 1)) → (rows [R,ncols], eps_scale [R] = 4^-e, scale_b [R,bcast] = 2^e or None)"""
    lib = _lib.load()
    _need_cuda(src, idx)
    if src.dtype != torch.float32 or src.dim() != 2 or (idx is not None and idx.dtype != torch.int64):
        raise TypeError("style_rows: fp32 [rows, ld] source and int64 indices expected")
    rows = src.shape[0] if idx is None else idx.shape[0]
    dst = torch.empty((rows, ncols), dtype=torch.float32, device=src.device)
    eps = torch.empty((rows,), dtype=torch.float32, device=src.device)
    sb = torch.empty((rows, bcast), dtype=torch.float32, device=src.device) if bcast else None
    if rows:
        _lib.check(lib.mnet_style_rows(_p(src), src.shape[0], src.shape[1], col0, ncols, _p(idx), rows, _p(dst), _p(eps), _p(sb), bcast,
                                       _stream()), "mnet_style_rows")
    return dst, eps, sb


def argmax_rows(x):
    lib = _lib.load()
    _need_cuda(x)
    rows, d = x.shape
    idx = torch.empty((rows,), dtype=torch.int64, device=x.device)
    _lib.check(lib.mnet_argmax_rows(_p(x), _p(idx), rows, d, _stream()), "mnet_argmax_rows")
    return idx


@_plumbing
def convert(x, dtype):
    if x.dtype == dtype:
        return x
    lib = _lib.load()
    _need_cuda(x)
    y = new_tensor(x.shape, dtype, x.device)
    stats.tail("convert", (x, y), lambda: _lib.check(lib.mnet_convert(_p(x), _dt(x), _p(y), _dt(y), x.numel(), _stream()), "mnet_convert"))
    return y


def fused_bias_act(x, bias, negative_slope=0.2, scale=2 ** 0.5):
    """basicsr.ops.fused_act.fused_leaky_relu semantics on a contiguous fp32 [N,C,...] tensor."""
    lib = _lib.load()
    _need_cuda(x, bias)
    if x.dtype != torch.float32:
        raise TypeError("fused_bias_act: fp32 expected")
    C = x.shape[1]
    inner = 1
    for s in x.shape[2:]:
        inner *= s
    y = tag(torch.empty_like(x))
    _lib.check(lib.mnet_fused_bias_act(_p(x), _p(bias), _p(y), x.numel(), C, inner, negative_slope, scale, _stream()),
               "mnet_fused_bias_act")
    return y


@_plumbing
def sr_postprocess(y_nhwc, u8=True):
    """test_sr.py:198-200 on the NHWC SR tensor [B,H,W,c_ld] (RGB in channels 0..2) → [B,H,W,3] BGR, uint8 (cv2.imwrite's
    rounding) or float32 (the array the script passes to cv2)."""
    lib = _lib.load()
    _need_cuda(y_nhwc)
    b, h, w, c_ld = y_nhwc.shape
    out = torch.empty((b, h, w, 3), dtype=torch.uint8 if u8 else torch.float32, device=y_nhwc.device)
    _lib.check(lib.mnet_sr_postprocess(_p(y_nhwc), _dt(y_nhwc), _p(out), 1 if u8 else 0, b * h * w, c_ld, _stream()),
               "mnet_sr_postprocess")
    return out


def nonfinite_flag(x, out=None):
    """mnet_nonfinite_flag: int32 [1] device tensor (``out``: a one-element int32 view to write into), 1 iff the fp32 / fp16 tensor
    ``x`` holds an inf or NaN (no synchronisation here)"""
    lib = _lib.load()
    _need_cuda(x, out)
    flag = torch.empty((1,), dtype=torch.int32, device=x.device) if out is None else out
    if flag.dtype != torch.int32 or flag.numel() != 1:
        raise TypeError("nonfinite_flag: out must be one int32 element")
    _lib.check(lib.mnet_nonfinite_flag(_p(x), _dt(x), x.numel(), _p(flag), _stream()), "mnet_nonfinite_flag")
    return flag


@_plumbing
def pack_weights(w, dtype, cout_pad=None, cin_pad=None, scale=1.0, sn_u=None, sn_v=None):
    """mnet_pack_weights: w fp32 [cout,cin,kh,kw] (or [out,in] for a Linear) on the device → [cout_pad,kh,kw,cin_pad] in ``dtype``,
    every element (w / sigma) * scale with sigma = uᵀ(W_mat v) when the spectral-norm vectors are given (models/networks.py:14)."""
    lib = _lib.load()
    w = w.detach()
    if w.dim() == 2:
        w = w.reshape(w.shape[0], w.shape[1], 1, 1)
    w = w.contiguous()
    if w.dtype != torch.float32:
        w = w.float()
    if sn_u is not None:
        sn_u, sn_v = sn_u.detach().float().contiguous(), sn_v.detach().float().contiguous()
    _need_cuda(w, sn_u, sn_v)
    cout, cin, kh, kw = w.shape
    cout_pad, cin_pad = cout_pad or cout, cin_pad or cin
    rows = mx_weight_rows(cout_pad, kh, kw, cin_pad) if dtype == MX_DTYPE else cout_pad
    out = new_tensor((rows, kh, kw, cin_pad), dtype, w.device, zero=dtype == MX_DTYPE)
    ws = torch.empty((cout + 1,), dtype=torch.float64, device=w.device) if sn_u is not None else None
    _lib.check(lib.mnet_pack_weights(_p(w), cout, cin, kh, kw, _p(sn_u), _p(sn_v), float(scale), _dt(out), cout_pad, cin_pad, _p(out),
                                     _p(ws), _stream()), "mnet_pack_weights")
    return out


def pack_wsq(w, scale):
    """mnet_pack_wsq: demodulation table [cin,cout] of a ModulatedConv2d weight [cout,cin,kh,kw] (fp32, device)"""
    lib = _lib.load()
    w = w.detach().float().contiguous()
    _need_cuda(w)
    cout, cin, kh, kw = w.shape
    out = torch.empty((cin, cout), dtype=torch.float32, device=w.device)
    _lib.check(lib.mnet_pack_wsq(_p(w), cout, cin, kh * kw, float(scale), _p(out), _stream()), "mnet_pack_wsq")
    return out


def gather_rows(src, col0=0, ncols=None, idx=None):
    """dst[r, :] = src[idx[r] (or r), col0:col0+ncols] — fp32 [rows, ncols] contiguous (mnet_gather_rows)"""
    lib = _lib.load()
    _need_cuda(src, idx)
    if src.dtype != torch.float32 or src.dim() != 2 or (idx is not None and idx.dtype != torch.int64):
        raise TypeError("gather_rows: fp32 [rows, ld] source and int64 indices expected")
    ld = src.shape[1]
    ncols = ld - col0 if ncols is None else ncols
    rows = src.shape[0] if idx is None else idx.shape[0]
    dst = torch.empty((rows, ncols), dtype=torch.float32, device=src.device)
    if rows:
        _lib.check(lib.mnet_gather_rows(_p(src), src.shape[0], ld, col0, ncols, _p(idx), rows, _p(dst), _stream()), "mnet_gather_rows")
    return dst


@_plumbing
def torgb(x, wgt, style, scale_b, bias, skip=None):
    """mnet_torgb (ToRGB.forward, models/networks.py:313-321): x NHWC [N,H,W,C] any storage dtype; wgt fp32 [3,C]; style fp32 [N,C];
    scale_b fp32 [N] / [N,1] or None; bias fp32 [>=3]; skip fp32 [N,H/2,W/2,4] or None → fp32 [N,H,W,4] (RGB0)"""
    lib = _lib.load()
    _need_cuda(x, wgt, style, scale_b, bias, skip)
    n, h, w, c = x.shape
    if wgt.dtype != torch.float32 or wgt.numel() != 3 * c or style.dtype != torch.float32 or tuple(style.shape) != (n, c):
        raise RuntimeError("torgb: weight [3,C] / style [N,C] fp32 expected")
    if skip is not None and (skip.dtype != torch.float32 or tuple(skip.shape) != (n, h // 2, w // 2, 4)):
        raise RuntimeError("torgb: skip must be fp32 [N,H/2,W/2,4]")
    out = torch.empty((n, h, w, 4), dtype=torch.float32, device=x.device)
    stats.tail("torgb", (x, skip, out), lambda: _lib.check(lib.mnet_torgb(_p(x), _dt(x), n, h, w, c, _p(wgt), _p(style), _p(scale_b), _p(bias), _p(skip), _p(out), _stream()), "mnet_torgb"))
    return out


@_plumbing
def conv3x3_rgb(x, wgt, bias, act=ACT_TANH, nhwc=True, nchw=False):
    """conv_final.6 (+ tanh): x NHWC [N,H,W,64]; wgt [3,3,3,64] same dtype (fp32 for a split-half x); bias fp32 [3] → NHWC
    [N,H,W,8] (same dtype; fp32 for a split-half x) and/or fp32 NCHW [N,3,H,W]; returns (y_nhwc or None, y_nchw or None)."""
    lib = _lib.load()
    _need_cuda(x, wgt, bias)
    n, h, w, c = x.shape
    odt = torch.float32 if is_split(x.dtype) else x.dtype            # split-half / fp16+8 input: fp32 weights, fp32 outputs
    if wgt.dtype != odt or wgt.numel() != 3 * 9 * c:
        raise RuntimeError("conv3x3_rgb: weight dtype/shape mismatch")
    y1 = torch.empty((n, h, w, 8), dtype=odt, device=x.device) if nhwc else None
    y2 = torch.empty((n, 3, h, w), dtype=torch.float32, device=x.device) if nchw else None
    stats.tail("conv3x3_rgb", (x, y1, y2), lambda: _lib.check(lib.mnet_conv3x3_rgb(_p(x), _dt(x), n, h, w, c, _p(wgt), _p(bias), act, _p(y1), _p(y2), _stream()), "mnet_conv3x3_rgb"))
    return y1, y2
