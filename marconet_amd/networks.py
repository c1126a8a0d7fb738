"""MI355X-native ``TextContextEncoderV2`` / ``TSPGAN`` / ``TSPSRNet`` — drop-in for the reference's
models/networks.py: same class names, constructor defaults, parameter/buffer names and shapes
(``load_state_dict(..., strict=True)`` with the reference checkpoints' key set), same ``forward()`` call forms
and return shapes (fp32, NCHW, on the input's device).

Internally nothing of the reference's execution survives: activations are NHWC in the compute dtype
(fp32 "parity" mode or fp16 "throughput" mode, ``set_precision`` / env MARCONET_PRECISION), every conv /
linear is one launch of the MFMA implicit-GEMM kernel with fused prologue/epilogue, weights are folded and
repacked once per load, the modulated conv is applied on the activation side, and TSPSRNet's per-glyph Python
loops become batched kernels over all glyphs of the batch.  The child ``nn.Module``s hold parameters only.
"""
import math
import os

import torch
import torch.nn as nn

from . import _lib, ops
from .glyphs import GlyphTables
from .packing import (PRECISIONS, SPLIT_DTYPE, PackCache, is_split, default_precision, equal_linear_scale, pack_conv_weight,
                      pack_linear_weight, pack_vec, pack_wsq, padded_cout, rgb_pad, torch_dtype)
from .resnet import resnet45stride as resnet45
from .textvit_arch import TextViT as TextEncoder

_STYLE_NORM = not bool(int(os.environ.get("MNET_NO_STYLE_NORM", "0")))     # A/B knob (tests): style rows normalised by a power of two
_FOLD_SKIP = not bool(int(os.environ.get("MNET_NO_FOLD_SKIP", "0")))       # A/B knob: ResTextBlockV2's 1x1 skip conv as extra K of its conv2
_FUSE_IMG_CONVERT = os.environ.get("MNET_NO_FUSE_IMG_CONVERT", "0") != "1"   # image-only levels: the f16 conversion rides in the up-sample's store (A/B knob)
_FUSE_CONV1_MOD = os.environ.get("MNET_NO_FUSE_CONV1_MOD", "0") != "1"      # conv1's style multiply in the SelectText gather (A/B knob)
RGB_PAD = 8      # 3-channel tensors are carried with 8 channels (one 16-byte fp16 chunk); 32 in the split-half mode (rgb_pad)


class _Precision:
    def set_precision(self, precision):
        """'fp32' (parity mode, exact fp32 MFMA), 'fp16x3' (split-half storage, three fp16 MFMA products per multiply:
        meets the same ≤1e-3 / bit-exact-argmax bar at several times the fp32 mode's rate) or 'fp16' (one half per element:
        fastest, ~1e-2 deviation).  Accumulation, statistics and the TextViT are fp32 in every mode."""
        if precision not in PRECISIONS:
            raise ValueError("precision must be one of %s" % ", ".join(PRECISIONS))
        self.precision = precision
        for m in self.children():
            if hasattr(m, "precision"):
                m.precision = precision
        return self


# =====================================================================================================
# 1) TextContextEncoderV2  (models/networks.py:27-45)
# =====================================================================================================
class TextContextEncoderV2(nn.Module, _Precision):
    """LR image → (character logits [B,64,6736], (left,right) locs [B,32], font style w [B,512])."""

    def __init__(self, dim=512, num_classes=6736):
        super().__init__()
        self.resnet = resnet45()
        self.transformer = TextEncoder(num_classes=num_classes, dim=512, max_length=16)
        self.precision = default_precision()
        self.resnet.precision = self.precision

    @staticmethod
    def resnet_precision(precision):
        """"fp16x2" (fp16+8 operands, ~3e-4 on the logits when the ResNet runs in it): the ResNet keeps the three-product split-half
        arithmetic — it is 5 % of the path's FLOPs, its logits decide the character indices (argmax over near ties) and its style
        vector w feeds every modulation of the generator"""
        return "fp16x3" if precision == "fp16x2" else precision

    def set_precision(self, precision):
        _Precision.set_precision(self, precision)
        self.resnet.precision = self.resnet_precision(precision)
        return self

    def forward(self, lq):
        with torch.no_grad(), ops.on_device(lq):
            rp = self.resnet_precision(self.precision)
            self.resnet.precision = rp
            dtype = torch_dtype(rp)
            x = ops.nchw_to_nhwc(lq.contiguous().float(), dtype, c_ld=rgb_pad(dtype))
            feat = self.resnet.forward_nhwc(x)
            feat = ops.convert(feat, torch.float32)          # the ViT always runs in fp32
            return self.transformer.forward_nhwc(feat)


# =====================================================================================================
# 2) TSPGAN  (models/networks.py:51-321) — parameter holders
# =====================================================================================================
class PixelNorm(nn.Module):
    pass


class EqualLinear(nn.Module):
    def __init__(self, in_channels, out_channels, bias=True, bias_init_val=0, lr_mul=1, activation=None):
        super().__init__()
        self.in_channels, self.out_channels, self.lr_mul, self.activation = in_channels, out_channels, lr_mul, activation
        self.scale = equal_linear_scale(in_channels, lr_mul)
        self.weight = nn.Parameter(torch.randn(out_channels, in_channels).div_(lr_mul))
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_channels).fill_(bias_init_val))
        else:
            self.register_parameter("bias", None)


class SelectText(nn.Module):
    def __init__(self, class_num, channel, size=4):
        super().__init__()
        self.size = size
        self.TextEmbeddings = nn.Parameter(torch.randn(class_num, channel, 1, 1))


class ModulatedConv2d(nn.Module):
    def __init__(self, in_channel, out_channel, kernel_size, style_dim, demodulate=True, upsample=False,
                 downsample=False, blur_kernel=(1, 3, 3, 1)):
        super().__init__()
        self.eps = 1e-8
        self.kernel_size, self.in_channel, self.out_channel = kernel_size, in_channel, out_channel
        self.upsample, self.downsample, self.demodulate = upsample, downsample, demodulate
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.padding = kernel_size // 2
        self.weight = nn.Parameter(torch.randn(1, out_channel, in_channel, kernel_size, kernel_size))
        self.modulation = EqualLinear(style_dim, in_channel, bias=True, bias_init_val=1, lr_mul=1, activation=None)


class FusedLeakyReLU(nn.Module):
    """parameter holder of basicsr's FusedLeakyReLU (bias [C]); the math is the conv epilogue ACT_LRELU_SQRT2."""

    def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel))
        self.negative_slope, self.scale = negative_slope, scale


class StyledConv(nn.Module):
    def __init__(self, in_channel, out_channel, kernel_size, style_dim, upsample=False, blur_kernel=(1, 3, 3, 1),
                 demodulate=True):
        super().__init__()
        self.conv = ModulatedConv2d(in_channel, out_channel, kernel_size, style_dim, upsample=upsample,
                                    blur_kernel=blur_kernel, demodulate=demodulate)
        self.bias = nn.Parameter(torch.zeros(1, out_channel, 1, 1))
        self.activate = FusedLeakyReLU(out_channel)


class ToRGB(nn.Module):
    def __init__(self, in_channel, style_dim, upsample=True, blur_kernel=(1, 3, 3, 1)):
        super().__init__()
        self.upsample = upsample
        self.conv = ModulatedConv2d(in_channel, 3, 1, style_dim, demodulate=False)
        self.bias = nn.Parameter(torch.zeros(1, 3, 1, 1))


def returned_image_precision(mode):
    """precision mode of TSPGAN's image-only level when the caller gets the image back (see TextGenerator.forward)"""
    return "fp16x3" if mode == "fp16x2" else None


class TextGenerator(nn.Module):
    """font style w + character labels → (structure image, prior64, prior32)   (models/networks.py:64-164)."""

    def __init__(self, size, style_dim, n_mlp, class_num, channel_multiplier=1, blur_kernel=(1, 3, 3, 1), lr_mlp=0.01):
        super().__init__()
        self.size, self.n_mlp, self.style_dim, self.class_num = size, n_mlp, style_dim, class_num
        self.style_mlp = nn.Sequential(PixelNorm(), *[
            EqualLinear(style_dim, style_dim, bias=True, bias_init_val=0, lr_mul=lr_mlp, activation="fused_lrelu")
            for _ in range(n_mlp)])
        cm = channel_multiplier
        self.channels = {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * cm, 128: 128 * cm, 256: 64 * cm, 512: 32 * cm,
                         1024: 16 * cm}
        self.input_text = SelectText(class_num, self.channels[4])
        self.conv1 = StyledConv(self.channels[4], self.channels[4], 3, style_dim, blur_kernel=blur_kernel)
        self.to_rgb1 = ToRGB(self.channels[4], style_dim, upsample=False)
        self.log_size = int(math.log(size, 2))
        self.convs, self.upsamples, self.to_rgbs = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        cin = self.channels[4]
        for i in range(3, self.log_size + 1):
            cout = self.channels[2 ** i]
            self.convs.append(StyledConv(cin, cout, 3, style_dim, upsample=True, blur_kernel=blur_kernel))
            self.convs.append(StyledConv(cout, cout, 3, style_dim, blur_kernel=blur_kernel))
            self.to_rgbs.append(ToRGB(cout, style_dim))
            cin = cout
        self.n_latent = self.log_size * 2 - 2
        self.precision = default_precision()
        self.module_call_fp16x3 = True               # forward() in the fp16x2 mode runs in fp16x3 (see forward); False: the mode's own arithmetic (bench.py --config gan times both)
        self._cache = PackCache()

    # ------------------------------------------------------------------ packing
    def _build(self, dtype):
        pk = {}
        f = lambda t: t.detach().float().contiguous()
        pk["mlp"] = []
        for i in range(1, self.n_mlp + 1):
            el = self.style_mlp[i]
            pk["mlp"].append((pack_linear_weight(el.weight, el.scale), f(el.bias.detach() * el.lr_mul)))   # networks.py:192-195
        pk["emb"] = f(self.input_text.TextEmbeddings.detach().reshape(self.class_num, -1))

        def styled(sc):
            mc = sc.conv
            w0 = mc.weight.detach()[0]                                             # [Cout,Cin,3,3]; scale folded in (networks.py:284)
            return dict(cin=mc.in_channel, cout=mc.out_channel, up=mc.upsample,
                        w=pack_conv_weight(w0, dtype, scale=mc.scale),
                        wsq_t=pack_wsq(w0, mc.scale),                              # [Cin,Cout] for the demod table
                        mod_w=pack_linear_weight(mc.modulation.weight, mc.modulation.scale), mod_b=f(mc.modulation.bias),
                        bias=f(sc.bias.detach().reshape(-1) + sc.activate.bias.detach()))   # :244 then :245

        def torgb(tr):
            mc = tr.conv
            # [3,Cin,1,1] → fp32 [3,Cin] with the layer's constant scale folded: the dedicated ToRGB kernel (mnet_torgb) does its three
            # dot products per pixel in fp32 in every precision mode
            return dict(cin=mc.in_channel, w=pack_conv_weight(mc.weight.detach()[0], torch.float32, cin_mult=1, cout_mult=1, scale=mc.scale).reshape(3, -1).contiguous(),
                        mod_w=pack_linear_weight(mc.modulation.weight, mc.modulation.scale), mod_b=f(mc.modulation.bias),
                        bias=pack_vec(tr.bias, 4))

        pk["conv1"] = styled(self.conv1)
        pk["rgb1"] = torgb(self.to_rgb1)
        pk["convs"] = [styled(c) for c in self.convs]
        pk["rgbs"] = [torgb(t) for t in self.to_rgbs]
        # all 17 modulation EqualLinears read the same latent (:141 repeats one w for every layer): one GEMM
        # [N,512] x [512, sum(Cin)] instead of 17 latency-bound launches; layer i takes columns mod_off[i] : +cin
        layers = [pk["conv1"], pk["rgb1"]] + pk["convs"] + pk["rgbs"]
        off = 0
        for L in layers:
            L["mod_off"] = off
            off += L["cin"]
        pk["mod_all_w"] = torch.cat([L["mod_w"] for L in layers], dim=0).contiguous()
        pk["mod_all_b"] = torch.cat([L["mod_b"] for L in layers], dim=0).contiguous()
        pk["mod_total"] = off
        return pk

    # ------------------------------------------------------------------ forward pieces
    def _mod(self, L, idx, bcast=0):
        """this layer's column window of the batched modulation EqualLinear (:283), one row per entry of ``idx`` (None: per style)
        → (rows, eps_scale, scale_b).  The rows are divided by 2^e (largest magnitude in [0.5, 1): mnet_style_rows), so the modulated
        activations x·s never exceed |x| in the half-precision storage modes; eps_scale = 4^-e makes the demodulation absorb the
        factor exactly, scale_b = 2^e [rows, bcast] undoes it for a conv without demodulation (ToRGB).  Power-of-two factors:
        bit-identical to the un-normalised evaluation (tests/test_modules_gpu.py::test_style_normalisation_is_exact)."""
        if _STYLE_NORM:
            return ops.style_rows(self._S, L["mod_off"], L["cin"], idx, bcast)
        return ops.gather_rows(self._S, L["mod_off"], L["cin"], idx), None, None

    def _style(self, L):
        """→ (modulation rows per glyph, demodulation table rsqrt(Σ (scale·W·s)² + 1e-8) per glyph (:286))"""
        s, eps, _ = self._mod(L, self._gidx)
        if self._gidx is None:
            return s, ops.demod(s, L["wsq_t"], eps)
        su, eps_u, _ = self._mod(L, None)                           # demodulation once per distinct style, gathered per glyph
        return s, ops.gather_rows(ops.demod(su, L["wsq_t"], eps_u), idx=self._gidx)

    @staticmethod
    def _styled(L, x, s, d, premodulated, post=None, out=None):
        """StyledConv with activation-side modulation.  ``premodulated``: x already carries ·s (applied once per
        element by the producer: the fused upsample or the previous conv's post_scale) — otherwise the conv
        prologue applies it.  ``post``: the NEXT StyledConv's style, multiplied into this conv's output."""
        return ops.conv2d(x, L["w"], L["cout"], 3, 3, (1, 1), (1, 1), in_scale=None if premodulated else s,
                          out_scale=d, bias=L["bias"], act=ops.ACT_LRELU_SQRT2, post_scale=post, out=out)

    def _to_rgb(self, L, x, skip):
        """ToRGB.forward (:313-321) as ONE streaming kernel: modulated 1x1 conv + bias + up-sampled skip (:318-319) + tanh → fp32 RGB0"""
        s, _, sb = self._mod(L, self._gidx, bcast=1)
        return ops.torgb(x, L["w"], s, sb, L["bias"], skip)

    def forward_nhwc(self, styles, labels, need_image=True, style_index=None, p64_out=None, p32_out=None, image_precision=None):
        """→ (image NHWC fp32 [N,128,128c,4] (RGB0), prior64 NHWC [N,64,64c,256], prior32 NHWC [N,32,32c,512]).
        ``style_index`` (int64 [N], optional): ``styles`` then holds only the DISTINCT style vectors (one per image in
        test_sr.py:183, where every glyph of an image gets the same w) and glyph i uses styles[style_index[i]] — the style
        MLP, the 17 modulations and the 11 demodulation tables run once per distinct style and are gathered per glyph.
        ``p64_out`` / ``p32_out``: preallocated NHWC [N,64,64,256] / [N,32,32,512] views the two prior levels are written into by
        the producing conv itself (the batched driver hands slices of its all-glyph buffers: no concatenation pass afterwards).
        ``need_image=False`` (opt-in, batched SR driver only) stops after the 64-px level: the 128-px level feeds nothing
        but the visualisation image (models/networks.py:148-164; 35 % of the generator's FLOPs) and ``image`` is None.
        ``image_precision`` (batched SR driver only, for an image it drops — its ``return_prior=True`` form keeps the mode's arithmetic): precision mode of the levels BEHIND the two
        prior levels — they feed nothing but the structure image (:161-164), so the driver keeps the reference's work but does not
        spend SR-grade arithmetic on it; prior64 / prior32 (and hence the SR output) are bit-identical with and without it."""
        pk = self._cache.get(self, self.precision, self._build)
        dtype = torch_dtype(self.precision)
        pk_img = pk if image_precision in (None, self.precision) else self._cache.get(self, image_precision, self._build)
        lat = ops.pixelnorm(styles)                                            # :170-171
        for w, b in pk["mlp"]:
            lat = ops.linear(lat, w, self.style_dim, bias=b, act=ops.ACT_LRELU_SQRT2)
        # every layer's modulation at once: [styles, Σ cin]; _mod() / _style() take per-layer column windows of it, per glyph
        self._S = ops.linear(lat, pk["mod_all_w"], pk["mod_total"], bias=pk["mod_all_b"])
        self._gidx = style_index
        s, d = self._style(pk["conv1"])
        # SelectText (:205-215) with conv1's modulation ·s riding in the gather (the gathered constant has no other reader): conv1 then runs
        # without a modulation prologue, i.e. on the LDS-DMA kernels like every other StyledConv (0.97 -> 0.2 ms per 1024 glyphs, 0.56 -> 0.1 ms
        # for one strip's 16) — `_FUSE_CONV1_MOD = False` keeps the prologue form
        x = ops.embed_gather(pk["emb"], labels, dtype, self.class_num, scale=s if _FUSE_CONV1_MOD else None)
        x = self._styled(pk["conv1"], x, s, d, premodulated=_FUSE_CONV1_MOD)
        skip = self._to_rgb(pk["rgb1"], x, None) if need_image else None
        p64 = p32 = None
        for lvl in range(len(pk["rgbs"])):
            if not need_image and p64 is not None and p32 is not None:
                return None, p64, p32
            pkl, up_dtype = pk, None
            if pk_img is not pk and p64 is not None and p32 is not None:      # image-only levels (see ``image_precision``)
                pkl = pk_img
                up_dtype = torch_dtype(image_precision)
                if not (_FUSE_IMG_CONVERT and up_dtype == torch.float16 and is_split(x.dtype)):
                    x = ops.convert(x, up_dtype)                              # (the general case: its own pass)
                # else: the up-sample below reads the prior level in the mode's storage and writes plain f16 (round 5: no convert pass)
            La, Lb = pkl["convs"][2 * lvl], pkl["convs"][2 * lvl + 1]
            sa, da = self._style(La)
            sb, db = self._style(Lb)
            xu = ops.upsample2x(x, scale=sa, out_dtype=up_dtype if up_dtype != x.dtype else None)   # bilinear ×2 (:293) with ·s_a fused
            xa = self._styled(La, xu, sa, da, premodulated=True, post=sb)      # emits x_a·s_b (x_a has no other reader)
            del xu
            wx = xa.shape[2]                                                   # this level's width (absolute, see below)
            x = self._styled(Lb, xa, sb, db, premodulated=True, out=p64_out if wx == 64 else (p32_out if wx == 32 else None))
            del xa
            if need_image:
                skip = self._to_rgb(pk["rgbs"][lvl], x, skip)
            if x.shape[2] == 64:              # ABSOLUTE width, like the reference (:155,158): with c characters per
                p64 = x                       # sample the map is 4c·2^k wide, so c = 2 hands out the 32x64 / 16x32 levels
            if x.shape[2] == 32:
                p32 = x
        if p64 is None or p32 is None:
            # the reference reaches `return image, prior_features64, prior_features32` with an unbound local here
            raise RuntimeError("no generator level is 64 / 32 pixels wide for %d characters per sample "
                               "(models/networks.py:155-160 selects the prior levels by absolute width)" % labels.shape[1])
        return skip, p64, p32

    def forward(self, styles, labels, noise=None):
        with torch.no_grad(), ops.on_device(styles):
            styles = styles.contiguous().float()
            labels = labels.to(styles.device).contiguous().long()
            if labels.dim() != 2 or labels.shape[0] != styles.shape[0]:
                raise ValueError("labels must be [N,c] with N == styles.size(0)")
            if labels.numel() and (int(labels.min()) < 0 or int(labels.max()) >= self.class_num):
                # the reference fails on label -1 (empty slice → torch.cat error, caught by test_sr.py:181-190)
                raise RuntimeError("label index out of range [0,%d)" % self.class_num)
            # this call form RETURNS the structure image (test_sr.py:183, test_w.py:108), and the image is where the fp16x2 mode has its thinnest margin: 6.8e-4 on the
            # trained-like regime, still 5.4e-4 with only the image's own 128-px level in the three-product arithmetic (round 6, measured) — every level's ToRGB adds into
            # it (models/networks.py:313-321).  So the MODULE call runs the whole generator in fp16x3 when its mode is fp16x2 (3.8e-5; 21 % of a strip's FLOPs at a third
            # instead of half the fp16 rate — paid by test_w.py / test_sr.py-style callers only: the batched driver, MarconetPipeline, calls forward_nhwc itself and keeps
            # the priors in the mode's arithmetic, with the image-only level in fp16x3 when it returns the image: returned_image_precision)
            mode = self.precision
            self.precision = "fp16x3" if (mode == "fp16x2" and self.module_call_fp16x3) else mode
            try:
                img, p64, p32 = self.forward_nhwc(styles, labels)
            finally:
                self.precision = mode
            out = ops.nhwc_to_nchw(img, c=3), ops.nhwc_to_nchw(p64), ops.nhwc_to_nchw(p32)
            # keep the NHWC originals reachable so TSPSRNet can skip the NCHW→NHWC round trip
            # (valid only while the NCHW tensor is unmodified: its version counter and address are recorded with the shadow)
            out[1]._mnet_nhwc, out[2]._mnet_nhwc = (p64, out[1]._version, out[1].data_ptr()), (p32, out[2]._version, out[2].data_ptr())
            return out


class TSPGAN(nn.Module, _Precision):
    def __init__(self, out_size=128, num_style_feat=512, class_num=6736, num_mlp=8):
        super().__init__()
        self.TextGenerator = TextGenerator(size=out_size, style_dim=num_style_feat, n_mlp=num_mlp, class_num=class_num)
        self.precision = default_precision()

    def forward(self, styles, labels, noise):
        self.TextGenerator.precision = self.precision
        return self.TextGenerator(styles, labels, noise)


# =====================================================================================================
# 3) TSPSRNet  (models/networks.py:328-533)
# =====================================================================================================
class _SNConv(nn.Module):
    """Holder with the old-style ``torch.nn.utils.spectral_norm`` parametrisation of an nn.Conv2d
    (networks.py:14): Parameters ``bias``, ``weight_orig``; buffers ``weight_u``, ``weight_v``."""

    def __init__(self, cin, cout, k=3, stride=1):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size, self.stride = cin, cout, k, stride
        conv = nn.Conv2d(cin, cout, k, stride, k // 2)
        self.bias = nn.Parameter(conv.bias.detach().clone())
        self.weight_orig = nn.Parameter(conv.weight.detach().clone())
        nrm = lambda t: t / (t.norm() + 1e-12)
        self.register_buffer("weight_u", nrm(torch.randn(cout)))
        self.register_buffer("weight_v", nrm(torch.randn(cin * k * k)))


def GroupNorm(in_channels):
    assert in_channels % 32 == 0
    return nn.GroupNorm(num_groups=in_channels // 32, num_channels=in_channels, eps=1e-6, affine=True)


# ---- module-level helpers of the reference's models/networks.py (:492-493, :518-533).  The scripts never call them — TSPSRNet.forward does the same
#      arithmetic inside the fused AdaIN / GroupNorm kernels — but `from models.networks import swish, calc_mean_std_4D, adaptive_instance_normalization`
#      resolves, with the reference's signatures (fp32 NCHW tensors in and out) and the statistics / elementwise passes on the HIP kernels.
def _as_rows32(x):
    """[b, c, ...] contiguous fp32 → an NHWC view [b*c, 1, m/32, 32] of the SAME memory (zero-padded copy when a channel's element count m is not a
    multiple of 32): a (b, c) plane becomes one 'image' whose 32 'channels' form one GroupNorm group"""
    b, c = x.shape[:2]
    flat = x.contiguous().float().reshape(b * c, -1)
    m = flat.shape[1]
    mp = (m + 31) // 32 * 32
    if mp != m:
        flat = torch.cat([flat, flat.new_zeros((flat.shape[0], mp - m))], dim=1)
    return flat.reshape(b * c, 1, mp // 32, 32), m, mp


def swish(x):
    """x * sigmoid(x) (models/networks.py:492-493) on mnet_affine_act_nhwc"""
    v = x.contiguous().float().reshape(1, -1)
    m = v.shape[1]
    mp = (m + 31) // 32 * 32
    if mp != m:
        v = torch.cat([v, v.new_zeros((1, mp - m))], dim=1)
    one = torch.ones((1, 32), dtype=torch.float32, device=x.device)
    y = ops.affine_act(v.reshape(1, 1, mp // 32, 32), one, None, swish=True)
    return y.reshape(-1)[:m].reshape(x.shape)


def calc_mean_std_4D(feat, eps=1e-5):
    """per-(b, c) mean and sqrt(unbiased variance + eps) of a 4-D feature (models/networks.py:518-525) → ([b,c,1,1], [b,c,1,1]).  The sums run in
    mnet_groupnorm_affine (fp64 partial sums) over the planes viewed as one-group images; the few scalars per plane are unpacked on the host side."""
    if feat.dim() != 4:
        raise AssertionError("The input feature should be 4D tensor.")
    b, c = feat.shape[:2]
    rows, m, mp = _as_rows32(feat)
    dev = feat.device
    tiny = 1e-30                                             # keeps a constant plane finite; removed again below
    scale, shift = ops.groupnorm_affine(rows, torch.ones(32, device=dev), torch.zeros(32, device=dev), tiny)
    rstd = scale[:, 0].double()
    mean_p = (-shift[:, 0].double() / rstd)                   # mean and biased variance over the PADDED count mp
    var_p = (1.0 / (rstd * rstd) - tiny).clamp_min(0.0)
    s1 = mean_p * mp
    s2 = (var_p + mean_p * mean_p) * mp
    mean = s1 / m
    var = ((s2 - m * mean * mean) / max(m - 1, 1)).clamp_min(0.0)          # torch.var: unbiased
    return mean.float().reshape(b, c, 1, 1), (var + eps).sqrt().float().reshape(b, c, 1, 1)


def adaptive_instance_normalization(prior_feat, lq_feat):
    """(prior - mean_p) / std_p * std_lq + mean_lq per (b, c) (models/networks.py:527-533): statistics as above, the apply pass on mnet_affine_act_nhwc"""
    lq_mean, lq_std = calc_mean_std_4D(lq_feat)
    p_mean, p_std = calc_mean_std_4D(prior_feat)
    g = (lq_std / p_std).reshape(-1, 1)
    sh = (lq_mean.reshape(-1, 1) - p_mean.reshape(-1, 1) * g)
    rows, m, mp = _as_rows32(prior_feat)
    y = ops.affine_act(rows, g.expand(-1, 32).contiguous(), sh.expand(-1, 32).contiguous(), swish=False)
    return y.reshape(rows.shape[0], mp)[:, :m].reshape(prior_feat.shape)


class ResTextBlockV2(nn.Module):
    def __init__(self, in_channels, out_channels=None):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = in_channels if out_channels is None else out_channels
        self.norm1 = GroupNorm(in_channels)
        self.conv1 = _SNConv(in_channels, self.out_channels)
        self.norm2 = GroupNorm(self.out_channels)
        self.conv2 = _SNConv(self.out_channels, self.out_channels)
        if self.in_channels != self.out_channels:
            self.conv_out = nn.Conv2d(in_channels, self.out_channels, kernel_size=1, stride=1, padding=0)


def _seq(*mods):
    return nn.Sequential(*mods)


class TSPSRNet(nn.Module, _Precision):
    """LR image + per-image structure priors (64², 32²) + glyph locations → SR image [B,3,128,2048]."""

    def __init__(self, in_channel=3, dim_channel=256):
        super().__init__()
        D = dim_channel
        I = nn.Identity            # index placeholders for the reference's parameter-free layers
        self.conv_first_32 = _seq(_SNConv(in_channel, D // 4), I())
        self.conv_first_16 = _seq(_SNConv(D // 4, D // 2, 3, 2), I())
        self.conv_first_8 = _seq(_SNConv(D // 2, D, 3, 2), I(), _SNConv(D, D))
        self.conv_body_16 = _seq(_SNConv(D + D // 2, D), I(), _SNConv(D, D))
        self.conv_body_32 = _seq(_SNConv(D + D // 4, D), I(), _SNConv(D, D))
        self.conv_up = _seq(I(), _SNConv(D, D), I(), ResTextBlockV2(D, D), _SNConv(D, D))
        self.conv_final = _seq(_SNConv(D, D // 2), I(), I(), _SNConv(D // 2, D // 4), I(),
                               ResTextBlockV2(D // 4, D // 4), _SNConv(D // 4, 3), I())
        self.conv_32_scale = _seq(_SNConv(D, D), I(), _SNConv(D, D))
        self.conv_32_shift = _seq(_SNConv(D, D), I(), _SNConv(D, D))
        self.conv_32_fuse = _seq(ResTextBlockV2(2 * D, D))
        self.conv_32_to256 = _seq(_SNConv(512, D), I(), _SNConv(D, D))
        self.conv_64_scale = _seq(_SNConv(D, D), I(), _SNConv(D, D))
        self.conv_64_shift = _seq(_SNConv(D, D), I(), _SNConv(D, D))
        self.conv_64_fuse = _seq(ResTextBlockV2(2 * D, D))
        self.dim = D
        self.precision = default_precision()
        # OPT-IN per-layer precision plan (tools/precision_plan.py): "fp16" runs the two conv_*_scale branches (models/networks.py:
        # 377-381,446) in plain fp16 — the layers whose demotion costs the SR output least on regular strips (+0.9e-4 / +1.1e-4 each at the 64-px
        # scale, emulated; 3.3e-4 instead of 2.0e-4 on the bench batch, measured).  NOT the default: on the reference's edge-window examples
        # (glyph windows clipped at the strip border) the same plan measures 1.09e-3 — over the bar.  None: off.
        self.scale_branch_precision = None
        self._cache = PackCache()

    def _scale_branch_pack(self):
        """packed weights the conv_*_scale branches run with, or None when they follow the module's precision mode"""
        want = self.scale_branch_precision
        if want is None or want == self.precision or self.precision == "fp32":
            return None
        return self._cache.get(self, want, self._build), torch_dtype(want)

    # ------------------------------------------------------------------ packing (SN fold, K18)
    def _build(self, dtype):
        pk = {}

        def sn(name, m, cout_mult=4):
            w = pack_conv_weight(m.weight_orig.detach(), dtype, cout_mult=cout_mult, sn=(m.weight_u, m.weight_v))
            cp = padded_cout(m.weight_orig.shape[0], dtype, cout_mult)   # whole 32-channel blocks in the split-half / fp16+8 modes
            pk[name] = dict(w=w, b=pack_vec(m.bias, cp), cout=cp, stride=(m.stride, m.stride))

        def res(name, m):
            sn(name + ".conv1", m.conv1)
            sn(name + ".conv2", m.conv2)
            f = lambda t: t.detach().float().contiguous()
            pk[name + ".norm1"] = (f(m.norm1.weight), f(m.norm1.bias))
            pk[name + ".norm2"] = (f(m.norm2.weight), f(m.norm2.bias))
            if hasattr(m, "conv_out"):
                pk[name + ".conv_out"] = dict(w=pack_conv_weight(m.conv_out.weight.detach(), dtype), b=f(m.conv_out.bias))
                if dtype != torch.float32:
                    # round 4: h + conv_out(x) (models/networks.py:514-515) folded into conv2's k-loop — one weight tensor [cout][3][3][C + Cin]
                    # whose second part is the 1x1 skip weight at the centre tap (MNET_CONV_ALGO_FLAG_X1_CENTER), one bias.  Layout plumbing only:
                    # the spectral-norm fold of conv2 is the library's (fp32 pack), the skip weights are copied into a zero tensor
                    c2 = m.conv2
                    w2 = pack_conv_weight(c2.weight_orig.detach(), torch.float32, cin_mult=1, cout_mult=1, sn=(c2.weight_u, c2.weight_v))   # [O][3][3][I], folded
                    w2 = w2.reshape(c2.out_channels, 3, 3, c2.in_channels).permute(0, 3, 1, 2)
                    wo = m.conv_out.weight.detach().float()
                    wz = torch.zeros((wo.shape[0], wo.shape[1], 3, 3), dtype=torch.float32, device=wo.device)
                    wz[:, :, 1, 1] = wo[:, :, 0, 0]
                    pk[name + ".conv2+out"] = dict(w=pack_conv_weight(torch.cat([w2, wz], dim=1).contiguous(), dtype),
                                                   b=pack_vec(c2.bias.detach() + m.conv_out.bias.detach(), pk[name + ".conv2"]["cout"]),
                                                   cout=pk[name + ".conv2"]["cout"])

        for name in ("conv_first_32", "conv_first_16"):
            sn(name + ".0", getattr(self, name)[0])
        for name in ("conv_first_8", "conv_body_16", "conv_body_32", "conv_32_scale", "conv_32_shift", "conv_32_to256",
                     "conv_64_scale", "conv_64_shift"):
            sn(name + ".0", getattr(self, name)[0])
            sn(name + ".2", getattr(self, name)[2])
        sn("conv_up.1", self.conv_up[1]); res("conv_up.3", self.conv_up[3]); sn("conv_up.4", self.conv_up[4])
        sn("conv_final.0", self.conv_final[0]); sn("conv_final.3", self.conv_final[3])
        res("conv_final.5", self.conv_final[5]); sn("conv_final.6", self.conv_final[6], cout_mult=rgb_pad(dtype))
        m6 = self.conv_final[6]                       # the same layer for the dedicated 64 → 3 kernel: [3][3][3][64], bias [3]
        rgb_dt = torch.float32 if is_split(dtype) else dtype          # (the split-half mode runs this 64 → 3 layer in fp32)
        pk["conv_final.6.rgb"] = (pack_conv_weight(m6.weight_orig.detach(), rgb_dt, cin_mult=1, cout_mult=1, sn=(m6.weight_u, m6.weight_v)),
                                  m6.bias.detach().float().contiguous())
        res("conv_32_fuse.0", self.conv_32_fuse[0]); res("conv_64_fuse.0", self.conv_64_fuse[0])
        return pk

    # ------------------------------------------------------------------ building blocks
    @staticmethod
    def _c(pk, name, x, act=ops.ACT_NONE, x1=None, valid_w=None, in_scale=None, in_shift=None, residual=None, gn_partial=None):
        L = pk[name]
        return ops.conv2d(x, L["w"], L["cout"], 3, 3, L["stride"], (1, 1), x1=x1, bias=L["b"], act=act, valid_w=valid_w,
                          in_scale=in_scale, in_shift=in_shift, in_swish=in_scale is not None, residual=residual, gn_partial=gn_partial)

    def _conv_gn(self, pk, name, x, norm, act=ops.ACT_NONE, x1=None, valid_w=None):
        """conv ``name`` whose output feeds GroupNorm ``norm`` (networks.py:487-493) → (y, (scale, shift) of that GroupNorm).  Round 5: in the
        fp16+8 mode the statistics are partial sums written by the conv's own epilogue (mnet_conv_desc.gn_partial, one wave-level fold per 32 pixels
        x 32 channels) and folded by mnet_groupnorm_affine_from_partial — the separate statistics pass over the map is gone; every other storage
        type / kernel keeps that pass (mnet_groupnorm_affine)."""
        L = pk[name]
        n, h, w, _ = x.shape
        if ops.can_emit_gn_partial(x, x1, L["cout"], L["stride"], h, w):
            # (can_emit_gn_partial has asked the planner: this launch goes to a kernel whose epilogue writes the sums; any error below is a real one)
            part = ops.gn_partial_buffer(n, h, w, L["cout"], x.device)
            y = self._c(pk, name, x, act, x1=x1, valid_w=valid_w, gn_partial=part)
            return y, ops.groupnorm_affine_from_partial(part, n, h, w, L["cout"], *pk[norm], 1e-6, valid_w)
        y = self._c(pk, name, x, act, x1=x1, valid_w=valid_w)
        return y, ops.groupnorm_affine(y, *pk[norm], 1e-6, valid_w)

    def _two(self, pk, name, x, x1=None, valid_w=None):
        """Sequential(SNconv, LeakyReLU(0.2), SNconv)."""
        h = self._c(pk, name + ".0", x, ops.ACT_LRELU, x1=x1, valid_w=valid_w)
        return self._c(pk, name + ".2", h, valid_w=valid_w)

    def _res_block(self, pk, name, x, valid_w=None, norm1_affine=None):
        """ResTextBlockV2 (networks.py:506-516): GN statistics → one elementwise normalise+swish pass → conv; the skip
        rides in the second conv — as extra K at the centre tap when the block has a 1x1 conv_out (the fuse blocks), as the
        epilogue's residual otherwise.  ``norm1_affine``: (scale, shift) of norm1 when the producer of x already has them (the
        AdaIN kernel for the fuse blocks).
        Masked-column contract (``valid_w``): the folded form reads x like every conv input — zeros at columns >= valid_w[n] — so
        there its output is conv2(h) + bias only, while the two-launch form adds conv_out(x) of whatever x holds there.  Both are
        outside the glyph's window: nothing downstream reads them (the scale / shift convs mask the same columns, the scatter
        writes columns < valid_w only), and the AdaIN kernel writes zeros there anyway."""
        s1, h1 = norm1_affine if norm1_affine is not None else ops.groupnorm_affine(x, *pk[name + ".norm1"], 1e-6, valid_w)
        xs = ops.affine_act(x, s1, h1, swish=True)              # GN apply + swish once per element
        h, (s2, h2) = self._conv_gn(pk, name + ".conv1", xs, name + ".norm2", valid_w=valid_w)
        del xs
        ops.affine_act(h, s2, h2, swish=True, out=h)            # in place: h has no other reader
        skip = x
        if (name + ".conv2+out") in pk and _FOLD_SKIP:
            L = pk[name + ".conv2+out"]       # the 1x1 skip conv as extra K of conv2: no separate launch, no residual read in the epilogue
            # only the LDS-DMA kernels walk a second source at one tap: ask the planner (cached per shape) whether this launch is theirs; one they do
            # not take runs the two-launch form below
            if ops.plan_is_lds_dma(ops.conv_plan(h, L["cout"], 3, 3, (1, 1), (1, 1), x1=x, algo=_lib.ALGO_FLAG_X1_CENTER)):
                return ops.conv2d(h, L["w"], L["cout"], 3, 3, (1, 1), (1, 1), x1=x, bias=L["b"], valid_w=valid_w, x1_center=True)
        if (name + ".conv_out") in pk:
            co = pk[name + ".conv_out"]
            skip = ops.conv2d(x, co["w"], co["b"].numel(), bias=co["b"])
        return self._c(pk, name + ".conv2", h, valid_w=valid_w, residual=skip)

    def _prior_transform(self, pk, tag, feat, prior, tab):
        """All glyphs of the batch at one scale (networks.py:421-449 / :455-482)."""
        if tab.G == 0:
            return feat
        # AdaIN + crop + cat, and norm1's GroupNorm affine of the result in closed form from the AdaIN statistics
        cat, s1, h1 = ops.adain_crop_concat_gn(prior, feat, tab.g_img, tab.g_x1, tab.g_y1, tab.g_w,
                                               *pk["conv_%s_fuse.0.norm1" % tag], 1e-6)
        fused = self._res_block(pk, "conv_%s_fuse.0" % tag, cat, valid_w=tab.g_w, norm1_affine=(s1, h1))
        del cat
        sb = self._scale_branch_pack()
        if sb is None:
            sc = self._two(pk, "conv_%s_scale" % tag, fused, valid_w=tab.g_w)
        else:                                                    # precision plan: this branch in another storage / arithmetic (see __init__)
            sc = ops.convert(self._two(sb[0], "conv_%s_scale" % tag, ops.convert(fused, sb[1]), valid_w=tab.g_w), fused.dtype)
        sh = self._two(pk, "conv_%s_shift" % tag, fused, valid_w=tab.g_w)
        return ops.glyph_scatter_affine(feat, sc, sh, tab.g_start, tab.g_x1, tab.g_w)        # ori + (f*scale+shift)

    @staticmethod
    def _gather_priors(priors, dtype, channels, size):
        """list of NCHW fp32 [n_b,C,S,S] (or tensors produced by our TSPGAN, which carry their NHWC original)."""
        parts = []
        for p in priors:
            if p.shape[0] == 0:
                continue
            if p.dim() != 4 or p.shape[1] != channels or p.shape[2] != size or p.shape[3] != size:
                raise ValueError("prior of shape %s, expected [n,%d,%d,%d]" % (tuple(p.shape), channels, size, size))
            sh = getattr(p, "_mnet_nhwc", None)
            # the NHWC original is used only if the caller has not touched the NCHW tensor since (in-place edits bump _version)
            if sh is not None and sh[0].dtype == dtype and sh[0].shape[0] == p.shape[0] and sh[1] == p._version and sh[2] == p.data_ptr():
                parts.append(sh[0])
            else:
                parts.append(ops.nchw_to_nhwc(p.contiguous().float(), dtype))
        if not parts:
            return None
        return ops.cat_rows(parts)

    # ------------------------------------------------------------------ forward
    def forward(self, lq, priors64, priors32, locs):
        """reference call form (test_sr.py:197): lists (one entry per image) of NCHW fp32 priors."""
        with torch.no_grad(), ops.on_device(lq):
            dtype = torch_dtype(self.precision)
            B = lq.shape[0]
            if len(priors32) > B or len(priors64) > B:
                raise IndexError("more prior lists (%d / %d) than images (%d)" % (len(priors64), len(priors32), B))
            counts32 = [int(p.shape[0]) for p in priors32] + [0] * (B - len(priors32))
            counts64 = [int(p.shape[0]) for p in priors64] + [0] * (B - len(priors64))
            p32 = self._gather_priors(priors32, dtype, 512, 32) if sum(counts32) else None
            p64 = self._gather_priors(priors64, dtype, 256, 64) if sum(counts64) else None
            return self.forward_packed(lq, p64, p32, counts64, counts32, locs, nchw_out=True)

    def forward_packed(self, lq, p64, p32, counts64, counts32, locs, nchw_out=False, tables=None):
        """batched entry: ``p64`` NHWC [ΣN,64,64,256] / ``p32`` NHWC [ΣN,32,32,512] hold the glyph priors of all
        images back to back (``counts*[b]`` glyphs for image b).  Returns NHWC [B,128,2048,8] (RGB in channels 0-2), or —
        ``nchw_out`` — the reference's fp32 NCHW [B,3,128,2048] written by the last conv itself.  ``tables``: prebuilt
        (GlyphTables@32, GlyphTables@64) instead of ``locs`` (the HIP-graph path keeps them at fixed device addresses)."""
        with torch.no_grad(), ops.on_device(lq):
            pk = self._cache.get(self, self.precision, self._build)
            dtype = torch_dtype(self.precision)
            x = ops.nchw_to_nhwc(lq.contiguous().float(), dtype, c_ld=rgb_pad(dtype))
            f32 = self._c(pk, "conv_first_32.0", x, ops.ACT_LRELU)                               # :412
            f16 = self._c(pk, "conv_first_16.0", f32, ops.ACT_LRELU)                             # :413
            f8 = self._c(pk, "conv_first_8.2", self._c(pk, "conv_first_8.0", f16, ops.ACT_LRELU))  # :414
            s16 = self._two(pk, "conv_body_16", ops.upsample2x(f8), x1=f16)                      # :415 (cat-free)
            s32 = self._two(pk, "conv_body_32", ops.upsample2x(s16), x1=f32)                     # :416
            del f8, f16, s16, f32, x

            # glyph windows: ONE device→host copy of locs, integer tables back
            n32, n64 = sum(counts32), sum(counts64)
            locs_host = locs.detach().float().cpu().numpy() if (n32 + n64) and tables is None else None
            if n32:
                tab32 = tables[0] if tables is not None else GlyphTables(locs_host, counts32, s32.shape[2], 16, lq.device)
                p32 = self._two(pk, "conv_32_to256", p32)                                        # :424
                s32 = self._prior_transform(pk, "32", s32, p32, tab32)                           # :425-449
                del p32

            h, aff = self._conv_gn(pk, "conv_up.1", ops.upsample2x(s32), "conv_up.3.norm1", ops.ACT_LRELU)   # conv_up :359-365
            del s32
            h = self._res_block(pk, "conv_up.3", h, norm1_affine=aff)
            s64 = self._c(pk, "conv_up.4", h)
            del h
            if n64:
                tab64 = tables[1] if tables is not None else GlyphTables(locs_host, counts64, s64.shape[2], 32, lq.device)
                s64 = self._prior_transform(pk, "64", s64, p64, tab64)                           # :455-482
                del p64

            h = self._c(pk, "conv_final.0", s64, ops.ACT_LRELU)                                  # conv_final :367-376
            del s64
            h, aff = self._conv_gn(pk, "conv_final.3", ops.upsample2x(h), "conv_final.5.norm1", ops.ACT_LRELU)
            h = self._res_block(pk, "conv_final.5", h, norm1_affine=aff)
            if h.shape[3] == 64:                      # conv_final.6 + tanh through the dedicated 3-output kernel
                wr, br = pk["conv_final.6.rgb"]
                y_nhwc, y_nchw = ops.conv3x3_rgb(h, wr, br, ops.ACT_TANH, nhwc=not nchw_out, nchw=nchw_out)
                return y_nchw if nchw_out else y_nhwc
            y = self._c(pk, "conv_final.6", h, ops.ACT_TANH)
            return ops.nhwc_to_nchw(y, c=3) if nchw_out else y
