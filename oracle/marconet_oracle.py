"""TEST INFRASTRUCTURE ONLY — CPU restatement (torch fp32, functional, state_dict-driven) of the
MARCONet ``test_sr.py`` / ``test_w.py`` inference forward.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
file.  The product path (``marconet_amd/``) never does: it raises if its HIP library is missing.

Pinning: ``tests/test_oracle.py`` checks this restatement (a) against the *real* reference modules
imported from /root/reference when that tree is present (build container), and (b) everywhere
against the golden vectors in ``tests/golden/`` that ``tests/golden/make_golden.py`` generated from
the real reference.  The reference itself ships no tests or golden tensors, so the
reference-run-here outputs are the pin.  One boundary stays *unpinned by the reference*:
``basicsr.ops.fused_act`` (third-party, un-vendored, un-pinned pip dependency, README.md:39) — its
published semantics ``sqrt(2)·leaky_relu(x + b[c], 0.2)`` are restated in ``fused_leaky_relu``.

Every function cites the reference lines it follows (paths relative to /root/reference).
All arithmetic is fp32 on CPU, NCHW, like the reference.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

SQRT2 = 2 ** 0.5


# =============================================================================================
# third-party op restated: basicsr.ops.fused_act (call sites models/networks.py:10,195,241-245)
# =============================================================================================
def fused_leaky_relu(x, bias, negative_slope=0.2, scale=SQRT2):
    """upstream kernel: x += b[(i/step_b)%C]; y = x>0 ? x : x*alpha; out = y*scale (act=3, grad=0)."""
    shape = [1, -1] + [1] * (x.dim() - 2)
    return F.leaky_relu(x + bias.reshape(shape), negative_slope) * scale


# =============================================================================================
# a2: ResNet-45, no BN, no bias (models/resnet.py:21-30, 63-74)
# =============================================================================================
_RESNET_CFG = [(3, (2, 1)), (4, (1, 1)), (6, (2, 1)), (6, (1, 1)), (3, (1, 1))]   # resnet.py:74


def resnet45_forward(sd, x, prefix="resnet."):
    x = F.relu(F.conv2d(x, sd[prefix + "conv1.weight"], padding=1))            # resnet.py:64-65
    for li, (nblk, stride) in enumerate(_RESNET_CFG, 1):
        for bi in range(nblk):
            p = "%slayer%d.%d." % (prefix, li, bi)
            s = stride if bi == 0 else (1, 1)
            y = F.relu(F.conv2d(x, sd[p + "conv1.weight"]))                    # resnet.py:23-24 (1x1)
            y = F.conv2d(y, sd[p + "conv2.weight"], stride=s, padding=1)       # resnet.py:25 (3x3, stride)
            if (p + "downsample.0.weight") in sd:                              # resnet.py:26-27,52-55
                x = F.conv2d(x, sd[p + "downsample.0.weight"], stride=s)
            x = F.relu(y + x)                                                  # resnet.py:28-29
    return x                                                                   # [B,512,8,512]


# =============================================================================================
# a3-a5: TextViT (models/textvit_arch.py)
# =============================================================================================
def posemb_sincos_1x64(dim=512, temperature=10000.0):
    """textvit_arch.py:170-181 for the fixed h=1, w=64 grid: cat(sin xω, cos xω, sin yω, cos yω)."""
    xs = torch.arange(64, dtype=torch.float32)
    ys = torch.zeros(64, dtype=torch.float32)
    omega = torch.arange(dim // 4) / (dim // 4 - 1)
    omega = 1.0 / (temperature ** omega)
    y = ys[:, None] * omega[None, :]
    x = xs[:, None] * omega[None, :]
    return torch.cat((x.sin(), x.cos(), y.sin(), y.cos()), dim=1).float()


def _ln(sd, key, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[key + ".weight"], sd[key + ".bias"], eps)


def _attention(sd, p, x, heads=8):
    """textvit_arch.py:104-112 — pre-LN MHSA, scale 64**-0.5, no biases."""
    B, N, D = x.shape
    h = _ln(sd, p + "norm", x)
    qkv = F.linear(h, sd[p + "to_qkv.weight"])
    q, k, v = [t.reshape(B, N, heads, D // heads).permute(0, 2, 1, 3) for t in qkv.chunk(3, dim=-1)]
    dots = torch.matmul(q, k.transpose(-1, -2)) * ((D // heads) ** -0.5)
    attn = dots.softmax(dim=-1)
    out = torch.matmul(attn, v).permute(0, 2, 1, 3).reshape(B, N, D)
    return F.linear(out, sd[p + "to_out.weight"])


def _feedforward(sd, p, x):
    """textvit_arch.py:84-91 — LN, Linear, exact-erf GELU, Linear."""
    h = _ln(sd, p + "net.0", x)
    h = F.gelu(F.linear(h, sd[p + "net.1.weight"], sd[p + "net.1.bias"]))
    return F.linear(h, sd[p + "net.3.weight"], sd[p + "net.3.bias"])


def _block(sd, p, x):
    x = _attention(sd, p + "0.", x) + x
    return _feedforward(sd, p + "1.", x) + x


def textvit_forward(sd, feat, prefix="transformer."):
    """textvit_arch.py:65-77 with Transformer.forward :146-164 inlined."""
    B = feat.shape[0]
    # Rearrange 'b c (h p1) (w p2) -> b h w (p1 p2 c)', p1=p2=8, h=1, w=64   (textvit_arch.py:33)
    tok = feat.reshape(B, 512, 8, 64, 8).permute(0, 3, 2, 4, 1).reshape(B, 64, 8 * 8 * 512)
    x = F.linear(tok, sd[prefix + "to_patch_embedding.1.weight"], sd[prefix + "to_patch_embedding.1.bias"])
    x = x + posemb_sincos_1x64()                                               # :67-68
    T = prefix + "transformer."
    x = _block(sd, T + "layers.0.", x)                                         # :147-149
    x = _block(sd, T + "layers.1.", x)
    x_cls = _block(sd, T + "layers_cls.0.", x)                                 # :151-153
    xt = x.permute(0, 2, 1)                                                    # :155 token axis last
    xt = F.linear(_ln(sd, T + "linear_seq_maxlen.0", xt), sd[T + "linear_seq_maxlen.1.weight"],
                  sd[T + "linear_seq_maxlen.1.bias"])
    x_16 = xt.permute(0, 2, 1)                                                 # [B,16,512]
    x_loc = _block(sd, T + "layers_locs.0.", x_16)                             # :156-158
    x_w = _block(sd, T + "layers_w.0.", x)                                     # :160-162
    logits = F.linear(_ln(sd, prefix + "linear_cls.0", x_cls), sd[prefix + "linear_cls.1.weight"],
                      sd[prefix + "linear_cls.1.bias"])                        # :71
    wt = x_w.permute(0, 2, 1)                                                  # :72
    wt = F.linear(_ln(sd, prefix + "linear_w_maxlen.0", wt), sd[prefix + "linear_w_maxlen.1.weight"],
                  sd[prefix + "linear_w_maxlen.1.bias"]).permute(0, 2, 1).reshape(B, 512)
    w = F.linear(_ln(sd, prefix + "linear_w.0", wt), sd[prefix + "linear_w.1.weight"], sd[prefix + "linear_w.1.bias"])
    h = _ln(sd, prefix + "linear_locs.0", x_loc)                               # :75, :44-50
    h = F.gelu(F.linear(h, sd[prefix + "linear_locs.1.weight"], sd[prefix + "linear_locs.1.bias"]))
    locs = torch.sigmoid(F.linear(h, sd[prefix + "linear_locs.3.weight"], sd[prefix + "linear_locs.3.bias"]))
    return logits, locs.reshape(B, -1), w


def encoder_forward(sd, lq):
    """a1: TextContextEncoderV2.forward (models/networks.py:42-45) → (logits, locs, w)."""
    return textvit_forward(sd, resnet45_forward(sd, lq))


def clear_labels(logits_1img, alphabet_size=6735):
    """test_w.py:34-40 — argmax per position, drop repeats, drop blank (index ≥ alphabet_size)."""
    idx = torch.max(logits_1img, 1)[1].tolist()
    out = []
    for i, v in enumerate(idx):
        if not (i > 0 and idx[i - 1] == v) and v < alphabet_size:
            out.append(v)
    return out


# =============================================================================================
# a6-a12: TSPGAN / TextGenerator (models/networks.py:64-321)
# =============================================================================================
_GAN_UPSAMPLE = [True, False] * 5          # convs.0..9: even index upsamples (networks.py:116-130)


def _modconv(x, weight, style, demodulate, upsample):
    """ModulatedConv2d.forward (networks.py:281-302): per-sample weights, grouped conv; bilinear×2
    (align_corners=False) applied to the *input* before the conv when upsample."""
    N, cin, H, W = x.shape
    cout, k = weight.shape[1], weight.shape[3]
    w = (1.0 / math.sqrt(cin * k * k)) * weight * style.reshape(N, 1, cin, 1, 1)
    if demodulate:
        d = torch.rsqrt(w.pow(2).sum([2, 3, 4]) + 1e-8)
        w = w * d.reshape(N, cout, 1, 1, 1)
    w = w.reshape(N * cout, cin, k, k)
    x = x.reshape(1, N * cin, H, W)
    if upsample:
        x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
    y = F.conv2d(x, w, padding=k // 2, groups=N)
    return y.reshape(N, cout, y.shape[2], y.shape[3])


def _styled_conv(sd, p, x, latent, upsample):
    """StyledConv.forward (networks.py:242-246): modconv → +bias[1,C,1,1] → FusedLeakyReLU(+bias[C])."""
    cin = x.shape[1]
    style = F.linear(latent, sd[p + ".conv.modulation.weight"] * (1.0 / math.sqrt(512)),
                     sd[p + ".conv.modulation.bias"])                          # EqualLinear lr_mul=1 (:188-197)
    y = _modconv(x, sd[p + ".conv.weight"], style, True, upsample)
    y = y + sd[p + ".bias"]
    return fused_leaky_relu(y, sd[p + ".activate.bias"])


def _to_rgb(sd, p, x, latent, skip):
    """ToRGB.forward (networks.py:313-321): 1×1 modconv (no demod) + bias (+ up(skip)) → tanh."""
    style = F.linear(latent, sd[p + ".conv.modulation.weight"] * (1.0 / math.sqrt(512)),
                     sd[p + ".conv.modulation.bias"])
    y = _modconv(x, sd[p + ".conv.weight"], style, False, False) + sd[p + ".bias"]
    if skip is not None:
        y = y + F.interpolate(skip, scale_factor=2, mode="bilinear", align_corners=False)
    return torch.tanh(y)


def style_mlp_forward(sd, styles, prefix="TextGenerator."):
    """PixelNorm + 8×EqualLinear(lr_mul=.01, fused_lrelu) (networks.py:83-89,170-171,188-198)."""
    x = styles * torch.rsqrt(torch.mean(styles ** 2, dim=1, keepdim=True) + 1e-8)
    scale = (1.0 / math.sqrt(512)) * 0.01
    for i in range(1, 9):
        x = F.linear(x, sd["%sstyle_mlp.%d.weight" % (prefix, i)] * scale)
        x = fused_leaky_relu(x, sd["%sstyle_mlp.%d.bias" % (prefix, i)] * 0.01)
    return x


def tspgan_forward(sd, styles, labels, prefix="TextGenerator."):
    """a6: TextGenerator.forward (networks.py:134-164) → (image, prior64, prior32). ``noise`` unused."""
    lat = style_mlp_forward(sd, styles, prefix)
    emb = sd[prefix + "input_text.TextEmbeddings"]                             # SelectText (:205-215)
    N, c = labels.shape
    x = torch.cat([emb[labels[:, j]].expand(N, 512, 4, 4) for j in range(c)], dim=3)
    x = _styled_conv(sd, prefix + "conv1", x, lat, False)
    skip = _to_rgb(sd, prefix + "to_rgb1", x, lat, None)
    p64 = p32 = None
    for lvl in range(5):
        x = _styled_conv(sd, "%sconvs.%d" % (prefix, 2 * lvl), x, lat, True)
        x = _styled_conv(sd, "%sconvs.%d" % (prefix, 2 * lvl + 1), x, lat, False)
        skip = _to_rgb(sd, "%sto_rgbs.%d" % (prefix, lvl), x, lat, skip)
        if x.shape[-1] == 64:                 # absolute width (networks.py:155,158)
            p64 = x
        if x.shape[-1] == 32:
            p32 = x
    return skip, p64, p32


# =============================================================================================
# a13-a16: TSPSRNet (models/networks.py:328-533)
# =============================================================================================
def sn_weight(sd, key):
    """a16: eval-mode old-style spectral norm: W_orig / (uᵀ (W_mat v)), no power iteration."""
    w = sd[key + ".weight_orig"]
    sigma = torch.dot(sd[key + ".weight_u"], torch.mv(w.reshape(w.shape[0], -1), sd[key + ".weight_v"]))
    return w / sigma


def _snconv(sd, key, x, stride=1):
    return F.conv2d(x, sn_weight(sd, key), sd[key + ".bias"], stride=stride, padding=1)


def _two_conv(sd, key, x):
    """Sequential(SNconv, LeakyReLU(0.2), SNconv) — conv_body_*, conv_*_scale/shift, conv_32_to256."""
    return _snconv(sd, key + ".2", F.leaky_relu(_snconv(sd, key + ".0", x), 0.2))


def _gn_swish(sd, key, x):
    c = x.shape[1]
    y = F.group_norm(x, c // 32, sd[key + ".weight"], sd[key + ".bias"], 1e-6)   # networks.py:487-490
    return y * torch.sigmoid(y)                                                  # :492-493


def res_text_block(sd, key, x):
    """a14: ResTextBlockV2.forward (networks.py:506-516)."""
    h = _snconv(sd, key + ".conv1", _gn_swish(sd, key + ".norm1", x))
    h = _snconv(sd, key + ".conv2", _gn_swish(sd, key + ".norm2", h))
    if (key + ".conv_out.weight") in sd:
        x = F.conv2d(x, sd[key + ".conv_out.weight"], sd[key + ".conv_out.bias"])
    return h + x


def adain(prior, lq, eps=1e-5):
    """a15: networks.py:518-533 — per-(sample,channel) mean / sqrt(unbiased var + eps)."""
    def ms(f):
        b, c = f.shape[:2]
        v = f.reshape(b, c, -1)
        return v.mean(dim=2).reshape(b, c, 1, 1), (v.var(dim=2) + eps).sqrt().reshape(b, c, 1, 1)
    lm, ls = ms(lq)
    pm, ps = ms(prior)
    return (prior - pm) / ps * ls + lm


def glyph_window(loc_center, feat_w, half):
    """SURVEY.md §3c / networks.py:426-441 (32-scale) and :460-474 (64-scale).
    center = trunc(fp32(loc)·fp32(W)); the width entry of ``locs`` is read but overwritten (:427-428)."""
    center = int(np.float32(loc_center) * np.float32(feat_w))        # .int() truncates toward zero
    x1 = 0 if center < half else center - half
    x2 = feat_w if center + half > feat_w else center + half
    y1 = half - int((x2 - x1) / 2)                                   # torch.div(..., rounding_mode='trunc')
    y2 = y1 + (x2 - x1)
    return x1, x2, y1, y2


def _prior_transform(sd, feat, priors, locs, scale_tag, half):
    """One of the two per-glyph loops (networks.py:421-449 / :455-482): reads from the unmodified
    ``feat``, writes into a zero ``res`` (later glyph index overwrites earlier), returns feat+res."""
    res = torch.zeros_like(feat)
    Wf = feat.shape[-1]
    for b, pr in enumerate(priors):
        if scale_tag == "32":
            pr = _two_conv(sd, "conv_32_to256", pr)                              # :424
        for c in range(pr.shape[0]):
            x1, x2, y1, y2 = glyph_window(float(locs[b, 2 * c]), Wf, half)
            cp = pr[c:c + 1, :, :, y1:y2]
            cl = feat[b:b + 1, :, :, x1:x2]
            fused = res_text_block(sd, "conv_%s_fuse.0" % scale_tag, torch.cat((adain(cp, cl), cl), dim=1))
            sc = _two_conv(sd, "conv_%s_scale" % scale_tag, fused)
            sh = _two_conv(sd, "conv_%s_shift" % scale_tag, fused)
            res[b, :, :, x1:x2] = cl[0] * sc[0] + sh[0]
    return feat + res


def _up2(x):
    return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)


def tspsr_forward(sd, lq, priors64, priors32, locs, return_intermediates=False):
    """a13: TSPSRNet.forward (networks.py:411-485)."""
    f32 = F.leaky_relu(_snconv(sd, "conv_first_32.0", lq), 0.2)                  # :412
    f16 = F.leaky_relu(_snconv(sd, "conv_first_16.0", f32, stride=2), 0.2)       # :413
    f8 = _snconv(sd, "conv_first_8.2", F.leaky_relu(_snconv(sd, "conv_first_8.0", f16, stride=2), 0.2))
    s16 = _two_conv(sd, "conv_body_16", torch.cat([_up2(f8), f16], dim=1))       # :415
    s32 = _two_conv(sd, "conv_body_32", torch.cat([_up2(s16), f32], dim=1))      # :416
    s32p = _prior_transform(sd, s32, priors32, locs, "32", 16)                   # :421-449
    # conv_up (:359-365): up, SNconv, lrelu, ResTextBlockV2, SNconv
    h = F.leaky_relu(_snconv(sd, "conv_up.1", _up2(s32p)), 0.2)
    h = res_text_block(sd, "conv_up.3", h)
    s64 = _snconv(sd, "conv_up.4", h)
    s64p = _prior_transform(sd, s64, priors64, locs, "64", 32)                   # :455-482
    # conv_final (:367-376)
    h = F.leaky_relu(_snconv(sd, "conv_final.0", s64p), 0.2)
    h = F.leaky_relu(_snconv(sd, "conv_final.3", _up2(h)), 0.2)
    h = res_text_block(sd, "conv_final.5", h)
    out = torch.tanh(_snconv(sd, "conv_final.6", h))
    if return_intermediates:
        return out, {"sq_f_32": s32, "sq_pf_32_out": s32p, "sq_f_64": s64, "sq_pf_64": s64p}
    return out


# =============================================================================================
# a17: driver glue (test_sr.py:146-201) for a batch of images — used by smoke()/bench cpu_baseline
# =============================================================================================
def end_to_end(sd_enc, sd_gan, sd_sr, lq, labels_per_image, locs):
    """encoder → per-image TSPGAN with the image's single style vector (test_sr.py:183) → SR."""
    with torch.no_grad():
        logits, enc_locs, w = encoder_forward(sd_enc, lq)
        p64, p32, imgs = [], [], []
        for b, lab in enumerate(labels_per_image):
            im, a, c = tspgan_forward(sd_gan, w[b:b + 1].repeat(lab.shape[0], 1), lab)
            imgs.append(im)
            p64.append(a)
            p32.append(c)
        sr = tspsr_forward(sd_sr, lq, p64, p32, locs)
    return {"logits": logits, "enc_locs": enc_locs, "w": w, "prior_images": imgs, "p64": p64, "p32": p32, "sr": sr}
