"""Seeded synthetic checkpoints and inputs for the MARCONet hot path (the build's replacement for the
reference's checkpoint fetcher, checkpoints/download_github.py — SURVEY.md §2 #9).  Pure numpy/torch-CPU data
generation; used by bench.py, __graft_entry__.smoke(), the tests and (re-exported as oracle/synth.py) the oracle.

The reference's real checkpoints are GitHub release assets (checkpoints/download_github.py:4-9) that
cannot be fetched (no network), so every parity claim in this repo is made on the synthetic
``state_dict``s produced here.  They satisfy ``load_state_dict(..., strict=True)`` of the reference
classes (test_sr.py:43-51): same keys, shapes and dtypes (schema pinned in
tests/golden/state_dict_schema.json, dumped from the reference classes themselves).

Determinism: all random numbers come from the *integer* output of numpy's PCG64 bit generator
(``random_raw``), turned into floats by exact arithmetic (sum of four 16-bit fields → Irwin–Hall
approximation of N(0,1); 53-bit mantissa fill for U[0,1)).  No libm call is involved, so the same
seed gives bit-identical fp32 tensors on the build container and on the GPU box.  The only
non-bit-reproducible step is the float64 power iteration for spectral-norm ``u``/``v`` (BLAS
summation order), which can move those vectors by an fp32 ulp — irrelevant at the 1e-4 tolerances
the golden fixtures are compared with.

Numerics are kept tame on purpose: a *fresh-init* reference ``TSPSRNet`` in eval()
produces NaN because spectral-norm u/v are random; here u/v are power-iterated to convergence so
sigma = uᵀWv is the true spectral norm, and the ResNet gain is reduced so features stay O(1..10).
"""
import hashlib
import math

import numpy as np
import torch

ALPHABET_SIZE = 6735          # utils/alphabets.py: 6735 characters, class 6735 = blank
NUM_CLASSES = 6736


# ----------------------------------------------------------------------------- PRNG
def _bitgen(seed, key):
    h = hashlib.sha256(("%d:%s" % (seed, key)).encode()).digest()
    return np.random.PCG64(int.from_bytes(h[:8], "little"))


def normal(seed, key, shape):
    """Approximately N(0,1) fp64 array, bit-reproducible (Irwin–Hall with 4 uniform 16-bit terms)."""
    n = int(np.prod(shape)) if len(shape) else 1
    raw = _bitgen(seed, key).random_raw(n)
    s = ((raw & 0xFFFF) + ((raw >> 16) & 0xFFFF) + ((raw >> 32) & 0xFFFF) + ((raw >> 48) & 0xFFFF)).astype(np.int64)
    # each term uniform on {0..65535}: mean 32767.5, var (65536^2-1)/12
    sd = math.sqrt(4.0 * (65536.0 ** 2 - 1.0) / 12.0)
    out = (s.astype(np.float64) - 4 * 32767.5) * (1.0 / sd)
    return out.reshape(shape)


def uniform01(seed, key, shape):
    n = int(np.prod(shape)) if len(shape) else 1
    raw = _bitgen(seed, key).random_raw(n)
    return ((raw >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)).reshape(shape)


def integers(seed, key, shape, lo, hi):
    """Uniform integers in [lo, hi) (tiny modulo bias is irrelevant here)."""
    n = int(np.prod(shape)) if len(shape) else 1
    raw = _bitgen(seed, key).random_raw(n)
    return (lo + (raw % np.uint64(hi - lo)).astype(np.int64)).reshape(shape)


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


# ----------------------------------------------------------------------------- weight regimes
# "tame" (default; every golden fixture): near-Gaussian / uniform weights of one scale per tensor — activations O(1) everywhere.
# "trained": a second seeded regime shaped like what training leaves behind, so that the parity margin of the
# half-range modes is reported on something other than the friendly strips:
#   * heavy-tailed conv weights: every element N(0,1)·exp(TAIL·N(0,1)) (log-normal magnitude mixing: kurtosis ~ 20 at TAIL = 0.6) and a
#     log-normal gain per OUTPUT channel (exp(ROW·N(0,1))), renormalised to the tame regime's variance — a few large taps per filter,
#     channels of very different magnitude inside one 32-channel storage block;
#   * StyleGAN-scale modulation: ``modulation.bias`` log-uniform over [10^-1.5, 10^+1.5] (a per-channel spread of 10^3 in the
#     style that multiplies the activations) instead of 1 +- 0.1;
#   * spectral-norm convs stored at sigma(weight_orig) log-uniform in [0.1, 10] (the fold divides by sigma: the effective weight keeps
#     spectral norm 1 but weight_orig, u and v live at very different scales per layer);
#   * GroupNorm / LayerNorm gains log-normal (a spread of x0.4 ... x2.5 around the median; TSPSRNet's GroupNorms around a median of 2.5,
#     which gives the SR output the tame regime's range), biases three times wider.
# The regime is still numerically sane in fp32 (finite, tanh not saturated) — that is checked where it is used.
REGIMES = ("tame", "trained")
_TAIL, _ROW = 0.6, 0.5


def _check_regime(regime):
    if regime not in REGIMES:
        raise ValueError("regime must be one of %s" % ", ".join(REGIMES))
    return regime == "trained"


def _heavy(seed, key, shape):
    """N(0,1)·exp(TAIL·z) per element times exp(ROW·z') per row (dim 0), scaled back to unit variance (fp64)"""
    z = normal(seed, key, shape) * np.exp(_TAIL * normal(seed, key + "#tail", shape))
    row = np.exp(_ROW * normal(seed, key + "#row", (shape[0],))).reshape((shape[0],) + (1,) * (len(shape) - 1))
    z = z * row
    return z / math.sqrt(float(np.mean(z * z)))


def _loguniform(seed, key, shape, lo, hi):
    return np.exp(math.log(lo) + uniform01(seed, key, shape) * (math.log(hi) - math.log(lo)))


# ----------------------------------------------------------------------------- encoder checkpoint
_RESNET_LAYERS = [(32, 3), (64, 4), (128, 6), (256, 6), (512, 3)]   # models/resnet.py:74 [3,4,6,6,3]


def make_encoder_state_dict(seed=1234, resnet_gain=None, input_gain=1.0, cls_gain=3.0, regime="tame"):
    """Keys of ``TextContextEncoderV2`` (models/networks.py:27-45).
    Stress variants (tests/test_stress_gpu.py): ``resnet_gain=1.0`` = the reference's own initialisation of every ResNet conv
    (models/resnet.py:45-48: features of std ~78, |max| ~610); ``input_gain`` multiplies conv1 — the BN-free,
    bias-free ReLU stack is positively homogeneous, so it scales every ResNet activation by that factor; ``cls_gain=1.0`` leaves
    near-ties among the 6736 logits (top-2 gaps down to 1e-4 and below)."""
    sd = {}
    trained = _check_regime(regime)

    def conv_w(key, cout, cin, k, gain):
        # reference init is N(0, sqrt(2/(k*k*cout))) (models/resnet.py:45-48); gain<1 keeps the
        # BN-free 45-layer stack from growing to |x|~600
        if resnet_gain is not None:
            gain = resnet_gain
        std = gain * math.sqrt(2.0 / (k * k * cout))
        sd[key] = _t((_heavy(seed, key, (cout, cin, k, k)) if trained else normal(seed, key, (cout, cin, k, k))) * std)

    conv_w("resnet.conv1.weight", 32, 3, 3, 1.0)
    if input_gain != 1.0:
        sd["resnet.conv1.weight"] = sd["resnet.conv1.weight"] * float(input_gain)
    inpl = 32
    for li, (planes, nblk) in enumerate(_RESNET_LAYERS, 1):
        for bi in range(nblk):
            p = "resnet.layer%d.%d." % (li, bi)
            cin = inpl if bi == 0 else planes
            conv_w(p + "conv1.weight", planes, cin, 1, 0.9)
            conv_w(p + "conv2.weight", planes, planes, 3, 0.68)
            if bi == 0:
                conv_w(p + "downsample.0.weight", planes, cin, 1, 0.7)
        inpl = planes

    def linear(key, out_f, in_f, bias=True, gain=1.0):
        b = gain / math.sqrt(in_f)
        sd[key + ".weight"] = _t((uniform01(seed, key + ".weight", (out_f, in_f)) * 2 - 1) * b)
        if bias:
            sd[key + ".bias"] = _t((uniform01(seed, key + ".bias", (out_f,)) * 2 - 1) * b)

    def layernorm(key, n):
        if trained:
            sd[key + ".weight"] = _t(np.exp(0.45 * normal(seed, key + ".weight", (n,))))
            sd[key + ".bias"] = _t(0.3 * normal(seed, key + ".bias", (n,)))
            return
        sd[key + ".weight"] = _t(1.0 + 0.1 * normal(seed, key + ".weight", (n,)))
        sd[key + ".bias"] = _t(0.1 * normal(seed, key + ".bias", (n,)))

    linear("transformer.to_patch_embedding.1", 512, 32768)
    for name, hidden in [("layers.0", 1024), ("layers.1", 1024), ("layers_cls.0", 1024),
                         ("layers_locs.0", 512), ("layers_w.0", 512)]:
        p = "transformer.transformer.%s." % name
        layernorm(p + "0.norm", 512)
        linear(p + "0.to_qkv", 1536, 512, bias=False)
        linear(p + "0.to_out", 512, 512, bias=False)
        layernorm(p + "1.net.0", 512)
        linear(p + "1.net.1", hidden, 512)
        linear(p + "1.net.3", 512, hidden)
    layernorm("transformer.transformer.linear_seq_maxlen.0", 64)
    linear("transformer.transformer.linear_seq_maxlen.1", 16, 64)
    layernorm("transformer.linear_cls.0", 512)
    linear("transformer.linear_cls.1", NUM_CLASSES, 512, gain=cls_gain)   # 3.0: wider top-2 logit gaps
    layernorm("transformer.linear_locs.0", 512)
    linear("transformer.linear_locs.1", 256, 512)
    linear("transformer.linear_locs.3", 2, 256)
    layernorm("transformer.linear_w.0", 512)
    linear("transformer.linear_w.1", 512, 512)
    layernorm("transformer.linear_w_maxlen.0", 64)
    linear("transformer.linear_w_maxlen.1", 1, 64)
    return sd


# ----------------------------------------------------------------------------- GAN checkpoint
_GAN_CONVS = [(512, 512)] * 6 + [(512, 256), (256, 256), (256, 128), (128, 128)]   # convs.0..9
_GAN_RGB_IN = [512, 512, 512, 256, 128]                                            # to_rgbs.0..4


def _apply_outliers(sd, outliers, skip=()):
    """stress variants (tests/test_stress_gpu.py): ``outliers`` = {state_dict key: (index along dim 0, gain)} — that slice of the tensor is
    multiplied by ``gain``, which puts ONE outlier channel into a 32-channel storage block of the consuming layer's activations"""
    for key, (idx, gain) in (outliers or {}).items():
        if key in skip:
            continue
        if key not in sd:
            raise KeyError("outlier key %s not in this state_dict" % key)
        sd[key] = sd[key].clone()
        sd[key][idx] *= float(gain)


def make_gan_state_dict(seed=1234, outliers=None, regime="tame"):
    """Keys of ``TSPGAN`` (models/networks.py:51-164), all under ``TextGenerator.``"""
    sd = {}
    trained = _check_regime(regime)
    wn = (lambda key, shape: _heavy(seed, key, shape)) if trained else (lambda key, shape: normal(seed, key, shape))
    # modulation bias: 1 +- 0.1, or — trained — log-uniform over three decades (the style multiplies the ACTIVATIONS on this implementation)
    mb = (lambda key, n: _loguniform(seed, key, (n,), 10.0 ** -1.5, 10.0 ** 1.5)) if trained else (lambda key, n: 1.0 + 0.1 * normal(seed, key, (n,)))
    P = "TextGenerator."
    for i in range(1, 9):
        k = P + "style_mlp.%d" % i
        # EqualLinear(lr_mul=0.01): weight = randn/lr_mul (models/networks.py:182)
        sd[k + ".weight"] = _t(normal(seed, k + ".weight", (512, 512)) * 100.0)
        sd[k + ".bias"] = _t(normal(seed, k + ".bias", (512,)) * 10.0)
    k = P + "input_text.TextEmbeddings"
    sd[k] = _t(normal(seed, k, (NUM_CLASSES, 512, 1, 1)))

    def styled(prefix, cin, cout):
        sd[prefix + ".bias"] = _t(0.1 * normal(seed, prefix + ".bias", (1, cout, 1, 1)))
        sd[prefix + ".conv.weight"] = _t(wn(prefix + ".conv.weight", (cout, cin, 3, 3)).reshape(1, cout, cin, 3, 3))
        sd[prefix + ".conv.modulation.weight"] = _t(wn(prefix + ".conv.modulation.weight", (cin, 512)))
        sd[prefix + ".conv.modulation.bias"] = _t(mb(prefix + ".conv.modulation.bias", cin))
        sd[prefix + ".activate.bias"] = _t(0.1 * normal(seed, prefix + ".activate.bias", (cout,)))

    def torgb(prefix, cin):
        sd[prefix + ".bias"] = _t(0.1 * normal(seed, prefix + ".bias", (1, 3, 1, 1)))
        sd[prefix + ".conv.weight"] = _t(wn(prefix + ".conv.weight", (3, cin, 1, 1)).reshape(1, 3, cin, 1, 1))
        sd[prefix + ".conv.modulation.weight"] = _t(wn(prefix + ".conv.modulation.weight", (cin, 512)))
        sd[prefix + ".conv.modulation.bias"] = _t(mb(prefix + ".conv.modulation.bias", cin))

    styled(P + "conv1", 512, 512)
    torgb(P + "to_rgb1", 512)
    for i, (cin, cout) in enumerate(_GAN_CONVS):
        styled(P + "convs.%d" % i, cin, cout)
    for i, cin in enumerate(_GAN_RGB_IN):
        torgb(P + "to_rgbs.%d" % i, cin)
    # reference key order: style_mlp, input_text, conv1, to_rgb1, convs, to_rgbs (insertion order above
    # differs only in that; load_state_dict is order-independent)
    _apply_outliers(sd, outliers)
    return sd


# ----------------------------------------------------------------------------- SR checkpoint
_SR_TRAINED = [False]
_GN_MEDIAN = [2.5]      # trained regime: GroupNorm gains of median 2.5 — calibrated so that the SR output spans the tanh range like the tame
                          # regime's (|sr| std 0.37, max 0.9; at median 1 the heavy-tailed spectral-norm convs leave it at std 0.16)
_SN_ROW_GAIN = {}      # make_sr_state_dict(outliers=...): weight_orig rows scaled BEFORE the power iteration (sigma stays the true spectral norm)


def _sn_conv(sd, seed, key, cout, cin, k=3, iters=40):
    """Old-style torch.nn.utils.spectral_norm parametrisation (models/networks.py:14,336-405):
    ``weight_orig`` (Parameter), ``weight_u`` / ``weight_v`` (buffers), ``bias``; eval-mode weight is
    weight_orig / (uᵀ · W_mat · v).  u, v are power-iterated here so sigma is the spectral norm."""
    fan_in = cin * k * k
    bound = 1.0 / math.sqrt(fan_in)
    trained = _SR_TRAINED[0]
    if trained:                         # heavy-tailed, of the uniform init's variance (bound^2 / 3)
        w = _heavy(seed, key + ".weight_orig", (cout, cin, k, k)) * (bound / math.sqrt(3.0))
    else:
        w = (uniform01(seed, key + ".weight_orig", (cout, cin, k, k)) * 2 - 1) * bound
    if key + ".weight_orig" in _SN_ROW_GAIN:
        row, gain = _SN_ROW_GAIN[key + ".weight_orig"]
        w[row] *= float(gain)
    if trained:                         # stored at sigma(weight_orig) log-uniform in [0.1, 10]: two SVD-free steps — the true sigma by a float64 power
        wm0 = np.ascontiguousarray(w.reshape(cout, -1).astype(np.float32)).astype(np.float64)           # iteration, then one rescale
        u0 = normal(seed, key + "#sigma", (cout,))
        for _ in range(iters):
            v0 = wm0.T @ u0; v0 /= np.linalg.norm(v0) + 1e-12
            u0 = wm0 @ v0; u0 /= np.linalg.norm(u0) + 1e-12
        sigma = float(u0 @ (wm0 @ v0))
        w = w * (float(_loguniform(seed, key + "#target", (1,), 0.1, 10.0)[0]) / sigma)
        sd[key + ".bias"] = _t(0.08 * normal(seed, key + ".bias", (cout,)))
    else:
        sd[key + ".bias"] = _t((uniform01(seed, key + ".bias", (cout,)) * 2 - 1) * bound)
    sd[key + ".weight_orig"] = _t(w)
    wm = np.ascontiguousarray(w.reshape(cout, -1).astype(np.float32)).astype(np.float64)
    u = normal(seed, key + ".weight_u", (cout,))
    u /= np.linalg.norm(u) + 1e-12
    v = None
    for _ in range(iters):
        v = wm.T @ u
        v /= np.linalg.norm(v) + 1e-12
        u = wm @ v
        u /= np.linalg.norm(u) + 1e-12
    sd[key + ".weight_u"] = _t(u)
    sd[key + ".weight_v"] = _t(v)


def _gn(sd, seed, key, c):
    if _SR_TRAINED[0]:
        sd[key + ".weight"] = _t(_GN_MEDIAN[0] * np.exp(0.45 * normal(seed, key + ".weight", (c,))))
        sd[key + ".bias"] = _t(0.3 * normal(seed, key + ".bias", (c,)))
        return
    sd[key + ".weight"] = _t(1.0 + 0.1 * normal(seed, key + ".weight", (c,)))
    sd[key + ".bias"] = _t(0.1 * normal(seed, key + ".bias", (c,)))


def _resblock(sd, seed, key, cin, cout):
    _gn(sd, seed, key + ".norm1", cin)
    _sn_conv(sd, seed, key + ".conv1", cout, cin)
    _gn(sd, seed, key + ".norm2", cout)
    _sn_conv(sd, seed, key + ".conv2", cout, cout)
    if cin != cout:
        bound = 1.0 / math.sqrt(cin)
        if _SR_TRAINED[0]:
            sd[key + ".conv_out.weight"] = _t(_heavy(seed, key + ".conv_out.weight", (cout, cin, 1, 1)) * (bound / math.sqrt(3.0)))
        else:
            sd[key + ".conv_out.weight"] = _t((uniform01(seed, key + ".conv_out.weight", (cout, cin, 1, 1)) * 2 - 1) * bound)
        sd[key + ".conv_out.bias"] = _t((uniform01(seed, key + ".conv_out.bias", (cout,)) * 2 - 1) * bound)


def make_sr_state_dict(seed=1234, outliers=None, regime="tame"):
    """Keys of ``TSPSRNet`` (models/networks.py:328-409).  ``outliers``: see _apply_outliers; a ``*.weight_orig`` entry scales that output
    row of a spectral-normalised conv before its u / v are power-iterated."""
    trained = _check_regime(regime)
    try:        # the regime / outlier switches of the helpers below are module state: restored on EVERY exit (an exception used to leave them set)
        _SR_TRAINED[0] = trained
        _SN_ROW_GAIN.clear()
        _SN_ROW_GAIN.update({k: v for k, v in (outliers or {}).items() if k.endswith(".weight_orig")})
        return _make_sr_state_dict(seed, outliers)
    finally:
        _SN_ROW_GAIN.clear()
        _SR_TRAINED[0] = False


def _make_sr_state_dict(seed, outliers):
    sd = {}
    D = 256
    _sn_conv(sd, seed, "conv_first_32.0", D // 4, 3)
    _sn_conv(sd, seed, "conv_first_16.0", D // 2, D // 4)
    _sn_conv(sd, seed, "conv_first_8.0", D, D // 2)
    _sn_conv(sd, seed, "conv_first_8.2", D, D)
    _sn_conv(sd, seed, "conv_body_16.0", D, D + D // 2)
    _sn_conv(sd, seed, "conv_body_16.2", D, D)
    _sn_conv(sd, seed, "conv_body_32.0", D, D + D // 4)
    _sn_conv(sd, seed, "conv_body_32.2", D, D)
    _sn_conv(sd, seed, "conv_up.1", D, D)
    _resblock(sd, seed, "conv_up.3", D, D)
    _sn_conv(sd, seed, "conv_up.4", D, D)
    _sn_conv(sd, seed, "conv_final.0", D // 2, D)
    _sn_conv(sd, seed, "conv_final.3", D // 4, D // 2)
    _resblock(sd, seed, "conv_final.5", D // 4, D // 4)
    _sn_conv(sd, seed, "conv_final.6", 3, D // 4)
    for s in ("32", "64"):
        _sn_conv(sd, seed, "conv_%s_scale.0" % s, D, D)
        _sn_conv(sd, seed, "conv_%s_scale.2" % s, D, D)
        _sn_conv(sd, seed, "conv_%s_shift.0" % s, D, D)
        _sn_conv(sd, seed, "conv_%s_shift.2" % s, D, D)
        _resblock(sd, seed, "conv_%s_fuse.0" % s, 2 * D, D)
    _sn_conv(sd, seed, "conv_32_to256.0", D, 512)
    _sn_conv(sd, seed, "conv_32_to256.2", D, D)
    missing = [k for k in _SN_ROW_GAIN if k not in sd]
    _apply_outliers(sd, outliers, skip=tuple(_SN_ROW_GAIN))
    if missing:
        raise KeyError("outlier keys %s not in this state_dict" % missing)
    return sd


# ----------------------------------------------------------------------------- inputs
def make_lq(seed, batch, content_widths=None):
    """LQ [B,3,32,512] in [-1,1]: uniform noise, 3x3 box low-pass, re-normalised per image to span
    [-1,1]; columns >= content width are -1 (test_sr.py:103-107 zero-pads *before* Normalize → -1)."""
    x = uniform01(seed, "lq", (batch, 3, 32, 512)) * 2 - 1
    p = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)), mode="edge")
    s = np.zeros_like(x)
    for dy in range(3):
        for dx in range(3):
            s += p[:, :, dy:dy + 32, dx:dx + 512]
    s /= 9.0
    mx = np.abs(s).reshape(batch, -1).max(axis=1).reshape(batch, 1, 1, 1)
    s = s / mx
    if content_widths is not None:
        for b, w in enumerate(content_widths):
            s[b, :, :, int(w):] = -1.0
    return _t(s)


def make_locs(n_glyphs, content_widths, max_glyphs=None):
    """locs[b,2c] = (c+0.5)/n·(w_b/512), locs[b,2c+1] = 0.5/n·(w_b/512)  (centre, half-width)/W —
    the meaning test_sr.py:121-135 gives to ``preds_locs``."""
    batch = len(n_glyphs)
    m = max_glyphs or max(n_glyphs)
    locs = np.zeros((batch, 2 * m), dtype=np.float32)
    for b in range(batch):
        n, w = n_glyphs[b], float(content_widths[b])
        for c in range(n):
            locs[b, 2 * c] = np.float32((c + 0.5) / n * (w / 512.0))
            locs[b, 2 * c + 1] = np.float32(0.5 / n * (w / 512.0))
    return torch.from_numpy(locs)


def make_labels(seed, n_total):
    return torch.from_numpy(integers(seed, "labels", (n_total, 1), 0, ALPHABET_SIZE))


def make_styles(seed, n):
    return _t(normal(seed, "styles", (n, 512)))
