"""One-time weight folding / repacking (host plumbing, runs once per (device, precision) after a
``load_state_dict`` / ``.to()``): spectral-norm sigma fold, EqualLinear / ModulatedConv scale fold,
OIHW fp32 → [O][KH][KW][I] in the compute dtype with channel padding, demodulation table.

Nothing here is on the per-forward hot path; it replaces work the reference redoes *every* forward
(spectral-norm ``W/σ`` — 211 ``addmv_`` + 275 ``div`` per SR forward, SURVEY.md §2a K18 — and the per-sample
weight modulation, models/networks.py:284-290).
"""
import math
import os
import warnings

import torch

# MNET_F16X2 ("split half", include/marconet_hip.h) tensors are tagged with torch.complex32: 4 bytes per LOGICAL element — a
# (hi, lo) pair of halves — so shapes, slicing along the outer dimensions, torch.cat / index_select over images and the caching
# allocator all work on logical NHWC shapes.  PyTorch never does arithmetic on them (only this package's kernels read the
# bytes; the hi / lo halves are blocked per 32 channels, not interleaved per element).
warnings.filterwarnings("ignore", message="ComplexHalf support is experimental")
SPLIT_DTYPE = torch.complex32
# MNET_F16M ("fp16+8", mode "fp16x2"; layout: mxfmt.py / include/marconet_hip.h) tensors are tagged with torch.uint32: again 4 bytes
# per logical element, blocked per 32 channels; PyTorch only allocates, slices and concatenates them.
MX_DTYPE = torch.uint32
SPLIT_WSCALE = 256.0          # MNET_SPLIT_WSCALE: split-half conv weights hold hi/lo of 256*W (the conv rescales by 2^-8)
PRECISIONS = ("fp32", "fp16", "fp16x3", "fp16x2")
HALF_PRECISIONS = ("fp16", "fp16x3", "fp16x2")        # modes whose stored values are bounded by the fp16 range


_TAG_GUARD = os.environ.get("MARCONET_TAG_GUARD", "1") != "0"


class BlockedTensor(torch.Tensor):
    """A tensor in one of the two 4-byte blocked storages (SPLIT_DTYPE / MX_DTYPE).  The dtype is only a TAG for bytes that this
    package's kernels interpret (32-channel blocks of halves / fp8 bytes / a scale exponent); any PyTorch arithmetic on them would be
    silently meaningless (``x + x`` on the complex32 tag adds the hi and lo halves as real and imaginary parts), so everything
    except storage plumbing — allocation, views, slicing, concatenation over outer dimensions, copies, device moves — raises."""

    _ALLOWED = frozenset((
        "__get__", "__getitem__", "__len__", "__repr__", "__str__", "__format__", "__deepcopy__", "__reduce_ex__", "__hash__",
        "view", "reshape", "contiguous", "flatten", "unsqueeze", "squeeze", "narrow", "select", "index_select", "cat", "stack",
        "chunk", "split", "unbind", "expand", "clone", "detach", "copy_", "to", "cpu", "cuda", "pin_memory", "data_ptr", "numel",
        "dim", "size", "stride", "element_size", "storage_offset", "untyped_storage", "is_contiguous", "get_device", "equal",
        "empty_like", "zeros_like", "new_empty", "new_zeros", "zero_", "record_stream", "is_floating_point", "is_complex",
        "is_pinned", "as_subclass", "_make_subclass", "requires_grad_", "is_shared", "share_memory_", "nbytes", "itemsize",
        "view_as", "reshape_as", "t_copy", "alias", "is_set_to", "_is_view", "is_same_size", "dim_order"))

    _LAYOUT_CHECKED = frozenset(("__getitem__", "view", "reshape", "flatten", "unsqueeze", "squeeze", "narrow", "select", "index_select",
                                 "chunk", "split", "unbind", "expand", "view_as", "reshape_as"))

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        name = getattr(func, "__name__", None) or str(func)
        if _TAG_GUARD and name == "to":          # device moves only: a dtype conversion would reinterpret the tag numerically
            for v in list(args[1:]) + list((kwargs or {}).values()):
                if isinstance(v, torch.dtype) and not is_split(v):
                    raise TypeError("marconet_amd: .to(%s) on a blocked-storage tensor — use ops.convert / packing.to_float" % v)
        if _TAG_GUARD and name not in cls._ALLOWED:
            raise TypeError("marconet_amd: %s() on a blocked-storage tensor (%s tag): these bytes are only meaningful to the HIP kernels "
                            "— convert with ops.convert(t, torch.float32) / packing.to_float(t) first" % (name, "split-half / fp16+8"))
        out = super().__torch_function__(func, types, args, kwargs or {})
        # plumbing is over OUTER dimensions only: the last dimension carries whole 32-channel / 128-byte blocks, so a result that is
        # still tagged must keep it (t[..., :8], t.narrow(-1, 0, 8), t.reshape(2, 192) would hand the kernels misaligned blocks)
        if _TAG_GUARD and name in cls._LAYOUT_CHECKED and args and isinstance(args[0], BlockedTensor):
            with torch._C.DisableTorchFunctionSubclass():
                last = args[0].shape[-1] if args[0].dim() else None
                for o in (out if isinstance(out, (tuple, list)) else (out,)):
                    if last is not None and isinstance(o, BlockedTensor) and is_split(o.dtype) and (
                            o.dim() == 0 or o.shape[-1] != last or (o.stride(-1) != 1 and o.shape[-1] > 1)):
                        raise TypeError("marconet_amd: %s() changed the channel dimension of a blocked-storage tensor (%d -> %s): blocks of "
                                        "32 channels / 128 bytes must stay whole — slice, view and concatenate over the outer dimensions only"
                                        % (name, last, tuple(o.shape)))
        # a view under another dtype (``t.view(torch.uint8)``, ``.view(torch.float16)``: raw bytes / halves for the host-side packers)
        # is an ordinary tensor again
        if isinstance(out, BlockedTensor):
            return out if is_split(out.dtype) else out.as_subclass(torch.Tensor)
        if isinstance(out, (tuple, list)):
            return type(out)(o.as_subclass(torch.Tensor) if isinstance(o, BlockedTensor) and not is_split(o.dtype) else o for o in out)
        return out


def tag(t):
    """mark a tensor of a blocked storage dtype (no copy); other tensors are returned unchanged"""
    return t.as_subclass(BlockedTensor) if is_split(t.dtype) and type(t) is torch.Tensor else t


def untag(t):
    """the same storage as a plain torch.Tensor (host-side packers / debugging helpers that do arithmetic on the raw halves)"""
    return t.as_subclass(torch.Tensor) if isinstance(t, BlockedTensor) else t


def new_tensor(shape, dtype, device, zero=False):
    """allocation of an activation / weight buffer in any storage dtype (blocked dtypes come back tagged)"""
    return tag((torch.zeros if zero else torch.empty)(tuple(shape), dtype=dtype, device=device))


def is_split(dtype):
    """the two 4-byte blocked storages (split half / fp16+8): 32-channel blocks, 128-byte aligned"""
    return dtype == SPLIT_DTYPE or dtype == MX_DTYPE


def default_precision():
    p = os.environ.get("MARCONET_PRECISION", "fp32").lower()
    if p not in PRECISIONS:
        raise ValueError("MARCONET_PRECISION must be one of %s, got %r" % (", ".join(PRECISIONS), p))
    return p


def torch_dtype(precision):
    """storage dtype of the activations / conv weights of a precision mode:
    fp32 — exact fp32 MFMA (parity mode); fp16 — one half per element (throughput mode, ~1e-2 deviation);
    fp16x3 — split half (hi, lo) per element, x*w = hi*hi + hi*lo + lo*hi on the fp16 MFMA: fp32-class accuracy (meets the
    1e-3 parity bar, argmax bit-exact) at a third of the fp16 MFMA rate instead of the fp32 MFMA's sixteenth;
    fp16x2 — "fp16+8": hi half + an e4m3 lo byte under a per-block scale, x*w = hi*hi (f16 MFMA) + one block-scaled fp8 MFMA for
    both correction products: 2 MFMA units per product, ~16 significant bits (≈3e-4 end to end: inside the 1e-3 bar)."""
    if precision not in PRECISIONS:
        raise ValueError("precision must be one of %s" % ", ".join(PRECISIONS))
    return {"fp32": torch.float32, "fp16": torch.float16, "fp16x3": SPLIT_DTYPE, "fp16x2": MX_DTYPE}[precision]


def rgb_pad(dtype):
    """channel count 3-channel tensors are carried with: one 16-byte chunk of halves / floats, or one 32-channel split block"""
    return 32 if is_split(dtype) else 8


def split_halves(t):
    """fp32 tensor [..., C] (C % 32 == 0) → split-half tensor of the same logical shape (host-side packing of weights):
    per 32-channel block 32 hi halves then 32 lo halves, hi = f16(v), lo = f16(v - hi)"""
    if t.shape[-1] % 32 != 0:
        raise ValueError("split-half tensors need a multiple of 32 channels, got %d" % t.shape[-1])
    t = t.detach().float().contiguous()
    if t.numel() and float(t.abs().max()) > 65000.0:
        raise OverflowError("value %.3g does not fit the half range of the fp16x3 mode" % float(t.abs().max()))
    hi = t.to(torch.float16)
    lo = (t - hi.float()).to(torch.float16)
    blk = t.shape[:-1] + (t.shape[-1] // 32, 32)
    both = torch.stack((hi.reshape(blk), lo.reshape(blk)), dim=-2).contiguous()          # [..., C/32, 2, 32] halves
    return tag(both.view(SPLIT_DTYPE).reshape(t.shape))


def to_float(t):
    """any storage dtype → fp32 (tests / debugging; host or device tensor)"""
    t = untag(t)
    if t.dtype == SPLIT_DTYPE:
        return unsplit_halves(t)
    if t.dtype == MX_DTYPE:
        from . import mxfmt
        return mxfmt.unpack_act(t.contiguous().view(torch.uint8), t.shape[-1])
    return t.float()


def from_float(t, dtype):
    """fp32 [..., C] → storage dtype (host-side reference packers; tests)"""
    if dtype == SPLIT_DTYPE:
        return split_halves(t)
    if dtype == MX_DTYPE:
        from . import mxfmt
        return tag(mxfmt.pack_act(t).view(MX_DTYPE))
    return t.to(dtype)


def unsplit_halves(t):
    """inverse of split_halves (tests / debugging): split-half tensor → fp32"""
    t = untag(t)
    c = t.shape[-1]
    h = t.contiguous().view(torch.float16).reshape(t.shape[:-1] + (c // 32, 2, 32)).float()
    return (h[..., 0, :] + h[..., 1, :]).reshape(t.shape)


def _round_up(v, m):
    return (v + m - 1) // m * m


def padded_cout(o, dtype, cout_mult=4):
    """output channels of pack_conv_weight(w [o, ...], dtype, cout_mult=...) (whole 32-channel blocks in the 4-byte blocked modes);
    for fp16+8 weights the packed tensor has MORE rows than this (the scale bytes): never read the channel count off its shape"""
    return _round_up(o, max(cout_mult, 32) if is_split(dtype) else cout_mult)


def pack_conv_weight(w, dtype, cin_mult=8, cout_mult=4, scale=1.0, sn=None):
    """w [O,I,KH,KW] fp32 → contiguous [O_pad,KH,KW,I_pad] in ``dtype`` (zero padded), every element (w / sigma) * scale.
    ``sn`` = (weight_u, weight_v): eval-mode old-style spectral norm, sigma = uᵀ(W_mat v) (models/networks.py:14).
    Parameters on a HIP device are packed by the library itself (mnet_pack_weights: no PyTorch arithmetic, no BLAS in the
    product process); host tensors (offline tooling, the CPU tests) take the equivalent torch path below."""
    o, i, kh, kw = w.shape
    if is_split(dtype):                 # whole 32-channel blocks on both sides
        cin_mult, cout_mult = max(cin_mult, 32), max(cout_mult, 32)
    op, ip = _round_up(o, cout_mult), _round_up(i, cin_mult)
    if w.is_cuda:
        from . import ops
        with ops.on_device(w):
            out = ops.pack_weights(w, dtype, op, ip, scale, None if sn is None else sn[0], None if sn is None else sn[1])
            if is_split(dtype):
                # same contract as the host path (split_halves raises OverflowError): 256 * W / sigma * scale must fit the half range.
                # Checked once per packing on the packed hi halves (bytes 0-63 of every 128-byte block)
                nblk = op * kh * kw * ip // 32
                hi = untag(out).contiguous().view(torch.uint8).reshape(-1)[: nblk * 128].reshape(nblk, 128)[:, :64].contiguous().view(torch.float16)
                if not bool(torch.isfinite(hi).all()):
                    raise OverflowError("a conv weight does not fit the half range of the %s mode (|256 W| >= 65504)"
                                        % ("fp16x3" if dtype == SPLIT_DTYPE else "fp16x2"))
            return out
    w = w.detach().float()
    if sn is not None:
        w = sn_fold(w, sn[0], sn[1])
    if scale != 1.0:
        w = w * scale
    if op != o or ip != i:
        wp = torch.zeros((op, ip, kh, kw), dtype=w.dtype, device=w.device)
        wp[:o, :i] = w
        w = wp
    if dtype == SPLIT_DTYPE:
        return split_halves(w.permute(0, 2, 3, 1).contiguous() * SPLIT_WSCALE)
    if dtype == MX_DTYPE:
        from . import mxfmt
        return mx_weight_tensor(mxfmt.pack_weight(w.permute(0, 2, 3, 1).contiguous()), op, kh, kw, ip)
    return w.permute(0, 2, 3, 1).contiguous().to(dtype)


def mx_weight_rows(cout_pad, kh, kw, cin_pad):
    """rows of the [rows, kh, kw, cin_pad] tensor that holds packed fp16+8 conv weights: cout_pad weight rows + the rows that
    carry the cout_pad per-channel scale bytes right behind them (include/marconet_hip.h)"""
    row_bytes = kh * kw * cin_pad * 4
    return cout_pad + (cout_pad + row_bytes - 1) // row_bytes


def mx_weight_tensor(flat_u8, cout_pad, kh, kw, cin_pad):
    """flat uint8 buffer of mxfmt.pack_weight → the [rows, kh, kw, cin_pad] MX_DTYPE tensor the conv wrappers take"""
    rows = mx_weight_rows(cout_pad, kh, kw, cin_pad)
    buf = torch.zeros((rows * kh * kw * cin_pad * 4,), dtype=torch.uint8, device=flat_u8.device)
    buf[: flat_u8.numel()] = flat_u8
    return tag(buf.view(MX_DTYPE).reshape(rows, kh, kw, cin_pad))


def pack_linear_weight(w, scale=1.0):
    """nn.Linear / EqualLinear weight [out,in] → fp32 contiguous, times the layer's constant scale (networks.py:180,192)"""
    if w.is_cuda:
        from . import ops
        with ops.on_device(w):
            return ops.pack_weights(w, torch.float32, scale=scale).reshape(w.shape[0], w.shape[1])
    return (w.detach().float() * scale).contiguous()


def pack_wsq(w, scale):
    """demodulation table [cin,cout] = Σ_k (scale·W[o,i,k])² of a ModulatedConv2d weight [cout,cin,kh,kw] (networks.py:284-287)"""
    if w.is_cuda:
        from . import ops
        with ops.on_device(w):
            return ops.pack_wsq(w, scale)
    ws = w.detach().float() * scale
    return (ws * ws).sum(dim=(2, 3)).t().contiguous()


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


# ------------------------------------------------------------------------------------------------ offline packed blob
# SURVEY.md §8(f) NEXT-3: the folded / repacked weights as a file, so that a serving process does not redo the spectral-norm
# fold and the layout change at start-up and does not need fp32 master weights resident twice while packing.  The file is a
# safetensors container: one entry per packed tensor, named "<root>[.<holder>]|<precision>|<path in the packed tree>", the tree
# structure and the non-tensor leaves as JSON metadata, and a SHA-256 of each holder's parameters so that a blob is never
# attached to different weights.
PACK_FORMAT = "marconet_amd.packed.v2"
PACK_LAYOUT = 6        # (6: ResTextBlockV2 emits '<block>.conv2+out', round 4; holders write every precision they have packed for) bump whenever any holder's _build() changes what it emits (keys, padding, layouts): a blob written by a
                       # different _build must not attach (it would fail with a KeyError mid-forward, or be read with another layout)


def _abi_version():
    from . import _lib
    return int(_lib.ABI_VERSION)


def _holders(root):
    """sub-modules that own a PackCache (ResNet45, TextViT, TextGenerator, TSPSRNet)"""
    return [(name, m) for name, m in root.named_modules() if isinstance(getattr(m, "_cache", None), PackCache) and hasattr(m, "_build")]


def _holder_precision(m):
    return getattr(m, "precision", "fp32")          # the TextViT has no precision switch: always fp32


def _weights_digest(m):
    import hashlib
    h = hashlib.sha256()
    for name, t in list(m.named_parameters()) + list(m.named_buffers()):
        h.update(("%s|%s|%s|" % (name, tuple(t.shape), t.dtype)).encode())
        h.update(t.detach().contiguous().cpu().numpy().tobytes())
    return h.hexdigest()


def _flatten(obj, path, tensors, tree):
    if torch.is_tensor(obj):
        split = is_split(obj.dtype)               # safetensors has no such dtype: stored as the raw halves, re-tagged on load
        tree[path] = {"t": "tensor", "split": split, "mx": obj.dtype == MX_DTYPE}
        t = untag(obj.detach()).contiguous().cpu()
        tensors[path] = (t.view(torch.float16) if split else t).clone()
    elif isinstance(obj, dict):
        if not all(isinstance(k, str) and "/" not in k for k in obj):
            raise TypeError("packed tree keys must be strings without '/' (at %r)" % path)
        tree[path] = {"t": "dict", "keys": list(obj)}
        for k, v in obj.items():
            _flatten(v, path + "/" + k, tensors, tree)
    elif isinstance(obj, (tuple, list)):
        tree[path] = {"t": "tuple" if isinstance(obj, tuple) else "list", "n": len(obj)}
        for i, v in enumerate(obj):
            _flatten(v, "%s/#%d" % (path, i), tensors, tree)
    elif obj is None or isinstance(obj, (bool, int, float, str)):
        tree[path] = {"t": "py", "v": obj}
    else:
        raise TypeError("cannot serialise %s in a packed tree (at %r)" % (type(obj).__name__, path))


def _unflatten(path, get_tensor, tree):
    node = tree[path]
    if node["t"] == "tensor":
        t = get_tensor(path)
        return tag(t.view(MX_DTYPE if node.get("mx") else SPLIT_DTYPE)) if node.get("split") else t
    if node["t"] == "dict":
        return {k: _unflatten(path + "/" + k, get_tensor, tree) for k in node["keys"]}
    if node["t"] in ("tuple", "list"):
        items = [_unflatten("%s/#%d" % (path, i), get_tensor, tree) for i in range(node["n"])]
        return tuple(items) if node["t"] == "tuple" else items
    return node["v"]


def save_packed(path, **roots):
    """save_packed("marconet.packed.safetensors", encoder=enc, gan=gan, sr=sr): packs (if not yet packed) every PackCache
    holder of the given modules in its current precision and writes one file."""
    import json
    from safetensors.torch import save_file
    tensors, trees, digests = {}, {}, {}
    for rname, root in roots.items():
        for hname, m in _holders(root):
            prec = _holder_precision(m)
            m._cache.get(m, prec, m._build)
            # the holder's own precision, plus every other precision it has ALREADY packed for (the batched driver runs the generator's
            # image-only levels in plain fp16 inside the fp16x2 / fp16x3 modes: that second pack belongs to the blob as well)
            digest = _weights_digest(m)
            for pr in [prec] + sorted(k for k in m._cache._store if k != prec):
                key = "%s|%s" % (rname + ("." + hname if hname else ""), pr)
                flat, tree = {}, {}
                _flatten(m._cache._store[pr], "", flat, tree)
                trees[key] = tree
                digests[key] = digest
                for p, t in flat.items():
                    tensors[key + "|" + p] = t
    if not tensors:
        raise ValueError("save_packed: no packed-weight holders in the given modules")
    save_file(tensors, path, metadata={"format": PACK_FORMAT, "layout": str(PACK_LAYOUT), "abi": str(_abi_version()),
                                       "trees": json.dumps(trees), "weights_sha256": json.dumps(digests)})
    return sorted(trees)


def load_packed(path, verify=True, **roots):
    """attach a blob written by save_packed to modules that carry the SAME weights (checked by SHA-256 unless verify=False);
    the next forward then uses the blob instead of re-packing.  Returns the list of holders that were attached."""
    import json
    from safetensors import safe_open
    attached = []
    with safe_open(path, framework="pt", device="cpu") as f:
        meta = f.metadata() or {}
        if meta.get("format") != PACK_FORMAT:
            raise ValueError("%s is not a %s file" % (path, PACK_FORMAT))
        if meta.get("layout") != str(PACK_LAYOUT) or meta.get("abi") != str(_abi_version()):
            raise ValueError("%s was written by packing layout %s / C-ABI %s; this build is layout %d / C-ABI %d — re-run save_packed"
                             % (path, meta.get("layout"), meta.get("abi"), PACK_LAYOUT, _abi_version()))
        trees, digests = json.loads(meta["trees"]), json.loads(meta["weights_sha256"])
        for rname, root in roots.items():
            for hname, m in _holders(root):
                prec = _holder_precision(m)
                key = "%s|%s" % (rname + ("." + hname if hname else ""), prec)
                if key not in trees:
                    raise KeyError("%s holds no packed weights for %s (has: %s)" % (path, key, ", ".join(sorted(trees))))
                if verify and _weights_digest(m) != digests[key]:
                    raise ValueError("packed blob %s was made from different weights than %s.%s carries" % (path, rname, hname))
                dev = next(m.parameters()).device
                sig = PackCache.signature(m)
                if m._cache._sig != sig:
                    m._cache._store, m._cache._sig = {}, sig
                stem = key.rsplit("|", 1)[0] + "|"
                for k2 in [key] + sorted(k for k in trees if k.startswith(stem) and k != key):     # further precisions of the same holder, if stored
                    m._cache._store[k2[len(stem):]] = _unflatten("", lambda p, k2=k2: f.get_tensor(k2 + "|" + p).to(dev), trees[k2])
                    attached.append(k2)
    return attached
