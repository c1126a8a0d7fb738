"""ctypes binding of libmarconet_hip.so (C-ABI declared in include/marconet_hip.h).

There is deliberately NO fallback: if the shared library is missing or a symbol cannot be resolved the
import of the ops fails loudly — the product path never routes through PyTorch eager ops or the CPU oracle.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MARCONET_HIP_LIB", os.path.join(_HERE, "lib", "libmarconet_hip.so"))

ABI_VERSION = 4               # MNET_ABI_VERSION of include/marconet_hip.h this binding was written against
MNET_F32, MNET_F16, MNET_F16X2, MNET_F16M = 0, 1, 2, 3
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_LRELU_SQRT2, ACT_TANH, ACT_GELU, ACT_SIGMOID = range(7)
ALGO_AUTO, ALGO_REG_STAGED, ALGO_LDS_DMA, ALGO_SKINNY, ALGO_DMA_CFG0, ALGO_STRIP_CFG0, ALGO_DMA_CFG16, ALGO_FLAG_ONE_TILE = 0, 1, 2, 3, 16, 32, 64, 256
ALGO_FLAG_X1_CENTER = 512     # x1 contributes through the filter's centre tap only (a 1x1 skip conv folded into the k-loop)

c_int, c_void_p, c_float, c_double, c_i64 = ctypes.c_int32, ctypes.c_void_p, ctypes.c_float, ctypes.c_double, ctypes.c_int64


class ConvDesc(ctypes.Structure):
    """mirror of ``mnet_conv_desc`` (include/marconet_hip.h)"""
    _fields_ = [
        ("dtype", c_int),
        ("x0", c_void_p), ("c0", c_int),
        ("x1", c_void_p), ("c1", c_int),
        ("n", c_int), ("h", c_int), ("w", c_int),
        ("wgt", c_void_p),
        ("cout", c_int), ("kh", c_int), ("kw", c_int), ("stride_h", c_int), ("stride_w", c_int),
        ("pad_h", c_int), ("pad_w", c_int),
        ("ho", c_int), ("wo", c_int),
        ("in_scale", c_void_p), ("in_shift", c_void_p), ("in_swish", c_int),
        ("valid_w", c_void_p),
        ("out_scale", c_void_p), ("bias", c_void_p),
        ("residual", c_void_p), ("res_mod", c_int),
        ("act", c_int),
        ("post_scale", c_void_p),
        ("y", c_void_p),
        ("gn_partial", c_void_p),
    ]


# name -> (restype, argtypes); every symbol include/marconet_hip.h declares
SYMBOLS = {
    "mnet_last_error": (ctypes.c_char_p, []),
    "mnet_abi_version": (c_int, []),
    "mnet_conv2d_nhwc": (c_int, [ctypes.POINTER(ConvDesc), c_void_p]),
    "mnet_conv2d_nhwc_ex": (c_int, [ctypes.POINTER(ConvDesc), c_int, c_void_p]),
    "mnet_conv2d_splitk": (c_int, [ctypes.POINTER(ConvDesc), c_int, c_void_p, c_void_p]),
    "mnet_conv2d_plan": (c_int, [ctypes.POINTER(ConvDesc), c_int]),
    "mnet_conv2d_flops": (c_double, [ctypes.POINTER(ConvDesc)]),
    "mnet_nchw_to_nhwc": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "mnet_nhwc_to_nchw": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "mnet_upsample2x_nhwc": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "mnet_upsample2x_scale_nhwc": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "mnet_upsample2x_convert_nhwc": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "mnet_affine_act_nhwc": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "mnet_groupnorm_affine": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                      c_float, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "mnet_groupnorm_affine_from_partial": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p]),
    "mnet_adain_crop_concat": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mnet_adain_crop_concat_gn": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p]),
    "mnet_adain_crop_concat_split": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                             c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                             c_void_p, c_void_p, c_int, c_void_p]),
    "mnet_glyph_scatter_affine": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                          c_void_p, c_void_p, c_void_p, c_void_p]),
    "mnet_layernorm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "mnet_token_mix": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                               c_int, c_float, c_void_p]),
    "mnet_attention": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "mnet_pixelnorm": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "mnet_embed_gather": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "mnet_embed_gather_scaled": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "mnet_demod": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "mnet_argmax_rows": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "mnet_convert": (c_int, [c_void_p, c_int, c_void_p, c_int, c_i64, c_void_p]),
    "mnet_fused_bias_act": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_int, c_int, c_float, c_float, c_void_p]),
    "mnet_torgb": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mnet_conv3x3_rgb": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "mnet_sr_postprocess": (c_int, [c_void_p, c_int, c_void_p, c_int, c_i64, c_int, c_void_p]),
    "mnet_nonfinite_flag": (c_int, [c_void_p, c_int, c_i64, c_void_p, c_void_p]),
    "mnet_pack_weights": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_int, c_int, c_int, c_void_p,
                                  c_void_p, c_void_p]),
    "mnet_pack_wsq": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    "mnet_demod_scaled": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "mnet_style_rows": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "mnet_gather_rows": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p]),
}

_lib = None


class MarconetHipError(RuntimeError):
    """Raised for a non-zero C-ABI status; test_sr.py's ``try/except ... continue`` (:181-190) still works."""


def load():
    """dlopen the library once and bind every declared symbol (raises if anything is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise ImportError(
            "marconet_amd: %s not found — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or marconet_amd/csrc/build.sh (there is no CPU/eager fallback)" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is absent
        fn.restype = res
        fn.argtypes = args
    if lib.mnet_abi_version() != ABI_VERSION:
        raise ImportError("marconet_amd: ABI version mismatch in %s" % LIB_PATH)
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        msg = load().mnet_last_error()
        raise MarconetHipError("%s failed (status %d): %s" % (what, status, msg.decode() if msg else "?"))
