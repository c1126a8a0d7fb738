"""not-gpu: the C-ABI library loads, exports every symbol include/marconet_hip.h declares, the ctypes mirror
of mnet_conv_desc has the C layout, argument validation works without a device, the host-side module tree has
the reference's state_dict schema, and the product package never imports the oracle."""
import ctypes
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "marconet_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mnet_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from marconet_amd import _lib
    lib = _lib.load()
    names = _header_functions()
    assert len(names) >= 19
    for n in names:
        assert n in _lib.SYMBOLS, "header symbol %s has no ctypes binding" % n
        assert getattr(lib, n) is not None
    assert set(_lib.SYMBOLS) == set(names)
    assert lib.mnet_abi_version() == _lib.ABI_VERSION == 4


def test_conv_desc_layout_matches_c():
    """compile a tiny C program against the header and compare sizeof/offsetof with the ctypes mirror"""
    from marconet_amd._lib import ConvDesc
    code = r'''
#include <stdio.h>
#include <stddef.h>
#include "marconet_hip.h"
int main(void){ printf("%zu %zu %zu %zu %zu %zu\n", sizeof(mnet_conv_desc), offsetof(mnet_conv_desc, x1),
  offsetof(mnet_conv_desc, wgt), offsetof(mnet_conv_desc, in_scale), offsetof(mnet_conv_desc, residual), offsetof(mnet_conv_desc, y)); return 0; }
'''
    import tempfile
    d = tempfile.mkdtemp(prefix="mnet_abi_")
    cpath, exe = os.path.join(d, "abi_probe.c"), os.path.join(d, "abi_probe")
    open(cpath, "w").write(code)
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), cpath, "-o", exe])
    got = [int(v) for v in subprocess.check_output([exe]).split()]
    exp = [ctypes.sizeof(ConvDesc), ConvDesc.x1.offset, ConvDesc.wgt.offset, ConvDesc.in_scale.offset,
           ConvDesc.residual.offset, ConvDesc.y.offset]
    assert got == exp


def test_argument_validation_without_device():
    from marconet_amd import _lib
    lib = _lib.load()
    d = _lib.ConvDesc()
    assert lib.mnet_conv2d_nhwc(None, None) == -1
    assert lib.mnet_conv2d_nhwc(ctypes.byref(d), None) == -1 and b"null" in lib.mnet_last_error()
    d.dtype = 1; d.x0 = 16; d.wgt = 16; d.y = 16; d.n = d.h = d.w = d.ho = d.wo = 4
    d.c0 = 12; d.cout = 8; d.kh = d.kw = 3; d.stride_h = d.stride_w = 1; d.pad_h = d.pad_w = 1
    assert lib.mnet_conv2d_nhwc(ctypes.byref(d), None) == -2          # c0 % 8 != 0 → MNET_E_ALIGN
    d.c0 = 16; d.ho = 5
    assert lib.mnet_conv2d_nhwc(ctypes.byref(d), None) == -1          # inconsistent output size
    d.ho = 4
    assert lib.mnet_conv2d_flops(ctypes.byref(d)) == 2.0 * 4 * 4 * 4 * 8 * 9 * 16
    assert lib.mnet_layernorm(None, None, None, None, 1, 1, 1e-5, None) == -1
    # the plan query resolves kernels without launching: an f16 3x3 conv with cin % 64 == 0 and >= 65536 pixels
    d2 = _lib.ConvDesc()
    d2.dtype = 1; d2.x0 = 16; d2.wgt = 16; d2.y = 16; d2.n = 64; d2.h = d2.ho = 32; d2.w = d2.wo = 32
    d2.c0 = 64; d2.kh = d2.kw = 3; d2.stride_h = d2.stride_w = 1; d2.pad_h = d2.pad_w = 1
    d2.cout = 256
    assert lib.mnet_conv2d_plan(ctypes.byref(d2), _lib.ALGO_AUTO) == _lib.ALGO_DMA_CFG16           # 256x256 LDS-DMA tile, 8-wave pipelined form (id 16)
    assert lib.mnet_conv2d_plan(ctypes.byref(d2), _lib.ALGO_DMA_CFG0 + 8) == _lib.ALGO_DMA_CFG0 + 8
    assert lib.mnet_conv2d_plan(ctypes.byref(d2), _lib.ALGO_STRIP_CFG0) == _lib.ALGO_STRIP_CFG0    # strip form by explicit request
    d2.cout = 64
    assert lib.mnet_conv2d_plan(ctypes.byref(d2), _lib.ALGO_AUTO) == _lib.ALGO_STRIP_CFG0 + 1      # strip kernel, 64x512 tile
    d2.cout = 128
    assert lib.mnet_conv2d_plan(ctypes.byref(d2), _lib.ALGO_AUTO) == _lib.ALGO_DMA_CFG0 + 9        # 128x512 tile (same form)
    # split-half (fp16x3) launches: 128-byte aligned tensors, the 8-wave tiles; no strip kernel
    d2.dtype = 2; d2.x0 = d2.wgt = d2.y = 128
    assert lib.mnet_conv2d_plan(ctypes.byref(d2), _lib.ALGO_AUTO) == _lib.ALGO_DMA_CFG0 + 9
    d2.cout = 256
    assert lib.mnet_conv2d_plan(ctypes.byref(d2), _lib.ALGO_AUTO) == _lib.ALGO_DMA_CFG0 + 11
    d2.cout = 64
    assert lib.mnet_conv2d_plan(ctypes.byref(d2), _lib.ALGO_AUTO) == _lib.ALGO_STRIP_CFG0 + 1      # strip kernel, split-half form
    assert lib.mnet_conv2d_plan(ctypes.byref(d2), _lib.ALGO_DMA_CFG0 + 5) == _lib.ALGO_DMA_CFG0 + 5    # the per-tap 64x512 tile, pinned
    d2.c0 = 48
    assert lib.mnet_conv2d_plan(ctypes.byref(d2), _lib.ALGO_AUTO) == -2 and b"split-half" in lib.mnet_last_error()   # c0 % 32
    d2.c0 = 32
    assert lib.mnet_conv2d_plan(ctypes.byref(d2), _lib.ALGO_AUTO) == _lib.ALGO_STRIP_CFG0 + 1      # one 32-channel block per k-slab
    d2.x0 = 16
    assert lib.mnet_conv2d_plan(ctypes.byref(d2), _lib.ALGO_AUTO) == -2                             # 128-byte alignment
    d2.dtype = 1; d2.x0 = d2.wgt = d2.y = 16; d2.c0 = 64; d2.cout = 128
    d2.n = 4                                                                                       # 4096 pixels: small-launch tile
    assert lib.mnet_conv2d_plan(ctypes.byref(d2), _lib.ALGO_AUTO) == _lib.ALGO_DMA_CFG0 + 10       # 16 tiles of 128x256 → 128x128 tiles
    d2.n = 56                                                                                      # 57344 pixels: 224 tiles of 128x256
    assert lib.mnet_conv2d_plan(ctypes.byref(d2), _lib.ALGO_AUTO) == _lib.ALGO_DMA_CFG0 + 2
    d2.n = 4
    d2.c0 = 32
    assert lib.mnet_conv2d_plan(ctypes.byref(d2), _lib.ALGO_AUTO) == _lib.ALGO_REG_STAGED          # cin % 64 != 0
    assert lib.mnet_conv2d_plan(ctypes.byref(d2), _lib.ALGO_LDS_DMA) == -1
    d2.c0 = 64; d2.dtype = 0
    assert lib.mnet_conv2d_plan(ctypes.byref(d2), _lib.ALGO_AUTO) == _lib.ALGO_REG_STAGED          # fp32
    # fp32 1x1 over a few tokens (the TextViT linears of a small batch) → the skinny kernel; above 512 pixels the general one
    d3 = _lib.ConvDesc()
    d3.dtype = 0; d3.x0 = 16; d3.wgt = 16; d3.y = 16; d3.n = d3.h = d3.ho = 1; d3.w = d3.wo = 64
    d3.c0 = 512; d3.cout = 2048; d3.kh = d3.kw = 1; d3.stride_h = d3.stride_w = 1
    assert lib.mnet_conv2d_plan(ctypes.byref(d3), _lib.ALGO_AUTO) == _lib.ALGO_SKINNY
    assert lib.mnet_conv2d_plan(ctypes.byref(d3), _lib.ALGO_REG_STAGED) == _lib.ALGO_REG_STAGED
    d3.w = d3.wo = 4096
    assert lib.mnet_conv2d_plan(ctypes.byref(d3), _lib.ALGO_AUTO) == _lib.ALGO_REG_STAGED
    assert lib.mnet_conv2d_plan(ctypes.byref(d3), _lib.ALGO_SKINNY) == -1 and b"skinny" in lib.mnet_last_error()
    d3.w = 512; d3.h = 8; d3.wo = 64; d3.kh = d3.kw = d3.stride_h = d3.stride_w = 8                # the TextViT patch embedding of one strip
    assert lib.mnet_conv2d_splitk(ctypes.byref(d3), 32, None, None) == -1                         # no workspace
    assert lib.mnet_conv2d_splitk(ctypes.byref(d3), 12, 16, None) == -1 and b"multiple of kh" in lib.mnet_last_error()
    d3.stride_w = 4; d3.wo = 127
    assert lib.mnet_conv2d_splitk(ctypes.byref(d3), 32, 16, None) == -1 and b"filter == stride" in lib.mnet_last_error()
    # argument checks of the newer entry points (all before any HIP call)
    assert lib.mnet_conv3x3_rgb(16, 1, 1, 8, 8, 32, 16, 16, 4, 16, None, None) == -1 and b"cin" in lib.mnet_last_error()
    assert lib.mnet_conv3x3_rgb(16, 1, 1, 8, 8, 64, 16, 16, 4, None, None, None) == -1            # no output requested
    assert lib.mnet_sr_postprocess(None, 1, 16, 1, 10, 8, None) == -1
    assert lib.mnet_sr_postprocess(16, 1, 16, 1, 10, 2, None) == -1                                # fewer than 3 channels
    assert lib.mnet_adain_crop_concat_gn(16, 16, 16, 1, 1, 32, 256, 512, 16, 16, 16, 16, None, None, 1e-6, None, None, None) == -1
    assert lib.mnet_adain_crop_concat_split(16, 16, 16, 1, 1, 32, 256, 512, 16, 16, 16, 16, None, None, 1e-6, None, None, None, 16, 16, None) == -1
    assert lib.mnet_adain_crop_concat_split(16, 16, 16, 1, 1, 32, 256, 512, 16, 16, 16, 16, 16, None, 1e-6, None, None, 16, 16, 16, None) == -1
    assert b"go together" in lib.mnet_last_error()
    # round-2 entry points
    assert lib.mnet_pack_weights(16, 8, 8, 3, 3, 16, None, 1.0, 1, 8, 8, 16, None, None) == -1 and b"go together" in lib.mnet_last_error()
    assert lib.mnet_pack_weights(16, 8, 8, 3, 3, 16, 16, 1.0, 1, 8, 8, 16, None, None) == -1 and b"workspace" in lib.mnet_last_error()
    assert lib.mnet_pack_weights(16, 8, 8, 3, 3, None, None, 1.0, 2, 8, 8, 128, None, None) == -2            # split-half: cin_pad % 32
    assert lib.mnet_pack_weights(16, 8, 8, 3, 3, None, None, 1.0, 1, 4, 8, 128, None, None) == -1            # cout_pad < cout
    assert lib.mnet_pack_wsq(None, 8, 8, 9, 1.0, 16, None) == -1
    assert lib.mnet_gather_rows(16, 4, 8, 6, 4, None, 4, 16, None) == -1                                       # window beyond ld
    assert lib.mnet_style_rows(16, 4, 8, 6, 4, None, 4, 16, None, None, 0, None) == -1                          # window beyond ld
    assert lib.mnet_style_rows(16, 4, 8, 0, 4, None, 4, 16, None, 16, 0, None) == -1 and b"bcast" in lib.mnet_last_error()
    assert lib.mnet_demod_scaled(None, 16, 16, 1, 8, 8, None, None) == -1
    assert lib.mnet_convert(128, 0, 256, 2, 24, None) == -2 and b"split-half" in lib.mnet_last_error()        # count % 32


def test_ops_refuse_cpu_tensors():
    from marconet_amd import ops
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.upsample2x(torch.zeros(1, 2, 2, 8))


def test_state_dict_schema_matches_reference():
    from marconet_amd import networks
    schema = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_schema.json")))
    for name, ref in schema.items():
        m = getattr(networks, name)()
        got = {k: list(v.shape) for k, v in m.state_dict().items()}
        assert list(got) == list(ref["state_dict"]) and got == ref["state_dict"], name
        assert [n for n, _ in m.named_parameters()] == ref["parameters"]
        assert [n for n, _ in m.named_buffers()] == ref["buffers"]
        assert sum(p.numel() for p in m.parameters()) == ref["num_parameters"]      # printed by test_sr.py:59-61


def test_synthetic_checkpoints_load_strict(ckpts):
    from marconet_amd import networks
    for cls, sd in zip((networks.TextContextEncoderV2, networks.TSPGAN, networks.TSPSRNet), ckpts):
        m = cls()
        m.load_state_dict(sd, strict=True)
        m.eval()


def test_glyph_tables_match_oracle_windows():
    from marconet_amd.glyphs import window
    from oracle.marconet_oracle import glyph_window
    rng = np.random.default_rng(0)
    for feat_w, half in ((512, 16), (1024, 32)):
        for loc in np.concatenate([rng.random(500).astype(np.float32), np.float32([0, 1e-4, 0.03, 0.97, 0.9999, 1.0])]):
            x1, x2, y1, y2 = glyph_window(loc, feat_w, half)
            assert window(loc, feat_w, half) == (x1, x2 - x1, y1)


def test_product_never_imports_oracle():
    """the oracle is test infrastructure: nothing under marconet_amd/ may import or reference it"""
    for dp, _, files in os.walk(os.path.join(ROOT, "marconet_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f
                assert "F.conv2d" not in src and "torch.nn.functional" not in src, f
    code = "import sys; import marconet_amd.networks, marconet_amd.ops; assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules)"
    subprocess.check_call([sys.executable, "-c", code], cwd=ROOT)


def test_missing_library_fails_loudly():
    code = ("import os; os.environ['MARCONET_HIP_LIB']='/nonexistent/lib.so';"
            "from marconet_amd import _lib\n"
            "try:\n    _lib.load()\nexcept ImportError as e:\n    print('LOUD', e)\n")
    out = subprocess.check_output([sys.executable, "-c", code], cwd=ROOT).decode()
    assert "LOUD" in out and "no CPU/eager fallback" in out


def test_glyph_tables_vectorised_equals_scalar_window():
    """the numpy-vectorised GlyphTables == the scalar window arithmetic (models/networks.py:425-441) for random locations,
    ragged glyph counts (including zero) and both scales; out-of-map windows raise like the reference"""
    import numpy as np
    import pytest
    from marconet_amd.glyphs import GlyphTables, window
    rng = np.random.default_rng(7)
    for _ in range(100):
        B = int(rng.integers(1, 7))
        counts = [int(rng.integers(0, 9)) for _ in range(B)]
        W, half = ((512, 16), (1024, 32), (256, 16), (640, 32))[int(rng.integers(0, 4))]
        locs = rng.random((B, 16)).astype(np.float32)
        t = GlyphTables(locs, counts, W, half, "cpu")
        exp = [(b,) + window(locs[b, 2 * c], W, half) for b, n in enumerate(counts) for c in range(n)]
        assert list(zip(t.g_img.tolist(), t.g_x1.tolist(), t.g_w.tolist(), t.g_y1.tolist())) == exp
        assert t.g_start.tolist() == [0] + list(np.cumsum(counts)) and t.G == sum(counts)
    with pytest.raises(ValueError):
        GlyphTables(np.array([[1.2, 0.0]], dtype=np.float32), [1], 512, 16, "cpu")      # centre beyond the map → empty window
    with pytest.raises(IndexError):
        GlyphTables(np.zeros((1, 4), dtype=np.float32), [3], 512, 16, "cpu")


def test_graph_path_host_pieces():
    """host-side parts of pipeline.GraphedForward that need no GPU: static glyph tables are refreshed in place (same storage,
    new contents), a different glyph signature is refused, labels are validated before anything is enqueued"""
    import numpy as np
    import pytest
    import torch
    from marconet_amd import networks
    from marconet_amd.glyphs import GlyphTables
    from marconet_amd.pipeline import MarconetPipeline
    counts = [3, 0, 2]
    static = GlyphTables(np.full((3, 6), 0.5, np.float32), counts, 512, 16, "cpu")
    ptrs = [getattr(static, n).data_ptr() for n in ("g_img", "g_x1", "g_y1", "g_w", "g_start")]
    locs = np.random.default_rng(3).random((3, 6)).astype(np.float32)
    fresh = GlyphTables(locs, counts, 512, 16, "cpu")
    fresh.copy_into(static)
    assert [getattr(static, n).data_ptr() for n in ("g_img", "g_x1", "g_y1", "g_w", "g_start")] == ptrs
    assert all(torch.equal(getattr(static, n), getattr(fresh, n)) for n in ("g_img", "g_x1", "g_y1", "g_w", "g_start"))
    with pytest.raises(ValueError):
        GlyphTables(locs, [3, 1, 2], 512, 16, "cpu").copy_into(static)
    # mixed-width bucketing: centres as in the 512-padded run (centre_w), windows clipped to the bucket width
    rng = np.random.default_rng(11)
    lh = (rng.random((4, 8)) * (180.0 / 512.0)).astype(np.float32)            # all centres left of 180 px
    full = GlyphTables(lh, [4, 4, 4, 4], 512, 16, "cpu")
    buck = GlyphTables(lh, [4, 4, 4, 4], 192, 16, "cpu", centre_w=512)
    assert torch.equal(full.g_x1, buck.g_x1) and torch.equal(full.g_y1[full.g_x1 + 32 <= 192], buck.g_y1[full.g_x1 + 32 <= 192])
    assert int((buck.g_x1 + buck.g_w).max()) <= 192
    renorm = GlyphTables((lh * np.float32(512.0 / 192.0)).astype(np.float32), [4, 4, 4, 4], 192, 16, "cpu")     # the fp32 re-normalisation it replaces
    assert int((renorm.g_x1 - buck.g_x1).abs().max()) <= 1
    pipe = MarconetPipeline(networks.TextContextEncoderV2(), networks.TSPGAN(), networks.TSPSRNet())
    lab, img_of = pipe._host_prep([torch.tensor([5, 6, 7]), torch.zeros(0, dtype=torch.long), torch.tensor([1, 2])], counts, "cpu")
    assert lab.shape == (5, 1) and img_of.tolist() == [0, 0, 0, 2, 2]
    assert pipe._host_prep([torch.zeros(0, dtype=torch.long)], [0], "cpu") == (None, None)
    with pytest.raises(RuntimeError):
        pipe._host_prep([torch.tensor([6736])], [1], "cpu")


def test_header_is_plain_c():
    """include/marconet_hip.h must be consumable by a C host (cgo / JNI / N-API style bindings): C99, no HIP headers needed"""
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "use.c")
        with open(src, "w") as f:
            f.write('#include "marconet_hip.h"\nint main(void) { mnet_conv_desc d; (void)d; return mnet_abi_version != 0 ? 0 : 1; }\n')
        subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), src])


def test_packed_blob_round_trip(ckpts, tmp_path):
    """SURVEY §8(f) NEXT-3: save_packed → load_packed on fresh modules with the same weights reproduces the packed tree exactly and
    the modules then never call their packer; a blob is refused for different weights"""
    import pytest
    import torch
    from marconet_amd import networks
    from marconet_amd.packing import load_packed, save_packed

    def same(a, b):
        if torch.is_tensor(a):
            if not (torch.is_tensor(b) and a.dtype == b.dtype and a.shape == b.shape):
                return False
            raw = (lambda t: t.view(torch.float16)) if a.dtype == torch.complex32 else (lambda t: t)     # split-half: compare the halves
            return torch.equal(raw(a), raw(b))
        if isinstance(a, dict):
            return isinstance(b, dict) and list(a) == list(b) and all(same(a[k], b[k]) for k in a)
        if isinstance(a, (tuple, list)):
            return type(a) is type(b) and len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
        return a == b and type(a) is type(b)

    enc, sr = networks.TextContextEncoderV2().eval(), networks.TSPSRNet().eval()
    enc.load_state_dict(ckpts[0], strict=True)
    sr.load_state_dict(ckpts[2], strict=True)
    enc.set_precision("fp16")
    sr.set_precision("fp16x3")                       # split-half packs travel through the blob as raw halves
    path = str(tmp_path / "marconet.packed.safetensors")
    keys = save_packed(path, encoder=enc, sr=sr)
    assert keys == ["encoder.resnet|fp16", "encoder.transformer|fp32", "sr|fp16x3"]
    enc2, sr2 = networks.TextContextEncoderV2().eval(), networks.TSPSRNet().eval()
    enc2.load_state_dict(ckpts[0], strict=True)
    sr2.load_state_dict(ckpts[2], strict=True)
    enc2.set_precision("fp16")
    sr2.set_precision("fp16x3")
    assert sorted(load_packed(path, encoder=enc2, sr=sr2)) == keys

    def boom(_dtype):
        raise AssertionError("packer called although a blob is attached")
    for a, b, prec in ((enc.resnet, enc2.resnet, "fp16"), (enc.transformer, enc2.transformer, "fp32"), (sr, sr2, "fp16x3")):
        got = b._cache.get(b, prec, boom)
        assert same(a._cache.get(a, prec, boom), got)
    with torch.no_grad():
        sr2.conv_final[6].bias.add_(1.0)             # different weights → the blob must be refused (and, if forced, re-packed on use)
    with pytest.raises(ValueError):
        load_packed(path, sr=sr2)
    sr2.set_precision("fp32")
    with pytest.raises(KeyError):
        load_packed(path, verify=False, sr=sr2)      # no fp32 pack of the SR net in this blob
