"""not-gpu: BASELINE configs[0] / SURVEY.md §8 a17 — the script plumbing around the networks.

 * marconet_amd/lq_io.py (vectorised host code of the product) against oracle/script_plumbing.py (the scripts' scalar
   statement, test_sr.py:24-35,98-135,198-201) on the reference's own test strips (tests/golden/pngs/), bit for bit;
 * the alphabet data file against its fingerprint (and against /root/reference/utils/alphabets.py when present);
 * the CPU oracle pushed through that plumbing against tests/golden/golden_png_v1.npz (generated from the REAL reference
   modules by tests/golden/make_golden_png.py): one PNG through the test_sr.py path on PyTorch-CPU, no GPU;
 * the drop-in ``models`` package: the reference's import line and constructor / load_state_dict block (test_sr.py:6,42-52)
   executed verbatim in a fresh interpreter.
"""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from marconet_amd import lq_io
from oracle import marconet_oracle as O
from oracle import script_plumbing as SP
from tests.golden import cases_png

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALPHABET_SHA256 = "0ab32c4bcaf8429550c5706a51174c6ca805ac0bccd4ace1aeab09d6946a89ae"


@pytest.fixture(scope="module")
def golden_png():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "golden_png_v1.npz")))


def _pngs():
    return sorted(f for f in os.listdir(cases_png.PNG_DIR) if f.endswith(".png"))


def test_alphabet_data_file():
    a = lq_io.alphabet()
    assert len(a) == 6735 and hashlib.sha256(a.encode()).hexdigest() == ALPHABET_SHA256
    ref = "/root/reference/utils/alphabets.py"
    if os.path.isfile(ref):
        ns = {}
        exec(open(ref, encoding="utf8").read(), ns)
        assert ns["alphabet"] == a
    t = "东北"                                               # two characters of the lqe01 strip's label
    assert lq_io.labels_from_text(t) == SP.get_labels_from_text(t, a) and min(lq_io.labels_from_text(t)) >= 0
    assert lq_io.labels_from_text("") == [-1]                  # alphabet.find -> -1 (the script then skips the strip)
    assert lq_io.text_from_labels(lq_io.labels_from_text(t)) == t == SP.get_text_from_labels(lq_io.labels_from_text(t), a)


@pytest.mark.parametrize("fname", _pngs())
def test_preprocessing_matches_script_statement(fname):
    img = lq_io.load_png(os.path.join(cases_png.PNG_DIR, fname))
    h, w, _ = img.shape
    lq, content_w, show_w = lq_io.lq_from_image(img)
    ref = SP.lq_tensor(img)
    assert lq.shape == (1, 3, 32, 512) and lq.dtype == torch.float32 and torch.equal(lq, ref)
    assert content_w == int(round(w * (32 / h))) and show_w == int(round(w * (128 / h)))
    if content_w < 512:
        assert float(lq[..., content_w:].max()) == -1.0               # the black canvas after Normalize
    for n in (1, 7, 16):
        boxes = lq_io.evenly_spaced_boxes(n, w, h)
        assert torch.equal(lq_io.locs_from_boxes(boxes, h), SP.preds_locs([np.array(b) for b in boxes], h))


def test_too_wide_strip_is_rejected_like_the_script():
    img = np.zeros((16, 16 * 17 + 1, 3), dtype=np.uint8)               # 32 x 546 after the resize
    assert SP.lq_tensor(img) is None
    with pytest.raises(lq_io.StripTooWide):
        lq_io.lq_from_image(img)


def test_resize_identity_and_constant():
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, size=(32, 77, 3)).astype(np.uint8)
    assert np.array_equal(lq_io.resize_cubic(img, 1.0, 1.0), img)      # scale 1: the cubic kernel interpolates its nodes
    flat = np.full((9, 30, 3), 137, dtype=np.uint8)
    assert (lq_io.resize_cubic(flat, 32 / 9, 32 / 9) == 137).all()      # fixed-point taps sum to exactly 2048


def test_panel_helpers_match_script_statement(tmp_path):
    """test_sr.py:207-232 (the saved visualisation): box marks, the structure-prior row squeezed to the preview's size, the stacking and
    the uint8 file — lq_io's vectorised forms against the scalar restatement in oracle/script_plumbing.py, on a reference strip with
    random "SR" and "prior" rows (the networks are not involved) and on marks that run over both edges of the preview"""
    rng = np.random.default_rng(5)
    fname = sorted(cases_png.SR_STRIPS.values())[0]
    s = lq_io.strip_from_png(os.path.join(cases_png.PNG_DIR, fname))
    n = int(s["labels"].shape[0])
    show = lq_io.show_lq(s["image"])
    assert show.shape == (128, s["show_w"], 3) and show.dtype == np.uint8
    # marks: the strip's own boxes, then boxes whose edges leave the preview on the left (negative slice ends: the script's own quirk) and right
    for locs in (s["locs"], torch.tensor([[0.0004, 0.002, 0.5, 0.01, 0.9995, 0.003]]), torch.tensor([[0.0, 0.004, 1.2, 0.05]])):
        k = locs.shape[1] // 2
        assert np.array_equal(lq_io.draw_locs(show, locs[0], k), SP.show_locs(show, locs, k))
    marked = lq_io.draw_locs(show, s["locs"][0], n)
    assert (marked != show).any() and (marked[:64, :, 0] == 255).any() and (marked[64:, :, 2] == 255).any()
    # bilinear squeeze of the prior row: shrink (the script's case), enlarge, identity
    prior_cha = torch.from_numpy(rng.uniform(-1, 1, (n, 3, 128, 128)).astype(np.float32))
    row = SP.prior_row(prior_cha)
    assert row.shape == (128, 128 * n, 3)
    for dw, dh in ((s["show_w"], 128), (128 * n + 37, 150), (128 * n, 128), (5, 3)):
        a, b = lq_io.resize_linear(row, dw, dh), SP.cv2_resize_linear_f32(row, dw, dh)
        assert a.shape == b.shape == (dh, dw, 3) and a.dtype == np.float32
        assert float(np.abs(a - b).max()) <= 2e-6
    assert np.array_equal(lq_io.resize_linear(row, 128 * n, 128), row)
    const = lq_io.resize_linear(np.full((7, 9, 3), 0.25, np.float32), 31, 17)
    assert float(np.abs(const - 0.25).max()) <= 1e-7
    # the stacked panel and the file
    show_sr = rng.uniform(0, 255, (128, 2048, 3)).astype(np.float32)
    pa = lq_io.panel(s["image"], s["locs"][0], n, show_sr, row)
    pb = SP.panel(show, s["locs"], n, show_sr[:, :show.shape[1], :], prior_cha)
    assert pa.shape == pb.shape == (4 * 128, s["show_w"], 3)
    assert float(np.abs(pa - pb).max()) <= 1e-3                      # (rows 1-3 identical; the prior row x255 carries the 2e-6 above)
    out = os.path.join(str(tmp_path), "panel.png")
    lq_io.save_panel(out, pa)
    back = lq_io.load_png(out)                                        # RGB
    assert np.array_equal(back[:, :, ::-1], SP.to_u8(pa))
    assert np.array_equal(back[:128], show)                           # the preview row survives the BGR round trip


@pytest.mark.parametrize("tag", list(cases_png.SR_STRIPS))
def test_png_through_cpu_path_vs_golden(tag, ckpts, golden_png):
    """configs[0]: one PNG through the test_sr.py path on PyTorch-CPU — product plumbing + CPU oracle vs the real reference"""
    s = lq_io.strip_from_png(os.path.join(cases_png.PNG_DIR, cases_png.SR_STRIPS[tag]))
    assert np.array_equal(s["labels"].numpy(), golden_png["sr.%s.labels" % tag])
    assert np.array_equal(s["locs"].numpy(), golden_png["sr.%s.locs" % tag])
    assert np.array_equal(np.rint((s["lq"][0].numpy() * 0.5 + 0.5) * 255).astype(np.uint8), golden_png["sr.%s.lq_u8" % tag])
    r = O.end_to_end(ckpts[0], ckpts[1], ckpts[2], s["lq"], [s["labels"]], s["locs"])
    assert np.array_equal(r["logits"].argmax(-1).numpy(), golden_png["sr.%s.argmax" % tag])
    assert float(np.abs(r["sr"][:, :, ::4, ::8].numpy() - golden_png["sr.%s.raw_s" % tag]).max()) <= 2e-4
    bgr = SP.postprocess(r["sr"])
    assert float(np.abs(cases_png.sample_bgr(bgr) - golden_png["sr.%s.bgr_f_s" % tag]).max()) <= 0.06      # 2e-4 * 255
    d = np.abs(cases_png.sample_bgr(SP.to_u8(bgr)).astype(int) - golden_png["sr.%s.bgr_u8_s" % tag].astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 0.01


def test_clear_labels_on_w_strip_vs_golden(ckpts, golden_png):
    """test_w.py:96-101 on Testsets/TestW/w1.png"""
    l1, _, _ = lq_io.lq_from_image(lq_io.load_png(os.path.join(cases_png.PNG_DIR, cases_png.W_STRIPS[0])))
    with torch.no_grad():
        logits, _, w1 = O.encoder_forward(ckpts[0], l1)
    assert [int(v) for v in O.clear_labels(logits[0])] == golden_png["w.labels"].tolist()
    assert float(np.abs(w1.numpy() - golden_png["w.w1"]).max()) <= 2e-4


DROPIN_SCRIPT = r'''
import sys, torch
sys.path.insert(0, sys.argv[1])
# ---- /root/reference/test_sr.py:6 and :42-52, verbatim
from models import networks, ocr
modelTSPGAN = networks.TSPGAN()
modelTSPGAN.load_state_dict(torch.load('./checkpoints/net_prior_generation.pth')['params'], strict=True)
modelTSPGAN.eval()

modelSR = networks.TSPSRNet()
modelSR.load_state_dict(torch.load('./checkpoints/net_sr.pth')['params'], strict=True)
modelSR.eval()

modelEncoder = networks.TextContextEncoderV2()
modelEncoder.load_state_dict(torch.load('./checkpoints/net_transformer_encoder.pth')['params'], strict=True)
modelEncoder.eval()
# ---- /root/reference/test_w.py:6 and models/networks.py:10,12,13 (what third-party code imports from the package)
from models import networks as n2
from models.textvit_arch import TextViT as TextEncoder
from models.resnet import resnet45stride as resnet45
from basicsr.ops.fused_act import FusedLeakyReLU, fused_leaky_relu
import marconet_amd.networks as hip
assert networks.TSPGAN is hip.TSPGAN and networks.TSPSRNet is hip.TSPSRNet and networks.TextContextEncoderV2 is hip.TextContextEncoderV2
assert networks.__file__.startswith(sys.argv[1]) and ocr.__file__.startswith(sys.argv[1])
assert fused_leaky_relu.__module__ == "marconet_amd.fused_act"
print("DROPIN-OK", type(modelSR).__module__, sum(p.numel() for p in modelTSPGAN.parameters()))
'''


def test_models_package_is_a_drop_in(ckpts, tmp_path):
    ck = tmp_path / "checkpoints"
    ck.mkdir()
    for name, sd in zip(("net_transformer_encoder", "net_prior_generation", "net_sr"), ckpts):
        torch.save({"params": sd}, str(ck / (name + ".pth")))
    r = subprocess.run([sys.executable, "-c", DROPIN_SCRIPT, ROOT], cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "DROPIN-OK marconet_amd.networks" in r.stdout
