"""-m gpu: BASELINE configs[0] on the HIP path — the reference's own test strips (tests/golden/pngs/) through the script
plumbing (marconet_amd/lq_io.py), the three HIP networks in the reference's call forms, and the K19 post-processing; against
tests/golden/golden_png_v1.npz (the REAL reference modules run on the same strips, tests/golden/make_golden_png.py) and
against the CPU oracle.  Also the test_w.py path on Testsets/TestW (clear_labels + style interpolation)."""
import os

import numpy as np
import pytest
import torch

from marconet_amd import lq_io
from oracle import marconet_oracle as O
from oracle import script_plumbing as SP
from tests.golden import cases_png

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def golden_png():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "golden_png_v1.npz")))


@pytest.fixture(scope="module")
def nets(ckpts):
    from models import networks                    # the drop-in package: test_sr.py:6
    enc, gan, sr = networks.TextContextEncoderV2(), networks.TSPGAN(), networks.TSPSRNet()
    enc.load_state_dict(ckpts[0], strict=True)
    gan.load_state_dict(ckpts[1], strict=True)
    sr.load_state_dict(ckpts[2], strict=True)
    return [m.eval().to(DEV).set_precision("fp32") for m in (enc, gan, sr)]


@pytest.mark.parametrize("tag", list(cases_png.SR_STRIPS))
def test_png_through_script_call_forms(tag, nets, golden_png):
    """test_sr.py:146-201 with the HIP modules, one strip per call exactly as the script does it"""
    enc, gan, sr = nets
    s = lq_io.strip_from_png(os.path.join(cases_png.PNG_DIR, cases_png.SR_STRIPS[tag]))
    LQ = s["lq"].to(DEV)
    with torch.no_grad():
        logits, _, w = enc(LQ)
        w0 = w[:1, ...].clone()
        labels = s["labels"]
        prior_cha, f64, f32 = gan(styles=w0.repeat(labels.size(0), 1), labels=labels, noise=None)
        y = sr(LQ, [f64], [f32], s["locs"].to(DEV))
    assert np.array_equal(logits.argmax(-1).cpu().numpy(), golden_png["sr.%s.argmax" % tag])
    err = float(np.abs(y[:, :, ::4, ::8].cpu().numpy() - golden_png["sr.%s.raw_s" % tag]).max())
    print("png %s: HIP fp32 vs real reference %.3e" % (tag, err))
    assert err <= 1e-3
    bgr = SP.postprocess(y)                        # the script's own post-processing on the HIP result
    d = np.abs(cases_png.sample_bgr(SP.to_u8(bgr)).astype(int) - golden_png["sr.%s.bgr_u8_s" % tag].astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 0.02


def test_png_batch_through_pipeline_vs_oracle(nets, ckpts):
    """both strips (+ one the script would skip) as one batch through MarconetPipeline.restore_strips → cropped uint8 BGR"""
    from marconet_amd.pipeline import MarconetPipeline
    pipe = MarconetPipeline(*nets, precision="fp32")
    strips = [lq_io.strip_from_png(os.path.join(cases_png.PNG_DIR, f)) for f in cases_png.SR_STRIPS.values()]
    bad = lq_io.strip_from_png(os.path.join(cases_png.PNG_DIR, "real_lq13.png"), text="ab")      # a character outside the alphabet
    assert int(bad["labels"].min()) == -1
    outs = pipe.restore_strips([strips[0], bad, strips[1]])
    assert outs[1] is None
    for s, o in zip(strips, (outs[0], outs[2])):
        r = O.end_to_end(ckpts[0], ckpts[1], ckpts[2], s["lq"], [s["labels"]], s["locs"])
        ref = SP.to_u8(SP.postprocess(r["sr"]))[:, :s["show_w"], :]
        assert o.shape == ref.shape == (128, s["show_w"], 3) and o.dtype == np.uint8
        d = np.abs(o.astype(int) - ref.astype(int))
        assert d.max() <= 1 and (d > 0).mean() < 0.02


def test_png_batch_with_structure_prior_row(nets, ckpts):
    """restore_strips(with_prior=True): next to ShowSR, the strip's structure images side by side — the array test_sr.py:208-211 builds from
    ``prior_cha`` for the panel's last row — against the oracle's generator images; ShowSR itself is the same bytes with and without it"""
    from marconet_amd.pipeline import MarconetPipeline
    pipe = MarconetPipeline(*nets, precision="fp32")
    strips = [lq_io.strip_from_png(os.path.join(cases_png.PNG_DIR, f)) for f in cases_png.SR_STRIPS.values()]
    plain = pipe.restore_strips(strips)
    outs = pipe.restore_strips(strips, with_prior=True)
    for s, o, q in zip(strips, outs, plain):
        show_sr, prior128 = o
        assert np.array_equal(show_sr, q)
        n = int(s["labels"].shape[0])
        r = O.end_to_end(ckpts[0], ckpts[1], ckpts[2], s["lq"], [s["labels"]], s["locs"])
        pc = (r["prior_images"][0] * 0.5 + 0.5).permute(0, 2, 3, 1).numpy()          # test_sr.py:208
        ref = pc[0]
        for i in range(1, len(pc)):                                                  # :209-211
            ref = np.hstack((ref, pc[i]))
        assert prior128.shape == ref.shape == (128, 128 * n, 3) and prior128.dtype == np.float32
        err = float(np.abs(prior128 - ref).max())
        print("prior row, %d glyphs: %.3e" % (n, err))
        assert err <= 5e-4                                                           # (x*0.5+0.5 halves the 1e-3 bar of the image)


def test_example_script_writes_the_script_panels(tmp_path):
    """examples/restore_strips.py -m on the reference's own strips: one panel PNG per strip under the script's file name (test_sr.py:232),
    4 x 128 rows — preview, box marks, the SR row MarconetPipeline.restore_strips gives in this process, the structure-prior row"""
    import subprocess
    import sys
    from marconet_amd import checkpoints
    from marconet_amd.pipeline import MarconetPipeline
    out = str(tmp_path / "panels")
    env = dict(os.environ)
    env.pop("MARCONET_CKPT_DIR", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "restore_strips.py"), "-i", cases_png.PNG_DIR, "-o", out, "-m", "--precision", "fp32"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    sde, sdg, sds, _ = checkpoints.load_state_dicts("")
    pipe = MarconetPipeline(*checkpoints.build_networks(sde, sdg, sds, DEV), precision="fp32")
    for fname in cases_png.SR_STRIPS.values():
        base = os.path.splitext(fname)[0]
        s = lq_io.strip_from_png(os.path.join(cases_png.PNG_DIR, fname))
        path = os.path.join(out, "%s_%s.png" % (base, s["text"]))
        assert os.path.isfile(path), (sorted(os.listdir(out)), r.stdout[-1500:])
        img = lq_io.load_png(path)                                               # RGB; the script's array is BGR
        assert img.shape == (4 * 128, s["show_w"], 3)
        show = lq_io.show_lq(s["image"])
        assert np.array_equal(img[:128], show)
        assert np.array_equal(img[128:256], lq_io.draw_locs(show, s["locs"][0], len(s["text"])))
        show_sr, prior128 = pipe.restore_strips([s], with_prior=True)[0]
        assert np.array_equal(img[256:384][:, :, ::-1], show_sr)
        want = SP.to_u8(lq_io.resize_linear(prior128, show.shape[1], 128) * 255)
        assert np.array_equal(img[384:][:, :, ::-1], want)


def test_example_w_script_writes_the_interpolation_frames(tmp_path):
    """examples/interpolate_w.py on Testsets/TestW: the script's eleven PNGs + w.gif (test_w.py:100-113); the middle frame against the same
    computation in this process (one generator call for all steps)"""
    import importlib.util
    import subprocess
    import sys
    from PIL import Image
    from marconet_amd import checkpoints
    out = str(tmp_path / "w")
    env = dict(os.environ)
    env.pop("MARCONET_CKPT_DIR", None)
    w1, w2 = (os.path.join(cases_png.PNG_DIR, f) for f in cases_png.W_STRIPS)
    script = os.path.join(ROOT, "examples", "interpolate_w.py")
    r = subprocess.run([sys.executable, script, "-w1", w1, "-w2", w2, "-o", out], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    names = sorted(os.listdir(out))
    assert names == sorted(["w_%.2f.png" % (i / 10) for i in range(11)] + ["w.gif"]), names
    spec = importlib.util.spec_from_file_location("interpolate_w_example", script)
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    sde, sdg, sds, _ = checkpoints.load_state_dicts("")
    enc, gan, _ = checkpoints.build_networks(sde, sdg, sds, DEV)
    enc.set_precision("fp32")
    gan.set_precision("fp32")
    rows = ex.frames(enc, gan, w1, w2, DEV)
    assert len(rows) == 11 and rows[0].shape[0] == 128 and rows[0].shape[1] % 128 == 0
    mid = lq_io.load_png(os.path.join(out, "w_0.50.png"))
    assert np.array_equal(mid[:, :, ::-1], SP.to_u8(rows[5] * 255.0))            # the RGB row written as if BGR (test_w.py:111)
    with Image.open(os.path.join(out, "w.gif")) as g:
        assert g.n_frames == 11 and g.size == (rows[0].shape[1], 128)


def test_w_strips_clear_labels_and_interpolation(nets, golden_png):
    """test_w.py:59-108 on Testsets/TestW/w1.png / w2.png"""
    from marconet_amd.pipeline import clear_labels_batch, w_interpolation
    enc, gan, _ = nets
    l1, _, _ = lq_io.lq_from_image(lq_io.load_png(os.path.join(cases_png.PNG_DIR, cases_png.W_STRIPS[0])))
    l2, _, _ = lq_io.lq_from_image(lq_io.load_png(os.path.join(cases_png.PNG_DIR, cases_png.W_STRIPS[1])))
    with torch.no_grad():
        p1, _, w1 = enc(l1.to(DEV))
        _, _, w2 = enc(l2.to(DEV))
    lab = clear_labels_batch(p1)[0]
    assert lab.flatten().tolist() == golden_png["w.labels"].tolist()
    assert float(np.abs(w1.cpu().numpy() - golden_png["w.w1"]).max()) <= 1e-3
    assert float(np.abs(w2.cpu().numpy() - golden_png["w.w2"]).max()) <= 1e-3
    labt = lab[:cases_png.W_MAX_GLYPHS]
    imgs = w_interpolation(gan, w1, w2, labt, steps=3)                # s = 0, 0.5, 1
    for k, scale in enumerate(cases_png.W_SCALES):
        err = float(np.abs(imgs[k][:, :, ::4, ::4].cpu().numpy() - golden_png["w.img_%.2f_s" % scale]).max())
        print("w interpolation %.2f: %.3e" % (scale, err))
        assert err <= 1e-3
