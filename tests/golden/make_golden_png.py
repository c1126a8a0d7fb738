"""Generates tests/golden/golden_png_v1.npz: the reference's own test strips (tests/golden/pngs/, copied from
/root/reference/Testsets/{LQsWithText,TestW,LQs}) pushed through the script plumbing (oracle/script_plumbing.py restates
test_sr.py:98-135,198-200 / test_w.py:34-40,59-108) and the REAL reference modules (/root/reference/models, imported
unmodified) with the seeded synthetic checkpoints.  BASELINE configs[0] / SURVEY.md §8c "plumbing config 1".

    python tests/golden/make_golden_png.py        (build container only: needs /root/reference)

Boxes: evenly spaced, one per character of the manual label in the file name (the YOLO / OCR front-end is out of scope).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from marconet_amd import lq_io  # noqa: E402  (file / alphabet helpers only: host code)
from oracle import script_plumbing as SP  # noqa: E402
from oracle import synth  # noqa: E402
from oracle.ref_loader import load_reference_networks  # noqa: E402
from tests.golden import cases_png  # noqa: E402


def main():
    torch.manual_seed(0)
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    nw = load_reference_networks()
    sde, sdg, sds = synth.make_encoder_state_dict(), synth.make_gan_state_dict(), synth.make_sr_state_dict()
    enc, gan, sr = nw.TextContextEncoderV2().eval(), nw.TSPGAN().eval(), nw.TSPSRNet().eval()
    enc.load_state_dict(sde, strict=True)
    gan.load_state_dict(sdg, strict=True)
    sr.load_state_dict(sds, strict=True)
    alphabet = lq_io.alphabet()
    out = {}
    with torch.no_grad():
        for tag, fname in cases_png.SR_STRIPS.items():
            path = os.path.join(cases_png.PNG_DIR, fname)
            img = lq_io.load_png(path)
            h, w, _ = img.shape
            text = lq_io.manual_text(path)
            LQ = SP.lq_tensor(img)
            boxes = [np.array(b) for b in lq_io.evenly_spaced_boxes(len(text), w, h)]
            locs = SP.preds_locs(boxes, h)
            labels = torch.Tensor(SP.get_labels_from_text(text, alphabet)).type(torch.LongTensor).unsqueeze(1)
            logits, _, wv = enc(LQ)
            w0 = wv[:1, ...].clone()
            cha, f64, f32 = gan(styles=w0.repeat(labels.size(0), 1), labels=labels, noise=None)
            y = sr(LQ, [f64], [f32], locs)
            bgr = SP.postprocess(y)
            out["sr.%s.labels" % tag] = labels.numpy()
            out["sr.%s.locs" % tag] = locs.numpy()
            out["sr.%s.lq_u8" % tag] = np.rint((LQ[0].numpy() * 0.5 + 0.5) * 255).astype(np.uint8)
            out["sr.%s.argmax" % tag] = logits.argmax(-1).numpy()
            out["sr.%s.bgr_f_s" % tag] = cases_png.sample_bgr(bgr).astype(np.float32)
            out["sr.%s.bgr_u8_s" % tag] = cases_png.sample_bgr(SP.to_u8(bgr))
            out["sr.%s.raw_s" % tag] = y[:, :, ::4, ::8].contiguous().numpy()
        # test_w.py: two styles, labels of strip 1 from clear_labels, interpolation at three weights
        l1 = SP.lq_tensor(lq_io.load_png(os.path.join(cases_png.PNG_DIR, cases_png.W_STRIPS[0])))
        l2 = SP.lq_tensor(lq_io.load_png(os.path.join(cases_png.PNG_DIR, cases_png.W_STRIPS[1])))
        p1, _, w1 = enc(l1)
        _, _, w2 = enc(l2)
        idx = torch.max(p1[0].detach(), 1)[1]
        lab = [int(idx[i]) for i in range(idx.size(0)) if (not (i > 0 and idx[i - 1] == idx[i])) and idx[i] < len(alphabet)]
        out["w.labels"] = np.array(lab, dtype=np.int64)
        out["w.w1"], out["w.w2"] = w1.numpy(), w2.numpy()
        labt = torch.Tensor(lab[:cases_png.W_MAX_GLYPHS]).type(torch.LongTensor).unsqueeze(1)
        for scale in cases_png.W_SCALES:
            new_w = w1 * scale + w2 * (1 - scale)
            cha, _, _ = gan(styles=new_w.repeat(labt.size(0), 1), labels=labt, noise=None)
            out["w.img_%.2f_s" % scale] = cha[:, :, ::4, ::4].contiguous().numpy()
    path = os.path.join(ROOT, "tests", "golden", "golden_png_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB;", len(out), "arrays; w labels:", len(lab))


if __name__ == "__main__":
    main()
