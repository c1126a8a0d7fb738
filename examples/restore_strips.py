#!/usr/bin/env python
"""The reference's test_sr.py as ONE batch per call on the HIP path: a directory of low-quality text strips in, one panel PNG per strip out
(preview | box marks | super-resolved strip | structure priors — the file test_sr.py:232 writes, under the same name).

    python examples/restore_strips.py -i <strips dir> -o <out dir> [-m] [--precision fp16x2] [--batch 64]

Same flags as the script (-i / -o / -m, test_sr.py:236-241).  What differs, and why:
  * the YOLO character detector and the modelscope OCR (test_sr.py:55-56,86-96) are not part of this build (SURVEY.md §8f NEXT-4).  With
    ``-m`` the text comes from the file name as in the script (:156-158) and the boxes are evenly spaced over the strip; without it both
    come from the encoder itself (its class logits and its (left, right) predictions: MarconetPipeline.forward_blind's sources);
  * weights: the three checkpoints are looked for in ``$MARCONET_CKPT_DIR`` (the names of checkpoints/download_github.py); without them
    the seeded synthetic weights run — the panel then shows the plumbing, not a restoration;
  * all strips of a batch go through the three networks in one call (the script: one strip at a time, :77).
Needs the GPU (there is no CPU path in this package)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from marconet_amd import checkpoints, lq_io                                      # noqa: E402
from marconet_amd.pipeline import MarconetPipeline, clear_labels_batch, locs_from_left_right   # noqa: E402


def manual_strips(paths):
    """-m: text after the last underscore of the file name (test_sr.py:156-158), one evenly spaced box per character"""
    out = []
    for p in paths:
        try:
            out.append(lq_io.strip_from_png(p))
        except lq_io.StripTooWide as e:                                          # test_sr.py:108-110
            print("Warning!!! %s: %s" % (os.path.basename(p), e))
            out.append(None)
    return out


def blind_strips(pipe, paths, dev, max_glyphs=16):
    """no -m: labels = the encoder's collapsed arg-max sequence (test_w.py:34-40), boxes = its (left, right) predictions"""
    pre, out = [], []
    for p in paths:
        img = lq_io.load_png(p)
        try:
            pre.append((img,) + tuple(lq_io.lq_from_image(img)))
        except lq_io.StripTooWide as e:
            print("Warning!!! %s: %s" % (os.path.basename(p), e))
            pre.append(None)
    ok = [i for i, v in enumerate(pre) if v is not None]
    if not ok:
        return [None] * len(paths)
    with torch.no_grad():
        logits, locs_lr, _ = pipe.encoder(torch.cat([pre[i][1] for i in ok]).to(dev))
    labels = clear_labels_batch(logits)
    locs = locs_from_left_right(locs_lr).float().cpu()
    k = 0
    for i, v in enumerate(pre):
        if v is None:
            out.append(None)
            continue
        lab = labels[k][:max_glyphs]
        n = int(lab.shape[0])
        out.append(dict(lq=v[1], labels=lab, locs=locs[k:k + 1, :2 * n].contiguous(), text=lq_io.text_from_labels(lab.flatten().tolist()),
                        content_w=v[2], show_w=v[3], image=v[0]))
        k += 1
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-i", "--test_path", type=str, default="./Testsets/LQs")
    ap.add_argument("-o", "--save_path", type=str, default=None)
    ap.add_argument("-m", "--manual", action="store_true")
    ap.add_argument("--precision", default="fp16x2", choices=["fp16x2", "fp16x3", "fp16", "fp32"])
    ap.add_argument("--batch", type=int, default=64, help="strips per call")
    a = ap.parse_args()
    save_path = a.save_path or a.test_path.rstrip("/") + "_" + time.strftime("%m-%d_%H-%M", time.localtime()) + "_MARCONet"
    os.makedirs(save_path, exist_ok=True)
    if not torch.cuda.is_available():
        raise SystemExit("restore_strips.py: no GPU visible — this package has no CPU path")
    dev = "cuda"
    sde, sdg, sds, source = checkpoints.load_state_dicts()
    print("%28s : %s" % ("Weights", source))
    pipe = MarconetPipeline(*checkpoints.build_networks(sde, sdg, sds, dev), precision=a.precision)
    names = sorted(f for f in os.listdir(a.test_path) if f.lower().endswith((".png", ".jpg", ".jpeg", ".bmp")))
    for s0 in range(0, len(names), a.batch):
        chunk = names[s0:s0 + a.batch]
        paths = [os.path.join(a.test_path, f) for f in chunk]
        strips = manual_strips(paths) if a.manual else blind_strips(pipe, paths, dev)
        live = [i for i, s in enumerate(strips) if s is not None and s["labels"].numel() > 0]
        for i, s in enumerate(strips):
            if s is not None and s["labels"].numel() == 0:
                print("Warning!!! No character is detected in %s. Continue..." % chunk[i])          # test_sr.py:168-170
        res = pipe.restore_strips([strips[i] for i in live], with_prior=True)
        for i, r in zip(live, res):
            s = strips[i]
            if r is None:                                                        # a character outside the alphabet (test_sr.py:181-190)
                print("Error in %s (a character outside the alphabet). Continue..." % chunk[i])
                continue
            show_sr, prior128 = r
            n = int(s["labels"].shape[0])
            base = os.path.splitext(chunk[i])[0]
            out = os.path.join(save_path, "%s_%s.png" % (base, s["text"]))       # test_sr.py:232
            lq_io.save_panel(out, lq_io.panel(s["image"], s["locs"][0], n, show_sr, prior128))
            print("Restoring %s. Using %s text: %s -> %s" % (chunk[i], "given" if a.manual else "predicted", s["text"], out))


if __name__ == "__main__":
    main()
