#!/usr/bin/env python
"""The reference's test_w.py on the HIP path: the font style of two low-quality strips, interpolated in eleven steps, drawn as structure
images of the first strip's characters — ``w_0.00.png … w_1.00.png`` and ``w.gif`` under ``-o`` (test_w.py:100-113), same flags as the script
(``-w1 / -w2 / -o``, :117-121).

    python examples/interpolate_w.py -w1 <strip 1> -w2 <strip 2> -o <out dir> [--precision fp32]

What differs from the script: the eleven generator passes are ONE call (every glyph of every step is independent,
``pipeline.w_interpolation``); the frames of the GIF are written with PIL (imageio is not part of this image).  Kept as the script has it:
characters come from the encoder's own collapsed arg-max of strip 1 (:34-40,100-101); the PNGs are the RGB rows handed to ``cv2.imwrite`` as if
they were BGR (so red and blue swap in the files, :111), the GIF frames are the same rows truncated to uint8 (:110).
Weights: ``$MARCONET_CKPT_DIR`` (checkpoints/download_github.py's file names) or the seeded synthetic ones.  Needs the GPU."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from marconet_amd import checkpoints, lq_io                                      # noqa: E402
from marconet_amd.pipeline import clear_labels_batch, w_interpolation            # noqa: E402


def frames(enc, gan, path1, path2, dev, steps=11):
    """→ list of ``steps`` float32 arrays [128, 128·n, 3] (``prior128`` of test_w.py:105-109), RGB, in [0, 1] up to overshoot"""
    lq = []
    for p in (path1, path2):
        try:
            lq.append(lq_io.lq_from_image(lq_io.load_png(p))[0])
        except lq_io.StripTooWide as e:                                          # test_w.py:69-70,86-87: the script exits
            raise SystemExit("LQ width is not normal... %s" % e)
    with torch.no_grad():
        p1, _, w1 = enc(lq[0].to(dev))
        _, _, w2 = enc(lq[1].to(dev))
        lab = clear_labels_batch(p1)[0]
        if lab.numel() == 0:
            raise SystemExit("interpolate_w.py: the encoder reads no character in %s" % path1)
        imgs = w_interpolation(gan, w1, w2, lab, steps=steps)                    # [steps, n, 3, 128, 128]
    rows = (imgs * 0.5 + 0.5).permute(0, 3, 1, 4, 2).cpu().numpy()              # [steps, 128, n, 128, 3]
    return [np.ascontiguousarray(r.reshape(128, -1, 3)) for r in rows]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-w1", "--w1_path", type=str, default="./Testsets/TestW/w1.png")
    ap.add_argument("-w2", "--w2_path", type=str, default="./Testsets/TestW/w2.png")
    ap.add_argument("-o", "--save_path", type=str, default="./Testsets/TestW")
    ap.add_argument("--precision", default="fp32", choices=["fp16x2", "fp16x3", "fp16", "fp32"])
    a = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("interpolate_w.py: no GPU visible — this package has no CPU path")
    os.makedirs(a.save_path, exist_ok=True)
    sde, sdg, sds, source = checkpoints.load_state_dicts()
    print("%16s : %s" % ("Weights", source))
    enc, gan, _ = checkpoints.build_networks(sde, sdg, sds, "cuda")
    enc.set_precision(a.precision)
    gan.set_precision(a.precision)
    rows = frames(enc, gan, a.w1_path, a.w2_path, "cuda")
    from PIL import Image
    gif = []
    for i, row in enumerate(rows):
        scale = i / (len(rows) - 1)
        print("Interpolating w1 and w2 with weight %.2f" % scale)
        lq_io.save_panel(os.path.join(a.save_path, "w_%.2f.png" % scale), row * 255.0)     # test_w.py:111 (cv2.imwrite: the row read as BGR)
        gif.append(Image.fromarray((row * 255.0).astype(np.uint8)))                        # :110
    gif[0].save(os.path.join(a.save_path, "w.gif"), save_all=True, append_images=gif[1:], duration=100, loop=0)   # :112 (0.1 s per frame)
    print("Finishing interpolation.")


if __name__ == "__main__":
    main()
