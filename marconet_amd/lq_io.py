"""Host-side plumbing of the reference's scripts around the three networks (BASELINE configs[0]):
what test_sr.py / test_w.py do to a PNG before ``modelEncoder(LQ)`` and to the boxes / text that come with it —

    resize to height 32 (bicubic) → paste into a 32x512 black canvas → ToTensor → Normalize(0.5, 0.5)     test_sr.py:98-115
    character boxes → ``preds_locs`` (centre, half-width) / 512                                           test_sr.py:121-135
    text → class indices through the 6735-character alphabet (−1 for an unknown character)                test_sr.py:24-35
    strip file name ``<anything>_<text>.png`` → the manual label                                          test_sr.py:156-158

    the saved panel: preview | preview with box marks | SR | structure priors, stacked                   test_sr.py:203-232

Pure host code (numpy / PIL): nothing here touches the GPU.  The YOLO + OCR front-end that produces boxes and text in the
reference (utils/yolo_ocr_xloc.py) is outside the path (SURVEY.md §8f NEXT-4); ``evenly_spaced_boxes`` stands in for it where
a harness needs boxes (plumbing config 1).
"""
import os

import numpy as np
import torch

LQ_H, LQ_W = 32, 32 * 16          # test_sr.py:104: the canvas every strip is pasted into

_ALPHABET = None


def alphabet():
    """the reference's class order (utils/alphabets.py: 6735 characters; class 6735 = blank) — a data file here, because
    the checkpoints' 6736-way classifier and TextEmbeddings table are indexed by it"""
    global _ALPHABET
    if _ALPHABET is None:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "alphabet.txt"), encoding="utf8", newline="") as f:
            _ALPHABET = f.read()
    return _ALPHABET


def labels_from_text(text):
    """test_sr.py:24-29 (``alphabet.find``: −1 for a character outside the alphabet — the generator then raises, and the
    script's try/except skips the image, test_sr.py:181-190)"""
    a = alphabet()
    return [a.find(t) for t in text]


def text_from_labels(labels):
    """test_sr.py:31-35"""
    a = alphabet()
    return "".join(a[int(i)] for i in labels)


def manual_text(path):
    """test_sr.py:156-158 (``-m``): the text after the last underscore of the file's base name"""
    base = os.path.splitext(os.path.basename(path))[0]
    return base.split("_")[-1]


class StripTooWide(ValueError):
    """test_sr.py:108-110: a strip wider than 512 px at height 32 is skipped by the script ("crop it into shorter segments")"""


# ------------------------------------------------------------------------------------------------ cv2.resize(INTER_CUBIC)
def _cubic_taps(n_dst, n_src, inv_scale):
    """OpenCV's cubic resampling table for one axis of an 8-bit image: source index of the first of 4 taps (before border
    replication) and the taps as 11-bit fixed point (A = −0.75, centre-aligned sampling, no antialiasing)."""
    scale = 1.0 / inv_scale
    d = np.arange(n_dst, dtype=np.float64)
    f = (d + 0.5) * scale - 0.5
    s = np.floor(f).astype(np.int64)
    x = (f - s).astype(np.float32)
    A = np.float32(-0.75)
    one = np.float32(1)
    c0 = ((A * (x + one) - np.float32(5) * A) * (x + one) + np.float32(8) * A) * (x + one) - np.float32(4) * A
    c1 = ((A + np.float32(2)) * x - (A + np.float32(3))) * x * x + one
    c2 = ((A + np.float32(2)) * (one - x) - (A + np.float32(3))) * (one - x) * (one - x) + one
    c3 = one - c0 - c1 - c2
    taps = np.stack([c0, c1, c2, c3], axis=1).astype(np.float32) * np.float32(2048)
    taps = np.clip(np.rint(taps), -32768, 32767).astype(np.int64)            # saturate_cast<short>
    idx = np.clip(s[:, None] - 1 + np.arange(4)[None, :], 0, n_src - 1)      # BORDER_REPLICATE
    return idx, taps


def resize_cubic(img, fx, fy):
    """``cv2.resize(img, (0, 0), fx=fx, fy=fy, interpolation=cv2.INTER_CUBIC)`` for a uint8 HxWxC image, following OpenCV's
    8-bit algorithm (size = round-half-even(w·fx) x round-half-even(h·fy); separable 4-tap filter, 11-bit fixed-point taps,
    horizontal pass kept in integers, one rounding shift by 22 bits at the end).  cv2 is not installable here, so this
    resampler is UNPINNED against cv2 itself; parity of the networks does not depend on it (HIP path and oracle consume the
    same tensor)."""
    img = np.asarray(img)
    if img.dtype != np.uint8 or img.ndim != 3:
        raise TypeError("resize_cubic: uint8 HxWxC image expected")
    h, w, _ = img.shape
    dw, dh = int(np.rint(w * fx)), int(np.rint(h * fy))
    if dw <= 0 or dh <= 0:
        raise ValueError("resize_cubic: empty output")
    xi, xt = _cubic_taps(dw, w, fx)
    yi, yt = _cubic_taps(dh, h, fy)
    src = img.astype(np.int64)
    hor = (src[:, xi, :] * xt[None, :, :, None]).sum(axis=2)                 # [h, dw, c]
    ver = (hor[yi, :, :] * yt[:, :, None, None]).sum(axis=1)                 # [dh, dw, c]
    return np.clip((ver + (1 << 21)) >> 22, 0, 255).astype(np.uint8)


def lq_from_image(img):
    """test_sr.py:98-115 / test_w.py:59-73.  img: uint8 RGB HxWx3 → (LQ float32 [1,3,32,512] on the host, content width at
    height 32, width of the 128-px-high preview the script crops the SR result to)."""
    img = np.asarray(img)
    h, w, _ = img.shape
    lq = resize_cubic(img, 32 / h, 32 / h)
    show_w = int(np.rint(w * (128 / h)))
    if lq.shape[1] > LQ_W:
        raise StripTooWide("strip is %d px wide at height 32 (limit %d): crop it into shorter segments" % (lq.shape[1], LQ_W))
    canvas = np.zeros((LQ_H, LQ_W, 3), dtype=np.uint8)
    canvas[:, :lq.shape[1], :] = lq
    t = torch.from_numpy(canvas).permute(2, 0, 1).contiguous().to(torch.float32).div(255)     # ToTensor
    t = t.sub_(0.5).div_(0.5)                                                                 # Normalize((.5,.5,.5),(.5,.5,.5))
    return t.unsqueeze(0), int(lq.shape[1]), show_w


# ------------------------------------------------------------------------------------------------ the saved panel (test_sr.py:203-232)
def show_lq(img):
    """test_sr.py:99: ``ShowLQ`` — the strip at height 128 (uint8 RGB [128, show_w, 3]), the panel's first row and the crop width of the SR row"""
    img = np.asarray(img)
    return resize_cubic(img, 128 / img.shape[0], 128 / img.shape[0])


def resize_linear(img, dst_w, dst_h):
    """``cv2.resize(img, (dst_w, dst_h))`` (INTER_LINEAR, the default) for a float32 HxWxC image — test_sr.py:212 squeezes the row of
    structure images to the preview's size with it.  OpenCV's float path: sample position (d + 0.5)·(src/dst) − 0.5, two taps per
    axis, positions outside the image clamped to the border pixel, no antialiasing when shrinking, float32 arithmetic (horizontal
    pass first).  UNPINNED against cv2 (not installable here), like ``resize_cubic``; it only shapes the visualisation row."""
    img = np.asarray(img, dtype=np.float32)
    h, w, _ = img.shape

    def taps(n_dst, n_src):
        f = (np.arange(n_dst, dtype=np.float64) + 0.5) * (n_src / n_dst) - 0.5
        i0 = np.floor(f).astype(np.int64)
        t = (f - i0).astype(np.float32)
        t[i0 < 0] = 0.0
        i0 = np.maximum(i0, 0)
        t[i0 >= n_src - 1] = 0.0
        i0 = np.minimum(i0, n_src - 1)
        return i0, np.minimum(i0 + 1, n_src - 1), t

    x0, x1, tx = taps(int(dst_w), w)
    y0, y1, ty = taps(int(dst_h), h)
    hor = img[:, x0, :] * (np.float32(1) - tx)[None, :, None] + img[:, x1, :] * tx[None, :, None]
    return hor[y0] * (np.float32(1) - ty)[:, None, None] + hor[y1] * ty[:, None, None]


def draw_locs(show, locs, n, img_max_width=16 * 128):
    """test_sr.py:214-231: ``ShowLocs`` — a copy of the preview with, per character, a 4-px red mark at its left edge in the upper half
    and a 2-px blue mark at its right edge in the lower half (RGB order; edges = int(centre·2048) ∓ int(half-width·2048), the two
    products truncated separately as the script does).  ``locs``: the strip's ``preds_locs`` row [≥ 2n]."""
    out = np.array(show, copy=True)
    loc = np.asarray(locs, dtype=np.float32).reshape(-1)
    pad, padr = 2, 1
    for c in range(int(n)):
        centre, width = int(float(loc[2 * c]) * img_max_width), int(float(loc[2 * c + 1]) * img_max_width)
        x, y = centre - width, centre + width
        a, b = max(0, x - pad), min(x + pad, img_max_width)
        r, t = max(0, y - padr), min(y + padr, img_max_width)
        out[:64, a:b, 0], out[:64, a:b, 1], out[:64, a:b, 2] = 255, 0, 0
        out[64:, r:t, 0], out[64:, r:t, 1], out[64:, r:t, 2] = 0, 0, 255
    return out


def panel(image, locs, n, show_sr, prior128):
    """test_sr.py:207-232: the array the script hands to ``cv2.imwrite`` — float [4·128, show_w, 3] in cv2's BGR order: the preview, the
    preview with the box marks (both flipped RGB→BGR), ``ShowSR`` (already BGR, cropped to the preview's width), and the row of
    structure images resized to the preview's size ×255 (NOT flipped: the script stacks that RGB array as it is, :212,232).
    ``image``: the strip as loaded (uint8 RGB); ``show_sr`` / ``prior128``: ``MarconetPipeline.restore_strips(with_prior=True)``."""
    show = show_lq(image)
    prior = resize_linear(prior128, show.shape[1], show.shape[0]) * 255
    return np.vstack((show[:, :, ::-1], draw_locs(show, locs, n)[:, :, ::-1], np.asarray(show_sr)[:, :show.shape[1], :], prior))


def save_panel(path, bgr):
    """``cv2.imwrite(path, bgr)`` for that array: saturate to uint8 with round-half-to-even (cv2's ``saturate_cast<uchar>``), BGR on
    the way in → RGB file"""
    from PIL import Image
    u8 = np.clip(np.rint(np.asarray(bgr, dtype=np.float64)), 0, 255).astype(np.uint8)
    Image.fromarray(np.ascontiguousarray(u8[:, :, ::-1])).save(path)


def load_png(path):
    """uint8 RGB HxWx3 (the array get_yolo_ocr_xloc hands to the script, utils/yolo_ocr_xloc.py:37,103)"""
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"))


def locs_from_boxes(boxes, img_h, lq_width=LQ_W):
    """test_sr.py:121-135: boxes [[x1,y1,x2,y2], …] in the ORIGINAL image → preds_locs float32 [1, 2n]:
    (centre, half-width), scaled to height 32, divided by the canvas width.  Double-precision scalar arithmetic rounded to
    fp32 on assignment, as the script's Python floats are."""
    b = np.asarray(boxes, dtype=np.float64).reshape(-1, 4)
    centre = (b[:, 0] + b[:, 2]) / 2.0
    half = (b[:, 2] - b[:, 0]) / 2.0
    out = np.zeros((1, 2 * b.shape[0]), dtype=np.float32)
    out[0, 0::2] = (centre * 32.0 / img_h / lq_width).astype(np.float32)
    out[0, 1::2] = (half * 32.0 / img_h / lq_width).astype(np.float32)
    return torch.from_numpy(out)


def evenly_spaced_boxes(n, img_w, img_h):
    """n integer boxes tiling the strip left to right — the stand-in for the YOLO character detector"""
    edges = np.rint(np.linspace(0, img_w, n + 1)).astype(np.int64)
    return [[int(edges[i]), 0, int(edges[i + 1]), int(img_h)] for i in range(n)]


def strip_from_png(path, text=None, boxes=None):
    """one iteration of test_sr.py's loop up to the network inputs: → dict(lq, labels int64 [n,1], locs [1,2n], text, show_w).
    ``text`` defaults to the manual label in the file name, ``boxes`` to evenly spaced ones (one per character)."""
    img = load_png(path)
    h, w, _ = img.shape
    text = manual_text(path) if text is None else text
    if len(text) < 1:
        raise ValueError("no character given for %s (test_sr.py:168-170 skips such strips)" % path)
    boxes = evenly_spaced_boxes(len(text), w, h) if boxes is None else boxes
    lq, lq_w, show_w = lq_from_image(img)
    labels = torch.tensor(labels_from_text(text), dtype=torch.float32).type(torch.LongTensor).unsqueeze(1)   # test_sr.py:179
    return dict(lq=lq, labels=labels, locs=locs_from_boxes(boxes, h), text=text, content_w=lq_w, show_w=show_w, image=img)
