"""TEST INFRASTRUCTURE ONLY — never imported by the product path (marconet_amd/).

CPU restatement of the host-side plumbing of the reference's scripts (SURVEY.md §8 a17, §8c "plumbing config 1"), written
the way the scripts write it (scalar Python / torch element assignment) so that the vectorised product code in
marconet_amd/lq_io.py can be checked against an independent statement:

    get_labels_from_text / get_text_from_labels      /root/reference/test_sr.py:24-35
    resize → canvas → ToTensor → Normalize            /root/reference/test_sr.py:98-115
    preds_locs from the detector's boxes              /root/reference/test_sr.py:121-135
    output post-processing                            /root/reference/test_sr.py:198-201
    the saved panel (box marks, prior row, stacking)  /root/reference/test_sr.py:207-232

``cv2.resize(..., INTER_CUBIC)`` is third-party arithmetic that is absent here (cv2 is not installed, un-pinned in
requirements.txt): it is restated from OpenCV's published 8-bit algorithm (4-tap cubic, A = −0.75, 11-bit fixed-point taps,
BORDER_REPLICATE, rounding shift by 22) as plain scalar loops — parity UNPINNED at that one call; everything downstream of
the resampled uint8 strip is pinned to the reference modules run here.
"""
import math

import numpy as np
import torch


def get_labels_from_text(text, alphabet):
    labels = []
    for t in text:
        labels.append(alphabet.find(t))
    return labels


def get_text_from_labels(preds, alphabet):
    s = ""
    for i in range(len(preds)):
        s = s + alphabet[preds[i]]
    return s


def _round_half_even(v):
    return int(round(v))          # Python's round() is round-half-even, like cvRound


def _taps(x):
    A = np.float32(-0.75)
    x = np.float32(x)
    one = np.float32(1)
    c = [((A * (x + one) - np.float32(5) * A) * (x + one) + np.float32(8) * A) * (x + one) - np.float32(4) * A,
         ((A + np.float32(2)) * x - (A + np.float32(3))) * x * x + one,
         ((A + np.float32(2)) * (one - x) - (A + np.float32(3))) * (one - x) * (one - x) + one]
    c.append(one - c[0] - c[1] - c[2])
    return [max(-32768, min(32767, int(np.rint(np.float32(v) * np.float32(2048))))) for v in c]


def cv2_resize_cubic_u8(img, fx, fy):
    h, w, ch = img.shape
    dw, dh = _round_half_even(w * fx), _round_half_even(h * fy)
    xs, ys = [], []
    for d in range(dw):
        f = (d + 0.5) * (1.0 / fx) - 0.5
        s = math.floor(f)
        xs.append((s, _taps(f - s)))
    for d in range(dh):
        f = (d + 0.5) * (1.0 / fy) - 0.5
        s = math.floor(f)
        ys.append((s, _taps(f - s)))
    src = img.astype(np.int64)
    hor = np.zeros((h, dw, ch), dtype=np.int64)
    for d, (s, t) in enumerate(xs):
        for k in range(4):
            hor[:, d, :] += src[:, min(max(s - 1 + k, 0), w - 1), :] * t[k]
    out = np.zeros((dh, dw, ch), dtype=np.uint8)
    for d, (s, t) in enumerate(ys):
        acc = np.zeros((dw, ch), dtype=np.int64)
        for k in range(4):
            acc += hor[min(max(s - 1 + k, 0), h - 1)] * t[k]
        out[d] = np.clip((acc + (1 << 21)) >> 22, 0, 255).astype(np.uint8)
    return out


def lq_tensor(img):
    """test_sr.py:98-115 on a uint8 RGB image → LQ [1,3,32,512] (or None where the script prints its warning and continues)"""
    h, w, c = img.shape
    LQ = cv2_resize_cubic_u8(img, 32 / h, 32 / h)
    bg = np.zeros((32, 32 * 16, 3)).astype(LQ.dtype)
    if LQ.shape[-2] <= 32 * 16:
        bg[:, :LQ.shape[-2], :] = bg[:, :LQ.shape[-2], :] + LQ
        LQ = bg
    else:
        return None
    t = torch.from_numpy(LQ.transpose(2, 0, 1).copy()).float().div(255)        # transforms.ToTensor()
    t = (t - 0.5) / 0.5                                                        # transforms.Normalize(0.5, 0.5)
    return t.unsqueeze(0)


def preds_locs(recognized_boxes, h, lq_width=512):
    """test_sr.py:121-135"""
    num_boxes = len(recognized_boxes)
    out = torch.zeros(1, num_boxes * 2).float()
    for i, box in enumerate(recognized_boxes):
        x1, y1, x2, y2 = box
        center = (x1 + x2) / 2.0
        width = (x2 - x1) / 2.0
        center_norm = center * 32.0 / h
        width_norm = width * 32.0 / h
        out[0, 2 * i] = center_norm / lq_width
        out[0, 2 * i + 1] = width_norm / lq_width
    return out


def postprocess(sr):
    """test_sr.py:198-200 → float HxWx3 BGR in [0,255] (what the script hands to cv2.imwrite)"""
    sr = sr * 0.5 + 0.5
    sr = sr.squeeze(0).permute(1, 2, 0).flip(2)
    return np.clip(sr.float().cpu().numpy(), 0, 1) * 255.0


def to_u8(img):
    """cv2.imwrite's float → uint8 conversion: saturate_cast<uchar> = round half to even"""
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def prior_row(prior_cha):
    """test_sr.py:208-211: the generator's images [n,3,128,128] → one float array [128, 128·n, 3], side by side"""
    pc = (prior_cha * 0.5 + 0.5).permute(0, 2, 3, 1).cpu().numpy()
    row = pc[0]
    for i in range(1, len(pc)):
        row = np.hstack((row, pc[i]))
    return row


def cv2_resize_linear_f32(img, dst_w, dst_h):
    """cv2.resize(img, (dst_w, dst_h)) with the default INTER_LINEAR on a float32 image (test_sr.py:212), restated from OpenCV's
    published float algorithm as two dense interpolation matrices (rows: output positions; two non-zero weights each; sample position
    (d + 0.5)·scale − 0.5, clamped to the border) — UNPINNED against cv2 like the cubic resampler above."""
    img = np.asarray(img, dtype=np.float32)
    h, w, c = img.shape

    def matrix(n_dst, n_src):
        m = np.zeros((n_dst, n_src), dtype=np.float32)
        scale = n_src / n_dst
        for d in range(n_dst):
            f = (d + 0.5) * scale - 0.5
            i = math.floor(f)
            t = f - i
            if i < 0:
                i, t = 0, 0.0
            if i >= n_src - 1:
                i, t = n_src - 1, 0.0
            m[d, i] += np.float32(1.0) - np.float32(t)
            if t:
                m[d, i + 1] += np.float32(t)
        return m

    mx, my = matrix(dst_w, w), matrix(dst_h, h)
    hor = np.einsum("dw,hwc->hdc", mx, img)
    return np.einsum("eh,hdc->edc", my, hor)


def show_locs(show, preds_locs, n_chars):
    """test_sr.py:214-231, one column at a time: which columns of the upper half turn red (the 4 px around each character's left edge)
    and which of the lower half turn blue (the 2 px around its right edge); RGB order, as ShowLQ is"""
    out = show.copy()
    width_limit = 16 * 128
    cols = out.shape[1]
    for c in range(n_chars):
        centre = int(preds_locs[0][2 * c].item() * width_limit)
        half = int(preds_locs[0][2 * c + 1].item() * width_limit)
        left, right = centre - half, centre + half
        red = range(cols)[max(0, left - 2):min(left + 2, width_limit)]
        blue = range(cols)[max(0, right - 1):min(right + 1, width_limit)]
        for col in red:
            out[:64, col] = (255, 0, 0)
        for col in blue:
            out[64:, col] = (0, 0, 255)
    return out


def panel(show, preds_locs, n_chars, show_sr, prior_cha):
    """test_sr.py:207-232: the stacked array handed to cv2.imwrite (BGR): preview, box marks, SR row, prior row"""
    prior = cv2_resize_linear_f32(prior_row(prior_cha), show.shape[1], show.shape[0]) * 255
    return np.vstack((show[:, :, ::-1], show_locs(show, preds_locs, n_chars)[:, :, ::-1], show_sr, prior))
