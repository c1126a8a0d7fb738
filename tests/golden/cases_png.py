"""PNG parity cases (BASELINE configs[0]): the reference's own test strips, committed under tests/golden/pngs/."""
import os

PNG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pngs")

SR_STRIPS = {"lqe01": "real_lqe01_清肺东北小木耳.png", "lqe02": "real_lqe02_开发区雨虹电子有限公司.png"}
W_STRIPS = ("w1.png", "w2.png")
W_SCALES = (0.0, 0.5, 1.0)
W_MAX_GLYPHS = 6          # the random-init encoder "reads" up to 64 characters from a strip; the generator check needs a few


def sample_bgr(a):
    return a[::4, ::8, :].copy()
