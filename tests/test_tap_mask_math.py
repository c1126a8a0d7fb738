"""CPU: the closed-form tap mask of the LDS-DMA kernels' per-tile set-up (marconet_amd/csrc/conv_igemm_dma.hip, setup()) against the
definition it replaced — bit t = r*kw + q is set iff input row ih0 + r lies in [0, H) and input column iw0 + q in [0, valid_w).
The device code computes (column range) * (row range of rowrep) with 32-bit shifts that are only defined below 32: the guards are part of what is checked."""
import itertools


def mask_by_definition(kh, kw, ih0, iw0, h, vw):
    m = 0
    for r in range(kh):
        for q in range(kw):
            if 0 <= ih0 + r < h and 0 <= iw0 + q < vw:
                m |= 1 << (r * kw + q)
    return m


def mask_closed_form(kh, kw, ih0, iw0, h, vw):
    u32 = 0xFFFFFFFF
    rowrep = 0
    for r in range(8):
        if r < kh:
            rowrep |= 1 << (r * kw)
    qlo, qhi = max(0, -iw0), min(kw, vw - iw0)
    rlo, rhi = max(0, -ih0), min(kh, h - ih0)
    assert 0 <= qlo <= 31 and max(qhi, 0) <= 31                      # shift counts of the device code
    cm = (((1 << max(qhi, 0)) - 1) & ~((1 << qlo) - 1)) & u32
    sh_hi, sh_lo = max(rhi, 0) * kw, rlo * kw
    hi = rowrep if sh_hi >= 32 else rowrep & ((1 << sh_hi) - 1)
    lo = 0 if sh_lo >= 32 else (~((1 << sh_lo) - 1)) & u32
    rr = hi & lo
    prod = cm * rr
    assert prod <= u32                                               # no carry out of 32 bits: kh * kw <= 32
    return prod


def test_closed_form_tap_mask_equals_the_definition():
    n = 0
    for kh, kw in itertools.product(range(1, 9), range(1, 9)):
        if kh * kw > 32:                                             # eligibility of the LDS-DMA path
            continue
        for ph, pw in ((0, 0), (kh // 2, kw // 2), (kh - 1, kw - 1), (min(7, kh), min(7, kw))):
            for h, w in ((1, 1), (3, 5), (4, 4), (8, 16)):
                for vw in sorted({0, 1, w // 2, w - 1, w} - {-1}):
                    for oh, ow, sh, sw in itertools.product(range(0, h + 2, max(1, h // 2)), range(0, w + 2, max(1, w // 2)), (1, 2), (1, 2)):
                        ih0, iw0 = oh * sh - ph, ow * sw - pw
                        assert mask_closed_form(kh, kw, ih0, iw0, h, vw) == mask_by_definition(kh, kw, ih0, iw0, h, vw), (kh, kw, ih0, iw0, h, vw)
                        n += 1
    assert n > 20000
