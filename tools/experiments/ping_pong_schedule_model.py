#!/usr/bin/env python
"""Functional model of the ping-pong slab schedule of tools/experiments/ping_pong.patch (conv_dma_kernel<..., PP = true>): 8 waves, 2 LDS
stages, every wave issues its own DMA pieces of every slab.  The per-wave programs below are transcribed from the kernel; waves run in random
interleavings between barriers, an issued DMA piece lands at a random moment between its issue and the issuing wave's next vmcnt(0).
Checked for every LOAD(s) of every wave: each piece of stage s & 1 holds slab s, and no DMA into that stage is in flight (from any wave)
while the stage may still be read — i.e. until every wave has passed the barrier that closes the phase of its LOAD.

    python tools/experiments/ping_pong_schedule_model.py        # a few thousand random schedules, prints "ok"
"""
import random
import sys

NW, NDMA = 8, 8


def program(grp, total):
    """yields the wave's events in program order (see the PP block of conv_igemm_dma.hip)"""
    issued = 0                                   # slabs issued so far by this wave (the stream has `total`)

    def issue():
        nonlocal issued
        if issued < total:
            for j in range(NDMA):
                yield ("piece", issued, j)
            issued += 1
    yield from issue()                           # slab 0
    yield ("vmcnt0",)
    yield ("bar",)                               # phase 0 starts
    if grp == 1:
        yield from issue()                       # group 1 idles through phase 0: its share of slab 1
        yield ("bar",)
    for s in range(total):
        yield ("load", s)
        if grp == 0:
            yield from issue()                   # phase 2s: slab s + 1
        yield ("lgkm0", s)                       # reads of slab s have returned
        if grp == 1:
            yield ("vmcnt0",)
        yield ("bar",)
        if grp == 1:
            yield from issue()                   # phase 2s + 2: slab s + 2, under the MFMAs
        yield ("compute", s)
        if grp == 0:
            yield ("vmcnt0",)
        yield ("bar",)
    if grp == 0:
        yield ("bar",)


def run(total, rng):
    waves = [program(w >> 2, total) for w in range(NW)]
    stage = [[[None] * NDMA for _ in range(NW)] for _ in range(2)]       # stage[st][owner wave][piece] = slab held
    pending = [[] for _ in range(NW)]                                     # per wave: (slab, piece) issued, not landed
    reading = {}                                                          # wave -> stage it has reads in flight on (load issued, lgkm0 not reached)
    at_bar = [False] * NW
    done = [False] * NW
    loaded = [-1] * NW

    def land(w, k):
        slab, j = pending[w].pop(k)
        st = slab & 1
        for r, rst in reading.items():
            assert rst != st, "DMA of slab %d lands in stage %d while wave %d is reading it" % (slab, st, r)
        stage[st][w][j] = slab

    while not all(done):
        # random DMA landings (in order per wave, like vmcnt)
        for w in range(NW):
            while pending[w] and rng.random() < 0.15:
                land(w, 0)
        runnable = [w for w in range(NW) if not done[w] and not at_bar[w]]
        if not runnable:
            assert all(at_bar[w] or done[w] for w in range(NW))
            assert not any(done[w] for w in range(NW)) or all(done[w] or at_bar[w] for w in range(NW))
            if any(done) and any(at_bar):
                raise AssertionError("barrier count mismatch: some waves finished while others wait")
            at_bar = [False] * NW
            continue
        w = rng.choice(runnable)
        try:
            ev = next(waves[w])
        except StopIteration:
            done[w] = True
            continue
        if ev[0] == "piece":
            _, slab, j = ev
            st = slab & 1
            # the stage must not be read by anyone from now on until the data is waited for: nobody may have reads in flight on it
            for r, rst in reading.items():
                assert rst != st, "wave %d issues slab %d into stage %d while wave %d reads it" % (w, slab, st, r)
            # ... and nobody may still NEED the old content (slab - 2): every wave has loaded it
            for r in range(NW):
                assert slab < 2 or loaded[r] >= slab - 2, "wave %d overwrites slab %d before wave %d loaded it" % (w, slab - 2, r)
            pending[w].append((slab, j))
        elif ev[0] == "vmcnt0":
            while pending[w]:
                land(w, 0)
        elif ev[0] == "load":
            s = ev[1]
            st = s & 1
            for o in range(NW):
                for j in range(NDMA):
                    assert stage[st][o][j] == s, "wave %d LOAD(%d): piece (%d,%d) of stage %d holds %r" % (w, s, o, j, st, stage[st][o][j])
                assert not any((p[0] & 1) == st for p in pending[o]), "wave %d LOAD(%d) with a DMA of wave %d in flight into its stage" % (w, s, o)
            reading[w] = st
            loaded[w] = s
        elif ev[0] == "lgkm0":
            reading.pop(w, None)
        elif ev[0] == "compute":
            assert loaded[w] == ev[1]
        elif ev[0] == "bar":
            at_bar[w] = True
    assert all(not p for p in pending)
    assert all(l == total - 1 for l in loaded)


def check_addresses():
    """the LOAD block reaches the other chunks of an LDS row by XOR on one address (4 address registers instead of 8): equal to
    swz_dma(row, chunk) of conv_dma_common.h for every lane, fragment, stage and both tile geometries"""
    def swz(row, chunk):
        return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4)
    for bc, bp, wc_n, wp_n in ((256, 256, 2, 4), (128, 512, 1, 8)):
        fa, fb, stage = bc // wc_n // 32, bp // wp_n // 32, (bc + bp) * 128
        for wave in range(8):
            wc, wp = wave // wp_n, wave % wp_n
            for lane in range(64):
                l32, h = lane & 31, lane >> 5
                ra, rb = wc * (bc // wc_n) + l32, wp * (bp // wp_n) + l32
                for st in (0, 1):
                    so = st * stage
                    aa, a8a, ba, bea = swz(ra, h) + so, swz(ra, 4 + 2 * h) + so, swz(rb, h) + so + bc * 128, swz(rb, 6) + so + bc * 128
                    for f in range(fa):
                        assert aa + f * 4096 == so + swz(ra + 32 * f, h) and (aa ^ 32) + f * 4096 == so + swz(ra + 32 * f, 2 + h)
                        assert a8a + f * 4096 == so + swz(ra + 32 * f, 4 + 2 * h) and (a8a ^ 16) + f * 4096 == so + swz(ra + 32 * f, 5 + 2 * h)
                    for f in range(fb):
                        sx = so + bc * 128
                        assert ba + f * 4096 == sx + swz(rb + 32 * f, h) and (ba ^ 32) + f * 4096 == sx + swz(rb + 32 * f, 2 + h)
                        assert (ba ^ 64) + f * 4096 == sx + swz(rb + 32 * f, 4 + h) and bea + f * 4096 == sx + swz(rb + 32 * f, 6)


def main():
    check_addresses()
    rng = random.Random(1234)
    for total in (1, 2, 3, 4, 5, 9, 18):
        for _ in range(400):
            run(total, rng)
    print("ok")


if __name__ == "__main__":
    sys.exit(main())
