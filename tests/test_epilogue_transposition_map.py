"""CPU: the slot map of the fp16+8 epilogue's store transposition (marconet_amd/csrc/conv_dma_common.h, dma_epilogue_mx) — what the comments there claim:
every 16-byte piece a lane drops into its wave's LDS scratch is read back exactly once by the lane that stores it; a store instruction then writes 16 runs of
64 contiguous bytes; and neither the ds_write_b128 (lanes served in groups of 8 consecutive lanes, 32 banks of 4 bytes) nor the ds_read_b128 (groups of 16
lanes, 64 banks: MI355X_MICROARCH.md, LDS table) has two lanes of a group on the same bank."""
READ_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
READ_GROUPS += [[l + 32 for l in g] for g in READ_GROUPS]


def wslot(L, c):                      # lane L drops piece c of its half block here (16-byte slots)
    return 4 * L + (c ^ ((L >> 1) & 3))


def rslot(P, j):                      # piece j of lane P's half block
    return 4 * P + (j ^ ((P >> 1) & 3))


def test_full_wave_round_4KiB():
    written = {}
    for L in range(64):
        for c in range(4):
            s = wslot(L, c)
            assert 0 <= s < 256 and s not in written
            written[s] = (L, c)
    seen = set()
    for k in range(4):                # read instruction k: lane l takes piece l % 4 of lane l / 4 + 16 k
        owners = []
        for l in range(64):
            P, j = (l >> 2) + 16 * k, l & 3
            assert written[rslot(P, j)] == (P, j)
            seen.add((P, j))
            owners.append((P, j))
        for q in range(0, 64, 4):     # the 4 lanes of a quad hold the 4 consecutive pieces of ONE block: 64 contiguous bytes per quad
            assert [o[0] for o in owners[q:q + 4]] == [owners[q][0]] * 4 and [o[1] for o in owners[q:q + 4]] == [0, 1, 2, 3]
        for g in READ_GROUPS:         # ds_read_b128: 16 lanes x 16 bytes = all 64 banks once
            assert len({rslot((l >> 2) + 16 * k, l & 3) % 16 for l in g}) == 16
    assert len(seen) == 256
    for c in range(4):                # ds_write_b128: 8 consecutive lanes x 16 bytes = all 32 banks once
        for g0 in range(0, 64, 8):
            assert len({wslot(L, c) % 8 for L in range(g0, g0 + 8)}) == 8


def test_quarter_wave_rounds_1KiB():
    for r in range(4):                # round r: the blocks of lanes 16 r .. 16 r + 15
        written = {}
        for L in range(16 * r, 16 * r + 16):
            Lq = L & 15
            for c in range(4):
                s = 4 * Lq + (c ^ ((L >> 1) & 3))
                assert 0 <= s < 64 and s not in written
                written[s] = (L, c)
        for l in range(64):
            Pq, j = l >> 2, l & 3
            assert written[4 * Pq + (j ^ ((Pq >> 1) & 3))] == (16 * r + Pq, j)
        for g in READ_GROUPS:
            assert len({(4 * (l >> 2) + ((l & 3) ^ (((l >> 2) >> 1) & 3))) % 16 for l in g}) == 16
        for c in range(4):
            for g0 in (16 * r, 16 * r + 8):
                assert len({(4 * (L & 15) + (c ^ ((L >> 1) & 3))) % 8 for L in range(g0, g0 + 8)}) == 8
