#!/usr/bin/env python
"""Static check of the hand-placed MFMAs of conv_dma_w4.hip (CPU only): hipcc pads nothing in front of an `asm volatile` MFMA, so a VALU instruction that
writes one of its operands (SrcA / SrcB / SrcC / scale registers) less than two wait states ahead is a silent wrong-result hazard (measured in round 6: a
compiler-inserted v_accvgpr_mov in front of an unpadded MFMA corrupted register 0 of every accumulator block).  For every v_mfma of the kernel this walks the
preceding instructions until two wait states are covered (s_nop N = N + 1 states, any other instruction = 1) and reports
   * HAZARD: a v_* instruction among them writes a register the MFMA reads;
   * UNKNOWN: a label (a join: some predecessor is not visible here) sits inside that window and the MFMA does not open with its own s_nop.
Also reports compiler-made accumulator-file traffic (v_accvgpr_write / _mov outside asm blocks), which must not exist: the blocks live in a[0:255] by construction.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only marconet_amd/csrc/conv_dma_w4.hip -o /tmp/w4.s;  python tools/isa_mfma_hazards.py /tmp/w4.s
exit status 1 on any finding."""
import re
import sys


def regs(tok):
    """'v[4:7]' / 'a12' / 's3' -> set of (file, index)"""
    tok = tok.strip()
    m = re.match(r"^([vas])\[(\d+):(\d+)\]$", tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.match(r"^([vas])(\d+)$", tok)
    if m:
        return {(m.group(1), int(m.group(2)))}
    return set()


def operands(line):
    body = line.strip().split(None, 1)
    if len(body) < 2:
        return []
    ops = re.split(r",\s*(?![^\[]*\])", body[1].split(" op_sel")[0].split(" cbsz")[0].split(" blgp")[0])
    return [o.strip() for o in ops]


def hot_scratch(body):
    """scratch accesses between the slab loop's barrier-carrying header and its back edge that an ordinary slab executes: those NOT behind a forward branch that skips
    more than 200 lines (the tile-closing epilogue and the tile crossing's set-up are such blocks)"""
    bar = max(i for i, l in enumerate(body) if l.strip() == "s_barrier")
    hdr = max(i for i in range(bar) if re.match(r"^\.LBB\d+_\d+:", body[i]))
    label = body[hdr].split(":")[0]
    back = max(i for i, l in enumerate(body) if re.search(r"s_c?branch\w*\s+" + re.escape(label) + r"\s*$", l.strip()))
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    skipped = set()
    for i in range(hdr, back):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", body[i])
        if m and m.group(1) in labels and i + 200 < labels[m.group(1)]:
            skipped.update(range(i, labels[m.group(1)]))
    def hot(l):
        t = l.strip()
        return "scratch_" in t or re.match(r"^v_accvgpr_(write|mov)", t)
    asm_lines, ia = set(), False
    for i, l in enumerate(body):
        if ";;#ASMSTART" in l:
            ia = True
        elif ";;#ASMEND" in l:
            ia = False
        elif ia:
            asm_lines.add(i)
    return hdr, back, [(i, body[i].strip()) for i in range(hdr, back) if hot(body[i]) and i not in skipped and i not in asm_lines]


def main():
    s = open(sys.argv[1]).read()
    name = sys.argv[2] if len(sys.argv) > 2 else "conv_dma_w4_kernel"
    rc = 0
    for m in re.finditer(r"^(_Z\w*%s\w*):" % name, s, re.M):
        rc |= check(s, m)
    return rc


def check(s, m):
    i = m.start()
    body = s[i:s.index(".Lfunc_end", i)].split("\n")
    instr = []                                      # (kind, text, in_asm): kind = 'i' instruction, 'l' label
    in_asm = False
    for l in body:
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith(";") or t.startswith("."):
            if re.match(r"^\.LBB\d+_\d+:", t):
                instr.append(("l", t, in_asm))
            continue
        instr.append(("i", t, in_asm))
    findings, n_mfma, spill_traffic = [], 0, []
    for k, (kind, t, ia) in enumerate(instr):
        if kind != "i":
            continue
        if re.match(r"^v_accvgpr_(write|mov)", t) and not ia:
            spill_traffic.append(t)        # hipcc parking a VGPR in an accumulator register whose block is dead (after the epilogue has read it): legal — the two-wait-state
                                           # walk below still covers every MFMA behind such a write; reported, and a finding only on the slab loop's ordinary path (hot_scratch)
        if not t.startswith("v_mfma"):
            continue
        n_mfma += 1
        ops = operands(t)
        reads = set()
        for o in ops[1:]:
            reads |= regs(o)
        states, j = 0, k - 1
        while j >= 0 and states < 2:
            kd, tt, _ = instr[j]
            if kd == "l":
                findings.append("UNKNOWN predecessor (label %s) %d wait state(s) before: %s" % (tt, states, t[:90]))
                break
            mm = re.match(r"^s_nop (\d+)", tt)
            if mm:
                states += int(mm.group(1)) + 1
            else:
                if tt.startswith("v_") and not tt.startswith("v_mfma") and not tt.startswith("v_cmp") and not tt.startswith("v_readlane") and not tt.startswith("v_readfirstlane"):
                    dst = operands(tt)[0] if operands(tt) else ""
                    if regs(dst) & reads:
                        findings.append("HAZARD %d wait state(s): `%s` writes an operand of `%s`" % (states, tt[:70], t[:90]))
                states += 1
            j -= 1
    hdr, back, hot = hot_scratch(body)
    for i_, t_ in hot:
        findings.append("HOT SCRATCH / ACCUMULATOR-FILE SPILL line %d of the slab loop (%d-%d): %s" % (i_, hdr, back, t_[:80]))
    print("%s: %d MFMAs checked, %d compiler accumulator-file moves outside the hot path, %d finding(s)" % (m.group(1), n_mfma, len(spill_traffic), len(findings)))
    for f in findings[:40]:
        print("   ", f)
    return 1 if findings else 0


if __name__ == "__main__":
    sys.exit(main())
