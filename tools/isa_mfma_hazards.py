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


def main():
    s = open(sys.argv[1]).read()
    name = sys.argv[2] if len(sys.argv) > 2 else "conv_dma_w4_kernel"
    m = re.search(r"^(_Z\w*%s\w*):" % name, s, re.M)
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
    findings, n_mfma = [], 0
    for k, (kind, t, ia) in enumerate(instr):
        if kind != "i":
            continue
        if re.match(r"^v_accvgpr_(write|mov)", t) and not ia:
            findings.append("COMPILER ACCUMULATOR TRAFFIC: %s" % t)
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
    print("%s: %d MFMAs checked, %d finding(s)" % (m.group(1), n_mfma, len(findings)))
    for f in findings[:40]:
        print("   ", f)
    return 1 if findings else 0


if __name__ == "__main__":
    sys.exit(main())
