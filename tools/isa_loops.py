#!/usr/bin/env python
"""Loop census of one kernel's gfx950 ISA (CPU only): per backward branch, the number of MFMAs, scratch accesses, LDS reads, DMA / buffer loads,
waits and barriers inside the loop body — the quick check that a change left no scratch traffic or stray waits in a k-loop.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only marconet_amd/csrc/conv_igemm_dma.hip -o /tmp/dma.s
    python tools/isa_loops.py /tmp/dma.s <mangled-name substring> [--dump-innermost]"""
import re
import sys


def main():
    s = open(sys.argv[1]).read()
    pat = sys.argv[2]
    names = [m.group(1) for m in re.finditer(r'^(_Z\w+):', s, re.M) if pat in m.group(1)]
    for name in names:
        i = s.index(name + ':')
        j = s.index('.Lfunc_end', i)
        body = s[i:j].split('\n')
        labels = {m.group(1): n for n, l in enumerate(body) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
        loops = []
        for n, l in enumerate(body):
            m = re.search(r's_cbranch\w*\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)', l)
            if m:
                t = m.group(1) or m.group(2)
                if t in labels and labels[t] < n:
                    loops.append((labels[t], n))
        print("%s: %d lines, %d loops" % (name, len(body), len(loops)))
        for a, b in loops:
            seg = body[a:b]
            c = lambda k: sum(k in x for x in seg)
            if c('v_mfma'):
                print("  lines %5d-%5d: mfma %3d  scratch %2d  ds_read %3d  buffer_load %2d  s_waitcnt %3d  s_barrier %d  v_cvt_scalef32 %2d  s_nop %d"
                      % (a, b, c('v_mfma'), c('scratch_'), c('ds_read') + c('ds_load'), c('buffer_load'), c('s_waitcnt'), c('s_barrier'), c('v_cvt_scalef32'), c('s_nop')))
        if "--dump-innermost" in sys.argv:
            inner = min((l for l in loops if sum('v_mfma' in x for x in body[l[0]:l[1]])), key=lambda l: l[1] - l[0])
            print("\n".join(body[inner[0]:inner[1] + 1]))


if __name__ == "__main__":
    main()
