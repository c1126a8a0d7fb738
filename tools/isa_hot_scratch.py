#!/usr/bin/env python
"""Scratch traffic on the HOT path of a software-pipelined / lock-step LDS-DMA conv kernel (CPU only; run after every change of these kernels).

hipcc's register allocation of the 256-VGPR tiles is global: a change in the epilogue or in the per-tile set-up can make it spill values of the
k-loop — every reload sits behind an `s_waitcnt vmcnt(0)` that also waits for the slab's LDS-DMA pieces, and a tile that passes every test runs
15 % slower (478 instead of 563 TFLOP/s).  This walks the kernel's innermost loop that holds the slab barrier and the slab's
MFMAs and counts scratch accesses outside the region the hot path branches around (the once-per-tile set-up / epilogue blocks).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only marconet_amd/csrc/conv_igemm_dma.hip -o /tmp/dma.s
    python tools/isa_hot_scratch.py /tmp/dma.s [mangled-name substring ...]     (default: the fp16+8 256x256 software-pipelined tile)
    hipcc ... -mllvm -greedy-reverse-local-assignment=1 -S --cuda-device-only marconet_amd/csrc/conv_dma_swp_gn.hip -o /tmp/sgn.s; python tools/isa_hot_scratch.py /tmp/sgn.s
        (the build of that tile with the GroupNorm-sum block: its own translation unit and flag, marconet_amd/csrc/build.sh)
exit status 1 if any hot scratch access is found."""
import re
import sys

DEFAULT = ["conv_dma_kernelILi256ELi256ELi2ELi4ELi2ELi32ELi0ELb1ELb0ELb0ELb1ELb1E"]      # (both builds: ...ELb0E = production, ...ELb1E = with the GroupNorm-sum block, conv_dma_swp_gn.hip)


def check(s, name):
    i = s.index(name + ":")
    j = s.index(".Lfunc_end", i)
    body = s[i:j].split("\n")
    labels = {m.group(1): k for k, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    loops = []
    for k, l in enumerate(body):
        m = re.search(r"s_cbranch\w*\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", l)
        if m:
            t = m.group(1) or m.group(2)
            if t in labels and labels[t] < k:
                seg = body[labels[t]:k]
                if sum("v_mfma" in x for x in seg) >= 16 and any("s_barrier" in x for x in seg):
                    loops.append((k - labels[t], labels[t], k))
    if not loops:
        print("%s: no loop with a barrier and >= 16 MFMAs found" % name[:70])
        return 0
    _, a, b = sorted(loops)[0]
    skipped = set()
    for k in range(a, b):          # forward branches over more than 300 lines inside the loop: the blocks an ordinary slab does not execute
        m = re.search(r"s_cbranch_scc1\s+(\.LBB\d+_\d+)", body[k])
        if m and m.group(1) in labels and k + 300 < labels[m.group(1)] <= b:
            skipped.update(range(k, labels[m.group(1)]))
    hot = [(k, body[k].strip()) for k in range(a, b) if "scratch_" in body[k] and k not in skipped]
    sp = re.search(r"\.name:\s+" + re.escape(name) + r"\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", s)
    print("%s\n    slab loop lines %d-%d (%d skipped as once-per-tile), spilled VGPRs %s, scratch accesses on the hot path: %d"
          % (name, a, b, len(skipped), sp.group(1) if sp else "?", len(hot)))
    for k, t in hot[:12]:
        print("      line %d: %s" % (k, t[:100]))
    return len(hot)


def main():
    s = open(sys.argv[1]).read()
    pats = sys.argv[2:] or DEFAULT
    names = [m.group(1) for m in re.finditer(r"^(_Z\w+):", s, re.M) if any(p in m.group(1) for p in pats)]
    bad = sum(check(s, n) for n in names)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
