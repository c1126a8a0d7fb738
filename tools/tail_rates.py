#!/usr/bin/env python
"""Achieved HBM rate per kernel = (2*FETCH_SIZE + WRITE_SIZE) KB per dispatch (PMC passes, tools/pmc_summary.py table of ALL kernels)
÷ the kernel's average duration (rocprofv3 --kernel-trace --stats summary of the same bench command).
usage: python tools/tail_rates.py <pmc_all_kernels.txt> <kernel_stats.txt> [out.txt]"""
import re
import sys


def main():
    txt = open(sys.argv[1]).read()
    stats = {}
    for l in open(sys.argv[2]):
        m = re.match(r"(.{110})\s+(\d+)\s+([\d.]+)\s+([\d.]+)", l)
        if m:
            stats[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)), float(m.group(4)))
    rows = []
    for b in re.split(r"\n(?=\S)", txt):
        lines = b.strip().split("\n")
        name, c = lines[0], {}
        for l in lines[1:]:
            q = re.match(r"\s+(\S+)\s+mean/dispatch\s+([\d.]+)\s+dispatches (\d+)", l)
            if q:
                c[q.group(1)] = float(q.group(2))
        key = [k for k in stats if k[:60] == name[:60]]
        if "FETCH_SIZE" not in c or not key:
            continue
        calls, tot, avg = stats[key[0]]
        rd, wr = 2 * c["FETCH_SIZE"] * 1024, c["WRITE_SIZE"] * 1024
        rows.append((tot, "%-96s %6d calls  avg %9.1f us  read %8.1f MB  write %8.1f MB  %5.2f TB/s"
                     % (name[:96], calls, avg, rd / 1e6, wr / 1e6, (rd + wr) / avg / 1e6)))
    out = "# HBM bytes per dispatch (2*FETCH_SIZE + WRITE_SIZE, gfx950 read-side correction) / average kernel duration; sorted by total time\n"
    out += "\n".join(r for _, r in sorted(rows, reverse=True)) + "\n"
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(out)
    print(out)


if __name__ == "__main__":
    main()
