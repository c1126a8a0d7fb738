#!/usr/bin/env python
"""Aggregates the CSVs written by tools/pmc_passes.sh into one per-kernel table (mean counter value per dispatch).
usage: python tools/pmc_summary.py <outdir> [<out.txt>] [kernel-name substring filter]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 else None
    filt = sys.argv[3] if len(sys.argv) > 3 else ""
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")
            if filt and filt not in k:
                continue
            c = acc[k][row["Counter_Name"]]
            c[0] += float(row["Counter_Value"]); c[1] += 1
    lines = []
    for k in sorted(acc):
        lines.append(k[:150])
        for name in sorted(acc[k]):
            tot, n = acc[k][name]
            lines.append("    %-34s mean/dispatch %18.1f   dispatches %d" % (name, tot / max(n, 1), n))
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
