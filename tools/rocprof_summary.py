#!/usr/bin/env python
"""Turns a rocprofv3 result (rocpd sqlite db, or the *_kernel_stats.csv of --output-format csv) into the plain-text
per-kernel summary committed under profiles/.
usage: python tools/rocprof_summary.py <results.db | kernel_stats.csv> <out.txt> [note]"""
import csv
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    if db.endswith(".csv"):       # "Name","Calls","TotalDurationNs","AverageNs","Percentage",...
        rows = [(r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, float(r["Percentage"]))
                for r in csv.DictReader(open(db))]
    else:
        c = sqlite3.connect(db)
        rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    tot = sum(r[2] for r in rows)
    with open(out, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats summary (durations in microseconds)\n# %s\n" % note)
        f.write("# total kernel time %.3f ms over %d kernels\n" % (tot / 1e3, len(rows)))
        f.write("%-110s %8s %14s %12s %7s\n" % ("kernel", "calls", "total_us", "avg_us", "pct"))
        for name, calls, total, avg, pct in rows:
            f.write("%-110s %8d %14.1f %12.2f %7.2f\n" % (name[:110], calls, total, avg, pct))
    print("wrote", out)


if __name__ == "__main__":
    main()
