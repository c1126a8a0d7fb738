"""Per-kernel summary of a rocprofv3 --kernel-trace CSV, normalised per forward:  python tools/trace_summary.py trace.csv N_FORWARDS [top]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
nf = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
tot, cnt = collections.Counter(), collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:78]
    tot[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    cnt[k] += 1
T = sum(tot.values())
print("%d kernel launches, %.3f ms of kernel time per forward (%g forwards)" % (len(rows), T / 1e6 / nf, nf))
for k, v in tot.most_common(top):
    print("%8.3f ms/fwd %6.1f calls/fwd %8.1f us  %s" % (v / 1e6 / nf, cnt[k] / nf, v / cnt[k] / 1e3, k))
