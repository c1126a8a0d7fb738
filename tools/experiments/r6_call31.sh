#!/bin/bash
# HBM-side bytes of the up-sample kernel (FETCH_SIZE / WRITE_SIZE) on the microbenchmark shapes: is the read amplified?
set -uo pipefail
O="$PWD/gpurun_out/${1:-r6ao}"; mkdir -p "$O"; export TMPDIR=/tmp
for g in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $g | cut -d' ' -f1)
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $g --output-format csv -d "$O/pmc_$n" -o pmc -- python "$GRAFT_REPO_ROOT/tools/experiments/tail_vs_torch_stream.py" ) > "$O/pmc_$n.log" 2>&1
done
python - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "upsample2x" in k or "affine_act" in k:
            acc[k[:60] + " grid " + r.get("Grid_Size", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k, {c: "%.3e (x%d)" % (sum(v) / len(v), len(v)) for c, v in d.items()})
PY
