#!/bin/bash
set -uo pipefail
O="$PWD/gpurun_out/${1:-r6aq}"; mkdir -p "$O"; export TMPDIR=/tmp
B="$PWD/tools/_build"
true
for v in nozz prod nozz prod nozz prod; do
  lib="$B/aux_$v/libmarconet_hip.so"; [ $v = prod ] && lib="$PWD/marconet_amd/lib/libmarconet_hip.so"
  MARCONET_HIP_LIB=$lib timeout 600 python bench.py --steps 4 --warmup 2 --cpu-images 0 --no-secondary 2>/dev/null > "$O/line.json"; echo "$v $(python tools/experiments/print_line.py $O/line.json)"
done | tee "$O/bench_torgb_tail_once_ab.txt"
