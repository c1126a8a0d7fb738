#!/bin/bash
# the one-wave tile's residual / GroupNorm-sums builds without the per-step scratch reload of lane & 31 (and its vmcnt(0)): tests, phases, bench A/B
set -uo pipefail
O="$PWD/gpurun_out/${1:-r6ax}"; mkdir -p "$O"; export TMPDIR=/tmp
B="$PWD/tools/_build"
timeout 1500 python -m pytest tests/test_mx_gpu.py tests/test_round6_gpu.py tests/test_modules_gpu.py -m gpu -q --tb=short > "$O/tests.log" 2>&1; tail -3 "$O/tests.log"
for e in plain gn residual scales; do echo "== epilogue terms: $e"; MARCONET_HIP_LIB=$B/w4_stamps/libmarconet_hip.so python tools/w4_phases.py --epilogue $e 2>&1 | grep "TFLOP\|per TILE\|crossing"; done | tee "$O/w4_phases_by_build.txt"
for v in prev prod prev prod; do
  lib="$B/w4_$v/libmarconet_hip.so"; [ $v = prod ] && lib="$PWD/marconet_amd/lib/libmarconet_hip.so"
  MARCONET_HIP_LIB=$lib timeout 600 python bench.py --steps 4 --warmup 2 --cpu-images 0 --no-secondary 2>/dev/null > "$O/line.json"; echo "$v $(python tools/experiments/print_line.py $O/line.json | cut -c1-110)"
done | tee "$O/bench_ab.txt"
