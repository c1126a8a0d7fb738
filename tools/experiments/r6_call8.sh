#!/bin/bash
set -uo pipefail
O="$PWD/gpurun_out/${1:-r6k}"; mkdir -p "$O"; export TMPDIR=/tmp
for cfg in 15 16 15 16; do
  MNET_MX_CFG256=$cfg timeout 600 python bench.py --steps 5 --warmup 2 --no-secondary --cpu-images 0 --no-regimes 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); r=d['roofline']
print('cfg256=$cfg', d['value'], 'img/s', d['ms_per_step'], 'ms | dominant', r['kernel'], r['achieved'], 'TFLOP/s', r['kernel_ms_per_step'], 'ms/step | tail', r['hbm_tail_ms_per_step'], 'ms | all convs', r['all_conv_kernels']['ms_per_step'])" | tee -a "$O/bench_ab.txt"
done
timeout 2400 python -m pytest tests -m gpu -q --tb=short -x > "$O/tests_all.log" 2>&1; tail -5 "$O/tests_all.log"
