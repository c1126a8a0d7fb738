#!/bin/bash
set -uo pipefail
O="$PWD/gpurun_out/${1:-r6am}"; mkdir -p "$O"; export TMPDIR=/tmp
true
for v in 0 64 0 64 128 32; do
  MNET_ADAIN_SLICE=$v timeout 600 python bench.py --steps 4 --warmup 2 --cpu-images 0 --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('slice $v', d['value'], 'img/s', d['ms_per_step'], 'ms | tail', r['hbm_tail_ms_per_step'], {k: round(v['ms_per_step'], 1) for k, v in r['hbm_tail']['by_kernel'].items()})"
done | tee "$O/bench_adain_slice_ab.txt"
