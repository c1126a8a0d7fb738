#!/bin/bash
set -uo pipefail
O="$PWD/gpurun_out/${1:-r6aj}"; mkdir -p "$O"; export TMPDIR=/tmp
B="$PWD/tools/_build"
for v in noxcd prod noxcd prod; do
  lib="$B/aux_$v/libmarconet_hip.so"; [ $v = prod ] && lib="$PWD/marconet_amd/lib/libmarconet_hip.so"
  MARCONET_HIP_LIB=$lib timeout 600 python bench.py --steps 5 --warmup 2 --cpu-images 0 --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$v', d['value'], 'img/s', d['ms_per_step'], 'ms |', r['achieved'], 'TFLOP/s | tail', r['hbm_tail_ms_per_step'], {k: round(v['ms_per_step'], 1) for k, v in r['hbm_tail']['by_kernel'].items()})"
done | tee "$O/bench_upsample_xcd_ab.txt"
