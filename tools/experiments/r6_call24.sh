#!/bin/bash
set -uo pipefail
O="$PWD/gpurun_out/${1:-r6ag}"; mkdir -p "$O"; export TMPDIR=/tmp
B="$PWD/tools/_build"
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_mx_gpu.py tests/test_split_gpu.py tests/test_modules_gpu.py -m gpu -q --tb=short -x > "$O/tests.log" 2>&1; tail -3 "$O/tests.log"
for v in cap1024 prod cap1024 prod; do
  lib="$B/aux_$v/libmarconet_hip.so"; [ $v = prod ] && lib="$PWD/marconet_amd/lib/libmarconet_hip.so"
  MARCONET_HIP_LIB=$lib timeout 600 python bench.py --steps 5 --warmup 2 --cpu-images 0 --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$v', d['value'], 'img/s', d['ms_per_step'], 'ms |', r['achieved'], 'TFLOP/s | tail', r['hbm_tail_ms_per_step'], {k: round(v['ms_per_step'], 1) for k, v in r.get('hbm_tail', {}).get('kernels', {}).items()} if isinstance(r.get('hbm_tail'), dict) else '')"
done | tee "$O/bench_affine_cap_ab.txt"
