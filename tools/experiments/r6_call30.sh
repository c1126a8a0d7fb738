#!/bin/bash
set -uo pipefail
O="$PWD/gpurun_out/${1:-r6an}"; mkdir -p "$O"; export TMPDIR=/tmp
B="$PWD/tools/_build"
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_modules_gpu.py tests/test_ab_forms_gpu.py tests/test_regimes_gpu.py -m gpu -q --tb=short -x > "$O/tests.log" 2>&1; tail -2 "$O/tests.log"
for v in shfl prod shfl prod; do
  lib="$B/aux_$v/libmarconet_hip.so"; [ $v = prod ] && lib="$PWD/marconet_amd/lib/libmarconet_hip.so"
  MARCONET_HIP_LIB=$lib timeout 600 python bench.py --steps 4 --warmup 2 --cpu-images 0 --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$v', d['value'], 'img/s', d['ms_per_step'], 'ms | tail', r['hbm_tail_ms_per_step'], {k: round(v['ms_per_step'], 1) for k, v in r['hbm_tail']['by_kernel'].items()})"
done | tee "$O/bench_torgb_dpp_ab.txt"
