#!/bin/bash
# the blocked storages' chunk loads / stores as GLOBAL instead of FLAT instructions (pointer arithmetic instead of an integer round trip in ldraw / straw): tests, kernel rates, bench A/B
set -uo pipefail
O="$PWD/gpurun_out/${1:-r6ai}"; mkdir -p "$O"; export TMPDIR=/tmp
B="$PWD/tools/_build"
timeout 1800 python -m pytest tests/test_kernels_gpu.py tests/test_mx_gpu.py tests/test_split_gpu.py tests/test_modules_gpu.py -m gpu -q --tb=short -x > "$O/tests.log" 2>&1; tail -2 "$O/tests.log"
{ for v in flat prod; do lib="$B/aux_$v/libmarconet_hip.so"; [ $v = prod ] && lib="$PWD/marconet_amd/lib/libmarconet_hip.so"
    echo "== $v"; MARCONET_HIP_LIB=$lib python tools/experiments/tail_vs_torch_stream.py 2>&1 | grep -v amdgpu.ids | sed "s/| torch.*//"; done; } | tee "$O/tail_kernels_flat_vs_global.txt"
for v in flat prod flat prod; do
  lib="$B/aux_$v/libmarconet_hip.so"; [ $v = prod ] && lib="$PWD/marconet_amd/lib/libmarconet_hip.so"
  MARCONET_HIP_LIB=$lib timeout 600 python bench.py --steps 5 --warmup 2 --cpu-images 0 --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$v', d['value'], 'img/s', d['ms_per_step'], 'ms |', r['achieved'], 'TFLOP/s | tail', r['hbm_tail_ms_per_step'], {k: round(v['ms_per_step'], 1) for k, v in r['hbm_tail']['by_kernel'].items()})"
done | tee "$O/bench_flat_vs_global.txt"
