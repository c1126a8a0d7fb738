#!/bin/bash
set -uo pipefail
O="$PWD/gpurun_out/${1:-r6n}"; mkdir -p "$O"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_mx_gpu.py tests/test_round6_gpu.py -m gpu -q --tb=line > "$O/tests_mx.log" 2>&1; tail -3 "$O/tests_mx.log"
tile() { MARCONET_HIP_LIB=$1 timeout 200 python tools/tile_power_ab.py --seconds 3 "${@:3}" 2>&1 | grep "TFLOP/s" | sed "s|^|$2 |"; }
PROD="$PWD/marconet_amd/lib/libmarconet_hip.so"
{
for rep in 1 2; do
  tile "$PROD" prod --only x2:15,x2:16
done
tile "$PROD" prod --only x2:15,x2:16 --shape 1024,64,64,512,256 --ragged
} 2>&1 | tee "$O/tile_rates.txt"
MARCONET_HIP_LIB=$PWD/tools/_build/w4_stamps/libmarconet_hip.so timeout 200 python tools/w4_phases.py 2>&1 | grep -v amdgpu.ids | tee "$O/w4_phases.txt"
for cfg in 15 16 15 16; do
  MNET_MX_CFG256=$cfg timeout 600 python bench.py --steps 5 --warmup 2 --no-secondary --cpu-images 0 --no-regimes 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); r=d['roofline']
print('cfg256=$cfg', d['value'], 'img/s', d['ms_per_step'], 'ms | dominant', r['kernel'], r['achieved'], 'TFLOP/s', r['kernel_ms_per_step'], 'ms/step | tail', r['hbm_tail_ms_per_step'], 'ms | all convs', r['all_conv_kernels']['ms_per_step'])" | tee -a "$O/bench_ab.txt"
done
