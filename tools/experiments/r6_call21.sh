#!/bin/bash
# the one-wave tile's epilogue with straight-line arithmetic (no runtime arms in the step): tests, phases, rates against the previous build (tools/_build/w4_prev)
set -uo pipefail
O="$PWD/gpurun_out/${1:-r6ab}"; mkdir -p "$O"; export TMPDIR=/tmp
B="$PWD/tools/_build"
timeout 1200 python -m pytest tests/test_mx_gpu.py tests/test_round6_gpu.py -m gpu -q --tb=short > "$O/tests_mx.log" 2>&1; tail -4 "$O/tests_mx.log"
for v in st1; do echo "== $v"; MARCONET_HIP_LIB=$B/w4_$v/libmarconet_hip.so timeout 200 python tools/w4_phases.py 2>&1 | grep -v amdgpu.ids | cat; done | tee "$O/w4_phases.txt"
for v in st2; do echo "== $v"; MARCONET_HIP_LIB=$B/w4_$v/libmarconet_hip.so timeout 200 python tools/w4_phases.py 2>&1 | grep -v amdgpu.ids | grep "epilogue\|TFLOP" | cat; done | tee -a "$O/w4_phases.txt"
tile() { MARCONET_HIP_LIB=$1 timeout 200 python tools/tile_power_ab.py --seconds 3 "${@:3}" 2>&1 | grep "TFLOP/s" | sed "s|^|$2 |"; }
{ for rep in 1 2; do
    tile "$B/w4_prev/libmarconet_hip.so" prev --only x2:16
    tile "$PWD/marconet_amd/lib/libmarconet_hip.so" new --only x2:16
  done
  tile "$B/w4_prev/libmarconet_hip.so" prev --only x2:16 --shape 1024,64,64,512,256 --ragged
  tile "$PWD/marconet_amd/lib/libmarconet_hip.so" new --only x2:16 --shape 1024,64,64,512,256 --ragged
} 2>&1 | tee "$O/tile_rates.txt"
for lib in "$B/w4_prev/libmarconet_hip.so" "$PWD/marconet_amd/lib/libmarconet_hip.so" "$B/w4_prev/libmarconet_hip.so" "$PWD/marconet_amd/lib/libmarconet_hip.so"; do
  MARCONET_HIP_LIB=$lib timeout 600 python bench.py --steps 5 --warmup 2 --cpu-images 0 --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$lib'.split('/')[-2], d['value'], 'img/s', d['ms_per_step'], 'ms |', r['achieved'], 'TFLOP/s | tail', r['hbm_tail_ms_per_step'])"
done | tee "$O/bench_ab.txt"
