#!/bin/bash
set -uo pipefail
O="$PWD/gpurun_out/${1:-r6s}"; mkdir -p "$O"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_mx_gpu.py tests/test_round6_gpu.py -m gpu -q --tb=short > "$O/tests_mx.log" 2>&1; tail -4 "$O/tests_mx.log"
tile() { MARCONET_HIP_LIB=$1 timeout 200 python tools/tile_power_ab.py --seconds 3 "${@:3}" 2>&1 | grep "TFLOP/s" | sed "s|^|$2 |"; }
PROD="$PWD/marconet_amd/lib/libmarconet_hip.so"
{ for rep in 1 2; do tile "$PROD" prod --only x2:15,x2:16; done; tile "$PROD" prod --only x2:15,x2:16 --shape 1024,64,64,512,256 --ragged; } 2>&1 | tee "$O/tile_rates.txt"
MARCONET_HIP_LIB=$PWD/tools/_build/w4_stamps/libmarconet_hip.so timeout 200 python tools/w4_phases.py 2>&1 | grep -v amdgpu.ids | cat | tee "$O/w4_phases.txt"
