#!/bin/bash
set -uo pipefail
O="$PWD/gpurun_out/${1:-r6m}"; mkdir -p "$O"; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --tb=short > "$O/tests_all.log" 2>&1; tail -5 "$O/tests_all.log"
tile() { MARCONET_HIP_LIB=$1 timeout 200 python tools/tile_power_ab.py --seconds 3 "${@:3}" 2>&1 | grep "TFLOP/s" | sed "s|^|$2 |"; }
PROD="$PWD/marconet_amd/lib/libmarconet_hip.so"
{
for rep in 1 2; do
  tile "$PROD" prod --only x2:15,x2:16
  tile "$PWD/tools/_build/w4_plain/libmarconet_hip.so" w4_plain --only x2:16
done
} 2>&1 | tee "$O/tile_rates.txt"
MARCONET_HIP_LIB=$PWD/tools/_build/w4_plain_stamps/libmarconet_hip.so timeout 200 python tools/w4_phases.py 2>&1 | grep -v amdgpu.ids | tee "$O/w4_phases_plain.txt"
