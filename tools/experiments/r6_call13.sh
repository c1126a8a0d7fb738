#!/bin/bash
set -uo pipefail
O="$PWD/gpurun_out/${1:-r6q}"; mkdir -p "$O"; export TMPDIR=/tmp
tile() { MARCONET_HIP_LIB=$1 timeout 200 python tools/tile_power_ab.py --seconds 3 "${@:3}" 2>&1 | grep "TFLOP/s" | sed "s|^|$2 |"; }
PROD="$PWD/marconet_amd/lib/libmarconet_hip.so"; VAR="$PWD/tools/_build/w4_actfirst/libmarconet_hip.so"
{
for rep in 1 2 3; do
  tile "$PROD" prod --only x2:16; tile "$VAR" actfirst --only x2:16
done
tile "$PROD" prod --only x2:16 --shape 1024,64,64,512,256 --ragged; tile "$VAR" actfirst --only x2:16 --shape 1024,64,64,512,256 --ragged
tile "$PROD" prod --only x2:16 --shape 1024,64,64,512,256 --ragged; tile "$VAR" actfirst --only x2:16 --shape 1024,64,64,512,256 --ragged
} 2>&1 | tee "$O/tile_rates_actfirst.txt"
MARCONET_HIP_LIB=$VAR timeout 300 python -m pytest tests/test_mx_gpu.py -m gpu -q --tb=line -k one_wave 2>&1 | tail -2
