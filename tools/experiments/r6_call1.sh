#!/bin/bash
# round 6, first GPU lease: the one-wave-per-SIMD tile (fp16+8 id 16) — byte equality tests, then interleaved rates against id 15 with power / clock
set -uo pipefail
O="$PWD/gpurun_out/r6a"; mkdir -p "$O"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_mx_gpu.py -m gpu -q --tb=short -x -k "one_wave or every_lds_dma or agree_bit" > "$O/tests_w4.log" 2>&1; tail -5 "$O/tests_w4.log"
timeout 600 python -m pytest tests/test_round6_gpu.py -m gpu -q --tb=short > "$O/tests_r6.log" 2>&1; tail -8 "$O/tests_r6.log"
tile() { MARCONET_HIP_LIB=$1 timeout 200 python tools/tile_power_ab.py --seconds 3 "${@:3}" 2>&1 | grep "TFLOP/s" | sed "s|^|$2 |"; }
PROD="$PWD/marconet_amd/lib/libmarconet_hip.so"
{
for rep in 1 2; do
  tile "$PROD" prod --only x2:15,x2:16
  tile "$PWD/tools/_build/w4_00/libmarconet_hip.so" w4_00 --only x2:16
  tile "$PWD/tools/_build/w4_10/libmarconet_hip.so" w4_10 --only x2:16
  tile "$PROD" prod --only x2:15,x2:16 --shape 1024,64,64,512,256 --ragged
done
tile "$PROD" prod-zeros --only x2:15,x2:16 --zeros
tile "$PWD/tools/_build/w4_00/libmarconet_hip.so" w4_00-zeros --only x2:16 --zeros
} 2>&1 | tee "$O/tile_rates.txt"
