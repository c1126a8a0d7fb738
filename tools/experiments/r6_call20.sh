#!/bin/bash
# A/B: the one-wave tile's epilogue storing every lane's own 128-byte block straight from registers (-DW4_DIRECT_STORE=1) instead of through the LDS transposition
set -uo pipefail
O="$PWD/gpurun_out/${1:-r6aa}"; mkdir -p "$O"; export TMPDIR=/tmp
B="$PWD/tools/_build"
MARCONET_HIP_LIB=$B/w4_direct/libmarconet_hip.so timeout 900 python -m pytest tests/test_mx_gpu.py -m gpu -q --tb=short -k "one_wave or every_lds_dma or agree" > "$O/tests_direct.log" 2>&1; tail -3 "$O/tests_direct.log"
for v in st1 st1_direct; do echo "== $v"; MARCONET_HIP_LIB=$B/w4_$v/libmarconet_hip.so timeout 200 python tools/w4_phases.py 2>&1 | grep -v amdgpu.ids | cat; done | tee "$O/w4_phases_direct_store.txt"
for v in st2 st2_direct; do echo "== $v"; MARCONET_HIP_LIB=$B/w4_$v/libmarconet_hip.so timeout 200 python tools/w4_phases.py 2>&1 | grep -v amdgpu.ids | grep "epilogue\|TFLOP" | cat; done | tee -a "$O/w4_phases_direct_store.txt"
tile() { MARCONET_HIP_LIB=$1 timeout 200 python tools/tile_power_ab.py --seconds 3 "${@:3}" 2>&1 | grep "TFLOP/s" | sed "s|^|$2 |"; }
{ for rep in 1 2; do
    tile "$PWD/marconet_amd/lib/libmarconet_hip.so" prod --only x2:16
    tile "$B/w4_direct/libmarconet_hip.so" direct --only x2:16
  done
  tile "$PWD/marconet_amd/lib/libmarconet_hip.so" prod --only x2:16 --shape 1024,64,64,512,256 --ragged
  tile "$B/w4_direct/libmarconet_hip.so" direct --only x2:16 --shape 1024,64,64,512,256 --ragged
} 2>&1 | tee "$O/tile_rates_direct_store.txt"
