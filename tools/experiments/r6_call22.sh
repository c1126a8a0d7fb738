#!/bin/bash
# A/B: DMA pieces of the next slab two per gap behind the first 8 scaled MFMAs (W4_PIECES_EARLY=1: reads unchanged; =5: the front reads behind MFMAs 4-15)
set -uo pipefail
O="$PWD/gpurun_out/${1:-r6ae}"; mkdir -p "$O"; export TMPDIR=/tmp
B="$PWD/tools/_build"
for v in early1 early5; do MARCONET_HIP_LIB=$B/w4_$v/libmarconet_hip.so timeout 600 python -m pytest tests/test_mx_gpu.py -m gpu -q --tb=line -k "one_wave" 2>&1 | tail -1; done
for v in st1_early1 st1_early5; do echo "== $v"; MARCONET_HIP_LIB=$B/w4_$v/libmarconet_hip.so timeout 200 python tools/w4_phases.py 2>&1 | grep -v amdgpu.ids | cat; done | tee "$O/w4_phases_pieces_early.txt"
tile() { MARCONET_HIP_LIB=$1 timeout 200 python tools/tile_power_ab.py --seconds 3 "${@:3}" 2>&1 | grep "TFLOP/s" | sed "s|^|$2 |"; }
{ for rep in 1 2; do
    tile "$PWD/marconet_amd/lib/libmarconet_hip.so" prod --only x2:16
    for v in early1 early5; do tile "$B/w4_$v/libmarconet_hip.so" $v --only x2:16; done
  done
  tile "$PWD/marconet_amd/lib/libmarconet_hip.so" prod --only x2:16 --shape 1024,64,64,512,256 --ragged
  for v in early1 early5; do tile "$B/w4_$v/libmarconet_hip.so" $v --only x2:16 --shape 1024,64,64,512,256 --ragged; done
} 2>&1 | tee "$O/tile_rates_pieces_early.txt"
