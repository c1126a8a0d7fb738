#!/bin/bash
# are the tile-closing output stores of the one-wave tile what its epilogue waits for (all CUs close their tiles in step: 64 MiB in one burst)?
#   st1 / st1_nostore / st1_stag4: phase stamps of the production tile | with the output descriptor empty (stores dropped) | with the workgroups entering in 8 phase groups
#   prod / nostore / stag2 / stag4: rates of the same builds without stamps
set -uo pipefail
O="$PWD/gpurun_out/${1:-r6w}"; mkdir -p "$O"; export TMPDIR=/tmp
B="$PWD/tools/_build"
for v in st1 st1_nostore st1_stag4; do echo "== $v"; MARCONET_HIP_LIB=$B/w4_$v/libmarconet_hip.so timeout 200 python tools/w4_phases.py 2>&1 | grep -v amdgpu.ids | cat; done | tee "$O/w4_phases_store_burst.txt"
tile() { MARCONET_HIP_LIB=$1 timeout 200 python tools/tile_power_ab.py --seconds 3 "${@:3}" 2>&1 | grep "TFLOP/s" | sed "s|^|$2 |"; }
{ for rep in 1 2; do
    tile "$PWD/marconet_amd/lib/libmarconet_hip.so" prod --only x2:16
    for v in nostore stag2 stag4; do tile "$B/w4_$v/libmarconet_hip.so" $v --only x2:16; done
  done
  tile "$PWD/marconet_amd/lib/libmarconet_hip.so" prod --only x2:16 --shape 1024,64,64,512,256 --ragged
  for v in stag2 stag4; do tile "$B/w4_$v/libmarconet_hip.so" $v --only x2:16 --shape 1024,64,64,512,256 --ragged; done
} 2>&1 | tee "$O/tile_rates_store_burst.txt"
