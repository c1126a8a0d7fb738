#!/bin/bash
set -uo pipefail
O="$PWD/gpurun_out/${1:-r6j}"; mkdir -p "$O"; export TMPDIR=/tmp
MARCONET_HIP_LIB=$PWD/tools/_build/w4_stamps/libmarconet_hip.so timeout 200 python tools/w4_phases.py 2>&1 | grep -v amdgpu.ids | tee "$O/w4_phases.txt"
MARCONET_HIP_LIB=$PWD/tools/_build/w4_stamps/libmarconet_hip.so timeout 200 python tools/w4_phases.py --shape 1024,64,64,512,256 2>&1 | grep -v amdgpu.ids | tee "$O/w4_phases_glyph.txt"
