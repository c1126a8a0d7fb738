#!/bin/bash
set -uo pipefail
O="$PWD/gpurun_out/${1:-r6f}"; mkdir -p "$O"; export TMPDIR=/tmp
MARCONET_HIP_LIB=$PWD/tools/_build/w4_stamps2/libmarconet_hip.so timeout 200 python tools/w4_phases.py 2>&1 | grep -v amdgpu.ids | tee "$O/w4_phases_epilogue.txt"
