#!/bin/bash
set -uo pipefail
O="$PWD/gpurun_out/r6b"; mkdir -p "$O"; export TMPDIR=/tmp
timeout 300 python tools/experiments/w4_debug.py > "$O/w4_debug.txt" 2>&1; cat "$O/w4_debug.txt" | grep -v amdgpu.ids | head -190
