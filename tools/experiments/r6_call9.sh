#!/bin/bash
set -uo pipefail
O="$PWD/gpurun_out/${1:-r6l}"; mkdir -p "$O"; export TMPDIR=/tmp
timeout 200 python tools/experiments/subnormal_debug.py 2>&1 | grep -v amdgpu.ids | tee "$O/subnormal_debug.txt"
timeout 2400 python -m pytest tests -m gpu -q --tb=short --deselect tests/test_round6_gpu.py::test_lo_byte_encoders_agree_on_subnormal_blocks > "$O/tests_all.log" 2>&1; tail -5 "$O/tests_all.log"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6
