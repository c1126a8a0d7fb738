#!/bin/bash
set -uo pipefail
O="$PWD/gpurun_out/${1:-r6r}"; mkdir -p "$O"; export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_regimes_gpu.py tests/test_bench_contract_gpu.py tests/test_modules_gpu.py -m gpu -q --tb=short -s -k "regime or contract or config_lines or gan or prior or graph" > "$O/tests_sel.log" 2>&1; grep "trained_like gan\|passed\|failed\|FAILED" "$O/tests_sel.log" | tail -12
cp gpurun_out/regime_parity.json "$O/" 2>/dev/null
