#!/bin/bash
export MNET_GIT_COMMIT=4294b4a
bash tools/round_profiles.sh r4zx fp16x2 2>&1 | tail -3
mkdir -p gpurun_out/r4zx
timeout 2000 python -m pytest tests -q -m gpu --tb=short 2>&1 | tail -4 | tee gpurun_out/r4zx/pytest_gpu_summary.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a gpurun_out/r4zx/pytest_gpu_summary.txt
