#!/bin/bash
export MNET_GIT_COMMIT=acedae3
bash tools/round_profiles.sh r4zy fp16x2 2>&1 | tail -3
mkdir -p gpurun_out/r4zy
timeout 2000 python -m pytest tests -q -m gpu --tb=short 2>&1 | tail -4 | tee gpurun_out/r4zy/pytest_gpu_summary.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee -a gpurun_out/r4zy/pytest_gpu_summary.txt
