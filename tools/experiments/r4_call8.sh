#!/bin/bash
O=$PWD/gpurun_out/r4h; mkdir -p $O; export TMPDIR=/tmp
timeout 500 python bench.py --steps 3 --warmup 1 > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-400
timeout 2400 python -m pytest tests -m gpu -q --tb=short -x 2>&1 | tail -15 | tee $O/pytest_gpu.log
