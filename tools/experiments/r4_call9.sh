#!/bin/bash
O=$PWD/gpurun_out/r4i; mkdir -p $O; export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -25 | tee $O/pytest_gpu.log
timeout 600 python bench.py --steps 3 --warmup 1 > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-300
