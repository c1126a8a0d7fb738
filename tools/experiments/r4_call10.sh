#!/bin/bash
O=$PWD/gpurun_out/r4j; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_mx_gpu.py tests/test_modules_gpu.py -m gpu -q --tb=short -k "skip_conv_folded or packed_blob or parity_bar or sr_parity or scale_branch or software_pipelined" 2>&1 | tail -25 | tee $O/tests.log
for v in 0 1 0 1; do
  MNET_NO_FOLD_SKIP=$v timeout 400 python bench.py --steps 3 --warmup 1 --no-secondary --cpu-images 0 > $O/bench_nofold$v.log 2>&1
  echo "MNET_NO_FOLD_SKIP=$v: $(grep -o '"value": [0-9.]*, "unit": "images/s"' $O/bench_nofold$v.log)"
done | tee $O/bench.txt
