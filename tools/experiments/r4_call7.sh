#!/bin/bash
O=$PWD/gpurun_out/r4g; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_mx_gpu.py -q --tb=short -x -k "software_pipelined" 2>&1 | tail -5 | tee $O/tests.log
{
  timeout 150 python tools/tile_power_ab.py --seconds 3 --only x2:11,x2:15,x2:16,x2:11,x2:15
  timeout 150 python tools/tile_power_ab.py --seconds 3 --only x2:11,x2:15 --shape 1024,64,64,512,256 --ragged
  timeout 150 python tools/tile_power_ab.py --seconds 3 --only x2:11,x2:15 --shape 1024,64,64,256,256 --ragged
  timeout 150 python tools/tile_power_ab.py --seconds 3 --only x2:11,x2:15 --shape 1024,32,32,512,512
} 2>&1 | grep "TFLOP/s\|rror\|--" | tee $O/rates.txt
for cfg in 11 15 11 15; do
  MNET_MX_CFG256=$cfg timeout 400 python bench.py --steps 3 --warmup 1 --no-secondary --cpu-images 0 > $O/bench_$cfg.log 2>&1
  echo "cfg256=$cfg: $(grep -o '"value": [0-9.]*, "unit": "images/s"' $O/bench_$cfg.log)"
done | tee $O/bench.txt
