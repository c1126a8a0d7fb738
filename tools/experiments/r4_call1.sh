#!/bin/bash
# round 4, GPU call 1: ping-pong tiles (ids 14 / 15) — correctness, then interleaved rates in three builds, then the step
O=$PWD/gpurun_out/r4a; mkdir -p $O
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_mx_gpu.py -q --tb=short -x -k "ping_pong or every_lds_dma or bit_for_bit" 2>&1 | tail -15 | tee $O/tests.log
if ! grep -q "passed" $O/tests.log || grep -q "failed\|error" $O/tests.log; then echo "ping-pong tiles are not correct"; PPBAD=1; fi
for v in default pp_dma_first pp_under_scaled; do
  if [ $v = default ]; then unset MARCONET_HIP_LIB; else export MARCONET_HIP_LIB=$PWD/tools/_build/$v/libmarconet_hip.so; fi
  echo "== build $v" | tee -a $O/rates.txt
  {
    timeout 120 python tools/tile_power_ab.py --seconds 3 --only x2:11,x2:14,x2:11,x2:14
    timeout 120 python tools/tile_power_ab.py --seconds 3 --only x2:11,x2:14 --shape 1024,64,64,512,256 --ragged
    timeout 120 python tools/tile_power_ab.py --seconds 3 --only x2:8,x2:15 --shape 64,128,2048,256,128
  } 2>&1 | grep "TFLOP/s\|rror" | tee -a $O/rates.txt
done
unset MARCONET_HIP_LIB
if [ -z "$PPBAD" ]; then
for cfg in "11 8" "14 8" "14 15"; do
  set -- $cfg
  MNET_MX_CFG256=$1 MNET_MX_CFG128=$2 timeout 400 python bench.py --steps 3 --warmup 1 --no-secondary --cpu-images 0 > $O/bench_$1_$2.log 2>&1
  echo "cfg256=$1 cfg128=$2: $(grep -o '"value": [0-9.]*, "unit": "images/s"' $O/bench_$1_$2.log)"
done | tee $O/bench.txt
fi
