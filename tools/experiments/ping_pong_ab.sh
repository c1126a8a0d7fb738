#!/bin/bash
# Next-round experiment: ping-pong tiles (ids 14 / 15) against the production fp16+8 tiles (ids 11 / 8).
# From the repo root on the GPU box AFTER `git apply tools/experiments/ping_pong.patch && bash marconet_amd/csrc/build.sh`:
#     bash tools/experiments/ping_pong_ab.sh            (about 6 GPU-minutes; every step under its own timeout: a schedule bug would hang a kernel)
O=$PWD/gpurun_out/ping_pong; mkdir -p $O
# 1. correctness first: byte-equal to the production tiles (multi-pass grids, one slab per tile, tails), every-tile and bit-for-bit tests
timeout 300 python -m pytest tests/test_mx_gpu.py -q --tb=short -x -k "ping_pong or every_lds_dma or bit_for_bit" 2>&1 | tail -15 | tee $O/tests.log
grep -q "passed" $O/tests.log && ! grep -q "failed\|error" $O/tests.log || { echo "ping-pong tiles are not correct: stop here"; exit 1; }
# 2. sustained rate, power and clock on the three shapes that carry the step (trunk 256->256, glyph 512->256, final 256->128)
{
  timeout 200 python tools/tile_power_ab.py --seconds 5 --only x2:11,x2:14
  timeout 200 python tools/tile_power_ab.py --seconds 5 --only x2:11,x2:14 --shape 1024,64,64,512,256 --ragged
  timeout 200 python tools/tile_power_ab.py --seconds 5 --only x2:8,x2:15 --shape 64,128,2048,256,128
} 2>&1 | grep "TFLOP/s" | tee $O/rates.txt
# 3. the whole step with the ping-pong tiles selected (A/B knobs of conv_dma_pick), against the default
for cfg in "11 8" "14 8" "14 15"; do
  set -- $cfg
  MNET_MX_CFG256=$1 MNET_MX_CFG128=$2 timeout 400 python bench.py --steps 3 --warmup 1 --no-secondary --cpu-images 0 > $O/bench_$1_$2.log 2>&1
  echo "cfg256=$1 cfg128=$2: $(grep -o '"value": [0-9.]*, "unit": "images/s"' $O/bench_$1_$2.log)"
done | tee $O/bench.txt
