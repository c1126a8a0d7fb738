#!/bin/bash
# the lock-step fp16+8 tiles whose stages fill the LDS (128x512): transposition scratch in the stage the last slab was read from instead of direct block stores
set -uo pipefail
O="$PWD/gpurun_out/${1:-r6av}"; mkdir -p "$O"; export TMPDIR=/tmp
B="$PWD/tools/_build"
timeout 1500 python -m pytest tests/test_mx_gpu.py tests/test_kernels_gpu.py tests/test_split_gpu.py -m gpu -q --tb=short > "$O/tests.log" 2>&1; tail -3 "$O/tests.log"
tile() { MARCONET_HIP_LIB=$1 timeout 200 python tools/tile_power_ab.py --seconds 3 "${@:3}" 2>&1 | grep "TFLOP/s" | sed "s|^|$2 |"; }
{ for rep in 1 2; do
    tile "$B/dma_epi0/libmarconet_hip.so" direct --only x2:8 --shape 256,128,128,256,128
    tile "$PWD/marconet_amd/lib/libmarconet_hip.so" staged --only x2:8 --shape 256,128,128,256,128
  done; } 2>&1 | tee "$O/tile_rates_128x512_epilogue.txt"
for v in epi0 prod epi0 prod; do
  lib="$B/dma_$v/libmarconet_hip.so"; [ $v = prod ] && lib="$PWD/marconet_amd/lib/libmarconet_hip.so"
  MARCONET_HIP_LIB=$lib timeout 600 python bench.py --steps 4 --warmup 2 --cpu-images 0 --no-secondary 2>/dev/null > "$O/line.json"; echo "$v $(python tools/experiments/print_line.py $O/line.json | cut -c1-100) $(python -c "
import json
d=json.loads([l for l in open('$O/line.json') if l.startswith('{')][-1]); print({k: round(v, 1) for k, v in d['roofline']['all_conv_kernels']['by_kernel_ms_per_step'].items() if '128,512,1,8' in k})")"
done | tee "$O/bench_128x512_epilogue_ab.txt"
