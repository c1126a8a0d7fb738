#!/bin/bash
# the 64x512 strip tile with the waves' DMA roles split (weights: waves 0-1, strips: waves 2-7; a strip wave waits only in front of its strip's first slab)
set -uo pipefail
O="$PWD/gpurun_out/${1:-r6at}"; mkdir -p "$O"; export TMPDIR=/tmp
B="$PWD/tools/_build"
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_mx_gpu.py tests/test_split_gpu.py -m gpu -q --tb=short -k "strip or fuzz or agree or cout64 or cout_64" > "$O/tests_strip.log" 2>&1; tail -3 "$O/tests_strip.log"
tile() { MARCONET_HIP_LIB=$1 timeout 200 python tools/tile_power_ab.py --seconds 3 "${@:3}" 2>&1 | grep "TFLOP/s" | sed "s|^|$2 |"; }
{ for rep in 1 2; do
    tile "$B/strip_noroles/libmarconet_hip.so" noroles --only x2:s1 --shape 64,128,2048,64,64
    tile "$PWD/marconet_amd/lib/libmarconet_hip.so" roles --only x2:s1 --shape 64,128,2048,64,64
  done
  tile "$B/strip_noroles/libmarconet_hip.so" noroles --only x3:s1 --shape 64,128,2048,64,64
  tile "$PWD/marconet_amd/lib/libmarconet_hip.so" roles --only x3:s1 --shape 64,128,2048,64,64
  tile "$B/strip_noroles/libmarconet_hip.so" noroles --only f16:s1 --shape 64,128,2048,64,64
  tile "$PWD/marconet_amd/lib/libmarconet_hip.so" roles --only f16:s1 --shape 64,128,2048,64,64
} 2>&1 | tee "$O/strip_roles_rates.txt"
for v in noroles prod noroles prod; do
  lib="$B/strip_$v/libmarconet_hip.so"; [ $v = prod ] && lib="$PWD/marconet_amd/lib/libmarconet_hip.so"
  MARCONET_HIP_LIB=$lib timeout 600 python bench.py --steps 4 --warmup 2 --cpu-images 0 --no-secondary 2>/dev/null > "$O/line.json"; echo "$v $(python tools/experiments/print_line.py $O/line.json | cut -c1-110) strip $(python -c "
import json
d=json.loads([l for l in open('$O/line.json') if l.startswith('{')][-1]); print({k: round(v, 1) for k, v in d['roofline']['all_conv_kernels']['by_kernel_ms_per_step'].items() if 'strip' in k})")"
done | tee "$O/bench_strip_roles_ab.txt"
