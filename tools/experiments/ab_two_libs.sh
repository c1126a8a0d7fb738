#!/bin/bash
# Same-box A/B of the production library against a variant build (tools/build_variant.sh), alternating the two:
#     bash tools/experiments/ab_two_libs.sh <tag> <variant name under tools/_build/> [reps]
# dominant-tile rates on the trunk and the glyph shape (tools/tile_power_ab.py), the strip / 128x512 tiles, the HBM-bound kernels
# (tools/tail_ab.py) and the bench line (no secondary figures) — everything into gpurun_out/<tag>/.
set -uo pipefail
TAG="$1"; VAR="$PWD/tools/_build/$2/libmarconet_hip.so"; REPS="${3:-2}"
O="$PWD/gpurun_out/$TAG"; mkdir -p "$O"; export TMPDIR=/tmp
PROD="$PWD/marconet_amd/lib/libmarconet_hip.so"
tile() { MARCONET_HIP_LIB=$1 timeout 200 python tools/tile_power_ab.py --seconds 2 "${@:3}" 2>&1 | grep "TFLOP/s" | sed "s|^|$2 |"; }
{
for rep in $(seq 1 "$REPS"); do
  tile "$PROD" prod --only x2:15; tile "$VAR" "$2" --only x2:15
  tile "$PROD" prod --only x2:15 --shape 1024,64,64,512,256 --ragged; tile "$VAR" "$2" --only x2:15 --shape 1024,64,64,512,256 --ragged
done
tile "$PROD" prod --only x2:8 --shape 64,64,1024,256,128; tile "$VAR" "$2" --only x2:8 --shape 64,64,1024,256,128
tile "$PROD" prod --only x2:s1 --shape 32,128,2048,64,64; tile "$VAR" "$2" --only x2:s1 --shape 32,128,2048,64,64
} 2>&1 | tee "$O/tile_rates_ab.txt"
for rep in $(seq 1 "$REPS"); do
  for which in prod "$2"; do
    lib="$PROD"; [ "$which" != "prod" ] && lib="$VAR"
    MARCONET_HIP_LIB="$lib" timeout 300 python bench.py --steps 5 --warmup 1 --no-secondary --cpu-images 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$which', d['value'], 'img/s', d['ms_per_step'], 'ms | dominant', r['achieved'], 'TFLOP/s | tail', r['hbm_tail_ms_per_step'], 'ms', {k:v['ms_per_step'] for k,v in r['hbm_tail']['by_kernel'].items()}, '| convs by kernel', r['all_conv_kernels']['by_kernel_ms_per_step'])"
  done
done 2>&1 | tee "$O/bench_ab.txt"
