#!/bin/bash
# A/B on one box: the fp16+8 256x256 tiles' epilogue stores through the LDS (production build) vs every lane storing its own blocks (-DMNET_MX_XPOSE=0)
O=$PWD/gpurun_out/r4s; mkdir -p $O; export TMPDIR=/tmp
V=tools/_build/no_xpose/libmarconet_hip.so
run() { MARCONET_HIP_LIB=$1 timeout 120 python tools/tile_power_ab.py --seconds 3 --only x2:15 "${@:2}" 2>&1 | grep "TFLOP/s" | sed "s|^|$(basename $(dirname $1)) |"; }
{
for rep in 1 2; do run marconet_amd/lib/libmarconet_hip.so; run $V; done
echo "-- 512->256 @ 64x64 x 1024 ragged"
for rep in 1 2; do run marconet_amd/lib/libmarconet_hip.so --shape 1024,64,64,512,256 --ragged; run $V --shape 1024,64,64,512,256 --ragged; done
} 2>&1 | tee $O/rates_ab.txt
timeout 400 python bench.py --steps 3 --warmup 1 --cpu-images 0 --no-secondary > $O/bench.json 2> $O/bench.err
MARCONET_HIP_LIB=$V timeout 400 python bench.py --steps 3 --warmup 1 --cpu-images 0 --no-secondary > $O/bench_no_xpose.json 2> $O/bench_no_xpose.err
timeout 400 python bench.py --steps 3 --warmup 1 --cpu-images 0 --no-secondary > $O/bench2.json 2> $O/bench2.err
python - <<'PY'
import json
for f in ('bench','bench_no_xpose','bench2'):
    d=json.loads(open('gpurun_out/r4s/%s.json'%f).read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['kernel_ms_per_step'])
PY
