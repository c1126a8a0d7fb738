#!/bin/bash
O=$PWD/gpurun_out/r4t; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_mx_gpu.py -q --tb=short -x 2>&1 | tail -3 | tee $O/tests.log
V=tools/_build/no_xpose/libmarconet_hip.so
run() { MARCONET_HIP_LIB=$1 timeout 120 python tools/tile_power_ab.py --seconds 3 --only x2:15 "${@:2}" 2>&1 | grep "TFLOP/s" | sed "s|^|$(basename $(dirname $1)) |"; }
{
for rep in 1 2; do run marconet_amd/lib/libmarconet_hip.so; run $V; done
echo "-- 512->256 @ 64x64 x 1024 ragged"
run marconet_amd/lib/libmarconet_hip.so --shape 1024,64,64,512,256 --ragged; run $V --shape 1024,64,64,512,256 --ragged
} 2>&1 | tee $O/rates_ab.txt
timeout 200 python tools/slab_phases.py --swp 2>&1 | grep -v "Warn\|amdgpu.ids" | tee $O/slab_phases.txt
