#!/bin/bash
O=$PWD/gpurun_out/r4z5; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_mx_gpu.py -q --tb=short -x -k "tile or pipelined or plain or fuzz" 2>&1 | tail -2 | tee $O/tests.log
V=tools/_build/prev/libmarconet_hip.so
run() { MARCONET_HIP_LIB=$1 timeout 120 python tools/tile_power_ab.py --seconds 2 --only x2:15 "${@:2}" 2>&1 | grep "TFLOP/s" | sed "s|^|$(basename $(dirname $1)) |"; }
{
for rep in 1 2; do run marconet_amd/lib/libmarconet_hip.so; run $V; done
run marconet_amd/lib/libmarconet_hip.so --shape 1024,64,64,256,256 --ragged; run $V --shape 1024,64,64,256,256 --ragged
} 2>&1 | tee $O/rates_ab.txt
