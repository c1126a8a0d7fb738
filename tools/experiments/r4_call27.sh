#!/bin/bash
O=$PWD/gpurun_out/r4z3; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_mx_gpu.py -q --tb=short -x 2>&1 | tail -3 | tee $O/tests.log
V=tools/_build/pre_setup/libmarconet_hip.so
run() { MARCONET_HIP_LIB=$1 timeout 120 python tools/tile_power_ab.py --seconds 3 --only x2:15 "${@:2}" 2>&1 | grep "TFLOP/s" | sed "s|^|$(basename $(dirname $1)) |"; }
{
for rep in 1 2 3; do run marconet_amd/lib/libmarconet_hip.so; run $V; done
} 2>&1 | tee $O/rates_ab.txt
