#!/bin/bash
O=$PWD/gpurun_out/r4l; mkdir -p $O; export TMPDIR=/tmp
for v in default swp_prio1 swp_prio2 default swp_prio1; do
  if [ $v = default ]; then unset MARCONET_HIP_LIB; else export MARCONET_HIP_LIB=$PWD/tools/_build/$v/libmarconet_hip.so; fi
  echo "== $v"
  timeout 150 python tools/tile_power_ab.py --seconds 3 --only x2:15 2>&1 | grep TFLOP
  timeout 150 python tools/tile_power_ab.py --seconds 3 --only x2:15 --shape 1024,64,64,512,256 --ragged 2>&1 | grep TFLOP
done | tee $O/rates.txt
