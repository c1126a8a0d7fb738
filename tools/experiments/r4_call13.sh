#!/bin/bash
# timing experiment: the software-pipelined fp16+8 tile with its scaled MFMAs typed fp6 (half the matrix-pipe cycles; 12-byte operand halves all read
# from the LDS, no conversions: what a stored-hi-copy fp16+6 format would execute) / with only the conversions replaced by an LDS read (exp_nc).
# Wrong results by construction — rates only.
O=$PWD/gpurun_out/r4m; mkdir -p $O; export TMPDIR=/tmp
run() { MARCONET_HIP_LIB=$1 timeout 120 python tools/tile_power_ab.py --seconds 3 --only x2:15 "${@:2}" 2>&1 | grep "TFLOP/s" | sed "s|^|$(basename $(dirname $1)) |"; }
B=marconet_amd/lib/libmarconet_hip.so
{
for rep in 1 2; do
  for v in $B tools/_build/exp_f6/libmarconet_hip.so tools/_build/exp_nc/libmarconet_hip.so; do run $v; done
done
echo "-- 512->256 @ 64x64 x 1024 ragged"
for v in $B tools/_build/exp_f6/libmarconet_hip.so $B tools/_build/exp_f6/libmarconet_hip.so; do run $v --shape 1024,64,64,512,256 --ragged; done
echo "-- zeros"
for v in $B tools/_build/exp_f6/libmarconet_hip.so; do run $v --zeros; done
} 2>&1 | tee $O/rates_fp6_typed.txt
