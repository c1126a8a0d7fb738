#!/bin/bash
O=$PWD/gpurun_out/r4d; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_mx_gpu.py -q --tb=short -x -k "software_pipelined or every_lds_dma or bit_for_bit" 2>&1 | tail -15 | tee $O/tests.log
{
  timeout 150 python tools/tile_power_ab.py --seconds 3 --only x2:11,x2:15,x2:11,x2:15
  timeout 150 python tools/tile_power_ab.py --seconds 3 --only x2:11,x2:15 --shape 1024,64,64,512,256 --ragged
  timeout 150 python tools/tile_power_ab.py --seconds 3 --only x2:8,x2:9 --shape 64,128,2048,256,128
  echo "-- zeros"
  timeout 150 python tools/tile_power_ab.py --seconds 3 --only x2:11,x2:15 --zeros
} 2>&1 | grep "TFLOP/s\|rror\|--" | tee $O/rates.txt
