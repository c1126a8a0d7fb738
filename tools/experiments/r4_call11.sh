#!/bin/bash
O=$PWD/gpurun_out/r4k; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_mx_gpu.py -q --tb=short -x -k "software_pipelined" 2>&1 | tail -5 | tee $O/tests.log
{
  timeout 150 python tools/tile_power_ab.py --seconds 3 --only x2:11,x2:15,x2:16,x2:15,x2:16
  timeout 150 python tools/tile_power_ab.py --seconds 3 --only x2:15,x2:16 --shape 1024,64,64,512,256 --ragged
  echo "-- zeros"
  timeout 150 python tools/tile_power_ab.py --seconds 3 --only x2:15,x2:16 --zeros
} 2>&1 | grep "TFLOP/s\|rror\|--" | tee $O/rates.txt
MNET_DIAG_SWP=2 timeout 200 python tools/slab_phases.py --swp 2>&1 | grep -v "Warn\|amdgpu" | tee $O/phases_cr.txt
