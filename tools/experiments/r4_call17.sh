#!/bin/bash
O=$PWD/gpurun_out/r4s; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_mx_gpu.py -q --tb=short -x 2>&1 | tail -5 | tee $O/tests.log
{
  timeout 150 python tools/tile_power_ab.py --seconds 3 --only x2:15,x2:11,x2:15
  timeout 150 python tools/tile_power_ab.py --seconds 3 --only x2:15 --shape 1024,64,64,512,256 --ragged
} 2>&1 | grep "TFLOP/s\|rror" | tee $O/rates.txt
timeout 200 python tools/slab_phases.py --swp 2>&1 | grep -v "Warn\|amdgpu.ids" | tee $O/slab_phases.txt
