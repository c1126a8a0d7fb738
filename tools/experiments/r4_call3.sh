#!/bin/bash
O=$PWD/gpurun_out/r4c; mkdir -p $O; export TMPDIR=/tmp
timeout 120 tools/_build/lds_dma_peak 2>&1 | tee $O/lds_dma_peak.txt
timeout 200 python tools/slab_phases.py 2>&1 | grep -v Warn | tee $O/slab_phases.txt
timeout 200 python tools/slab_phases.py --zeros 2>&1 | grep -v Warn | tee $O/slab_phases_zeros.txt
timeout 200 python tools/slab_phases.py --shape 1024,64,64,512,256 2>&1 | grep -v Warn | tee $O/slab_phases_glyph.txt
