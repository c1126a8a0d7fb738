#!/bin/bash
O=$PWD/gpurun_out/r4e; mkdir -p $O; export TMPDIR=/tmp; R=$PWD
for g in sq lds; do
  case $g in
    sq) C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU";;
    lds) C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE";;
  esac
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/$g -o pmc -- python $R/tools/tile_power_ab.py --launches 4 --only x2:11,x2:15 ) > $O/$g.log 2>&1
done
python tools/pmc_summary.py $O $O/pmc_swp_vs_lockstep.txt conv_dma
cat $O/pmc_swp_vs_lockstep.txt
rm -rf $O/sq $O/lds
