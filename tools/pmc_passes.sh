#!/bin/bash
# Hardware-counter passes for one command (rocprofv3 --pmc; counters only with --kernel-trace, each group in its own
# run because gfx950 has 8 SQ / 4 TCC slots per pass — MI355X_MICROARCH.md §rocprofv3 PMC slots).
# usage: tools/pmc_passes.sh <outdir> -- <command...>      (run from anywhere; outputs CSV under <outdir>/<group>/)
set -uo pipefail
OUT="$1"; shift; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
declare -A PMCG=(
  [sq]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVES"
  [lds]="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"
  [fetch]="FETCH_SIZE"
  [write]="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
)
for g in sq lds fetch write; do
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc ${PMCG[$g]} --output-format csv -d "$OUT/$g" -o pmc -- "$@" ) > "$OUT/$g.log" 2>&1 || echo "[pmc] group $g failed (see $OUT/$g.log)"
done
