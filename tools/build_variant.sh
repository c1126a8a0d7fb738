#!/bin/bash
# A/B build of ONE kernel source with extra flags, linked against the production objects:
#     tools/build_variant.sh <name> <source-stem> <hipcc flags...>   → tools/_build/<name>/libmarconet_hip.so
#     tools/build_variant.sh <name> all <hipcc flags...>             (every source with the flags: knobs that live in a shared header)
# select it at run time with MARCONET_HIP_LIB=tools/_build/<name>/libmarconet_hip.so (marconet_amd/_lib.py)
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
NAME=$1; STEM=$2; shift 2
SRC="$ROOT/marconet_amd/csrc"; LIB="$ROOT/marconet_amd/lib"; OUT="$ROOT/tools/_build/$NAME"
mkdir -p "$OUT"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
ALL="api conv_igemm conv_igemm_dma conv_dma_swp_gn conv_dma_w4 conv_strip_dma conv_skinny aux_kernels vit_kernels pack_kernels"
xf() { [ "$1" = conv_dma_swp_gn ] && echo "-mllvm -greedy-reverse-local-assignment=1"; }
if [ "$STEM" = "all" ]; then
  PIDS=()
  for f in $ALL; do ( "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $(xf "$f") "$@" -c "$SRC/$f.hip" -o "$OUT/$f.o" 2>/dev/null ) & PIDS+=($!); done
  for p in "${PIDS[@]}"; do wait "$p"; done
else
  "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $(xf "$STEM") "$@" -c "$SRC/$STEM.hip" -o "$OUT/$STEM.o"
fi
OBJS=""
for f in $ALL; do
  if [ "$f" = "$STEM" ] || [ "$STEM" = "all" ]; then OBJS="$OBJS $OUT/$f.o"; else OBJS="$OBJS $LIB/$f.o"; fi
done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC $OBJS -o "$OUT/libmarconet_hip.so"
echo "[variant] $OUT/libmarconet_hip.so"
