#!/bin/bash
# A/B build of the whole library from a PATCHED copy of marconet_amd/csrc (the tree stays untouched):
#     tools/build_variant_patch.sh <name> <patch file> [hipcc flags...]   → tools/_build/<name>/libmarconet_hip.so
# select it at run time with MARCONET_HIP_LIB=$PWD/tools/_build/<name>/libmarconet_hip.so (marconet_amd/_lib.py)
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
NAME=$1; PATCH=$(realpath "$2"); shift 2
W="$ROOT/tools/_build/$NAME"; rm -rf "$W"; mkdir -p "$W/src/marconet_amd" "$W/src/include"
cp -r "$ROOT/marconet_amd/csrc" "$W/src/marconet_amd/csrc"; cp "$ROOT/include/"*.h "$W/src/include/"
( cd "$W/src" && patch -p1 --quiet < "$PATCH" )
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
PIDS=()
xf() { [ "$1" = conv_dma_swp_gn ] && echo "-mllvm -greedy-reverse-local-assignment=1"; }
for f in api conv_igemm conv_igemm_dma conv_dma_swp_gn conv_strip_dma conv_skinny aux_kernels vit_kernels pack_kernels; do
  ( "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $(xf "$f") "$@" -c "$W/src/marconet_amd/csrc/$f.hip" -o "$W/$f.o" ) &
  PIDS+=($!)
done
for p in "${PIDS[@]}"; do wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "$W"/*.o -o "$W/libmarconet_hip.so"
rm -rf "$W/src" "$W"/*.o
echo "[variant] $W/libmarconet_hip.so"
