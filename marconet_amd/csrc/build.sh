#!/bin/bash
# Builds libmarconet_hip.so for gfx950 in-tree (the .so travels to the GPU box with the snapshot).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../lib"
mkdir -p "$OUT"
# MNET_CLEAN=1: throw the objects away first, so that the build PROVES a compile (the default is incremental on mtimes: a tree
# that already holds fresh objects only re-links)
if [ "${MNET_CLEAN:-0}" != "0" ]; then rm -f "$OUT"/*.o "$OUT"/*.o.tmp "$OUT"/libmarconet_hip.so; echo "[build] MNET_CLEAN: objects removed, full compile"; fi
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
PIDS=()
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
for f in api conv_igemm conv_igemm_dma conv_strip_dma conv_skinny aux_kernels vit_kernels pack_kernels; do
  if [ ! -f "$OUT/$f.o" ] || [ "$HERE/$f.hip" -nt "$OUT/$f.o" ] || [ "$HERE/common.h" -nt "$OUT/$f.o" ] || [ "$HERE/conv_args.h" -nt "$OUT/$f.o" ] || [ "$HERE/conv_dma_common.h" -nt "$OUT/$f.o" ] \
     || [ "$HERE/../../include/marconet_hip.h" -nt "$OUT/$f.o" ]; then
    echo "[build] hipcc $f.hip"
    ( "$HIPCC" $FLAGS ${EXTRA_HIPCC_FLAGS:-} -c "$HERE/$f.hip" -o "$OUT/$f.o.tmp" && mv "$OUT/$f.o.tmp" "$OUT/$f.o" ) &
    PIDS+=($!)
  fi
done
for p in "${PIDS[@]:-}"; do        # a failed compile must fail the build (a stale object would otherwise be linked)
  if [ -n "$p" ]; then wait "$p"; fi
done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "$OUT"/api.o "$OUT"/conv_igemm.o "$OUT"/conv_igemm_dma.o "$OUT"/conv_strip_dma.o "$OUT"/conv_skinny.o "$OUT"/aux_kernels.o "$OUT"/vit_kernels.o "$OUT"/pack_kernels.o -o "$OUT/libmarconet_hip.so"
echo "[build] $OUT/libmarconet_hip.so"
