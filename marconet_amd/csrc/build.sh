#!/bin/bash
# Builds libmarconet_hip.so for gfx950 in-tree (the .so travels to the GPU box with the snapshot).
# An object is rebuilt whenever the CONTENT of what it is made from changes: every object carries a stamp (<name>.o.sha) = SHA-256 over its
# source, the shared headers, the compiler flags and the compiler's version string.  (Round 4 keyed on mtimes: a checkout, a `git stash` or a
# copied tree could leave a stale object that linked silently.)  MNET_CLEAN=1 throws everything away first, so that the build PROVES a compile.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../lib"
mkdir -p "$OUT"
if [ "${MNET_CLEAN:-0}" != "0" ]; then rm -f "$OUT"/*.o "$OUT"/*.o.tmp "$OUT"/*.o.sha "$OUT"/libmarconet_hip.so "$OUT"/libmarconet_hip.so.sha; echo "[build] MNET_CLEAN: objects removed, full compile"; fi
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
PIDS=()
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
SRCS="api conv_igemm conv_igemm_dma conv_dma_swp_gn conv_dma_w4 conv_strip_dma conv_skinny aux_kernels vit_kernels pack_kernels"
# per-source extra flags (conv_dma_swp_gn: see the note at its top)
extra() { case "$1" in conv_dma_swp_gn) echo "-mllvm -greedy-reverse-local-assignment=1" ;; *) echo "" ;; esac; }
HDRS="$HERE/common.h $HERE/conv_args.h $HERE/conv_dma_common.h $HERE/../../include/marconet_hip.h"
CCVER="$("$HIPCC" --version 2>/dev/null | head -3 | tr '\n' ' ')"
stamp() { { cat "$HERE/$1.hip" $HDRS; [ "$1" = conv_dma_swp_gn ] && cat "$HERE/conv_igemm_dma.hip"; echo "$FLAGS $(extra "$1") ${EXTRA_HIPCC_FLAGS:-} | $CCVER"; } | sha256sum | cut -d' ' -f1; }
for f in $SRCS; do
  want="$(stamp "$f")"
  if [ ! -f "$OUT/$f.o" ] || [ ! -f "$OUT/$f.o.sha" ] || [ "$(cat "$OUT/$f.o.sha")" != "$want" ]; then
    echo "[build] hipcc $f.hip"
    rm -f "$OUT/$f.o.sha"
    ( "$HIPCC" $FLAGS $(extra "$f") ${EXTRA_HIPCC_FLAGS:-} -c "$HERE/$f.hip" -o "$OUT/$f.o.tmp" && mv "$OUT/$f.o.tmp" "$OUT/$f.o" && echo "$want" > "$OUT/$f.o.sha" ) &
    PIDS+=($!)
  fi
done
for p in "${PIDS[@]:-}"; do        # a failed compile must fail the build (a stale object would otherwise be linked)
  if [ -n "$p" ]; then wait "$p"; fi
done
OBJS=""; ALL=""
for f in $SRCS; do OBJS="$OBJS $OUT/$f.o"; ALL="$ALL$(cat "$OUT/$f.o.sha")"; done
LSHA="$(echo "$ALL" | sha256sum | cut -d' ' -f1)"
if [ ! -f "$OUT/libmarconet_hip.so" ] || [ ! -f "$OUT/libmarconet_hip.so.sha" ] || [ "$(cat "$OUT/libmarconet_hip.so.sha")" != "$LSHA" ]; then
  "$HIPCC" --offload-arch=gfx950 -shared -fPIC $OBJS -o "$OUT/libmarconet_hip.so"
  echo "$LSHA" > "$OUT/libmarconet_hip.so.sha"
fi
echo "[build] $OUT/libmarconet_hip.so (sources $LSHA)"
