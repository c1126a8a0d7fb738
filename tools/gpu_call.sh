#!/bin/bash
# One GPU lease = one call of this script ON THE GPU BOX, from the repo root:   bash tools/gpu_call.sh <tag> <step> [<step> ...]
# (replaces round 4's 29 one-off tools/experiments/r4_call*.sh; they are in the history at commit 273fd17).  Everything a step writes goes to
# gpurun_out/<tag>/ — copy what is to be judged into profiles/.  Steps:
#   tests            the whole GPU tier (pytest -m gpu) + smoke()
#   tests:<expr>     pytest -m gpu -k "<expr>"
#   bench            python bench.py --steps 5 --warmup 2            (the line the driver produces, shorter)
#   bench:<args>     python bench.py <args>                           (spaces as '+': bench:--steps+3+--glyph-chunk+2048)
#   tail             tools/tail_ab.py: the HBM-bound kernels at bench shapes, every A/B form
#   tile:<args>      tools/tile_power_ab.py <args>
#   profiles         tools/round_profiles.sh <tag> (bench lines, rocprofv3 --stats summaries, PMC passes, traffic JSON, graph latency)
#   py:<script+args> python <script> <args>
set -uo pipefail
TAG="$1"; shift
O="$PWD/gpurun_out/$TAG"; mkdir -p "$O"; export TMPDIR=/tmp
k=0
for step in "$@"; do
  k=$((k + 1)); name="${step%%:*}"; arg=""; [ "$step" != "$name" ] && arg="${step#*:}"; arg="${arg//+/ }"
  t0=$(date +%s)
  case "$name" in
    tests)
      if [ -n "$arg" ]; then timeout 1500 python -m pytest tests -m gpu -q --tb=short -x -k "$arg" > "$O/${k}_tests.log" 2>&1
      else { timeout 2400 python -m pytest tests -m gpu -q --tb=short; timeout 600 python -c "import __graft_entry__ as g; g.smoke()"; } > "$O/${k}_tests.log" 2>&1; fi
      tail -4 "$O/${k}_tests.log" ;;
    bench)
      [ -z "$arg" ] && arg="--steps 5 --warmup 2"
      timeout 1500 python bench.py $arg > "$O/${k}_bench.json" 2> "$O/${k}_bench.err"; tail -c 600 "$O/${k}_bench.json"; echo ;;
    tail) timeout 900 python tools/tail_ab.py $arg > "$O/${k}_tail_ab.txt" 2>&1; tail -40 "$O/${k}_tail_ab.txt" ;;
    tile) timeout 900 python tools/tile_power_ab.py $arg > "$O/${k}_tile_ab.txt" 2>&1; grep -i "tflop" "$O/${k}_tile_ab.txt" | tail -20 ;;
    profiles) bash tools/round_profiles.sh "$TAG" > "$O/${k}_profiles.log" 2>&1; tail -5 "$O/${k}_profiles.log" ;;
    py) timeout 1500 python $arg > "$O/${k}_py.log" 2>&1; tail -30 "$O/${k}_py.log" ;;
    *) echo "[gpu_call] unknown step $step" ;;
  esac
  echo "[gpu_call] step $k ($step): $(( $(date +%s) - t0 )) s"
done
