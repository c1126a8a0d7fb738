#!/bin/bash
# Measurement set of a round, run ON THE GPU BOX from the repo root:  bash tools/round_profiles.sh <tag> [precision]
# writes everything under gpurun_out/<tag>/ (copy what is to be judged into profiles/).
#   bench JSON lines (default / gan / mixed / force-gather), rocprofv3 --kernel-trace --stats summaries, the four PMC passes
#   (tools/pmc_passes.sh: each counter group in its own run, --kernel-trace only) and the traffic JSON bench.py reads.
set -uo pipefail
TAG="$1"; PREC="${2:-fp16x2}"
R="$PWD"; O="$R/gpurun_out/$TAG"; mkdir -p "$O"
export TMPDIR=/tmp
stats() {   # <name> <note> -- <bench args...>
  local name="$1" note="$2"; shift 3
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_$name" -o run -- python "$R/bench.py" "$@" ) > "$O/prof_$name.log" 2>&1
  local f; f=$(find "$O/prof_$name" -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then python "$R/tools/rocprof_summary.py" "$f" "$O/${name}_kernel_stats.txt" "$note" > /dev/null; else echo "[round_profiles] no kernel stats for $name"; fi
}
timeout 600 python bench.py --steps 5 --warmup 1 --precision "$PREC" > "$O/bench_${PREC}.json" 2> "$O/bench_${PREC}.err"
stats "bench_b256_n16_${PREC}" "rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --cpu-images 0 --no-secondary --precision $PREC (batch 256, 16 glyphs; 3 timed + 2 instrumented + 1 warm-up steps)" -- --steps 3 --warmup 1 --cpu-images 0 --no-secondary --precision "$PREC"
bash tools/pmc_passes.sh "$O/pmc" -- python "$R/bench.py" --steps 1 --warmup 1 --cpu-images 0 --no-secondary --precision "$PREC"
python tools/pmc_summary.py "$O/pmc" "$O/pmc_all_kernels_b256_${PREC}.txt" > /dev/null
python tools/pmc_summary.py "$O/pmc" "$O/pmc_conv_b256_${PREC}.txt" conv_ > /dev/null
python tools/pmc_traffic_json.py "$O/pmc_all_kernels_b256_${PREC}.txt" "$O/pmc_traffic_${PREC}.json" 256 "$PREC" > /dev/null
timeout 400 python bench.py --config gan --steps 3 --precision "$PREC" > "$O/bench_gan_${PREC}.json" 2> "$O/bench_gan.err"
stats "bench_gan_b256x16_${PREC}" "rocprofv3 --kernel-trace --stats -- python bench.py --config gan --steps 2 --cpu-images 0 --precision $PREC (configs[3]: 4096 glyphs)" -- --config gan --steps 2 --cpu-images 0 --precision "$PREC"
timeout 400 python bench.py --config mixed --steps 3 --precision "$PREC" > "$O/bench_mixed_${PREC}.json" 2> "$O/bench_mixed.err"
stats "bench_mixed_b256_${PREC}" "rocprofv3 --kernel-trace --stats -- python bench.py --config mixed --steps 2 --cpu-images 0 --precision $PREC (configs[4]: widths 128..512)" -- --config mixed --steps 2 --cpu-images 0 --precision "$PREC"
timeout 400 python bench.py --steps 3 --force-gather --cpu-images 0 --precision "$PREC" > "$O/bench_force_gather_${PREC}.json" 2> "$O/bench_force_gather.err"
for p in fp16x2 fp16x3 fp16; do timeout 200 python tools/graph_latency.py 16 50 $p; done > "$O/graph_latency.txt" 2>&1
timeout 400 python tools/tile_power_ab.py --seconds 5 --only x3:11,x2:6,x2:11,x2:15,x2:16,f16:16 > "$O/tile_power_ab.txt" 2>&1
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$O/pmc_tiles" -o pmc -- python "$R/tools/tile_power_ab.py" --launches 6 --only x3:11,x2:11,x2:15,x2:16 ) > "$O/pmc_tiles.log" 2>&1
python tools/pmc_summary.py "$O/pmc_tiles" "$O/pmc_tiles_x3_mf16_mf32_x2.txt" conv_dma > /dev/null
# per-launch conv table of one step, per-phase cycles of the dominant tile (lock-step and software-pipelined form), micro-benchmarks
timeout 400 python tools/profile_layers.py --batch 256 --glyphs 16 --precision "$PREC" --out "$O/conv_layers_b256_n16_${PREC}.txt" > /dev/null 2>&1
{ timeout 200 python tools/slab_phases.py; timeout 200 python tools/slab_phases.py --swp; timeout 200 python tools/slab_phases.py --swp --zeros; } 2>&1 | grep -v "Warn\|amdgpu.ids" > "$O/slab_phases.txt"
# the one-wave-per-SIMD tile (id 16): phase stamps need the diagnostic build (tools/build_variant.sh w4_stamps conv_dma_w4 -DW4_STAMPS=1, built before the lease)
[ -f tools/_build/w4_stamps/libmarconet_hip.so ] && { MARCONET_HIP_LIB=$R/tools/_build/w4_stamps/libmarconet_hip.so timeout 200 python tools/w4_phases.py; MARCONET_HIP_LIB=$R/tools/_build/w4_stamps/libmarconet_hip.so timeout 200 python tools/w4_phases.py --zeros;
  MARCONET_HIP_LIB=$R/tools/_build/w4_stamps/libmarconet_hip.so timeout 200 python tools/w4_phases.py --shape 1024,64,64,512,256; } 2>&1 | grep -v "Warn\|amdgpu.ids" > "$O/w4_phases.txt"
[ -x tools/_build/lds_dma_peak ] && timeout 120 tools/_build/lds_dma_peak > "$O/lds_dma_peak_microbench.txt" 2>&1
[ -x tools/_build/mfma_slab ] && { timeout 120 tools/_build/mfma_slab z; timeout 120 tools/_build/mfma_slab; } > "$O/mfma_slab_microbench.txt" 2>&1
rm -rf "$O"/prof_*/ "$O"/pmc/*/ "$O/pmc_tiles"/*/ 2>/dev/null      # raw traces are large; the summaries stay
ls -la "$O" | head -40
