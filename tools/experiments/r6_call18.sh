#!/bin/bash
# final check of the round: PMC passes on the final kernel sources, the whole GPU tier, smoke(), the driver's bench command
set -uo pipefail
R="$PWD"; O="$R/gpurun_out/${1:-r6x}"; mkdir -p "$O"; export TMPDIR=/tmp
export MNET_GIT_COMMIT="${2:-}"
bash tools/pmc_passes.sh "$O/pmc" -- python "$R/bench.py" --steps 1 --warmup 1 --cpu-images 0 --no-secondary --precision fp16x2
python tools/pmc_summary.py "$O/pmc" "$O/pmc_all_kernels_b256_fp16x2.txt" > /dev/null
python tools/pmc_summary.py "$O/pmc" "$O/pmc_conv_b256_fp16x2.txt" conv_ > /dev/null
python tools/pmc_traffic_json.py "$O/pmc_all_kernels_b256_fp16x2.txt" "$O/pmc_traffic_fp16x2.json" 256 fp16x2 | grep -A7 "w4_kernel"
rm -rf "$O"/pmc/*/
cp "$O/pmc_traffic_fp16x2.json" profiles/pmc_traffic_fp16x2.json
timeout 2400 python -m pytest tests -m gpu -q --tb=short > "$O/tests_all.log" 2>&1; tail -3 "$O/tests_all.log"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke.log" 2>&1; tail -2 "$O/smoke.log"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o run -- python "$R/bench.py" --steps 3 --warmup 1 --cpu-images 0 --no-secondary ) > "$O/prof.log" 2>&1
f=$(find "$O/prof" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python tools/rocprof_summary.py "$f" "$O/bench_b256_n16_fp16x2_kernel_stats.txt" "rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --cpu-images 0 --no-secondary (batch 256, 16 glyphs; 3 timed + 2 instrumented + 1 warm-up steps)" > /dev/null; rm -rf "$O/prof"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > "$O/bench_driver_command.json" 2> "$O/bench_driver_command.err"; python -c "
import json
d=json.loads([l for l in open('$O/bench_driver_command.json') if l.startswith('{')][-1]); r=d['roofline']
print('driver command:', d['value'], 'img/s', d['ms_per_step'], 'ms |', r['kernel'], r['achieved'], 'TFLOP/s frac', r['frac'], '| tail', r['hbm_tail_ms_per_step'], '| traffic', r.get('traffic'), '| all-levels', d.get('value_all_levels_in_mode_precision'), 'no-image', d.get('value_without_prior_image'))"
