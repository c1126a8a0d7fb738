#!/bin/bash
# PMC passes + traffic JSON only (the kernel sources changed after the round's measurement set)
set -uo pipefail
R="$PWD"; O="$R/gpurun_out/${1:-r6y}"; mkdir -p "$O"; export TMPDIR=/tmp
export MNET_GIT_COMMIT="${2:-}"
bash tools/pmc_passes.sh "$O/pmc" -- python "$R/bench.py" --steps 1 --warmup 1 --cpu-images 0 --no-secondary --precision fp16x2
python tools/pmc_summary.py "$O/pmc" "$O/pmc_all_kernels_b256_fp16x2.txt" > /dev/null
python tools/pmc_summary.py "$O/pmc" "$O/pmc_conv_b256_fp16x2.txt" conv_ > /dev/null
python tools/pmc_traffic_json.py "$O/pmc_all_kernels_b256_fp16x2.txt" "$O/pmc_traffic_fp16x2.json" 256 fp16x2 | grep -A6 "w4_kernel"
rm -rf "$O"/pmc/*/
cp "$O/pmc_traffic_fp16x2.json" profiles/pmc_traffic_fp16x2.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > "$O/bench_driver_command.json" 2> "$O/bench_driver_command.err"; python -c "
import json
d=json.loads([l for l in open('$O/bench_driver_command.json') if l.startswith('{')][-1]); r=d['roofline']
print('driver command:', d['value'], 'img/s', d['ms_per_step'], 'ms |', r['kernel'], r['achieved'], 'TFLOP/s frac', r['frac'], '| tail', r['hbm_tail_ms_per_step'], '| traffic', r.get('traffic'), r.get('traffic_note'))"
