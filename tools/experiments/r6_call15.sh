#!/bin/bash
set -uo pipefail
O="$PWD/gpurun_out/${1:-r6z}"; mkdir -p "$O"; export TMPDIR=/tmp
export MNET_GIT_COMMIT="${2:-}"
timeout 2400 python -m pytest tests -m gpu -q --tb=short > "$O/tests_all.log" 2>&1; tail -4 "$O/tests_all.log"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke.log" 2>&1; tail -3 "$O/smoke.log"
bash tools/round_profiles.sh "${1:-r6z}" > "$O/round_profiles.log" 2>&1; tail -3 "$O/round_profiles.log"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > "$O/bench_driver_command.json" 2> "$O/bench_driver_command.err"; python -c "
import json
d=json.loads([l for l in open('$O/bench_driver_command.json') if l.startswith('{')][-1]); r=d['roofline']
print('driver command:', d['value'], 'img/s', d['ms_per_step'], 'ms |', r['kernel'], r['achieved'], 'TFLOP/s frac', r['frac'], '| tail', r['hbm_tail_ms_per_step'], '| all-levels', d.get('value_all_levels_in_mode_precision'), 'no-image', d.get('value_without_prior_image'), '| traffic', r.get('traffic'))"
