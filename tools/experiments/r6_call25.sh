#!/bin/bash
# the up-sample kernel walking runs of input rows (3 chunk loads per input pixel instead of 9): tests, kernel-level rates for run lengths 2 / 4 / 8 / 16 against the old form, bench A/B
set -uo pipefail
O="$PWD/gpurun_out/${1:-r6ah}"; mkdir -p "$O"; export TMPDIR=/tmp
B="$PWD/tools/_build"
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_mx_gpu.py tests/test_split_gpu.py -m gpu -q --tb=short -k "upsample or up_sample or bilinear" > "$O/tests_ups.log" 2>&1; tail -3 "$O/tests_ups.log"
{ for v in old run2 prod run8 run16; do lib="$B/aux_$v/libmarconet_hip.so"; [ $v = prod ] && lib="$PWD/marconet_amd/lib/libmarconet_hip.so"
    echo "== $v"; MARCONET_HIP_LIB=$lib python tools/experiments/tail_vs_torch_stream.py 2>&1 | grep "up-sample" | sed "s/| torch.*//"; done; } | tee "$O/upsample_runs.txt"
for v in old prod old prod; do
  lib="$B/aux_$v/libmarconet_hip.so"; [ $v = prod ] && lib="$PWD/marconet_amd/lib/libmarconet_hip.so"
  MARCONET_HIP_LIB=$lib timeout 600 python bench.py --steps 5 --warmup 2 --cpu-images 0 --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$v', d['value'], 'img/s', d['ms_per_step'], 'ms |', r['achieved'], 'TFLOP/s | tail', r['hbm_tail_ms_per_step'])"
done | tee "$O/bench_upsample_ab.txt"
