#!/bin/bash
# A/B builds of the streaming (HBM-bound) kernels: non-temporal stores / loads, one trip per thread in GroupNorm apply — kernel level, then the bench step
set -uo pipefail
O="$PWD/gpurun_out/${1:-r6af}"; mkdir -p "$O"; export TMPDIR=/tmp
B="$PWD/tools/_build"
{ echo "== prod"; python tools/experiments/tail_vs_torch_stream.py 2>&1 | grep -v amdgpu.ids | sed "s/| torch.*//"
  for v in ntst ntld ntboth trip1 all3; do echo "== $v"; MARCONET_HIP_LIB=$B/aux_$v/libmarconet_hip.so python tools/experiments/tail_vs_torch_stream.py 2>&1 | grep -v amdgpu.ids | sed "s/| torch.*//"; done
} | tee "$O/tail_kernels_nt_ab.txt"
for v in prod ntst ntld ntboth all3 prod; do
  lib="$B/aux_$v/libmarconet_hip.so"; [ $v = prod ] && lib="$PWD/marconet_amd/lib/libmarconet_hip.so"
  MARCONET_HIP_LIB=$lib timeout 600 python bench.py --steps 4 --warmup 2 --cpu-images 0 --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$v', d['value'], 'img/s', d['ms_per_step'], 'ms |', r['achieved'], 'TFLOP/s | tail', r['hbm_tail_ms_per_step'], r.get('hbm_tail_by_kernel_ms') or '')"
done | tee "$O/bench_nt_ab.txt"
