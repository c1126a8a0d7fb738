#!/bin/bash
O=$PWD/gpurun_out/r4n; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -q --tb=short -x -k "affine_act or upsample" 2>&1 | tail -5 | tee $O/tests.log
timeout 400 python bench.py --steps 2 --warmup 1 --cpu-images 0 --no-secondary > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4n/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
t=d['roofline']['hbm_tail']
print(t['ms_per_step'], t['GB_per_s'])
for k,v in t['by_kernel'].items(): print(k, v)
PY
