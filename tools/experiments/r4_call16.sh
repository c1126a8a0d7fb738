#!/bin/bash
O=$PWD/gpurun_out/r4q2; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_mx_gpu.py tests/test_split_gpu.py tests/test_modules_gpu.py -q --tb=short -x 2>&1 | tail -6 | tee $O/tests.log
for p in fp16x2 fp16; do timeout 200 python tools/graph_latency.py 16 50 $p; MNET_NO_FUSE_CONV1_MOD=1 timeout 200 python tools/graph_latency.py 16 50 $p; done 2>&1 | grep batch | tee $O/graph_latency.txt
timeout 400 python bench.py --steps 3 --warmup 1 --cpu-images 0 --no-secondary > $O/bench.json 2> $O/bench.err
MNET_NO_FUSE_CONV1_MOD=1 timeout 400 python bench.py --steps 3 --warmup 1 --cpu-images 0 --no-secondary > $O/bench_nofuse.json 2> $O/bench_nofuse.err
python - <<'PY'
import json
for f in ('bench','bench_nofuse'):
    d=json.loads(open('gpurun_out/r4q2/%s.json'%f).read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], {k:v for k,v in d.items() if 'max_abs' in k})
PY
