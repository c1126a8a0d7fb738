#!/bin/bash
# one box: the production build against the build of before the epilogue work (tools/_build/no_xpose: direct epilogue stores, residual loads one by one,
# strip tile without the transposition), alternating; then the full default bench line for the record
O=$PWD/gpurun_out/r4x; mkdir -p $O; export TMPDIR=/tmp
V=tools/_build/no_xpose/libmarconet_hip.so
for i in 1 2; do
  timeout 400 python bench.py --steps 3 --warmup 1 --cpu-images 0 --no-secondary > $O/bench_new_$i.json 2> $O/err
  MARCONET_HIP_LIB=$V timeout 400 python bench.py --steps 3 --warmup 1 --cpu-images 0 --no-secondary > $O/bench_old_$i.json 2> $O/err
done
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
for f in ('bench_new_1','bench_old_1','bench_new_2','bench_old_2','bench_default'):
    d=json.loads(open('gpurun_out/r4x/%s.json'%f).read().strip().splitlines()[-1])
    r=d['roofline']
    print(f, d['value'], d['ms_per_step'], r['achieved'], r['kernel_ms_per_step'], r['all_conv_kernels']['ms_per_step_all_dtypes'], r['hbm_tail']['ms_per_step'])
PY
