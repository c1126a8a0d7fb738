#!/bin/bash
O=$PWD/gpurun_out/r4z4; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_mx_gpu.py tests/test_kernels_gpu.py -q --tb=short -x 2>&1 | tail -3 | tee $O/tests.log
V=tools/_build/pre_setup/libmarconet_hip.so
run() { MARCONET_HIP_LIB=$1 timeout 120 python tools/tile_power_ab.py --seconds 3 --only x2:15 "${@:2}" 2>&1 | grep "TFLOP/s" | sed "s|^|$(basename $(dirname $1)) |"; }
{
for rep in 1 2 3; do run marconet_amd/lib/libmarconet_hip.so; run $V; done
} 2>&1 | tee $O/rates_ab.txt
for i in 1 2; do
timeout 400 python bench.py --steps 3 --warmup 1 --cpu-images 0 --no-secondary > $O/bench_new_$i.json 2> $O/err
MARCONET_HIP_LIB=$V timeout 400 python bench.py --steps 3 --warmup 1 --cpu-images 0 --no-secondary > $O/bench_old_$i.json 2> $O/err
done
python - <<'PY'
import json
for f in ('bench_new_1','bench_old_1','bench_new_2','bench_old_2'):
    d=json.loads(open('gpurun_out/r4z4/%s.json'%f).read().strip().splitlines()[-1]); r=d['roofline']
    print(f, d['value'], d['ms_per_step'], r['achieved'], r['kernel_ms_per_step'], r['hbm_tail']['ms_per_step'])
PY
