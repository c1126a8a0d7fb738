#!/bin/bash
O=$PWD/gpurun_out/r4v; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_mx_gpu.py tests/test_modules_gpu.py -q --tb=short -x 2>&1 | tail -3 | tee $O/tests.log
timeout 400 python bench.py --steps 3 --warmup 1 --cpu-images 0 --no-secondary > $O/bench.json 2> $O/bench.err
timeout 400 python tools/profile_layers.py --batch 256 --glyphs 16 --precision fp16x2 --out $O/conv_layers.txt > /dev/null 2>&1
grep "128x2048\|64x1024 c=256+0   ->  128\|32x512  c= 32\|32x512  c= 64" $O/conv_layers.txt | head -8
python - <<'PY'
import json
for f in ('bench',):
    d=json.loads(open('gpurun_out/r4v/%s.json'%f).read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['kernel_ms_per_step'], d['roofline']['all_conv_kernels']['by_kernel_ms_per_step'])
PY
