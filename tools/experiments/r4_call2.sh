#!/bin/bash
# round 4, GPU call 2: new tests, strip 256x256 fp16+8 tile A/B, zero-operand control, bench with the f16 image level, per-XCD pass barrier
O=$PWD/gpurun_out/r4b; mkdir -p $O; export TMPDIR=/tmp; R=$PWD
timeout 900 python -m pytest tests/test_mx_gpu.py tests/test_modules_gpu.py tests/test_stress_gpu.py -q --tb=short -m gpu \
   -k "strip_256x256 or generator_chunks or prior_image_precision or outlier_channel or without_prior_image" 2>&1 | tail -40 | tee $O/tests.log
{
  timeout 150 python tools/tile_power_ab.py --seconds 3 --only x2:11,x2:s0,x2:11,x2:s0
  timeout 150 python tools/tile_power_ab.py --seconds 3 --only x2:11,x2:s0 --shape 1024,64,64,512,256 --ragged
  timeout 150 python tools/tile_power_ab.py --seconds 3 --only x2:11,x2:s0 --shape 1024,32,32,512,512
  echo "-- zero-filled operands"
  timeout 150 python tools/tile_power_ab.py --seconds 3 --only x2:11,x3:11,f16:16 --zeros
  echo "-- random operands"
  timeout 150 python tools/tile_power_ab.py --seconds 3 --only x2:11,x3:11,f16:16
} 2>&1 | grep "TFLOP/s\|rror\|--" | tee $O/rates.txt
timeout 500 python bench.py --steps 3 --warmup 1 > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-600
MNET_MX_STRIP256=1 timeout 400 python bench.py --steps 3 --warmup 1 --no-secondary --cpu-images 0 > $O/bench_strip256.log 2>&1; tail -1 $O/bench_strip256.log | cut -c1-300
# per-XCD pass barrier (variant build): FETCH_SIZE and rate, production arm and MNET_XCD_SYNC=1
export MARCONET_HIP_LIB=$R/tools/_build/xcd_sync/libmarconet_hip.so
for arm in base sync; do
  case $arm in base) E="X=1";; sync) E="MNET_XCD_SYNC=1";; esac
  for s in "glyph64_512to256:--shape 1024,64,64,512,256 --ragged" "trunk:"; do
    tag=${s%%:*}; args=${s#*:}
    ( cd /tmp && env $E timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/xcd_${arm}_${tag} -o pmc -- \
        python $R/tools/tile_power_ab.py --launches 4 --only x2:11 $args ) > $O/xcd_${arm}_${tag}.log 2>&1
    echo "== $arm $tag FETCH_SIZE (KB raw per launch; x2 for bytes at the fabric)"; python tools/pmc_summary.py $O/xcd_${arm}_${tag} /dev/null conv_dma | grep -v "^$"
    rm -rf $O/xcd_${arm}_${tag}/
    env $E timeout 150 python tools/tile_power_ab.py --seconds 3 --only x2:11 $args 2>&1 | grep "TFLOP/s" | sed "s/^/   $arm $tag  /"
  done
done 2>&1 | tee $O/xcd_summary.txt
