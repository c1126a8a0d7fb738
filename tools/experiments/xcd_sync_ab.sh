#!/bin/bash
# Next-round experiment: does a per-XCD pass barrier remove the fp16+8 tile's L2 misses on 512-channel inputs, and what does it do
# to the rate?  Run from the repo root on the GPU box AFTER `git apply tools/experiments/xcd_sync.patch && bash marconet_amd/csrc/build.sh`:
#     bash tools/experiments/xcd_sync_ab.sh            (about 3 GPU-minutes)
# Arms: production | MNET_XCD_SYNC=1 | MNET_DBG_WEIGHT_WINDOW=1 (diagnostic, wrong results: weights from a 1 MiB window, always L2 hits)
O=$PWD/gpurun_out/xcd_sync; mkdir -p $O; export TMPDIR=/tmp; R=$PWD
export MNET_ALLOW_DIAGNOSTIC_KERNELS=1
SHAPES=("glyph64_512to256:--shape 1024,64,64,512,256 --ragged" "glyph64_256to256:--shape 1024,64,64,256,256 --ragged" "trunk:")
for arm in base sync wwin; do
  case $arm in base) E="";; sync) E="MNET_XCD_SYNC=1";; wwin) E="MNET_DBG_WEIGHT_WINDOW=1";; esac
  for s in "${SHAPES[@]}"; do
    tag=${s%%:*}; args=${s#*:}
    ( cd /tmp && env $E timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${arm}_${tag} -o pmc -- \
        python $R/tools/tile_power_ab.py --launches 4 --only x3:11,x2:11 $args ) > $O/${arm}_${tag}.log 2>&1
    echo "== $arm $tag FETCH_SIZE (KB raw per launch; x2 for bytes at the fabric)"; python tools/pmc_summary.py $O/${arm}_${tag} /dev/null conv_dma | grep -v "^$"
    rm -rf $O/${arm}_${tag}/
    env $E timeout 200 python tools/tile_power_ab.py --seconds 4 --only x3:11,x2:11 $args 2>&1 | grep "TFLOP/s" | sed "s/^/   $arm $tag  /"
  done
done 2>&1 | tee $O/summary.txt
