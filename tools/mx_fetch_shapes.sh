#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per launch of the split-half (id 11) and fp16+8 (id 11) 256x256 tiles on glyph-shaped layers
O=$PWD/gpurun_out/r3m; mkdir -p $O; export TMPDIR=/tmp; R=$PWD
run() {  # <tag> <tile_power_ab args...>
  local tag=$1; shift
  for grp in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
    t=$(echo $grp | cut -d" " -f1)
    ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/${tag}_$t -o pmc -- python $R/tools/tile_power_ab.py --launches 4 --only x3:11,x2:11 "$@" ) > $O/${tag}_$t.log 2>&1
    echo "== $tag $t"; python tools/pmc_summary.py $O/${tag}_$t /dev/null conv_dma | grep -v "^$"
  done
  rm -rf $O/${tag}_*/
}
run glyph64_512to256 --shape 1024,64,64,512,256 --ragged
run glyph64_256to256 --shape 1024,64,64,256,256 --ragged
run gan32_512 --shape 1024,32,32,512,512
run trunk_cat --shape 64,32,512,256,256
