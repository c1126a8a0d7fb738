O=$PWD/gpurun_out/r3h; mkdir -p $O; export TMPDIR=/tmp; R=$PWD
for pad in 0 1; do
  for grp in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $grp | cut -d" " -f1)
    ( cd /tmp && MNET_MX_FETCH_PAD=$pad timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pad${pad}_$tag -o pmc -- python $R/tools/tile_power_ab.py --launches 6 --only x2:11,x3:11 ) > $O/pad${pad}_$tag.log 2>&1
  done
  python tools/pmc_summary.py $O "$O/pad${pad}.txt.tmp" conv_dma > /dev/null
  MNET_MX_FETCH_PAD=$pad timeout 200 python tools/tile_power_ab.py --seconds 5 --only x2:11 > $O/rate_pad${pad}.txt 2>&1
done
for pad in 0 1; do for t in FETCH_SIZE WRITE_SIZE; do echo "== pad $pad $t"; python tools/pmc_summary.py $O/pad${pad}_$t /dev/null conv_dma | grep -A4 "conv_dma"; done; cat $O/rate_pad${pad}.txt | tail -1; done
rm -rf $O/pad*_*/
