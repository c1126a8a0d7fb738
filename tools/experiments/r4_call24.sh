#!/bin/bash
# one strip at a time (batch 1, 16 glyphs, fp16x2): kernel trace of 23 eager forwards
O=$PWD/gpurun_out/r4o; mkdir -p $O; export TMPDIR=/tmp; R=$PWD
cat > /tmp/b1.py <<PY
import sys, torch
sys.path.insert(0, "$R")
from marconet_amd import networks, synthetic
from marconet_amd.pipeline import MarconetPipeline
nets = [networks.TextContextEncoderV2(), networks.TSPGAN(), networks.TSPSRNet()]
for m, sd in zip(nets, (synthetic.make_encoder_state_dict(), synthetic.make_gan_state_dict(), synthetic.make_sr_state_dict())):
    m.load_state_dict(sd, strict=True)
nets = [m.eval().cuda() for m in nets]
pipe = MarconetPipeline(*nets, precision="fp16x2", check_finite=False)
lq = synthetic.make_lq(1234, 1, [512]).cuda(); labels = [synthetic.make_labels(1234, 16).cuda()]; locs = synthetic.make_locs([16], [512]).cuda()
for _ in range(23): pipe.forward_batch(lq, labels, locs)
torch.cuda.synchronize()
PY
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b1 -o run -- python /tmp/b1.py ) > $O/prof_b1.log 2>&1
f=$(find $O/prof_b1 -name "*kernel_stats.csv" | head -1)
python tools/rocprof_summary.py "$f" $O/b1_n16_fp16x2_kernel_stats.txt "rocprofv3 --kernel-trace --stats: 23 eager forwards of one strip (batch 1, 16 glyphs), fp16x2" > /dev/null
rm -rf $O/prof_b1
head -45 $O/b1_n16_fp16x2_kernel_stats.txt | cut -c1-165
