#!/usr/bin/env python
"""Where does the fp16 throughput mode's deviation come from?  Runs the pipeline with each of the three nets (and parts of
them) switched between fp16 and fp32 and reports the SR max-abs / mean-abs deviation from the CPU oracle (test
infrastructure) on a small batch.  python tools/precision_probe.py"""
import itertools
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from marconet_amd import networks, synthetic
    from marconet_amd.pipeline import MarconetPipeline
    from oracle import marconet_oracle as O
    dev = torch.device("cuda:0")
    sde, sdg, sds = synthetic.make_encoder_state_dict(), synthetic.make_gan_state_dict(), synthetic.make_sr_state_dict()
    enc, gan, sr = networks.TextContextEncoderV2(), networks.TSPGAN(), networks.TSPSRNet()
    enc.load_state_dict(sde); gan.load_state_dict(sdg); sr.load_state_dict(sds)
    pipe = MarconetPipeline(enc.eval().to(dev), gan.eval().to(dev), sr.eval().to(dev), precision="fp16")
    B, n = 2, 16
    lq = synthetic.make_lq(1234, B, [512] * B)
    labels = [synthetic.make_labels(1234 + b, n) for b in range(B)]
    locs = synthetic.make_locs([n] * B, [512] * B)
    ref = O.end_to_end(sde, sdg, sds, lq, labels, locs)["sr"]
    lqd, labd, locd = lq.to(dev), [l.to(dev) for l in labels], locs.to(dev)
    print("%-8s %-8s %-8s   max-abs     mean-abs" % ("encoder", "gan", "sr"))
    for pe, pg, ps in itertools.product(("fp16", "fp32"), repeat=3):
        pipe.encoder.set_precision(pe); pipe.gan.set_precision(pg); pipe.sr.set_precision(ps)
        pipe.precision = pg          # forward_batch passes this to the generator
        y = pipe.forward_batch(lqd, labd, locd).cpu()
        d = (y - ref).abs()
        print("%-8s %-8s %-8s   %.3e   %.3e" % (pe, pg, ps, d.max().item(), d.mean().item()))


if __name__ == "__main__":
    main()
