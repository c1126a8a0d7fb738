import sys, time, torch
sys.path.insert(0, ".")
from marconet_amd import networks, synthetic
from marconet_amd.pipeline import MarconetPipeline
dev = torch.device("cuda:0")
enc, gan, sr = networks.TextContextEncoderV2(), networks.TSPGAN(), networks.TSPSRNet()
enc.load_state_dict(synthetic.make_encoder_state_dict()); gan.load_state_dict(synthetic.make_gan_state_dict()); sr.load_state_dict(synthetic.make_sr_state_dict())
pipe = MarconetPipeline(enc.eval().to(dev), gan.eval().to(dev), sr.eval().to(dev), precision="fp16")
for B in (1, 4, 64):
    lq = synthetic.make_lq(5, B, [512] * B).to(dev)
    labels = [synthetic.make_labels(6 + b, 16) for b in range(B)]
    locs = synthetic.make_locs([16] * B, [512] * B)
    for _ in range(2):
        pipe.forward_batch(lq, labels, locs)
    torch.cuda.synchronize()
    ts, tt = [], []
    for _ in range(5):
        t0 = time.perf_counter()
        pipe.forward_batch(lq, labels, locs)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ts.append(t1 - t0); tt.append(t2 - t0)
    print("B=%d host enqueue %.1f ms, total %.1f ms" % (B, 1e3 * min(ts), 1e3 * min(tt)))
