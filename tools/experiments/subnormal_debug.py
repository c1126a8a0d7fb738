import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from marconet_amd import mxfmt, ops, packing
cout, cin, n, h, w = 256, 64, 1, 8, 32
g = torch.Generator().manual_seed(77)
exps = (-24, -23, -22, -20, -18, -16, -15, -14.2)
mags = torch.tensor([2.0 ** e for e in exps])
bias = ((torch.rand((8, 32), generator=g) * 2 - 1) * mags[:, None]).reshape(-1).float()
bias[5] = 0.0
want = bias[None, None, None, :].expand(n, h, w, cout).contiguous()
host = mxfmt.pack_act(want).reshape(n, h, w, 8, 128)
stream = ops.convert(want.cuda(), packing.MX_DTYPE).cpu().view(torch.uint8).reshape(n, h, w, 8, 128)
xd = ops.convert(torch.zeros((n, h, w, cin)).cuda(), packing.MX_DTYPE)
wp = packing.pack_conv_weight(torch.zeros((cout, cin, 3, 3)), packing.MX_DTYPE).cuda()
y = ops.conv2d(xd, wp, cout, 3, 3, (1, 1), (1, 1), bias=bias.cuda()).cpu().view(torch.uint8).reshape(n, h, w, 8, 128)
for name, t in (("streaming", stream), ("conv epilogue", y)):
    for b in range(8):
        d = (t[0, 0, 0, b] != host[0, 0, 0, b])
        if d.any():
            idx = d.nonzero().flatten().tolist()
            print("%s block %d (max |v| ~ 2^%s): %d bytes differ at %s" % (name, b, exps[b], len(idx), idx[:40]))
            print("    host  :", [int(host[0, 0, 0, b, i]) for i in idx[:24]], "scale byte", int(host[0, 0, 0, b, 96]))
            print("    device:", [int(t[0, 0, 0, b, i]) for i in idx[:24]], "scale byte", int(t[0, 0, 0, b, 96]))
    print(name, "equal over all pixels:", bool(torch.equal(t, host)))
print("stream == conv:", bool(torch.equal(stream, y)))
