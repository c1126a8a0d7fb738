"""debug: fp16+8 id 16 (one wave per SIMD) against id 6 on tiny launches — where do the outputs differ?"""
import math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from marconet_amd import ops, packing, _lib

def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale

def run(n, h, w, cin, cout, k, detail=(), **kw):
    x = ops.convert(rnd((n, h, w, cin), 1).cuda(), packing.MX_DTYPE)
    wp = packing.pack_conv_weight(rnd((cout, cin, k, k), 2, 1.0 / math.sqrt(cin * k * k)), packing.MX_DTYPE).cuda()
    outs = []
    for algo in (_lib.ALGO_DMA_CFG0 + 6, _lib.ALGO_DMA_CFG16 + 0):
        y = ops.conv2d(x, wp, cout, k, k, (1, 1), (k // 2, k // 2), algo=algo, **kw)
        torch.cuda.synchronize()
        outs.append(ops.convert(y, torch.float32).cpu())
        raws = locals().setdefault("raws", [])
        raws.append(packing.untag(y).view(torch.uint8).reshape(-1, 128).cpu())
    a, b = outs
    db = (raws[0] != raws[1])
    if db.any():
        i = db.nonzero()
        print("  RAW bytes differ: %d bytes in %d blocks; by byte offset in the block: %s; first: block %d byte %d ref 0x%02x got 0x%02x" % (
            int(db.sum()), int(db.any(1).sum()), sorted(set(int(v) for v in i[:, 1]))[:16], int(i[0, 0]), int(i[0, 1]), int(raws[0][i[0, 0], i[0, 1]]), int(raws[1][i[0, 0], i[0, 1]])))
        blk = int(i[0, 0])
        print("   ref block:", " ".join("%02x" % int(v) for v in raws[0][blk][:100]))
        print("   got block:", " ".join("%02x" % int(v) for v in raws[1][blk][:100]))
    npix = n * h * w
    a2, b2 = a.reshape(npix, cout), b.reshape(npix, cout)
    bad = ~((a2 == b2) | (torch.isnan(a2) & torch.isnan(b2)))
    nan = torch.isnan(b2)
    print("shape n%d h%d w%d cin%d cout%d k%d %s: mismatching %d of %d, NaN in id16: %d, max|d| %.3e" % (n, h, w, cin, cout, k, sorted(kw), int(bad.sum()), bad.numel(), int(nan.sum()),
          float((a2 - b2).nan_to_num(1e9).abs().max())))
    if bad.any():
        P, C = (npix + 31) // 32, cout // 32
        pad = torch.zeros((P * 32 - npix, cout), dtype=torch.bool)
        grid = torch.cat([bad, pad]).reshape(P, 32, C, 32).sum(dim=(1, 3))
        print("  mismatches per (32-pixel block [rows], 32-channel block [cols]) — first 16 rows:")
        for r in range(min(P, 16)):
            print("   ", " ".join("%4d" % int(v) for v in grid[r]))
        if detail:
            b4 = torch.cat([bad, pad]).reshape(P, 32, C, 32)
            d4 = torch.cat([(a2 - b2).nan_to_num(9.0), torch.zeros((P * 32 - npix, cout))]).reshape(P, 32, C, 32)
            for (r, c) in detail:
                print("  block (pixel block %d, channel block %d): rows = pixel in block, cols = channel in block; '.' equal, 'x' differs, 'N' NaN" % (r, c))
                for px_ in range(32):
                    print("    %2d " % px_ + "".join("." if not b4[r, px_, c, ch] else ("N" if d4[r, px_, c, ch] == 9.0 else "x") for ch in range(32)))
        i = bad.nonzero()[0]
        print("  first mismatch pixel %d channel %d: ref %r got %r" % (int(i[0]), int(i[1]), float(a2[i[0], i[1]]), float(b2[i[0], i[1]])))

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "scales":          # which epilogue term of the SC build differs?
        n, cout = 2, 256
        osc, psc, bias = (rnd((n, cout), 6).abs() + 0.5).cuda(), (rnd((n, cout), 7).abs() + 0.5).cuda(), rnd((cout,), 5).cuda()
        run(n, 16, 32, 64, cout, 3)
        run(n, 16, 32, 64, cout, 3, bias=bias)
        run(n, 16, 32, 64, cout, 3, out_scale=osc)
        run(n, 16, 32, 64, cout, 3, post_scale=psc)
        run(n, 16, 32, 64, cout, 3, out_scale=osc, bias=bias)
        run(n, 16, 32, 64, cout, 3, post_scale=psc, bias=bias)
        run(n, 16, 32, 64, cout, 3, post_scale=psc, act=2)
        run(n, 16, 32, 64, cout, 3, post_scale=psc, act=3)
        run(n, 16, 32, 64, cout, 3, out_scale=osc, post_scale=psc, bias=bias, act=3)
        n = 6
        osc, psc = (rnd((n, cout), 173).abs() + 0.5).cuda(), (rnd((n, cout), 174).abs() + 0.5).cuda()
        run(n, 64, 64, 64, cout, 3, out_scale=osc, post_scale=psc, bias=bias, act=3)
        run(n, 64, 64, 64, cout, 3)
        sys.exit(0)
    run(1, 16, 16, 32, 256, 1, detail=[(1, 3), (2, 0), (2, 1), (3, 0)])
    sys.exit(0)
    run(1, 16, 16, 64, 256, 1)
    run(1, 16, 16, 32, 256, 3)
    run(1, 16, 16, 128, 256, 3)
    run(2, 16, 40, 128, 288, 3)
    bias = rnd((288,), 5).cuda()
    run(2, 16, 40, 128, 288, 3, bias=bias)
    run(2, 16, 40, 128, 288, 3, bias=bias, act=3)
    run(2, 16, 40, 128, 288, 3, out_scale=(rnd((2, 288), 6).abs() + 0.5).cuda())
    run(2, 16, 40, 128, 288, 3, post_scale=(rnd((2, 288), 7).abs() + 0.5).cuda())
    res = ops.convert(rnd((2, 16, 40, 288), 8).cuda(), packing.MX_DTYPE)
    run(2, 16, 40, 128, 288, 3, residual=res)
    run(2, 16, 40, 128, 288, 3, valid_w=torch.tensor([40, 23], dtype=torch.int32).cuda())
    run(4, 128, 160, 64, 256, 3)
