"""TextViT on the HIP kernels — same module tree / state_dict keys / ``forward(img) -> (out_cls, out_locs_16,
out_w)`` as the reference's models/textvit_arch.py.  Everything here runs in fp32 (0.2 % of the path's FLOPs):
GEMMs on the fp32 MFMA path of mnet_conv2d_nhwc, LayerNorm / token-mix / attention in dedicated kernels.

``nn.Linear`` / ``nn.LayerNorm`` children are parameter holders only (never called).
"""
import torch
import torch.nn as nn

from . import ops
from .packing import PackCache, pack_vec, posemb_sincos_1x64

EMBED_KSPLIT = 32      # K-slices of the patch embedding at <= 512 tokens (fixed: the fp32 fold order must not depend on the batch)


class FeedForward(nn.Module):
    def __init__(self, dim, hidden_dim):
        super().__init__()
        self.net = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, hidden_dim), nn.GELU(), nn.Linear(hidden_dim, dim))


class Attention(nn.Module):
    def __init__(self, dim, heads=8, dim_head=64):
        super().__init__()
        inner = dim_head * heads
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.norm = nn.LayerNorm(dim)
        self.to_qkv = nn.Linear(dim, inner * 3, bias=False)
        self.to_out = nn.Linear(inner, dim, bias=False)


class Transformer(nn.Module):
    """2 shared blocks + cls / locs / w branch blocks (textvit_arch.py:115-164)."""

    def __init__(self, dim, depth, heads, dim_head, mlp_dim):
        super().__init__()
        mk = lambda hid: nn.ModuleList([Attention(dim, heads=heads, dim_head=dim_head), FeedForward(dim, hid)])
        self.layers = nn.ModuleList([mk(mlp_dim) for _ in range(depth - 1)])
        self.layers_cls = nn.ModuleList([mk(mlp_dim)])
        self.layers_locs = nn.ModuleList([mk(mlp_dim // 2)])
        self.layers_w = nn.ModuleList([mk(mlp_dim // 2)])
        self.linear_seq_maxlen = nn.Sequential(nn.LayerNorm(64), nn.Linear(64, 16))


class TextViT(nn.Module):
    def __init__(self, num_classes, dim, max_length=16):
        super().__init__()
        # the reference pins these regardless of its arguments (textvit_arch.py:15-22)
        depth, heads, mlp_dim, channels, dim_head, patch, max_length = 3, 8, 1024, 512, 64, 8, 16
        self.max_length = max_length
        self.num_classes = num_classes
        self.dim = dim
        patch_dim = channels * patch * patch
        # index 0 of the reference Sequential is the parameter-free einops Rearrange
        self.to_patch_embedding = nn.Sequential(nn.Identity(), nn.Linear(patch_dim, dim))
        self.transformer = Transformer(dim, depth, heads, dim_head, mlp_dim)
        self.to_latent = nn.Identity()
        self.linear_cls = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, num_classes))
        seq = int(32 * max_length / 8)
        self.linear_locs = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, dim // 2), nn.GELU(),
                                         nn.Linear(dim // 2, 2), nn.Sigmoid())
        self.linear_w = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, 512))
        self.linear_w_maxlen = nn.Sequential(nn.LayerNorm(seq), nn.Linear(seq, 1))
        self._cache = PackCache()

    # ------------------------------------------------------------------ packed (all fp32)
    def _build(self, _dtype):
        f = lambda t: t.detach().float().contiguous()
        pk = {"pe": posemb_sincos_1x64(self.linear_cls[1].weight.device)}
        pk["embed.w"] = f(self.to_patch_embedding[1].weight)
        pk["embed.b"] = f(self.to_patch_embedding[1].bias)
        T = self.transformer
        for name, ml in (("layers.0", T.layers[0]), ("layers.1", T.layers[1]), ("cls", T.layers_cls[0]),
                         ("locs", T.layers_locs[0]), ("w", T.layers_w[0])):
            at, ff = ml[0], ml[1]
            pk[name] = dict(ln1=(f(at.norm.weight), f(at.norm.bias)), qkv=f(at.to_qkv.weight), out=f(at.to_out.weight),
                            ln2=(f(ff.net[0].weight), f(ff.net[0].bias)), w1=f(ff.net[1].weight), b1=f(ff.net[1].bias),
                            w2=f(ff.net[3].weight), b2=f(ff.net[3].bias), hidden=ff.net[1].out_features)
        sm = T.linear_seq_maxlen
        pk["seq"] = (f(sm[0].weight), f(sm[0].bias), f(sm[1].weight), f(sm[1].bias))
        wm = self.linear_w_maxlen
        pk["wmax"] = (f(wm[0].weight), f(wm[0].bias), f(wm[1].weight), f(wm[1].bias))
        pk["cls"]["head"] = (f(self.linear_cls[0].weight), f(self.linear_cls[0].bias), f(self.linear_cls[1].weight),
                            f(self.linear_cls[1].bias))
        pk["whead"] = (f(self.linear_w[0].weight), f(self.linear_w[0].bias), f(self.linear_w[1].weight), f(self.linear_w[1].bias))
        ll = self.linear_locs
        w3 = torch.zeros((4, ll[3].in_features), dtype=torch.float32, device=ll[3].weight.device)
        w3[:2] = ll[3].weight.detach()
        pk["lhead"] = (f(ll[0].weight), f(ll[0].bias), f(ll[1].weight), f(ll[1].bias), w3, pack_vec(ll[3].bias, 4))
        return pk

    # ------------------------------------------------------------------ pieces
    @staticmethod
    def _encoder_block(blk, x, B, N):
        """pre-LN MHSA + pre-LN MLP with residuals (textvit_arch.py:104-112, 84-91, 147-162). x [B*N,512]."""
        h = ops.layernorm(x, *blk["ln1"])
        qkv = ops.linear(h, blk["qkv"], 1536)
        a = ops.attention(qkv, B, N, 8, 0.125)
        x = ops.linear(a, blk["out"], 512, residual=x)
        h = ops.layernorm(x, *blk["ln2"])
        h = ops.linear(h, blk["w1"], blk["hidden"], bias=blk["b1"], act=ops.ACT_GELU)
        return ops.linear(h, blk["w2"], 512, bias=blk["b2"], residual=x)

    def forward_nhwc(self, feat):
        """feat: NHWC fp32 [B,8,512,512] (ResNet output) → (logits [B,64,C], locs [B,32], w [B,512])."""
        pk = self._cache.get(self, "fp32", self._build)
        B = feat.shape[0]
        # patchify + Linear == 8x8 / stride-8 conv in NHWC ('(p1 p2 c)' is exactly the NHWC window order), + bias + pos-emb
        # up to 8 strips (512 tokens) the weight stream (64 MiB) is the whole cost: split-K over 32 slices spreads it over the chip
        x = ops.conv2d(feat, pk["embed.w"], 512, 8, 8, (8, 8), (0, 0), bias=pk["embed.b"],
                       residual=pk["pe"].reshape(1, 1, 64, 512), res_mod=64,
                       splitk=EMBED_KSPLIT if B * 64 <= 512 else 0).reshape(B * 64, 512)
        x = self._encoder_block(pk["layers.0"], x, B, 64)
        x = self._encoder_block(pk["layers.1"], x, B, 64)
        x_cls = self._encoder_block(pk["cls"], x, B, 64)
        x16 = ops.token_mix(x.reshape(B, 64, 512), *pk["seq"]).reshape(B * 16, 512)     # LN(64)+Linear(64,16) over tokens
        x_loc = self._encoder_block(pk["locs"], x16, B, 16)
        x_w = self._encoder_block(pk["w"], x, B, 64)
        g, b, w, bb = pk["cls"]["head"]
        logits = ops.linear(ops.layernorm(x_cls, g, b), w, self.num_classes, bias=bb).reshape(B, 64, self.num_classes)
        xw = ops.token_mix(x_w.reshape(B, 64, 512), *pk["wmax"]).reshape(B, 512)          # LN(64)+Linear(64,1)
        g, b, w, bb = pk["whead"]
        out_w = ops.linear(ops.layernorm(xw, g, b), w, 512, bias=bb)
        g, b, w1, b1, w3, b3 = pk["lhead"]
        h = ops.linear(ops.layernorm(x_loc, g, b), w1, 256, bias=b1, act=ops.ACT_GELU)
        locs = ops.linear(h, w3, 4, bias=b3, act=ops.ACT_SIGMOID)                         # cout padded 2 → 4
        locs = locs[:, :2].reshape(B, 32)
        return logits, locs, out_w

    def forward(self, img):
        """img: NCHW fp32 [B,512,8,512] like the reference (textvit_arch.py:65-77)."""
        with torch.no_grad(), ops.on_device(img):
            feat = ops.nchw_to_nhwc(img.contiguous().float(), torch.float32)
            return self.forward_nhwc(feat)
