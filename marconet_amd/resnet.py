"""ResNet-45 LR backbone (no BN, no bias) on the HIP implicit-GEMM kernel — mirrors the module tree and
state_dict keys of the reference's models/resnet.py (``ResNet``, ``BasicBlock``, ``resnet45stride``).

The ``nn.Conv2d`` children are *parameter holders* (same names / shapes / default init as the reference so
``load_state_dict(strict=True)`` round-trips); their own ``forward`` is never called.
"""
import math

import torch
import torch.nn as nn

from . import ops
from .packing import PackCache, default_precision, pack_conv_weight, rgb_pad, torch_dtype


def _conv(cin, cout, k, stride=1):
    return nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=k // 2, bias=False)


class BasicBlock(nn.Module):
    """1x1 → ReLU → 3x3(stride) → + residual (1x1 strided projection on the first block of a stage) → ReLU
    (models/resnet.py:11-30)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv(inplanes, planes, 1)
        self.conv2 = _conv(planes, planes, 3, stride)
        self.downsample = downsample
        self.stride = stride if isinstance(stride, tuple) else (stride, stride)


class ResNet(nn.Module):
    def __init__(self, block, layers, strides=(2, 1, 2, 1, 1)):
        super().__init__()
        self.inplanes = 32
        self.conv1 = _conv(3, 32, 3)
        for i, (planes, nblk, st) in enumerate(zip((32, 64, 128, 256, 512), layers, strides), 1):
            setattr(self, "layer%d" % i, self._make_layer(block, planes, nblk, st))
        for m in self.modules():                         # same init law as models/resnet.py:45-48
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / n))
        self.precision = default_precision()
        self._cache = PackCache()

    def _make_layer(self, block, planes, blocks, stride):
        downsample = None
        if stride != 1 or self.inplanes != planes:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes, kernel_size=1, stride=stride, bias=False))
        seq = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes
        seq += [block(planes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*seq)

    # ------------------------------------------------------------------ packed weights
    def _build(self, dtype):
        pk = {"conv1": pack_conv_weight(self.conv1.weight.detach(), dtype)}
        for li in range(1, 6):
            for bi, blk in enumerate(getattr(self, "layer%d" % li)):
                p = "layer%d.%d." % (li, bi)
                pk[p + "conv1"] = pack_conv_weight(blk.conv1.weight.detach(), dtype)
                pk[p + "conv2"] = pack_conv_weight(blk.conv2.weight.detach(), dtype)
                if blk.downsample is not None:
                    pk[p + "down"] = pack_conv_weight(blk.downsample[0].weight.detach(), dtype)
        return pk

    # ------------------------------------------------------------------ forward
    def forward_nhwc(self, x):
        """x: NHWC [B,32,512,8] (RGB zero-padded to 8 channels) in the compute dtype → NHWC [B,8,512,512]."""
        pk = self._cache.get(self, self.precision, self._build)
        x = ops.conv2d(x, pk["conv1"], 32, 3, 3, (1, 1), (1, 1), act=ops.ACT_RELU)
        for li in range(1, 6):
            for bi, blk in enumerate(getattr(self, "layer%d" % li)):
                p = "layer%d.%d." % (li, bi)
                planes = blk.conv1.out_channels
                y = ops.conv2d(x, pk[p + "conv1"], planes, act=ops.ACT_RELU)
                res = x
                if blk.downsample is not None:
                    res = ops.conv2d(x, pk[p + "down"], planes, 1, 1, blk.stride, (0, 0))
                x = ops.conv2d(y, pk[p + "conv2"], planes, 3, 3, blk.stride, (1, 1), residual=res, act=ops.ACT_RELU)
        return x

    def forward(self, x):
        """NCHW fp32 in → NCHW fp32 out, like the reference (models/resnet.py:63-71)."""
        with torch.no_grad(), ops.on_device(x):
            dtype = torch_dtype(self.precision)
            h = ops.nchw_to_nhwc(x.contiguous().float(), dtype, c_ld=rgb_pad(dtype))
            return ops.nhwc_to_nchw(self.forward_nhwc(h))


def resnet45stride():
    return ResNet(BasicBlock, [3, 4, 6, 6, 3], [(2, 1), 1, (2, 1), 1, 1])
