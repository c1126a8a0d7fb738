"""Provider for the one operator boundary the reference already has:
``from basicsr.ops.fused_act import FusedLeakyReLU, fused_leaky_relu`` (models/networks.py:10).

Upstream basicsr ships this op as a CUDA-only extension (``fused_act_ext``; README.md:61-72 documents the
``NameError`` users hit without it).  Here it is served by the gfx950 kernel ``mnet_fused_bias_act``.
``install()`` registers this module as ``basicsr.ops.fused_act`` so the reference's import line works unmodified.
(On the HIP modules' own hot path the same math is the conv epilogue ``MNET_ACT_LRELU_SQRT2``.)
"""
import sys
import types

import torch
import torch.nn as nn

from . import ops


def fused_leaky_relu(input, bias, negative_slope=0.2, scale=2 ** 0.5):
    """out = scale * leaky_relu(input + bias.view(1, C, 1, ...), negative_slope); inference only (no autograd)."""
    with torch.no_grad(), ops.on_device(input):
        x = input.contiguous().float()
        b = None if bias is None else bias.detach().contiguous().float()
        return ops.fused_bias_act(x, b, negative_slope, scale)


class FusedLeakyReLU(nn.Module):
    def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel))
        self.negative_slope = negative_slope
        self.scale = scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)


def install():
    """make ``import basicsr.ops.fused_act`` resolve to this provider (no-op if a real basicsr is importable)."""
    if "basicsr.ops.fused_act" in sys.modules:
        return
    pkg = sys.modules.get("basicsr") or types.ModuleType("basicsr")
    opsm = sys.modules.get("basicsr.ops") or types.ModuleType("basicsr.ops")
    fa = types.ModuleType("basicsr.ops.fused_act")
    fa.fused_leaky_relu, fa.FusedLeakyReLU = fused_leaky_relu, FusedLeakyReLU
    pkg.ops, opsm.fused_act = opsm, fa
    sys.modules.update({"basicsr": pkg, "basicsr.ops": opsm, "basicsr.ops.fused_act": fa})
