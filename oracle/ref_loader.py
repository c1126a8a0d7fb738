"""TEST INFRASTRUCTURE ONLY — never imported by the product path (marconet_amd/).

Imports the *real* reference modules from /root/reference (read-only, present only in the build
container, absent on the GPU box) so that the CPU restatement in oracle/marconet_oracle.py can be
pinned against them and golden vectors can be generated (tests/golden/make_golden.py).

The reference imports one third-party CUDA-only op, ``basicsr.ops.fused_act``
(/root/reference/models/networks.py:10); basicsr is not installed and un-pinned (README.md:39),
so a pure-torch stub with upstream semantics is registered in ``sys.modules`` first:
    out = leaky_relu(x + bias.view(1, C, 1, ...), negative_slope) * scale      (act=3, grad=0)
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("MARCONET_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "models", "networks.py"))


def _install_fused_act_stub():
    if "basicsr.ops.fused_act" in sys.modules:
        return
    import torch
    import torch.nn as nn
    import torch.nn.functional as F

    def fused_leaky_relu(input, bias, negative_slope=0.2, scale=2 ** 0.5):
        shape = [1, -1] + [1] * (input.dim() - 2)
        return F.leaky_relu(input + bias.view(*shape), negative_slope) * scale

    class FusedLeakyReLU(nn.Module):
        def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5):
            super().__init__()
            self.bias = nn.Parameter(torch.zeros(channel))
            self.negative_slope = negative_slope
            self.scale = scale

        def forward(self, input):
            return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)

    pkg = types.ModuleType("basicsr")
    ops = types.ModuleType("basicsr.ops")
    fa = types.ModuleType("basicsr.ops.fused_act")
    fa.fused_leaky_relu = fused_leaky_relu
    fa.FusedLeakyReLU = FusedLeakyReLU
    pkg.ops = ops
    ops.fused_act = fa
    sys.modules["basicsr"] = pkg
    sys.modules["basicsr.ops"] = ops
    sys.modules["basicsr.ops.fused_act"] = fa


def load_reference_networks():
    """Returns the reference's ``models.networks`` module (unmodified source, imported in place)."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    _install_fused_act_stub()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import importlib
    return importlib.import_module("models.networks")
