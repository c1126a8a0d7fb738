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


def _fused_act_stub_modules():
    """pure-torch ``basicsr`` / ``basicsr.ops`` / ``basicsr.ops.fused_act`` module objects with upstream semantics"""
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
    return {"basicsr": pkg, "basicsr.ops": ops, "basicsr.ops.fused_act": fa}


_REF_NETWORKS = None


def load_reference_networks():
    """Returns the reference's ``models.networks`` module (unmodified source, imported in place).

    The repo root carries its own drop-in ``models`` package (the HIP classes) and may have registered the HIP
    ``basicsr.ops.fused_act`` provider; the reference is therefore imported with ``sys.modules['models*']`` and
    ``sys.modules['basicsr*']`` swapped out for the duration of the import and restored afterwards — the reference module
    keeps the names it bound at import time, the rest of the process keeps seeing the drop-in package."""
    global _REF_NETWORKS
    if _REF_NETWORKS is not None:
        return _REF_NETWORKS
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    import importlib
    swapped = {k: sys.modules.pop(k) for k in list(sys.modules)
               if k == "models" or k.startswith("models.") or k == "basicsr" or k.startswith("basicsr.")}
    sys.modules.update(_fused_act_stub_modules())
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        importlib.invalidate_caches()
        # the reference's models/ has no __init__.py (a namespace package): a regular package of the same name anywhere on
        # sys.path — the drop-in at the repo root — would win the lookup, so the package object is pinned to the reference
        # directory explicitly
        ref_pkg = types.ModuleType("models")
        ref_pkg.__path__ = [os.path.join(REFERENCE_ROOT, "models")]
        sys.modules["models"] = ref_pkg
        mod = importlib.import_module("models.networks")
        if not os.path.abspath(mod.__file__).startswith(os.path.abspath(REFERENCE_ROOT)):
            raise RuntimeError("models.networks resolved to %s, not the reference tree" % mod.__file__)
    finally:
        sys.path.remove(REFERENCE_ROOT)
        for k in [k for k in sys.modules if k == "models" or k.startswith("models.") or k == "basicsr" or k.startswith("basicsr.")]:
            del sys.modules[k]
        sys.modules.update(swapped)
    _REF_NETWORKS = mod
    return mod
