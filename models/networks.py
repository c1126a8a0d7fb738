"""``models.networks`` of the reference (models/networks.py) → the HIP-backed classes of ``marconet_amd.networks``.

Same class names, constructor defaults, ``state_dict`` keys/shapes and ``forward()`` call forms:
``modelEncoder(LQ)`` → (logits, locs, w) (test_sr.py:146); ``modelTSPGAN(styles=…, labels=…, noise=None)`` →
(image, prior64, prior32) (test_sr.py:183, test_w.py:108); ``modelSR(LQ, [p64], [p32], locs)`` → SR (test_sr.py:197).
"""
from marconet_amd.networks import (EqualLinear, FusedLeakyReLU, GroupNorm, ModulatedConv2d, PixelNorm,  # noqa: F401
                                   ResTextBlockV2, SelectText, StyledConv, TextContextEncoderV2, TextGenerator, ToRGB,
                                   TSPGAN, TSPSRNet, adaptive_instance_normalization, calc_mean_std_4D, swish)
from marconet_amd.fused_act import fused_leaky_relu  # noqa: F401
from marconet_amd.resnet import resnet45stride as resnet45  # noqa: F401
from marconet_amd.textvit_arch import TextViT as TextEncoder  # noqa: F401

__all__ = ["TextContextEncoderV2", "TSPGAN", "TSPSRNet", "TextGenerator", "PixelNorm", "EqualLinear", "SelectText",
           "StyledConv", "ModulatedConv2d", "ToRGB", "ResTextBlockV2", "GroupNorm", "FusedLeakyReLU", "fused_leaky_relu",
           "swish", "calc_mean_std_4D", "adaptive_instance_normalization"]
