"""``models.textvit_arch`` of the reference (models/textvit_arch.py) → ``marconet_amd.textvit_arch``:
``TextViT(num_classes, dim, max_length=16)`` with ``forward(img) -> (out_cls, out_locs_16, out_w)`` (:12-77)."""
from marconet_amd.textvit_arch import Attention, FeedForward, TextViT, Transformer  # noqa: F401

__all__ = ["TextViT", "Transformer", "Attention", "FeedForward"]
