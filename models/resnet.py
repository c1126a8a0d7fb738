"""``models.resnet`` of the reference (models/resnet.py) → ``marconet_amd.resnet``: ``resnet45stride()`` (:73-74)."""
from marconet_amd.resnet import BasicBlock, ResNet, resnet45stride  # noqa: F401

__all__ = ["ResNet", "BasicBlock", "resnet45stride"]
