"""Counterparts of the two helpers of the reference's utils.py that the driver needs
(reference: src/utils.py:24-25 InterruptException, :76-93 extract_layers)."""
import torch.nn as nn

SUPPORTED_LAYER_TYPE = {nn.Linear, nn.Conv2d}

# The reference whitelists nn.Sequential plus torchvision's block classes by exact type
# (utils.py:9-22).  torchvision is optional here: the same classes are whitelisted when it is importable.
SUPPORTED_BLOCK_TYPE = {nn.Sequential}
try:  # pragma: no cover - depends on the installation
    from torchvision.models.resnet import BasicBlock, Bottleneck, ResNet
    from torchvision.models.googlenet import BasicConv2d, Inception, InceptionAux
    from torchvision.models.efficientnet import Conv2dNormActivation, SqueezeExcitation, MBConv
    from torchvision.models.mobilenetv2 import InvertedResidual
    SUPPORTED_BLOCK_TYPE |= {BasicBlock, Bottleneck, ResNet, BasicConv2d, Inception, InceptionAux,
                             Conv2dNormActivation, SqueezeExcitation, MBConv, InvertedResidual}
except Exception:  # torchvision absent: only nn.Sequential containers (and user-registered types) recurse
    pass


class InterruptException(Exception):
    """Raised by the capture hooks to abort a forward pass at the hooked layer (utils.py:24-25)."""


def register_block_type(cls):
    """Let extract_layers recurse into a user container type (the reference hard-codes its list)."""
    SUPPORTED_BLOCK_TYPE.add(cls)
    return cls


def extract_layers(model, layer_list, supported_block_type=None, supported_layer_type=None):
    """Collect Linear/Conv2d leaves in registration order, recursing only through whitelisted container
    types matched by exact type() (utils.py:76-93)."""
    blocks = SUPPORTED_BLOCK_TYPE if supported_block_type is None else supported_block_type
    leaves = SUPPORTED_LAYER_TYPE if supported_layer_type is None else supported_layer_type
    for layer in model.children():
        if type(layer) in blocks:
            extract_layers(layer, layer_list, blocks, leaves)
        if not list(layer.children()) and type(layer) in leaves:
            layer_list.append(layer)
