"""Builder-owned ResNet architectures (random init) for runs on a box without torchvision.

The driver only cares about three properties of a model, and these classes reproduce them for torchvision's
`resnet18` / `resnet50` (the models of BASELINE.json configs 1 and 3; reference main.py:61-62 loads them from
torchvision):

  * the REGISTRATION ORDER of the Linear / Conv2d leaves, which is the order `extract_layers` visits and therefore
    the order layers are quantized in (reference utils.py:76-93): stem conv, then per block conv1, conv2[, conv3],
    and the block's downsample conv LAST (it is registered after the main path), then fc;
  * the container types extract_layers must recurse through by EXACT type (utils.py:17-22 whitelists torchvision's
    BasicBlock / Bottleneck / ResNet): the classes here are registered with `register_block_type`;
  * the layer geometries (strided 3x3 convs, 1x1 stride-2 downsample convs, 7x7 stride-2 stem), which the capture
    hooks see (quantize_neural_net.py:295-350 ignores the layer's own stride and samples a kernel-strided grid).

Weights are random (Kaiming-normal convs, default Linear init, BatchNorm at identity in eval mode): there is no
network access for checkpoints, and the hot path's cost does not depend on the values.
"""
import torch
import torch.nn as nn

from .utils import register_block_type


def _conv3x3(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, 3, stride=stride, padding=1, bias=False)


def _conv1x1(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, 1, stride=stride, bias=False)


@register_block_type
class BasicBlockArch(nn.Module):
    """Two 3x3 convs + identity / downsample shortcut (ResNet-18/34); registration order conv1, bn1, relu, conv2, bn2,
    downsample."""
    expansion = 1

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv3x3(cin, planes, stride)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = _conv3x3(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


@register_block_type
class BottleneckArch(nn.Module):
    """1x1 reduce, 3x3 (carries the stride), 1x1 expand + shortcut (ResNet-50/101/152); registration order conv1, bn1,
    conv2, bn2, conv3, bn3, relu, downsample -- the downsample conv is quantized AFTER conv3 although it runs first."""
    expansion = 4

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv1x1(cin, planes)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = _conv3x3(planes, planes, stride)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = _conv1x1(planes, planes * self.expansion)
        self.bn3 = nn.BatchNorm2d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


@register_block_type
class ResNetArch(nn.Module):
    """conv1 7x7/2, bn1, relu, maxpool, layer1..4 (nn.Sequential of blocks), avgpool, fc -- torchvision's attribute
    names and registration order."""

    def __init__(self, block, layers, num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for mod in self.modules():
            if isinstance(mod, nn.Conv2d):
                nn.init.kaiming_normal_(mod.weight, mode="fan_out", nonlinearity="relu")

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(_conv1x1(self.inplanes, planes * block.expansion, stride),
                                       nn.BatchNorm2d(planes * block.expansion))
        seq = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        seq += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*seq)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


def resnet18(num_classes=1000):
    """20 convs + fc = 21 quantizable layers (BASELINE.json config 1)."""
    return ResNetArch(BasicBlockArch, [2, 2, 2, 2], num_classes)


def resnet50(num_classes=1000):
    """53 convs + fc = 54 quantizable layers (BASELINE.json config 3; its sixteen 3x3 conv2 layers are the headline)."""
    return ResNetArch(BottleneckArch, [3, 4, 6, 3], num_classes)


ARCHITECTURES = {"resnet18": resnet18, "resnet50": resnet50}
