"""Builder-owned architectures (random init) for runs on a box without torchvision: ResNet-18 / ResNet-50, VGG-16 and
EfficientNet-B1 -- the models of BASELINE.json configs 1-4 (reference main.py:61-62 loads them from torchvision).

The driver only cares about three properties of a model, and these classes reproduce them for torchvision's
`resnet18` / `resnet50` / `vgg16` / `efficientnet_b1`:

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


# ------------------------------------------------------------------------------------------------------------------
# VGG-16 (BASELINE.json config 2).  torchvision's VGG is `features` (nn.Sequential of conv / ReLU / max-pool), `avgpool`,
# `classifier` (nn.Sequential: Linear 25088 -> 4096, ReLU, Dropout, Linear, ReLU, Dropout, Linear -> classes): only
# nn.Sequential containers, which extract_layers recurses through without any registration (reference utils.py:17).
# ------------------------------------------------------------------------------------------------------------------
class VGGArch(nn.Module):
    CFG16 = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M")

    def __init__(self, cfg=CFG16, num_classes=1000):
        super().__init__()
        seq, cin = [], 3
        for v in cfg:
            if v == "M":
                seq.append(nn.MaxPool2d(2, 2))
            else:
                seq += [nn.Conv2d(cin, v, 3, padding=1), nn.ReLU(inplace=True)]
                cin = v
        self.features = nn.Sequential(*seq)
        self.avgpool = nn.AdaptiveAvgPool2d((7, 7))
        self.classifier = nn.Sequential(nn.Linear(512 * 7 * 7, 4096), nn.ReLU(True), nn.Dropout(),
                                        nn.Linear(4096, 4096), nn.ReLU(True), nn.Dropout(), nn.Linear(4096, num_classes))
        for mod in self.modules():
            if isinstance(mod, nn.Conv2d):
                nn.init.kaiming_normal_(mod.weight, mode="fan_out", nonlinearity="relu")
                nn.init.zeros_(mod.bias)

    def forward(self, x):
        return self.classifier(torch.flatten(self.avgpool(self.features(x)), 1))


def vgg16(num_classes=1000):
    """13 convs + 3 fc = 16 quantizable layers (BASELINE.json config 2; fc6 has 25 088 input features)."""
    return VGGArch(VGGArch.CFG16, num_classes)


# ------------------------------------------------------------------------------------------------------------------
# EfficientNet-B1 (BASELINE.json config 4: sparse GPFQ, reg = 'L1', 2 bits).  What the reference's whitelist names
# (utils.py:13, :20: Conv2dNormActivation, SqueezeExcitation, MBConv) and what the driver meets inside them:
#   Conv2dNormActivation  an nn.Sequential SUBCLASS (conv, BatchNorm, activation): matched by exact type, so it needs its
#                         own registration;
#   MBConv                `block` = nn.Sequential[expand 1x1 (absent when the expand ratio is 1), DEPTHWISE k x k (groups =
#                         channels, stride 1 or 2, k = 3 or 5: quantize_neural_net.py:165-193 hands StepAlgorithm
#                         groups = layer.groups, step_algorithm.py:221-247 loops the groups), SqueezeExcitation, project
#                         1x1 without activation], then `stochastic_depth` (identity in eval mode);
#   SqueezeExcitation     avgpool, fc1, fc2 (1x1 Conv2d WITH bias on a 1x1 map: m = batch), activation, scale_activation.
# Registration order = torchvision's; weights random.
# ------------------------------------------------------------------------------------------------------------------
def _make_divisible(v, divisor=8):
    n = max(divisor, int(v + divisor / 2) // divisor * divisor)
    return n + divisor if n < 0.9 * v else n


@register_block_type
class ConvNormActivationArch(nn.Sequential):
    def __init__(self, cin, cout, kernel_size=3, stride=1, groups=1, activation=True):
        layers = [nn.Conv2d(cin, cout, kernel_size, stride, (kernel_size - 1) // 2, groups=groups, bias=False),
                  nn.BatchNorm2d(cout)]
        if activation:
            layers.append(nn.SiLU(inplace=True))
        super().__init__(*layers)


@register_block_type
class SqueezeExcitationArch(nn.Module):
    def __init__(self, channels, squeeze):
        super().__init__()
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc1 = nn.Conv2d(channels, squeeze, 1)
        self.fc2 = nn.Conv2d(squeeze, channels, 1)
        self.activation = nn.SiLU(inplace=True)
        self.scale_activation = nn.Sigmoid()

    def forward(self, x):
        scale = self.scale_activation(self.fc2(self.activation(self.fc1(self.avgpool(x)))))
        return scale * x


@register_block_type
class MBConvArch(nn.Module):
    def __init__(self, cin, cout, expand_ratio, kernel, stride):
        super().__init__()
        self.use_res_connect = stride == 1 and cin == cout
        exp = _make_divisible(cin * expand_ratio)
        layers = []
        if exp != cin:
            layers.append(ConvNormActivationArch(cin, exp, 1))
        layers.append(ConvNormActivationArch(exp, exp, kernel, stride, groups=exp))
        layers.append(SqueezeExcitationArch(exp, max(1, cin // 4)))
        layers.append(ConvNormActivationArch(exp, cout, 1, activation=False))
        self.block = nn.Sequential(*layers)
        self.stochastic_depth = nn.Identity()        # (torchvision: StochasticDepth, the identity in eval mode)

    def forward(self, x):
        out = self.block(x)
        return x + self.stochastic_depth(out) if self.use_res_connect else out


class EfficientNetArch(nn.Module):
    """`features` (stem, seven stages of MBConv blocks as nn.Sequential, 1x1 head), `avgpool`, `classifier` (Dropout,
    Linear) -- torchvision's attribute names and registration order."""
    # (expand ratio, kernel, stride, in, out, blocks at depth 1.0)
    STAGES = ((1, 3, 1, 32, 16, 1), (6, 3, 2, 16, 24, 2), (6, 5, 2, 24, 40, 2), (6, 3, 2, 40, 80, 3),
              (6, 5, 1, 80, 112, 3), (6, 5, 2, 112, 192, 4), (6, 3, 1, 192, 320, 1))

    def __init__(self, width_mult=1.0, depth_mult=1.1, num_classes=1000):
        super().__init__()
        import math
        ch = lambda c: _make_divisible(c * width_mult)                      # noqa: E731
        feats = [ConvNormActivationArch(3, ch(32), 3, 2)]
        for (e, k, s, cin, cout, n) in self.STAGES:
            blocks = []
            for b in range(int(math.ceil(n * depth_mult))):
                blocks.append(MBConvArch(ch(cin) if b == 0 else ch(cout), ch(cout), e, k, s if b == 0 else 1))
            feats.append(nn.Sequential(*blocks))
        last = ch(self.STAGES[-1][4])
        feats.append(ConvNormActivationArch(last, 4 * last, 1))
        self.features = nn.Sequential(*feats)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.classifier = nn.Sequential(nn.Dropout(0.2, inplace=True), nn.Linear(4 * last, num_classes))
        for mod in self.modules():
            if isinstance(mod, nn.Conv2d):
                nn.init.kaiming_normal_(mod.weight, mode="fan_out")
                if mod.bias is not None:
                    nn.init.zeros_(mod.bias)

    def forward(self, x):
        return self.classifier(torch.flatten(self.avgpool(self.features(x)), 1))


def efficientnet_b1(num_classes=1000):
    """116 quantizable layers: stem, 23 MBConv blocks (2 without an expand conv: 4 layers each; 21 with: 5), head, fc
    (BASELINE.json config 4)."""
    return EfficientNetArch(1.0, 1.1, num_classes)


ARCHITECTURES = {"resnet18": resnet18, "resnet50": resnet50, "vgg16": vgg16, "efficientnet_b1": efficientnet_b1}
