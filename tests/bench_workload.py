"""The benchmark workload (BASELINE.json / SURVEY.md 8d): layer shapes and synthetic inputs.
Shared by bench.py and the full-size GPU tests."""
import math

import torch

# ResNet-50, 224x224 input, calibration batch B, retain_rate 0.25: m = B * int(0.25*L + 1) with
# L = (floor((H + 2p - (k-1) - 1)/k) + 1)^2 (unfold stride = kernel size, quantize_neural_net.py:320)
def conv_m(B, H, k, pad, retain=0.25):
    L = ((H + 2 * pad - (k - 1) - 1) // k + 1) ** 2
    return B * int(retain * L + 1 if retain != 1 else L)


def resnet50_3x3_layers(B=1024):
    """[(name, N, d, m, groups, (C_in, H, k, pad))] for the sixteen 3x3 conv2 layers; the last entry is
    the geometry of the layer's input feature map (used by bench.py --capture)."""
    layers = []
    spec = [("layer1", 64, 3, 56, 56), ("layer2", 128, 4, 56, 28), ("layer3", 256, 6, 28, 14), ("layer4", 512, 3, 14, 7)]
    for name, planes, blocks, h_first, h_rest in spec:
        for b in range(blocks):
            H = h_first if b == 0 else h_rest       # conv2 of block 0 sees the pre-stride map
            layers.append(("%s.%d.conv2" % (name, b), planes, planes * 9, conv_m(B, H, 3, 1), 1, (planes, H, 3, 1)))
    return layers


def resnet50_all_convs(B=1024):
    """[(name, N, d, m)] for all 53 conv layers of ResNet-50 in extract_layers order (conv1, conv2, conv3,
    downsample per Bottleneck; torchvision v1.5: the stride sits on conv2).  m follows the capture rule: the unfold
    stride is the kernel size, so a 1x1 conv sees every position of its INPUT map whatever its own stride is."""
    layers = [("conv1", 64, 3 * 49, conv_m(B, 224, 7, 3))]
    inplanes, H = 64, 56
    for name, planes, blocks, stride in (("layer1", 64, 3, 1), ("layer2", 128, 4, 2), ("layer3", 256, 6, 2), ("layer4", 512, 3, 2)):
        for b in range(blocks):
            s = stride if b == 0 else 1
            Hout = H // s
            layers.append(("%s.%d.conv1" % (name, b), planes, inplanes, conv_m(B, H, 1, 0)))
            layers.append(("%s.%d.conv2" % (name, b), planes, planes * 9, conv_m(B, H, 3, 1)))
            layers.append(("%s.%d.conv3" % (name, b), planes * 4, planes, conv_m(B, Hout, 1, 0)))
            if b == 0:
                layers.append(("%s.0.downsample.0" % name, planes * 4, inplanes, conv_m(B, H, 1, 0)))
            inplanes, H = planes * 4, Hout
    return layers


def resnet50_all_layers(B=1024):
    """BASELINE.json configs[3]: all 54 layers of ResNet-50 (the 53 convs in extract_layers order + the fc)."""
    return resnet50_all_convs(B) + [("fc", 1000, 2048, B)]


def resnet18_layers(B=256):
    """[(name, N, d, m)] for the 21 conv + fc layers of ResNet-18 in extract_layers order (BASELINE.json configs[1]:
    4-bit, calibration batch 256).  BasicBlock: conv1 (carries the stride), conv2, downsample."""
    layers = [("conv1", 64, 3 * 49, conv_m(B, 224, 7, 3))]
    inplanes, H = 64, 56
    for name, planes, stride in (("layer1", 64, 1), ("layer2", 128, 2), ("layer3", 256, 2), ("layer4", 512, 2)):
        for b in range(2):
            s = stride if b == 0 else 1
            Hout = H // s
            layers.append(("%s.%d.conv1" % (name, b), planes, inplanes * 9, conv_m(B, H, 3, 1)))
            layers.append(("%s.%d.conv2" % (name, b), planes, planes * 9, conv_m(B, Hout, 3, 1)))
            if b == 0 and (s != 1 or inplanes != planes):
                layers.append(("%s.0.downsample.0" % name, planes, inplanes, conv_m(B, H, 1, 0)))
            inplanes, H = planes, Hout
    layers.append(("fc", 1000, 512, B))
    return layers


def vgg16_layers(B=512):
    """[(name, N, d, m)] for the 13 conv + 3 fc layers of VGG-16 (BASELINE.json configs[2]: 4-bit, calibration
    batch 512): first convs with m = 720 384 calibration rows, fc6 with 25 088 columns."""
    layers = []
    cin, H, i = 3, 224, 0
    for v in (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"):
        if v == "M":
            H //= 2
            continue
        layers.append(("features.conv%d" % i, v, cin * 9, conv_m(B, H, 3, 1)))
        cin, i = v, i + 1
    layers += [("classifier.fc6", 4096, 512 * 7 * 7, B), ("classifier.fc7", 4096, 4096, B), ("classifier.fc8", 1000, 4096, B)]
    return layers


def efficientnet_b1_layers(B=1024):
    """BASELINE.json configs[4]: the 116 Conv2d / Linear layers of torchvision's efficientnet_b1 (width 1.0, depth 1.1,
    224x224 input as data_loaders.py:58-59 crops) in extract_layers order -- stem, then per MBConv block: expand 1x1
    (absent when the expand ratio is 1), depthwise kxk (groups = channels), squeeze-excite fc1 / fc2 (1x1 convs on a 1x1
    map: m = B), project 1x1 -- then the 1x1 head and the classifier.  Entries: (name, N, d_g, m, groups)."""
    def _c(v):                                       # torchvision _make_divisible(v, 8)
        n = max(8, int(v + 4) // 8 * 8)
        return n + 8 if n < 0.9 * v else n
    layers = [("features.0.0", 32, 27, conv_m(B, 224, 3, 1), 1)]
    H = 112
    # (expand ratio, kernel, stride, in, out, layers at depth 1.0); depth multiplier 1.1 -> ceil
    stages = [(1, 3, 1, 32, 16, 1), (6, 3, 2, 16, 24, 2), (6, 5, 2, 24, 40, 2), (6, 3, 2, 40, 80, 3),
              (6, 5, 1, 80, 112, 3), (6, 5, 2, 112, 192, 4), (6, 3, 1, 192, 320, 1)]
    for si, (e, k, s, cin, cout, n) in enumerate(stages):
        cin, cout = _c(cin), _c(cout)
        for b in range(int(math.ceil(n * 1.1))):
            ci = cin if b == 0 else cout
            st = s if b == 0 else 1
            exp = _c(ci * e)
            pre = "features.%d.%d.block" % (si + 1, b)
            j = 0
            if exp != ci:
                layers.append(("%s.%d.0" % (pre, j), exp, ci, conv_m(B, H, 1, 0), 1))
                j += 1
            layers.append(("%s.%d.0" % (pre, j), exp, k * k, conv_m(B, H, k, (k - 1) // 2), exp))      # depthwise
            Hout = H // st
            sq = max(1, ci // 4)
            layers.append(("%s.%d.fc1" % (pre, j + 1), sq, exp, conv_m(B, 1, 1, 0), 1))
            layers.append(("%s.%d.fc2" % (pre, j + 1), exp, sq, conv_m(B, 1, 1, 0), 1))
            layers.append(("%s.%d.0" % (pre, j + 2), cout, exp, conv_m(B, Hout, 1, 0), 1))
            H = Hout
    layers.append(("features.8.0", 1280, _c(320), conv_m(B, H, 1, 0), 1))
    layers.append(("classifier.1", 1000, 1280, B, 1))
    return layers


def normalize_layers(layers):
    """(name, N, d, m) -> (name, N, d_g, m, groups=1); 5-tuples pass through."""
    return [tuple(l) if len(l) >= 5 else tuple(l) + (1,) for l in layers]


WORKLOADS = {
    # name: (layer list function, named calibration batch, description[, quantizer config: bits / reg / lamb])
    "r50_3x3": (resnet50_3x3_layers, 1024, "ResNet-50 sixteen 3x3 conv2 layers"),
    "r50_all_convs": (resnet50_all_convs, 1024, "ResNet-50 all 53 conv layers"),
    "r50_all": (resnet50_all_layers, 1024, "ResNet-50 all 54 conv + fc layers"),
    "effnet_b1": (efficientnet_b1_layers, 1024, "EfficientNet-B1 all 116 conv + fc layers (23 depthwise, 46 squeeze-excite)",
                  dict(bits=2, reg="L1", lamb=0.1)),
    "r18": (resnet18_layers, 256, "ResNet-18 all 21 conv + fc layers"),
    "vgg16": (vgg16_layers, 512, "VGG-16 all 16 conv + fc layers"),
}


def algorithmic_bytes(N, d, m, groups=1):
    """per greedy step of one group 8*N_g*m + 8*m + 8*N_g bytes; per layer groups*d_g times that."""
    Ng = N // groups
    return groups * d * (8 * Ng * m + 8 * m + 8 * Ng)


def synthetic_layer(N, d, m, seed, first_layer=False, d_limit=None, rows_d=None):
    """W = randn*sqrt(2/d); A = relu(pre); X = relu(pre + 0.05 randn); every 97th column of X zero.
    CPU generator so that every path sees identical bits.  d_limit keeps only the first columns.
    Grouped layers: d = groups*d_g input features, rows_d = d_g columns of W (the generator stream of an ungrouped
    layer, rows_d None or == d, is unchanged)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    dd = d if d_limit is None else min(d, d_limit)
    wd = dd if rows_d is None or rows_d == d else rows_d
    W = torch.randn(N, wd, generator=g) * math.sqrt(2.0 / (d if wd == dd else rows_d))
    pre = torch.randn(m, dd, generator=g)
    if first_layer:
        A = pre
        X = pre.clone()
    else:
        A = torch.relu(pre)
        X = torch.relu(pre + 0.05 * torch.randn(m, dd, generator=g))
    X[:, ::97] = 0.0
    return W, A, X


def synthetic_capture_layer(layer, B, seed, device="cuda", retain=0.25):
    """Inputs of one conv layer as the real driver sees them (quantize_neural_net.py:325-350): the input feature maps
    of the analog and the quantized network (B, C, H, H) and the sampled patch rows (per image int(retain*L + 1) of the
    L kernel-sized patches, drawn with replacement).  Generated on the device (feature maps are GBs); W on the host
    like synthetic_layer.  Returns (W, fmap_a, fmap_x, (kh, kw, ph, pw), patch_index int64 [m])."""
    name, N, d, m, groups, (C, H, k, pad) = layer[:6]
    g = torch.Generator(device="cpu").manual_seed(seed)
    W = torch.randn(N, d, generator=g) * math.sqrt(2.0 / d)
    gd = torch.Generator(device=device).manual_seed(seed)
    pre = torch.randn(B, C, H, H, generator=gd, device=device)
    fa = torch.relu(pre)
    fx = torch.relu(pre + 0.05 * torch.randn(B, C, H, H, generator=gd, device=device))
    del pre
    side = (H + 2 * pad - (k - 1) - 1) // k + 1
    L = side * side
    keep = int(retain * L + 1 if retain != 1 else L)
    assert B * keep == m, (name, B * keep, m)
    sel = (torch.arange(B)[:, None] * L + torch.randint(0, L, (B, keep), generator=g)).reshape(-1)
    return W, fa, fx, (k, k, pad, pad), sel


def layer_step(W, scalar=1.16, K=8):
    """alphabet step for percentile 1: (scalar/K) * mean(rowmax |W|)  (step_algorithm.py:191-192)"""
    return float((scalar / K) * W.abs().max(dim=1).values.mean())
