"""The builder-owned ResNet-18 / ResNet-50 definitions (quantized_neural_nets_amd/arch.py) present the driver with what
torchvision's models present the reference with: the same Linear / Conv2d leaves in the same REGISTRATION order
(extract_layers, reference utils.py:76-93 -- a Bottleneck's downsample conv comes after its conv3), the same weight
shapes, and input feature maps whose sampled-patch counts are the m of the benchmark's layer tables (bench_workload,
SURVEY 6.2)."""
import numpy as np
import pytest
import torch
import torch.nn as nn

import bench_workload as bw
from quantized_neural_nets_amd import arch
from quantized_neural_nets_amd.utils import extract_layers


def _named_leaves(model):
    layers = []
    extract_layers(model, layers)
    names = {id(mod): name for name, mod in model.named_modules()}
    return [(names[id(l)], l) for l in layers]


@pytest.mark.parametrize("name,table,batch", [("resnet18", bw.resnet18_layers, 256), ("resnet50", bw.resnet50_all_layers, 1024),
                                              ("vgg16", bw.vgg16_layers, 512), ("efficientnet_b1", bw.efficientnet_b1_layers, 1024)])
def test_layer_order_shapes_and_sample_counts_match_the_workload_tables(name, table, batch):
    torch.manual_seed(0)
    model = arch.ARCHITECTURES[name]().eval()
    leaves = _named_leaves(model)
    want = bw.normalize_layers(table(batch))
    if name == "vgg16":                      # (the table names VGG's layers by role, torchvision by position in the Sequential)
        assert [n for n, _ in leaves] == ["features.%d" % i for i in (0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28)] + \
            ["classifier.0", "classifier.3", "classifier.6"]
    else:
        assert [n for n, _ in leaves] == [w[0] for w in want]                # registration order, downsample last in its block
    # input side of every quantizable layer from ONE forward of a single image
    seen = {}
    hooks = [l.register_forward_hook(lambda mod, inp, out, key=n: seen.__setitem__(key, tuple(inp[0].shape))) for n, l in leaves]
    with torch.no_grad():
        out = model(torch.randn(1, 3, 224, 224))
    for h in hooks:
        h.remove()
    assert out.shape == (1, 1000)
    for (lname, layer), (wname, N, dg, m, groups) in zip(leaves, (w[:5] for w in want)):
        Wt = layer.weight
        assert Wt.shape[0] == N and Wt[0].numel() == dg, (lname, tuple(Wt.shape), N, dg)
        if isinstance(layer, nn.Conv2d):
            assert layer.groups == groups
            _, C, H, Wd = seen[lname]
            k, p = layer.kernel_size[0], layer.padding[0]
            assert bw.conv_m(batch, H, k, p) == m, (lname, H, k, p, m)       # quantize_neural_net.py:340-345: B * int(p*L + 1)
        else:
            assert m == batch
    if name == "vgg16":
        assert len(leaves) == 16 and leaves[13][1].in_features == 25088
        return
    if name == "efficientnet_b1":
        convs = [l for _, l in leaves if isinstance(l, nn.Conv2d)]
        dw = [l for l in convs if l.groups > 1]
        assert len(leaves) == 116 and len(dw) == 23 and all(l.groups == l.in_channels == l.out_channels for l in dw)
        assert sorted({(l.kernel_size[0], l.stride[0]) for l in dw}) == [(3, 1), (3, 2), (5, 1), (5, 2)]
        se = [n for n, _ in leaves if n.endswith(".fc1") or n.endswith(".fc2")]
        assert len(se) == 46 and all(l.bias is not None for n, l in leaves if n in se)
        return
    # three strided 3x3 convs and three (r18) / four (r50: layer1.0 widens without a stride) 1x1 downsample convs exist
    strided = [n for n, l in leaves if isinstance(l, nn.Conv2d) and l.stride[0] == 2 and l.kernel_size[0] == 3]
    down = [n for n, l in leaves if "downsample" in n]
    assert len(strided) == 3 and len(down) == (3 if name == "resnet18" else 4)


def test_cli_falls_back_to_the_builder_owned_architectures_without_torchvision():
    from quantized_neural_nets_amd import main as cli
    try:
        import torchvision  # noqa: F401
        pytest.skip("torchvision is installed: load_model takes its models")
    except ImportError:
        pass
    model, pretrained = cli.load_model("resnet18")
    assert isinstance(model, arch.ResNetArch) and pretrained is False
    layers = []
    extract_layers(model, layers)
    assert len(layers) == 21
