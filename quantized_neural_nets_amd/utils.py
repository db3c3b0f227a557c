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


# ---- helpers around the path that main.py of the reference imports from utils (main.py:9) -----------------
def fusion_layers_inplace(model, device):
    '''Fold every BatchNorm2d that directly follows a Conv2d (in extract_layers order) into the conv weights and
    leave the BN as "identity plus bias" (utils.py:96-130): w <- w * gamma/sqrt(var+eps); the BN keeps
    running_mean 0, running_var 1, weight 1 and the folded bias.  The reference sets eps = 0, which current
    torch rejects ("batch_norm eps must be positive"); 1e-12 is below fp32 resolution next to var = 1, so the
    arithmetic is the same.'''
    import torch
    layers = []
    extract_layers(model, layers, supported_layer_type=[nn.Conv2d, nn.BatchNorm2d])
    for conv, bn in zip(layers[:-1], layers[1:]):
        if not (isinstance(conv, nn.Conv2d) and isinstance(bn, nn.BatchNorm2d)):
            continue
        inv_std = 1.0 / torch.sqrt(bn.running_var + bn.eps)
        scale = bn.weight.data * inv_std
        shift = bn.bias.data - bn.weight.data * bn.running_mean * inv_std
        conv.weight.data = conv.weight.data * scale[:, None, None, None]
        n = bn.num_features
        bn.running_var = torch.ones(n, device=device)
        bn.running_mean = torch.zeros(n, device=device)
        bn.weight.data = torch.ones(n, device=device)
        bn.eps = 1e-12
        if conv.bias is None:
            bn.bias.data = shift
        else:
            conv.bias.data = conv.bias.data * scale + shift
            bn.bias.data = torch.zeros(n, device=device)


def eval_sparsity(model):
    '''Fraction of exactly-zero entries among the weights and biases of the quantizable layers, rounded to 4
    digits (utils.py:133-159).'''
    import numpy as np
    layers = []
    extract_layers(model, layers)
    total = zeros = 0
    for layer in layers:
        for prm in (layer.weight, layer.bias):
            if prm is not None:
                total += prm.numel()
                zeros += int(prm.eq(0).sum().item())
    return np.around(zeros / total, 4)


def test_accuracy(model, test_dl, device, topk=(1, )):
    '''Top-k accuracy over a loader (utils.py:54-73).'''
    import numpy as np
    import torch
    model.eval()
    maxk = max(topk)
    hits = np.zeros(len(topk))
    for x, target in test_dl:
        with torch.no_grad():
            pred = torch.topk(model(x.to(device)), maxk, dim=1).indices
        match = pred.eq(target.to(device).view(-1, 1))
        for i, k in enumerate(topk):
            hits[i] += int(match[:, :k].sum().item())
    return hits / len(test_dl.dataset)


test_accuracy.__test__ = False      # not a pytest test


def parse_imagenet_val_labels(data_dir):
    '''Class labels of the ILSVRC2012 validation images in file order (utils.py:28-51; the reference's
    data_loaders.py:12 imports this name from `utils`).  Reads the devkit's meta.mat (leaf synsets: ILSVRC id -> wnid),
    ILSVRC2012_validation_ground_truth.txt (one ILSVRC id per image) and wnid_to_label.pickle (wnid -> class index).'''
    import os
    import pickle

    import numpy as np
    import scipy.io as sio
    synsets = sio.loadmat(os.path.join(data_dir, 'meta.mat'), squeeze_me=True)['synsets']
    wnid_of = {int(s[0]): str(s[1]) for s in synsets if int(s[4]) == 0}          # leaves only (num_children == 0)
    truth = np.loadtxt(os.path.join(data_dir, 'ILSVRC2012_validation_ground_truth.txt'))
    with open(os.path.join(data_dir, 'wnid_to_label.pickle'), 'rb') as f:
        label_of = pickle.load(f)
    return np.array([label_of[wnid_of[int(i)]] for i in truth])
