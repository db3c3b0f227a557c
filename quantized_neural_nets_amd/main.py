"""Command line of the quantization run -- counterpart of the reference's src/main.py with the same flags
(main.py:17-41) and the same hyper-parameter grid (main.py:182-193), driving QuantizeNeuralNet on the MI355X.

    python -m quantized_neural_nets_amd.main -model alexnet -b 4 -bs 32 -s 1.16 --synthetic

Models come from torchvision when it is installed (pretrained=True needs its checkpoint cache); without
torchvision `alexnet`, `resnet18` and `resnet50` are available as randomly initialised copies of the architectures
(arch.py), which together with --synthetic (random calibration batches instead of the ImageNet loader) is enough to run the whole plumbing.
Like the reference, every run appends one row to a CSV log (main.py:166-177; same 20 columns, written with the
header when the file is new) and, with --save_dir, saves the quantized nn.Module under the reference's file name
(main.py:127-136).  Accuracy columns are evaluated when a test loader exists (main.py:138-159); with --synthetic
there is none and they stay empty.
"""
import argparse
import csv
import os
from datetime import datetime

import numpy as np
import torch
import torch.nn as nn

from .quantize_neural_net import QuantizeNeuralNet
from .utils import eval_sparsity, fusion_layers_inplace, test_accuracy


def build_parser():
    p = argparse.ArgumentParser(description='GPFQ post-training quantization (MI355X)')
    p.add_argument('--bits', '-b', default=[4], type=int, nargs='+', help='number of bits for quantization')
    p.add_argument('--scalar', '-s', default=[1.16], type=float, nargs='+',
                   help='the scalar C used to determine the radius of alphabets')
    p.add_argument('--batch_size', '-bs', default=[128], type=int, nargs='+', help='batch size used for quantization')
    p.add_argument('--percentile', '-p', default=[1], type=float, nargs='+', help='percentile of weights')
    p.add_argument('--num_worker', '-w', default=8, type=int, help='number of workers for data loader')
    p.add_argument('--data_set', '-ds', default='ILSVRC2012', choices=['ILSVRC2012', 'CIFAR10'])
    p.add_argument('-model', default='resnet18', help='model name')
    p.add_argument('--stochastic_quantization', '-sq', action='store_true', help='use stochastic quantization')
    p.add_argument('--retain_rate', '-rr', default=0.25, type=float, help='subsampling probability p for conv layers')
    p.add_argument('--regularizer', '-reg', default=None, choices=['L0', 'L1'], help='regularization mode')
    p.add_argument('--lamb', '-l', default=[0.1], type=float, nargs='+', help='regularization term')
    p.add_argument('--ignore_layer', '-ig', default=[], type=int, nargs='+', help='indices of unquantized layers')
    p.add_argument('-seed', default=0, type=int, help='set random seed')
    p.add_argument('--fusion', '-f', action='store_true', help='fusing CNN and BN layers')
    # additions of this build
    p.add_argument('--synthetic', action='store_true', help='random calibration batches instead of the dataset loader')
    p.add_argument('--image_size', default=224, type=int, help='side of the synthetic images')
    p.add_argument('--data_loaders_dir', default=None,
                   help="directory holding the reference's data_loaders.py (its src/): the dataset loader is imported from "
                        "exactly there, never from whatever module of that name sys.path happens to offer")
    p.add_argument('--save_dir', default=None, help='where to torch.save the quantized model, as main.py:127-136 does under '
                                                   '../quantized_models/ (default: not saved)')
    p.add_argument('--log_file', default=os.path.join('logs', 'Quantization_Log.csv'),
                   help="CSV log every run appends a row to (the reference's ../logs/Quantization_Log.csv); '' disables")
    p.add_argument('--save_packed', default=None, help='write the quantized model as packed alphabet indices + steps to this file')
    return p


class AlexNetArch(nn.Sequential):
    """The AlexNet layer stack (5 convs, 3 fully connected), randomly initialised: stands in for
    torchvision.models.alexnet when torchvision is not installed."""

    def __init__(self, num_classes=1000):
        super().__init__(
            nn.Conv2d(3, 64, 11, stride=4, padding=2), nn.ReLU(inplace=True), nn.MaxPool2d(3, 2),
            nn.Conv2d(64, 192, 5, padding=2), nn.ReLU(inplace=True), nn.MaxPool2d(3, 2),
            nn.Conv2d(192, 384, 3, padding=1), nn.ReLU(inplace=True),
            nn.Conv2d(384, 256, 3, padding=1), nn.ReLU(inplace=True),
            nn.Conv2d(256, 256, 3, padding=1), nn.ReLU(inplace=True), nn.MaxPool2d(3, 2),
            nn.AdaptiveAvgPool2d((6, 6)), nn.Flatten(),
            nn.Dropout(), nn.Linear(256 * 6 * 6, 4096), nn.ReLU(inplace=True),
            nn.Dropout(), nn.Linear(4096, 4096), nn.ReLU(inplace=True), nn.Linear(4096, num_classes))


def load_model(name, data_set='ILSVRC2012'):
    """Returns (model, pretrained): ILSVRC2012 -> torchvision's pretrained model (main.py:61-62); CIFAR10 -> the
    checkpoint pretrained_cifar10/<model>_cifar10.pt (main.py:76-79).  pretrained is False for the random-init AlexNet
    stand-in used when torchvision is not installed."""
    if data_set == 'CIFAR10':
        path = os.path.join('pretrained_cifar10', name + '_cifar10.pt')
        if not os.path.isfile(path):
            raise SystemExit("-ds CIFAR10 needs the checkpoint %s (main.py:76-79 of the reference loads it)" % path)
        return torch.load(path, map_location=torch.device('cpu'), weights_only=False).module, True
    try:
        import torchvision
        return getattr(torchvision.models, name)(pretrained=True), True
    except ImportError:
        if name == 'alexnet':
            return AlexNetArch(), False
        from .arch import ARCHITECTURES
        if name in ARCHITECTURES:                   # builder-owned definitions (arch.py), random init
            return ARCHITECTURES[name](), False
        raise SystemExit("torchvision is not installed: only `-model alexnet` and %s (random init) are available" % ", ".join(sorted(ARCHITECTURES)))


class SyntheticLoader:
    """Endless (images, labels) batches from a seeded generator -- the loader surface QuantizeNeuralNet needs.
    device None: drawn on the host (bit-reproducible anywhere; 0.5 s per batch of 1024 -- the reference hides that cost
    in DataLoader worker processes, data_loaders.py:75).  device given: drawn there by that device's generator (no host
    time, no host-to-device copy: what a prefetching loader looks like to the driver)."""

    def __init__(self, batch_size, image_size, seed, device=None):
        self.bs, self.hw, self.device = batch_size, image_size, device
        self.gen = torch.Generator(device=device if device is not None else "cpu").manual_seed(seed)

    def __iter__(self):
        while True:
            yield (torch.randn(self.bs, 3, self.hw, self.hw, generator=self.gen, device=self.device),
                   torch.zeros(self.bs, dtype=torch.long))


# the columns of the reference's log (main.py:13-14 `fields`, logs/Quantization_Log.csv), in its order
LOG_FIELDS = ['Model Name', 'Dataset', 'Quantization Batch Size', 'Original Top1 Accuracy', 'Quantized Top1 Accuracy',
              'Original Top5 Accuracy', 'Quantized Top5 Accuracy', 'Bits', 'MLP_Alphabet_Scalar', 'CNN_Alphabet_Scalar',
              'MLP_Percentile', 'CNN_Percentile', 'Stochastic Quantization', 'Regularizer', 'Lambda', 'Original Sparsity',
              'Quantized Sparsity', 'Retain_rate', 'Fusion', 'Seed']
# (exactly the reference's 20 columns: rows appended to a log the reference created stay readable by its schema; layers redone
# after a cooperative kernel gave up waiting are reported on stdout and in quantizer.layer_reports, not in the CSV)

# FP32 top-1 / top-5 of the un-quantized torchvision models, as the reference tabulates them (main.py:65-74)
ORIGINAL_ACCURACY = {
    'alexnet': (.56522, .79066), 'vgg16': (.71592, .90382), 'resnet18': (.69758, .89078), 'googlenet': (.69778, .89530),
    'resnet50': (.7613, .92862), 'efficientnet_b1': (.7761, .93596), 'efficientnet_b7': (.84122, .96908),
    'mobilenet_v2': (.71878, .90286)}


def saved_model_name(args, bits, bs, mlp_s, cnn_s, mlp_per, cnn_per, lamb):
    """File name of the saved quantized model, main.py:127-129 (the reference's f-string carries the indentation of
    its continuation lines into the name; that whitespace is not reproduced)."""
    return (f'ds{args.data_set}_b{bits}_batch{bs}_mlpscalar{mlp_s}_cnnscalar{cnn_s}_mlppercentile{mlp_per}'
            f'_cnnpercentile{cnn_per}_retain_rate{args.retain_rate}_reg{args.regularizer}_lambda{lamb}.pt')


def append_log_row(path, row):
    """One row per run, main.py:166-177; the header goes in when the file is new (the reference writes it once at
    import time, main.py:13-16 / logs/Quantization_Log.csv)."""
    assert len(row) == len(LOG_FIELDS)
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    new = not os.path.exists(path) or os.path.getsize(path) == 0
    with open(path, 'a', newline='') as f:
        w = csv.writer(f)
        if new:
            w.writerow(LOG_FIELDS)
        w.writerow(row)


def run(args, bits, mlp_s, cnn_s, bs, mlp_per, cnn_per, lamb):
    if not torch.cuda.is_available():
        raise SystemExit("this build runs on the MI355X only (no CPU path)")
    device = torch.device("cuda:0")
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    model, pretrained = load_model(args.model, args.data_set)
    model = model.to(device).eval()
    if args.fusion:
        fusion_layers_inplace(model, device)
        print('CNN and BN layers are fused before quantization!\n')
    mode = ('stochastic quantization, i.e. SGPFQ' if args.stochastic_quantization else
            f'sparse quantization using {args.regularizer} norm with lambda {lamb}' if args.regularizer else 'GPFQ')
    print(f'Quantization mode: {mode}')
    print(f'Quantizing {args.model} on {device}: bits {bits}, mlp_scalar {mlp_s}, cnn_scalar {cnn_s}, '
          f'percentiles {mlp_per}/{cnn_per}, retain_rate {args.retain_rate}, batch_size {bs}\n')
    test_loader = None
    if args.synthetic:
        train_loader = SyntheticLoader(bs, args.image_size, args.seed)
    else:
        if not args.data_loaders_dir:
            raise SystemExit("no dataset loader: use --synthetic, or --data_loaders_dir <the reference's src/> for its ImageNet loader")
        import importlib.util
        path = os.path.join(args.data_loaders_dir, "data_loaders.py")
        if not os.path.isfile(path):
            raise SystemExit("no data_loaders.py in %s" % args.data_loaders_dir)
        spec = importlib.util.spec_from_file_location("gpfq_reference_data_loaders", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)                      # (main.py:10, :80: data_loader(data_set, batch_size, num_worker))
        train_loader, test_loader = mod.data_loader(args.data_set, bs, args.num_worker)
    quantizer = QuantizeNeuralNet(model, args.model, bs, train_loader, mlp_bits=bits, cnn_bits=bits,
                                  ignore_layers=args.ignore_layer, mlp_alphabet_scalar=mlp_s, cnn_alphabet_scalar=cnn_s,
                                  mlp_percentile=mlp_per, cnn_percentile=cnn_per, reg=args.regularizer, lamb=lamb,
                                  retain_rate=args.retain_rate, stochastic_quantization=args.stochastic_quantization,
                                  device=device)
    start = datetime.now()
    quantized_model = quantizer.quantize_network().to(device)
    torch.cuda.synchronize()
    print(f'\nTime used for quantization: {datetime.now() - start}\n')
    if args.save_dir:
        os.makedirs(os.path.join(args.save_dir, args.model), exist_ok=True)
        torch.save(quantized_model, os.path.join(args.save_dir, args.model,
                                                 saved_model_name(args, bits, bs, mlp_s, cnn_s, mlp_per, cnn_per, lamb)))
    if args.save_packed:
        from . import packed
        info = packed.save(args.save_packed, quantizer)
        print("Packed checkpoint %s: %d layers, %.2f MB of indices for %.2f MB of fp32 weights"
              % (args.save_packed, info["layers"], info["packed_bytes"] / 1e6, info["fp32_bytes"] / 1e6))
    orig_acc, acc = ('', ''), ('', '')
    if test_loader is not None:
        # the tabulated FP32 accuracies are those of torchvision's ImageNet weights (main.py:65-74: the table exists for
        # ILSVRC2012 only); any other model is evaluated (main.py:143-146)
        tabulated = ORIGINAL_ACCURACY.get(args.model) if (args.data_set == 'ILSVRC2012' and pretrained) else None
        orig_acc = tabulated or test_accuracy(model, test_loader, device, (1, 5))
        print(f'Top-1 accuracy of {args.model} is {orig_acc[0]}.')
        print(f'Top-5 accuracy of {args.model} is {orig_acc[1]}.')
        acc = test_accuracy(quantized_model, test_loader, device, (1, 5))
        print(f'Top-1 accuracy of quantized {args.model} is {acc[0]}.')
        print(f'Top-5 accuracy of quantized {args.model} is {acc[1]}.')
    original_sparsity, quantized_sparsity = eval_sparsity(model), eval_sparsity(quantized_model)
    print("Sparsity: Org: {}, Quant: {}".format(original_sparsity, quantized_sparsity))
    if args.log_file:
        append_log_row(args.log_file, [args.model, args.data_set, bs, orig_acc[0], acc[0], orig_acc[1], acc[1], bits, mlp_s, cnn_s,
                                       mlp_per, cnn_per, args.stochastic_quantization, args.regularizer, lamb,
                                       original_sparsity, quantized_sparsity, args.retain_rate, args.fusion, args.seed])
    print("Layers redone after a cooperative timeout: {}".format(sum(len(r.get('timeouts', ())) for r in quantizer.layer_reports)))
    return quantizer


def main(argv=None):
    args = build_parser().parse_args(argv)
    out = None
    for b in args.bits:                       # the grid of main.py:182-189
        for s in args.scalar:
            for bs in args.batch_size:
                for mlp_per in args.percentile:
                    for cnn_per in args.percentile:
                        for lamb in args.lamb:
                            out = run(args, b, s, s, bs, mlp_per, cnn_per, lamb)
    return out


if __name__ == '__main__':
    main()
