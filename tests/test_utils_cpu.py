"""CPU checks of the small utilities around the path (counterparts of the reference's utils.py helpers)."""
import numpy as np
import torch
import torch.nn as nn


def _net():
    torch.manual_seed(0)
    net = nn.Sequential(nn.Conv2d(3, 5, 3, padding=1, bias=False), nn.BatchNorm2d(5), nn.ReLU(),
                        nn.Conv2d(5, 4, 3, padding=1, bias=True), nn.BatchNorm2d(4), nn.ReLU(),
                        nn.Flatten(), nn.Linear(4 * 6 * 6, 3))
    for m in net:
        if isinstance(m, nn.BatchNorm2d):
            m.running_mean.uniform_(-0.5, 0.5)
            m.running_var.uniform_(0.5, 2.0)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.uniform_(-0.3, 0.3)
    return net.eval()


def test_fusion_preserves_the_function():
    from quantized_neural_nets_amd.utils import fusion_layers_inplace
    net = _net()
    x = torch.randn(7, 3, 6, 6)
    want = net(x)
    fusion_layers_inplace(net, torch.device("cpu"))
    assert torch.allclose(net(x), want, atol=1e-5)
    bn = net[1]
    assert torch.equal(bn.weight.data, torch.ones(5)) and bn.eps <= 1e-12 and torch.equal(bn.running_mean, torch.zeros(5))
    assert float(net[4].bias.detach().abs().sum()) == 0.0           # conv with a bias absorbs the shift


def test_eval_sparsity_and_accuracy():
    from quantized_neural_nets_amd.utils import eval_sparsity, test_accuracy
    net = _net()
    with torch.no_grad():
        net[0].weight[:2] = 0.0
    layers = [net[0], net[3], net[7]]
    total = sum(l.weight.numel() + (l.bias.numel() if l.bias is not None else 0) for l in layers)
    assert eval_sparsity(net) == np.around(2 * 3 * 9 / total, 4)
    xs = torch.randn(10, 3, 6, 6)
    ys = net(xs).argmax(1)
    ds = torch.utils.data.TensorDataset(xs, ys)
    acc = test_accuracy(net, torch.utils.data.DataLoader(ds, batch_size=4), torch.device("cpu"), topk=(1, 2))
    assert acc[0] == 1.0 and acc[1] == 1.0


def test_cli_log_row_and_model_name(tmp_path):
    """The CSV log (main.py:166-177 / logs/init_log.py:7-11: exactly its 20 columns in its order, header once -- a row
    appended to a log the reference created must fit its header) and the saved-model name (main.py:127-129)."""
    import csv
    from quantized_neural_nets_amd import main as cli
    args = cli.build_parser().parse_args(["-model", "resnet50", "-b", "4", "-bs", "1024", "-s", "1.16", "-reg", "L1", "-l", "0.05"])
    assert cli.saved_model_name(args, 4, 1024, 1.16, 1.16, 1, 1, 0.05) == (
        "dsILSVRC2012_b4_batch1024_mlpscalar1.16_cnnscalar1.16_mlppercentile1_cnnpercentile1_retain_rate0.25_regL1_lambda0.05.pt")
    log = str(tmp_path / "sub" / "log.csv")
    row = ["resnet50", "ILSVRC2012", 1024, 0.7613, 0.74, 0.92862, 0.92, 4, 1.16, 1.16, 1, 1, False, "L1", 0.05, 0.0, 0.31, 0.25, False, 0]
    cli.append_log_row(log, row)
    cli.append_log_row(log, row)
    rows = list(csv.reader(open(log)))
    assert len(rows) == 3 and rows[0] == cli.LOG_FIELDS and rows[1] == [str(v) for v in row] == rows[2]
    assert len(cli.LOG_FIELDS) == 20 and cli.LOG_FIELDS[0] == "Model Name" and cli.LOG_FIELDS[19] == "Seed"
    # appending to a log whose 20-column header the reference wrote (logs/init_log.py) keeps every row at 20 values
    ref_log = str(tmp_path / "Quantization_Log.csv")
    with open(ref_log, "w", newline="") as f:
        csv.writer(f).writerow(cli.LOG_FIELDS)
    cli.append_log_row(ref_log, row)
    assert [len(r) for r in csv.reader(open(ref_log))] == [20, 20]
    assert cli.ORIGINAL_ACCURACY["resnet50"] == (.7613, .92862)
