"""Driver-level parity on the MI355X: QuantizeNeuralNet(...).quantize_network() on the toy network of the G5
fixture must reproduce the reference's quantized weights (the fixture was produced by running the reference's
own quantize_neural_net.py on the CPU with the same seeds, batches and np.random stream)."""
import json
import os

import numpy as np
import pytest
import torch

import golden_inputs as gi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("prefetch", [False, True], ids=["serial", "analog_one_layer_ahead"])
@pytest.mark.parametrize("ci", [0, 1, 2])
def test_quantize_network_matches_reference(ci, prefetch, capsys):
    """... and with the analog capture of layer i+1 put into the column layout on a side stream while layer i is
    quantized (prefetch_analog: same batches, same numpy draws, mixed prepared / matrix inputs for Linear layers): the
    same bits."""
    from quantized_neural_nets_amd import QuantizeNeuralNet
    fx = np.load(os.path.join(gi.GOLDEN_DIR, "g5_driver.npz"))
    meta = json.loads(str(fx["meta"]))["configs"][ci]
    cfg = meta["cfg"]
    assert cfg == gi.DRIVER_CONFIGS[ci]
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(meta["net_seed"])
    net = gi.toy_net(rng).to(dev)
    batches = gi.toy_batches(rng, meta["batch"], meta["nlayers"])
    np.random.seed(meta["np_seed"])
    torch.manual_seed(meta["np_seed"])
    quant = QuantizeNeuralNet(net, "toy", meta["batch"], batches, mlp_bits=cfg["bits"], cnn_bits=cfg["bits"],
                              ignore_layers=[], mlp_alphabet_scalar=1.16, cnn_alphabet_scalar=1.16,
                              mlp_percentile=1, cnn_percentile=1, reg=cfg["reg"], lamb=cfg["lamb"],
                              retain_rate=cfg["retain_rate"], stochastic_quantization=False, device=dev)
    assert quant.prefetch_analog is False            # (off by default: measured slower)
    quant.prefetch_analog = prefetch
    qnet = quant.quantize_network()
    assert qnet is quant.quantized_network and len(quant.quantized_network_layers) == meta["nlayers"]
    for li, layer in enumerate(quant.quantized_network_layers):
        got = layer.weight.detach().cpu().numpy()
        want = fx["c%d_layer%d_weight" % (ci, li)]
        assert got.shape == want.shape
        assert np.array_equal(got, want), "layer %d of config %d differs from the reference" % (li, ci)
    # analog network untouched, biases untouched
    rng2 = np.random.default_rng(meta["net_seed"])
    ref_net = gi.toy_net(rng2)
    for a, b in zip(quant.analog_network.parameters(), ref_net.parameters()):
        assert torch.equal(a.cpu(), b)
    out = capsys.readouterr().out
    assert "Quantizing layer with index: 4" in out and "The relative quantization error of layer 0" in out


def test_ignore_layers_and_unsupported_layer_type():
    from quantized_neural_nets_amd import QuantizeNeuralNet
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(3)
    net = gi.toy_net(rng).to(dev)
    batches = gi.toy_batches(rng, 4, 5)
    np.random.seed(0)
    quant = QuantizeNeuralNet(net, "toy", 4, batches, 4, 4, [0, 3], 1.16, 1.16, 1, 1, None, 0.1, 0.25, False, dev)
    quant.quantize_network()
    assert torch.equal(quant.quantized_network_layers[0].weight, quant.analog_network_layers[0].weight)
    assert torch.equal(quant.quantized_network_layers[3].weight, quant.analog_network_layers[3].weight)
    assert not torch.equal(quant.quantized_network_layers[1].weight, quant.analog_network_layers[1].weight)
    w = quant.quantized_network_layers[4].weight
    assert torch.unique(w).numel() <= 17


@pytest.mark.parametrize("cfg", [
    # B, C, H, W, kernel, padding, dilation, retain
    (5, 4, 12, 12, 3, 1, 1, 0.25),
    (3, 6, 14, 10, (3, 2), (1, 0), 1, 0.5),
    (4, 3, 17, 17, 7, 3, 1, 0.25),
    (2, 8, 9, 9, 1, 0, 1, 1),
    (3, 5, 16, 16, 3, 2, 2, 0.3),
    (2, 70, 8, 8, 5, 2, 1, 0.25),        # more than 64 features per patch column block
    (3, 6, 12, 12, 3, 1, 1, 0.5),        # 3 x 3, padding 1: patch rows inside the image (one 12-byte load) and across its edge
    (2, 4, 23, 23, 5, 2, 1, 0.5),        # 5 x 5: 16 + 4 bytes per row
    (2, 3, 30, 30, 7, 3, 1, 1),          # 7 x 7 stem: 16 + 12 bytes per row, every patch kept
])
def test_fused_capture_equals_unfold_path(cfg):
    """gpfq_gather_patches_f32 (patches straight into the column layout) == unfold -> transpose -> reshape -> index
    (quantize_neural_net.py:334-347), bit for bit, including the zero padding of the columns."""
    import quantized_neural_nets_amd.quantize_neural_net as qnn
    from quantized_neural_nets_amd.step_algorithm import PreparedColumns
    from quantized_neural_nets_amd.utils import InterruptException
    B, C, H, W, k, pad, dil, retain = cfg
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, C, H, W, generator=g).to(dev)
    outs = {}
    for fused in (True, False):
        qnn.FUSED_CAPTURE = fused
        np.random.seed(123)
        hook = qnn.SaveInputConv2d(kernel_size=k, dilation=dil, padding=pad, stride=1, groups=1, retain_rate=retain)
        for xin in (x, x * 2.0):
            with pytest.raises(InterruptException):
                hook(None, (xin,), None)
        outs[fused] = hook.inputs
    qnn.FUSED_CAPTURE = True
    for a, b in zip(outs[True], outs[False]):
        assert isinstance(a, PreparedColumns) and tuple(a.shape) == tuple(b.shape)
        assert torch.equal(a.matrix(), b)
        assert a.T.shape[1] % 1024 == 0 and float(a.T[:, a.m:].abs().sum()) == 0.0


def test_driver_matches_reference_without_fused_capture():
    import quantized_neural_nets_amd.quantize_neural_net as qnn
    qnn.FUSED_CAPTURE = False
    try:
        fx = np.load(os.path.join(gi.GOLDEN_DIR, "g5_driver.npz"))
        meta = json.loads(str(fx["meta"]))["configs"][0]
        cfg = meta["cfg"]
        dev = torch.device("cuda:0")
        rng = np.random.default_rng(meta["net_seed"])
        net = gi.toy_net(rng).to(dev)
        batches = gi.toy_batches(rng, meta["batch"], meta["nlayers"])
        np.random.seed(meta["np_seed"])
        quant = qnn.QuantizeNeuralNet(net, "toy", meta["batch"], batches, cfg["bits"], cfg["bits"], [], 1.16, 1.16, 1, 1,
                                      cfg["reg"], cfg["lamb"], cfg["retain_rate"], False, dev)
        quant.quantize_network()
        for li, layer in enumerate(quant.quantized_network_layers):
            assert np.array_equal(layer.weight.detach().cpu().numpy(), fx["c0_layer%d_weight" % li])
    finally:
        qnn.FUSED_CAPTURE = True


def test_cli_alexnet_plumbing_config(capsys, tmp_path):
    """BASELINE.json configs[0] on the GPU: `main.py -model alexnet -b 4 -bs 32 -s 1.16` (random-init AlexNet
    architecture, synthetic calibration batches): all 8 layers are quantized to the 17-level 4-bit alphabet; the run
    appends its row to the CSV log (main.py:166-177) and saves the model under the reference's name (main.py:127-136)."""
    import csv
    from quantized_neural_nets_amd import main as cli
    log = str(tmp_path / "logs" / "Quantization_Log.csv")
    q = cli.main(["-model", "alexnet", "-b", "4", "-bs", "32", "-s", "1.16", "--synthetic", "--log_file", log,
                  "--save_dir", str(tmp_path / "quantized_models")])
    rows = list(csv.reader(open(log)))
    assert rows[0] == cli.LOG_FIELDS and len(rows) == 2 and len(rows[1]) == 20          # the reference's 20 columns
    assert rows[1][:3] == ["alexnet", "ILSVRC2012", "32"] and rows[1][7:10] == ["4", "1.16", "1.16"]
    assert rows[1][12:15] == ["False", "", "0.1"] and rows[1][17:20] == ["0.25", "False", "0"]
    assert 0.0 <= float(rows[1][15]) <= float(rows[1][16]) < 1.0            # quantization only adds zeros
    saved = tmp_path / "quantized_models" / "alexnet" / ("dsILSVRC2012_b4_batch32_mlpscalar1.16_cnnscalar1.16_mlppercentile1"
                                                         "_cnnpercentile1_retain_rate0.25_regNone_lambda0.1.pt")
    assert saved.exists()
    reloaded = torch.load(saved, weights_only=False)
    assert torch.equal(reloaded[0].weight.cpu(), q.quantized_network_layers[0].weight.cpu())
    layers = q.quantized_network_layers
    assert len(layers) == 8 and len(q.layer_reports) == 8
    for rep in q.layer_reports:
        assert np.isfinite(rep["relative_quantize_error"]) and 0 < rep["relative_quantize_error"] < 1.0
    for layer in (layers[0], layers[4], layers[7]):
        vals = torch.unique(layer.weight.detach())
        assert vals.numel() <= 17
        step = float(vals.detach()[vals.detach() > 0].min())
        k = vals / step
        assert torch.allclose(k, torch.round(k), atol=1e-3)
    out = capsys.readouterr().out
    assert "Time used for quantization" in out and "Sparsity" in out
    assert "Layers redone after a cooperative timeout: 0" in out


@pytest.mark.parametrize("ci", [0, 2])
def test_packed_checkpoint_round_trip_is_bitwise(ci, tmp_path):
    """packed.save() keeps the alphabet indices + steps; packed.load() rebuilds the quantized network bit for bit
    (packed on-disk format instead of the fp32 torch.save of main.py:136)."""
    from quantized_neural_nets_amd import QuantizeNeuralNet, packed
    fx = np.load(os.path.join(gi.GOLDEN_DIR, "g5_driver.npz"))
    meta = json.loads(str(fx["meta"]))["configs"][ci]
    cfg = meta["cfg"]
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(meta["net_seed"])
    net = gi.toy_net(rng).to(dev)
    batches = gi.toy_batches(rng, meta["batch"], meta["nlayers"])
    np.random.seed(meta["np_seed"])
    quant = QuantizeNeuralNet(net, "toy", meta["batch"], batches, mlp_bits=cfg["bits"], cnn_bits=cfg["bits"],
                              ignore_layers=[1], mlp_alphabet_scalar=1.16, cnn_alphabet_scalar=1.16,
                              mlp_percentile=1, cnn_percentile=1, reg=cfg["reg"], lamb=cfg["lamb"],
                              retain_rate=cfg["retain_rate"], stochastic_quantization=False, device=dev)
    qnet = quant.quantize_network()
    path = str(tmp_path / "toy.gpfq")
    info = packed.save(path, quant)
    assert info["layers"] == meta["nlayers"] - 1 and info["packed_bytes"] * 4 < info["fp32_bytes"]
    fresh = gi.toy_net(np.random.default_rng(123))          # same architecture, other weights
    packed.load(path, fresh)
    want = qnet.state_dict()
    got = fresh.state_dict()
    assert set(want) == set(got)
    for k in want:
        assert torch.equal(want[k].cpu(), got[k]), k


def test_two_quantizers_interleaved_layer_by_layer():
    """No class-level state travels between StepAlgorithm calls: two QuantizeNeuralNet instances whose layers are
    quantized alternately (config 0 and config 1 of the G5 fixture) each reproduce the reference's weights, and each
    records ITS OWN indices for the packed checkpoint."""
    from quantized_neural_nets_amd import QuantizeNeuralNet, dist as qd
    fx = np.load(os.path.join(gi.GOLDEN_DIR, "g5_driver.npz"))
    metas = json.loads(str(fx["meta"]))["configs"]
    dev = torch.device("cuda:0")
    quants, streams = [], []
    for ci in (0, 1):
        meta, cfg = metas[ci], metas[ci]["cfg"]
        rng = np.random.default_rng(meta["net_seed"])
        net = gi.toy_net(rng).to(dev)
        batches = gi.toy_batches(rng, meta["batch"], meta["nlayers"])
        quants.append(QuantizeNeuralNet(net, "toy", meta["batch"], batches, mlp_bits=cfg["bits"], cnn_bits=cfg["bits"],
                                        ignore_layers=[], mlp_alphabet_scalar=1.16, cnn_alphabet_scalar=1.16,
                                        mlp_percentile=1, cnn_percentile=1, reg=cfg["reg"], lamb=cfg["lamb"],
                                        retain_rate=cfg["retain_rate"], stochastic_quantization=False, device=dev))
        # each instance consumes its own np.random stream, as its own un-interleaved run would
        st = np.random.RandomState(meta["np_seed"])
        streams.append(st.get_state())
    # drive the two layer loops alternately: layer L of A, layer L of B, layer L+1 of A, ...
    for li in range(metas[0]["nlayers"]):
        for qi, quant in enumerate(quants):
            np.random.set_state(streams[qi])
            saved = quant.ignore_layers
            quant.ignore_layers = [i for i in range(metas[qi]["nlayers"]) if i != li]
            # the loader hands one batch per quantized layer (quantize_neural_net.py:227): skip none
            quant.quantize_network()
            quant.ignore_layers = saved
            streams[qi] = np.random.get_state()
    for qi, quant in enumerate(quants):
        for li, layer in enumerate(quant.quantized_network_layers):
            assert np.array_equal(layer.weight.detach().cpu().numpy(), fx["c%d_layer%d_weight" % (qi, li)]), (qi, li)
        assert [e["layer"] for e in quant.layer_indices] == list(range(metas[qi]["nlayers"]))
        for e, layer in zip(quant.layer_indices, quant.quantized_network_layers):
            q = qd.rebuild_q(e["idx"], e["step"], e["K"], e["mode"], e["lamb"]).reshape(layer.weight.shape)
            assert torch.equal(q, layer.weight.detach().cpu()), "instance %d recorded another layer's indices" % qi


def test_cooperative_timeout_falls_back_to_whole_row_streaming(monkeypatch, oracle_mod):
    """A cooperative launch that gives up waiting for a peer (forced: spin limit 0) never hands its outputs on: the
    status word is read before anything is consumed and the layer is redone on GPFQ_PLAN_STREAM_ROWS."""
    from quantized_neural_nets_amd import StepAlgorithm as SA, _lib
    case = dict(name="timeout", N=16, d=24, m=40000, bits=4, scalar=1.16, percentile=1.0, reg=None, lamb=0.0, groups=1,
                first_layer=False, zero_every=0, seed=4)
    W, A, X = gi.make_inputs(case)
    dev = torch.device("cuda:0")
    assert _lib.describe_plan(16, 24, 40000).startswith("coop")
    monkeypatch.setenv("GPFQ_COOP_SPIN_LIMIT", "0")
    r = SA._quantize_layer_ex(torch.from_numpy(W).to(dev), torch.from_numpy(A).to(dev), torch.from_numpy(X).to(dev),
                              40000, 1.16 / 8, 8, 1.0, None, 0.0, 1, False, dev)
    monkeypatch.delenv("GPFQ_COOP_SPIN_LIMIT")
    assert r["timeouts"] == [(16, 24, 40000)]
    o = oracle_mod.quantize_layer(W, A, X, 1.16 / 8, 8)
    assert np.array_equal(r["idx"].cpu().numpy().astype(np.int16), o["idx"])
    assert np.array_equal(r["U"].cpu().numpy(), o["U"])
    assert abs(float(r["quantize_error"]) - o["quantize_error"]) <= 1e-4 * o["quantize_error"]
    _lib.check_status(dev)                          # the status word was consumed by the fallback
    # the in-place surface: U must be restored before the redo
    Q = torch.zeros(16, 24, device=dev)
    U = torch.zeros(16, 40000, device=dev)
    monkeypatch.setenv("GPFQ_COOP_SPIN_LIMIT", "0")
    monkeypatch.setenv("GPFQ_STREAM_C", "4"); monkeypatch.setenv("GPFQ_STREAM_RT", "4")
    SA._quantization(torch.from_numpy(W).to(dev), Q, U, torch.from_numpy(A).to(dev), torch.from_numpy(X).to(dev), SA._msq,
                     float(r["step"]), 8, 0.0)
    assert np.array_equal(U.cpu().numpy(), o["U"]) and np.array_equal(Q.cpu().numpy(), o["Q"])


def test_cooperative_launches_on_two_streams_are_serialised():
    """A cooperative grid is sized to be co-resident on an otherwise idle chip, so two of them must not run at once: the
    scratch area is one per device, and a launch on another stream than the previous user's first waits (on the device)
    for that stream.  Two chip-filling cooperative layers issued back to back on two streams, WITHOUT the host-side
    status read in between: both finish without a timeout, with the results of the serial runs."""
    from quantized_neural_nets_amd import StepAlgorithm as SA, _lib
    import bench_workload as bw
    dev = torch.device("cuda:0")
    N, d, m = 64, 96, 93184
    assert _lib.describe_plan(N, d, m).startswith("coop") and "grid=256" in _lib.describe_plan(N, d, m)   # a chip-filling cooperative grid
    layers = []
    for seed in (11, 12):
        W, A, X = bw.synthetic_layer(N, d, m, seed, first_layer=False)
        layers.append((W.to(dev), A.to(dev), X.to(dev), bw.layer_step(W)))
    ref = [SA._quantize_layer_ex(W, A, X, m, 1.16 / 8, 8, 1, None, 0.1, 1, False, dev, compute_errors=False, step_override=st)
           for W, A, X, st in layers]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    out = []
    for (W, A, X, st), s in zip(layers, streams):
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            out.append(SA._quantize_layer_ex(W, A, X, m, 1.16 / 8, 8, 1, None, 0.1, 1, False, dev, compute_errors=False,
                                             step_override=st, check_status=False))
    for s in streams:
        s.synchronize()
    _lib.check_status(dev)                           # raises if either launch gave up waiting for a peer workgroup
    for r, q in zip(ref, out):
        assert torch.equal(r["idx"], q["idx"]) and torch.equal(r["U"], q["U"])


def test_cooperative_layers_from_two_host_threads_keep_their_granules_and_status_apart():
    """Two HOST THREADS, each on a stream of its own, each quantizing chip-filling cooperative layers through the default
    surface (status read behind every launch): the scratch area -- exchange granules and status words -- is one per device,
    so `scratch() -> launch -> status read` runs under the device's lock (_lib.exclusive).  Without it both threads can
    pass scratch() before either launches: two grids on the same granules, one thread's memset under the other's exchange,
    either thread consuming the other's timeout.  Both threads must return the serial results, with no timeout."""
    import threading
    from quantized_neural_nets_amd import StepAlgorithm as SA, _lib
    import bench_workload as bw
    dev = torch.device("cuda:0")
    N, d, m = 64, 48, 93184
    assert _lib.describe_plan(N, d, m).startswith("coop")
    layers = []
    for seed in range(21, 27):
        W, A, X = bw.synthetic_layer(N, d, m, seed, first_layer=False)
        layers.append((W.to(dev), A.to(dev), X.to(dev), bw.layer_step(W)))
    ref = [SA._quantize_layer_ex(W, A, X, m, 1.16 / 8, 8, 1, None, 0.1, 1, False, dev, compute_errors=False, step_override=st)
           for W, A, X, st in layers]
    torch.cuda.synchronize()
    out, errors = {}, []
    start = threading.Barrier(2)

    def worker(which):
        try:
            s = torch.cuda.Stream(device=dev)
            start.wait()
            with torch.cuda.stream(s):
                for i in range(which, len(layers), 2):
                    W, A, X, st = layers[i]
                    out[i] = SA._quantize_layer_ex(W, A, X, m, 1.16 / 8, 8, 1, None, 0.1, 1, False, dev, compute_errors=False,
                                                   step_override=st)
            s.synchronize()
        except Exception as e:                       # noqa: BLE001  (reported by the main thread)
            errors.append(e)

    ts = [threading.Thread(target=worker, args=(w,)) for w in (0, 1)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    _lib.check_status(dev)
    for i, r in enumerate(ref):
        assert out[i]["timeouts"] == [] and torch.equal(r["idx"], out[i]["idx"]) and torch.equal(r["U"], out[i]["U"])
