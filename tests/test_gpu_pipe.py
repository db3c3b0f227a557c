"""The PIPELINED cooperative kernels (gpfq_pipe_kernels.h, round 4): a workgroup's rows in four groups, a step in four phases,
publisher / gatherer roles in waves of their own, three exchanges in flight under the fourth group's sweep, granules published
XCD-locally where every member of a tile runs on one XCD.  Same recurrence (reference step_algorithm.py:107-148), same
canonical arithmetic: every configuration must reproduce the CPU oracle -- and the streaming plan -- bit for bit: indices,
Q, U, and the per-segment sums of squares of the epilogue."""
import numpy as np
import pytest
import torch

import bench_workload as bw

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(W, A, X, m, plan, mode="msq", seed=None, K=8, step=None):
    from quantized_neural_nets_amd import StepAlgorithm as SA
    reg = {"msq": None, "soft": "L1", "hard": "L0"}.get(mode)
    r = SA._quantize_layer_ex(W.to(DEV), A.to(DEV), X.to(DEV), m, 1.16 / 8, K, 1, reg, 0.05, 1, mode == "stochastic",
                              torch.device(DEV), step_override=step, plan=plan, seed=seed, compute_errors=False)
    torch.cuda.synchronize()
    return r


# (N, d, m), forced configuration ({} = what AUTO picks), the plan that must result, what the case is there for
CASES = [
    ((64, 16, 93184), {}, "coop RT=4 C=16 waves=6 S=91 grid=256 pipe=1", "four single rows, 16 tiles on the eight XCDs: XCD-local publishing"),
    ((128, 12, 93184), {}, "coop RT=8 C=16 waves=6 S=91 grid=256 pipe=1", "four interleaved pairs, six sweep waves + publisher + gatherer"),
    ((130, 12, 93184), {"GPFQ_COOP_RT": "8"}, "coop RT=8 C=16 waves=6 S=91 grid=256 rounds=2 pipe=1", "two rounds, the last tile with 2 valid rows of 8"),
    ((256, 10, 26624), {}, "coop RT=8 C=8 waves=4 S=26 grid=256 pipe=1", "four sweep waves"),
    ((128, 10, 26624), {}, "coop RT=4 C=8 waves=4 S=26 grid=256 pipe=1", "ResNet-50 layer2.{1,2,3}.conv2's plan: four single rows x 8 members, four sweep waves"),
    ((128, 10, 26624), {"GPFQ_COOP_PIPE": "1", "GPFQ_COOP_RT": "8"}, "coop RT=8 C=16 waves=2 S=26 grid=256 pipe=1", "two sweep waves, members of 1 and 2 segments"),
    ((24, 20, 26624), {"GPFQ_COOP_PIPE": "1", "GPFQ_COOP_RT": "8", "GPFQ_COOP_C": "8"}, "coop RT=8 C=8 waves=4 S=26 grid=24 pipe=1",
     "3 tiles: the grid is padded to 8 tiles so that each tile's members share an XCD; the workgroups of tiles 3..7 leave at once"),
    ((24, 20, 26624), {"GPFQ_COOP_PIPE": "1", "GPFQ_COOP_RT": "8", "GPFQ_COOP_C": "8", "GPFQ_COOP_XCD_TILES": "0"}, "coop RT=8 C=8 waves=4 S=26 grid=24 pipe=1",
     "the same without placement: members spread over the XCDs -- the first gather finds it, device-scope publishing throughout"),
    ((112, 9, 51200), {}, "coop RT=8 C=16 waves=4 S=50 grid=224 pipe=1", "14 tiles padded to 16 (EfficientNet-B1's 112-channel project convs)"),
    ((300, 24, 51200), {"GPFQ_COOP_PIPEL": "0"}, "coop RT=8 C=8 waves=7 S=50 grid=256 rounds=2 pipe=1",
     "SEVEN sweep waves: one wave for both reducer roles (vmcnt(1) behind its own store); the last round partial"),
    ((2048, 6, 13312), {}, "coop RT=8 C=2 waves=7 S=13 grid=256 rounds=2 pipe=1", "two members per tile, the one-wave reducer"),
    ((70, 16, 201728), {"GPFQ_COOP_PIPEL": "0"}, "coop RT=8 C=32 waves=7 S=197 grid=256 rounds=2 pipe=1", "64 granules per gather (two rows x 32 members: 32 lanes per row)"),
    ((70, 12, 26624), {"GPFQ_COOP_PIPE": "1", "GPFQ_COOP_RT": "4", "GPFQ_COOP_C": "4"}, "coop RT=4 C=4 waves=7 S=26 grid=72 pipe=1", "single rows with the one-wave reducer"),
    ((9, 4, 400000), {"GPFQ_COOP_PIPE": "1", "GPFQ_COOP_RT": "4"}, "coop RT=4 C=64 waves=7 S=391 grid=192 pipe=1",
     "64 members of one row per gather (all 64 lanes), the last tile with one valid row"),
    ((21, 6, 803840), {"GPFQ_COOP_PIPE": "1", "GPFQ_COOP_C": "128", "GPFQ_COOP_PIPEL": "0"}, "coop RT=8 C=128 waves=7 S=785 grid=256 rounds=2 pipe=1",
     "two rows x 128 members per gather: FOUR granules per lane, members on four XCDs (device-scope publishing); the last tile has 5 valid rows"),
    ((16, 1, 26624), {"GPFQ_COOP_PIPE": "1", "GPFQ_COOP_RT": "8"}, "coop RT=8 C=16 waves=2 S=26 grid=32 pipe=1", "ONE column: the pipeline is all fill and drain"),
    ((16, 2, 26624), {"GPFQ_COOP_PIPE": "1", "GPFQ_COOP_RT": "8"}, "coop RT=8 C=16 waves=2 S=26 grid=32 pipe=1", "two columns"),
    ((16, 7, 26624), {"GPFQ_COOP_PIPE": "1", "GPFQ_COOP_RT": "8"}, "coop RT=8 C=16 waves=2 S=26 grid=32 pipe=1", "seven columns: the buffer rotation past one period"),
    ((16, 131, 20000), {"GPFQ_COOP_PIPE": "1", "GPFQ_COOP_C": "4", "GPFQ_COOP_RT": "8"}, "coop RT=8 C=4 waves=5 S=20 grid=8 pipe=1", "three Q / idx history flushes, the last partial"),
]


@pytest.mark.parametrize("shape,env,plan_desc,why", CASES, ids=["%dx%dx%d_%s" % (c[0] + ("_".join(c[2].split()[1:4]),)) for c in CASES])
def test_pipelined_kernels_equal_oracle_and_streaming(oracle_mod, monkeypatch, shape, env, plan_desc, why):
    from quantized_neural_nets_amd import _lib
    N, d, m = shape
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    assert _lib.describe_plan(N, d, m).startswith(plan_desc), _lib.describe_plan(N, d, m)
    W, A, X = bw.synthetic_layer(N, d, m, 4242 + N + d, first_layer=False)
    step = bw.layer_step(W)
    r = _run(W, A, X, m, 0, step=step)
    assert r["timeouts"] == []
    Q, idx, U = oracle_mod.quantization(W.numpy(), A.numpy(), X.numpy(), float(r["step"]), 8)
    assert np.array_equal(r["idx"].cpu().numpy().astype(np.int16), idx)
    assert np.array_equal(r["Q"].cpu().numpy().view(np.uint32), Q.view(np.uint32))
    assert np.array_equal(r["U"].cpu().numpy(), U)
    st = _run(W, A, X, m, 1, step=step)
    assert torch.equal(st["idx"], r["idx"]) and torch.equal(st["U"], r["U"]) and torch.equal(st["usq_seg"], r["usq_seg"])
    # device-scope publishing throughout (GPFQ_PIPE_LOCAL=0) gives the same bits
    monkeypatch.setenv("GPFQ_PIPE_LOCAL", "0")
    r2 = _run(W, A, X, m, 0, step=step)
    assert r2["timeouts"] == [] and torch.equal(r2["idx"], r["idx"]) and torch.equal(r2["U"], r["U"])


@pytest.mark.parametrize("mode", ["soft", "hard", "stochastic"])
def test_pipelined_kernels_other_quantizers_and_global_row_keys(oracle_mod, monkeypatch, mode):
    """soft / hard / stochastic through both row groupings and both reducer arrangements, in rounds: the stochastic
    quantizer's Philox key is the GLOBAL row number (oracle keyed the same way)."""
    from quantized_neural_nets_amd import _lib
    omode = {"soft": oracle_mod.MODE_SOFT, "hard": oracle_mod.MODE_HARD, "stochastic": oracle_mod.MODE_STOCHASTIC}[mode]
    lmode = {"soft": _lib.MODE_SOFT, "hard": _lib.MODE_HARD, "stochastic": _lib.MODE_STOCHASTIC}[mode]
    monkeypatch.setenv("GPFQ_COOP_PIPEL", "0")       # (the twelve-row family would take the first two in one round: tests/test_gpu_pipel.py)
    for (N, d, m) in ((130, 9, 93184), (300, 8, 51200), (64, 11, 93184)):
        desc = _lib.describe_plan(N, d, m, 1, 0, lmode)
        assert "pipe=1" in desc and desc == _lib.describe_plan(N, d, m, 1, 0, _lib.MODE_MSQ), desc
        W, A, X = bw.synthetic_layer(N, d, m, 17 + N, first_layer=False)
        step = bw.layer_step(W)
        r = _run(W, A, X, m, 0, mode=mode, seed=99, step=step)
        assert r["timeouts"] == []
        Q, idx, U = oracle_mod.quantization(W.numpy(), A.numpy(), X.numpy(), float(r["step"]), 8, mode=omode, lamb=0.05, seed=99)
        assert np.array_equal(r["idx"].cpu().numpy().astype(np.int16), idx)
        assert np.array_equal(r["U"].cpu().numpy(), U)


def test_pipelined_timeout_is_reported_and_the_layer_redone(oracle_mod, monkeypatch):
    """A gather that gives up (spin limit 0: the first unanswered look) raises the status word; the launch runs out, its
    outputs are dropped and the layer is redone on the whole-row streaming plan -- correct results, the timeout reported."""
    from quantized_neural_nets_amd import _lib
    N, d, m = 64, 12, 93184
    assert "pipe=1" in _lib.describe_plan(N, d, m)
    W, A, X = bw.synthetic_layer(N, d, m, 5, first_layer=False)
    step = bw.layer_step(W)
    monkeypatch.setenv("GPFQ_COOP_SPIN_LIMIT", "0")
    r = _run(W, A, X, m, 0, step=step)
    assert r["timeouts"] == [(N, d, m)]
    Q, idx, U = oracle_mod.quantization(W.numpy(), A.numpy(), X.numpy(), float(r["step"]), 8)
    assert np.array_equal(r["idx"].cpu().numpy().astype(np.int16), idx) and np.array_equal(r["U"].cpu().numpy(), U)
    monkeypatch.delenv("GPFQ_COOP_SPIN_LIMIT")
    assert _run(W, A, X, m, 0, step=step)["timeouts"] == []
    _lib.lib.gpfq_clear_contention()                     # (the forced timeout threw the device's launch-API switch)


def test_first_timeout_switches_the_device_to_the_cooperative_launch_api(oracle_mod, monkeypatch, capfd):
    """Contention fails fast without taxing the common case: an undisturbed process launches its cooperative grids plainly;
    the first timeout its status read reports on a device (here forced: spin limit 0) is logged once and switches every
    later cooperative launch on that device to hipLaunchCooperativeKernel -- same kernels, same bits -- until
    gpfq_clear_contention."""
    from quantized_neural_nets_amd import _lib
    L = _lib.lib
    monkeypatch.delenv("GPFQ_COOP_LAUNCH_API", raising=False)
    L.gpfq_clear_contention()
    assert L.gpfq_coop_launch_api_active() == 0
    results = {}
    for (N, d, m, want) in ((64, 12, 93184, "pipe=1"), (64, 9, 23296, "coop RT=2 C=8")):
        assert want in _lib.describe_plan(N, d, m)
        W, A, X = bw.synthetic_layer(N, d, m, 31 + N + d, first_layer=False)
        step = bw.layer_step(W)
        results[(N, d, m)] = (W, A, X, step, _run(W, A, X, m, 0, step=step))
        assert results[(N, d, m)][4]["timeouts"] == [] and L.gpfq_coop_launch_api_active() == 0
    # the first cooperative launch times out -> redone on the streaming plan, the switch is thrown and logged ONCE
    (N, d, m), (W, A, X, step, ref) = next(iter(results.items()))
    monkeypatch.setenv("GPFQ_COOP_SPIN_LIMIT", "0")
    capfd.readouterr()
    r = _run(W, A, X, m, 0, step=step)
    assert r["timeouts"] == [(N, d, m)] and torch.equal(r["idx"], ref["idx"]) and torch.equal(r["U"], ref["U"])
    r = _run(W, A, X, m, 0, step=step)                   # a second timeout: no second line
    assert r["timeouts"] == [(N, d, m)]
    monkeypatch.delenv("GPFQ_COOP_SPIN_LIMIT")
    assert capfd.readouterr().err.count("hipLaunchCooperativeKernel from now on") == 1
    assert L.gpfq_coop_launch_api_active() == 1
    # the next cooperative launches go through the cooperative API: same bits, both families, oracle-equal
    for (N, d, m), (W, A, X, step, ref) in results.items():
        r = _run(W, A, X, m, 0, step=step)
        assert r["timeouts"] == [] and torch.equal(r["idx"], ref["idx"]) and torch.equal(r["U"], ref["U"])
        Q, idx, U = oracle_mod.quantization(W.numpy(), A.numpy(), X.numpy(), float(r["step"]), 8)
        assert np.array_equal(r["idx"].cpu().numpy().astype(np.int16), idx) and np.array_equal(r["U"].cpu().numpy(), U)
    monkeypatch.setenv("GPFQ_COOP_LAUNCH_API", "-1")     # never
    assert L.gpfq_coop_launch_api_active() == 0
    monkeypatch.delenv("GPFQ_COOP_LAUNCH_API")
    L.gpfq_clear_contention()
    assert L.gpfq_coop_launch_api_active() == 0


def test_classic_lock_step_kernels_stay_selectable(monkeypatch):
    from quantized_neural_nets_amd import _lib
    monkeypatch.setenv("GPFQ_COOP_PIPE", "0")
    assert _lib.describe_plan(64, 576, 93184).startswith("coop RT=2 C=8 waves=12 S=91 grid=256 d=")
    assert _lib.describe_plan(1024, 256, 51200).startswith("coop RT=4 C=4 waves=13 S=50 grid=256 rounds=4 d=")


def test_runtime_enforced_coresidency_launch_gives_the_same_bits(monkeypatch):
    """GPFQ_COOP_LAUNCH_API=1: grids whose workgroups wait for each other go through hipLaunchCooperativeKernel (the runtime
    refuses a grid that cannot be co-resident instead of letting it spin) -- same kernels, same outputs, on the pipelined
    and on the lock-step family."""
    from quantized_neural_nets_amd import _lib
    for (N, d, m, want) in ((64, 9, 93184, "pipe=1"), (64, 9, 23296, "coop RT=2 C=8")):
        assert want in _lib.describe_plan(N, d, m) and ("pipe=1" in want or "pipe=1" not in _lib.describe_plan(N, d, m))
        W, A, X = bw.synthetic_layer(N, d, m, 8 + N, first_layer=False)
        step = bw.layer_step(W)
        monkeypatch.delenv("GPFQ_COOP_LAUNCH_API", raising=False)
        ref = _run(W, A, X, m, 0, step=step)
        monkeypatch.setenv("GPFQ_COOP_LAUNCH_API", "1")
        r = _run(W, A, X, m, 0, step=step)
        assert r["timeouts"] == [] and torch.equal(r["idx"], ref["idx"]) and torch.equal(r["U"], ref["U"])
