"""The pipelined cooperative kernels with LDS-staged columns and twelve rows per workgroup (gpfq_pipel_kernels.h, round 5):
three row groups of four rows, three phases per step, seven sweep waves + one reducer wave, five column buffers per sweep wave
in LDS, a coalesced gather of up to 128 members.  Same recurrence (reference step_algorithm.py:107-148), same canonical
arithmetic: every configuration must reproduce the CPU oracle -- and the streaming plan -- bit for bit: indices, Q (sign of
zero included), U, and the per-segment sums of squares of the epilogue."""
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


FORCE = {"GPFQ_COOP_PIPEL": "1"}
# (N, d, m), extra environment, the plan that must result, what the case is there for
CASES = [
    ((24, 6, 803840), {}, "coop RT=12 C=128 waves=7 S=785 grid=256 pipel=1",
     "ResNet-50's 56 x 56 maps: 128 members, FOUR granules of a row pair per lane of the gather, members on four XCDs"),
    ((60, 5, 803840), {}, "coop RT=12 C=128 waves=7 S=785 grid=256 rounds=3 pipel=1", "three rounds, the last with one tile"),
    ((30, 10, 201728), {"GPFQ_COOP_C": "32"}, "coop RT=12 C=32 waves=7 S=197 grid=96 pipel=1", "32 members: one load per lane; the last tile has 6 valid rows of 12"),
    ((100, 12, 51200), {"GPFQ_COOP_C": "8"}, "coop RT=12 C=8 waves=7 S=50 grid=72 pipel=1", "8 members (a quarter of the gather's lanes); the last tile has 4 valid rows"),
    ((400, 9, 51200), {}, "coop RT=12 C=8 waves=7 S=50 grid=256 rounds=2 pipel=1", "two rounds of 32 and 2 tiles"),
    ((1536, 8, 13312), {}, "coop RT=12 C=2 waves=7 S=13 grid=256 pipel=1", "two members per tile: 6 and 7 segments, a sweep wave idles in one of them"),
    ((12, 5, 425984), {"GPFQ_COOP_C": "64"}, "coop RT=12 C=64 waves=7 S=416 grid=64 pipel=1", "64 members: two granules of a row pair per lane"),
    ((24, 9, 102400), {"GPFQ_COOP_C": "16"}, "coop RT=12 C=16 waves=7 S=100 grid=32 pipel=1", "16 members"),
    ((36, 9, 26624), {"GPFQ_COOP_C": "4"}, "coop RT=12 C=4 waves=7 S=26 grid=12 pipel=1", "4 members of 6 / 7 segments"),
    ((36, 9, 9216), {"GPFQ_COOP_C": "2", "plan": "3"}, "coop RT=12 C=2 waves=5 S=9 grid=6 pipel=1", "five sweep waves, members of 4 and 5 segments"),
    ((13, 9, 3072), {"GPFQ_COOP_C": "2", "plan": "3"}, "coop RT=12 C=2 waves=2 S=3 grid=4 pipel=1", "two sweep waves, members of 1 and 2 segments; 1 valid row in the last tile"),
    ((24, 1, 51200), {"GPFQ_COOP_C": "8"}, "coop RT=12 C=8 waves=7 S=50 grid=16 pipel=1", "ONE column: the pipeline is all fill and drain"),
    ((24, 2, 51200), {"GPFQ_COOP_C": "8"}, "coop RT=12 C=8 waves=7 S=50 grid=16 pipel=1", "two columns"),
    ((24, 7, 51200), {"GPFQ_COOP_C": "8"}, "coop RT=12 C=8 waves=7 S=50 grid=16 pipel=1", "seven columns: both buffer rings past one period"),
    ((12, 131, 20000), {"GPFQ_COOP_C": "4"}, "coop RT=12 C=4 waves=5 S=20 grid=4 pipel=1", "three Q / idx history flushes, the last partial"),
]


@pytest.mark.parametrize("shape,env,plan_desc,why", CASES, ids=["%dx%dx%d_%s" % (c[0] + ("_".join(c[2].split()[2:4]),)) for c in CASES])
def test_lds_staged_pipelined_kernels_equal_oracle_and_streaming(oracle_mod, monkeypatch, shape, env, plan_desc, why):
    from quantized_neural_nets_amd import _lib
    N, d, m = shape
    env = dict(env)
    plan = int(env.pop("plan", "0"))                 # (3 = the cooperative family asked for: rows short enough for the resident plan)
    for k, v in dict(FORCE, **env).items():
        monkeypatch.setenv(k, v)
    assert _lib.describe_plan(N, d, m, 1, plan).startswith(plan_desc), _lib.describe_plan(N, d, m, 1, plan)
    W, A, X = bw.synthetic_layer(N, d, m, 777 + N + d, first_layer=False)
    step = bw.layer_step(W)
    r = _run(W, A, X, m, plan, step=step)
    assert r["timeouts"] == []
    Q, idx, U = oracle_mod.quantization(W.numpy(), A.numpy(), X.numpy(), float(r["step"]), 8)
    assert np.array_equal(r["idx"].cpu().numpy().astype(np.int16), idx)
    assert np.array_equal(r["Q"].cpu().numpy().view(np.uint32), Q.view(np.uint32))
    assert np.array_equal(r["U"].cpu().numpy(), U)
    st = _run(W, A, X, m, 1, step=step)
    assert torch.equal(st["idx"], r["idx"]) and torch.equal(st["U"], r["U"]) and torch.equal(st["usq_seg"], r["usq_seg"])
    # and a second run of the same layer (another launch number in the epoch words, granules of the first still in the scratch)
    r2 = _run(W, A, X, m, plan, step=step)
    assert r2["timeouts"] == [] and torch.equal(r2["idx"], r["idx"]) and torch.equal(r2["U"], r["U"])


@pytest.mark.parametrize("mode", ["soft", "hard", "stochastic"])
def test_lds_staged_pipelined_kernels_other_quantizers_and_global_row_keys(oracle_mod, monkeypatch, mode):
    """soft / hard / stochastic, in rounds too: the stochastic quantizer's Philox key is the GLOBAL row number."""
    from quantized_neural_nets_amd import _lib
    monkeypatch.setenv("GPFQ_COOP_PIPEL", "1")
    omode = {"soft": oracle_mod.MODE_SOFT, "hard": oracle_mod.MODE_HARD, "stochastic": oracle_mod.MODE_STOCHASTIC}[mode]
    lmode = {"soft": _lib.MODE_SOFT, "hard": _lib.MODE_HARD, "stochastic": _lib.MODE_STOCHASTIC}[mode]
    for (N, d, m) in ((400, 8, 51200), (80, 9, 201728), (26, 5, 803840)):
        desc = _lib.describe_plan(N, d, m, 1, 0, lmode)
        assert "pipel=1" in desc, desc
        W, A, X = bw.synthetic_layer(N, d, m, 29 + N, first_layer=False)
        step = bw.layer_step(W)
        r = _run(W, A, X, m, 0, mode=mode, seed=77, step=step)
        assert r["timeouts"] == []
        Q, idx, U = oracle_mod.quantization(W.numpy(), A.numpy(), X.numpy(), float(r["step"]), 8, mode=omode, lamb=0.05, seed=77)
        assert np.array_equal(r["idx"].cpu().numpy().astype(np.int16), idx)
        assert np.array_equal(r["U"].cpu().numpy(), U)


def test_lds_staged_pipelined_timeout_is_reported_and_the_layer_redone(oracle_mod, monkeypatch):
    """A gather that gives up (spin limit 0) raises the status word; the launch runs out, its outputs are dropped and the layer
    is redone on the whole-row streaming plan -- correct results, the timeout reported."""
    from quantized_neural_nets_amd import _lib
    monkeypatch.setenv("GPFQ_COOP_PIPEL", "1")
    N, d, m = 48, 12, 51200
    monkeypatch.setenv("GPFQ_COOP_C", "8")
    assert "pipel=1" in _lib.describe_plan(N, d, m)
    W, A, X = bw.synthetic_layer(N, d, m, 5, first_layer=False)
    step = bw.layer_step(W)
    monkeypatch.setenv("GPFQ_COOP_SPIN_LIMIT", "0")
    r = _run(W, A, X, m, 0, step=step)
    assert r["timeouts"] == [(N, d, m)]
    Q, idx, U = oracle_mod.quantization(W.numpy(), A.numpy(), X.numpy(), float(r["step"]), 8)
    assert np.array_equal(r["idx"].cpu().numpy().astype(np.int16), idx) and np.array_equal(r["U"].cpu().numpy(), U)
    monkeypatch.delenv("GPFQ_COOP_SPIN_LIMIT")
    assert _run(W, A, X, m, 0, step=step)["timeouts"] == []


def test_the_other_cooperative_families_stay_selectable(monkeypatch):
    from quantized_neural_nets_amd import _lib
    monkeypatch.setenv("GPFQ_COOP_PIPEL", "0")
    assert "pipel" not in _lib.describe_plan(256, 64, 803840) and _lib.describe_plan(256, 64, 803840).startswith("coop RT=4 C=64 waves=13")
    assert "pipe=1" in _lib.describe_plan(1024, 256, 51200)
