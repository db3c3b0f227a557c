"""XCD-local publishing in the LOCK-STEP cooperative kernels (gpfq_loop_kernels.h reducer_section, round 4): once step 0's
granules have shown every member of a row tile on this workgroup's XCD, the later steps publish with plain stores (the
XCD's L2 is the members' coherence point) instead of device-scope write-throughs.  Speed only: every configuration must give
the bits of the CPU oracle (reference step_algorithm.py:107-148) and of the same launch with device-scope publishing
throughout (GPFQ_COOP_LOCAL=0), and a tile whose members are NOT on one XCD must never switch."""
import numpy as np
import pytest
import torch

import bench_workload as bw

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
LOCK = {"GPFQ_COOP_PIPE": "0"}


def _run(W, A, X, m, plan, mode="msq", seed=None, K=8, step=None):
    from quantized_neural_nets_amd import StepAlgorithm as SA
    reg = {"msq": None, "soft": "L1", "hard": "L0"}.get(mode)
    r = SA._quantize_layer_ex(W.to(DEV), A.to(DEV), X.to(DEV), m, 1.16 / 8, K, 1, reg, 0.05, 1, mode == "stochastic",
                              torch.device(DEV), step_override=step, plan=plan, seed=seed, compute_errors=False)
    torch.cuda.synchronize()
    return r


# (N, d, m), forced configuration on top of GPFQ_COOP_PIPE=0, the plan that must result, what the case is there for
CASES = [
    ((128, 12, 26624), {}, "coop RT=2 C=4 waves=7 S=26 grid=256", "layer2.{1,2,3}.conv2's plan: 64 tiles of four members, eight tiles per XCD"),
    ((256, 10, 26624), {}, "coop RT=4 C=4 waves=7 S=26 grid=256", "four rows per tile"),
    ((64, 9, 93184), {}, "coop RT=2 C=8 waves=12 S=91 grid=256", "twelve sweep waves: wave 0 doubles as the reducer"),
    ((128, 9, 93184), {}, "coop RT=4 C=8 waves=12 S=91 grid=256", "the four-row twelve-wave variant (ABORTWORD protocol: never local, epoch word unchanged)"),
    ((64, 10, 26624), {"GPFQ_COOP_RT": "1", "GPFQ_COOP_C": "4"}, "coop RT=1 C=4 waves=7 S=26 grid=256", "one row per tile"),
    ((16, 10, 93184), {"GPFQ_COOP_RT": "2", "GPFQ_COOP_C": "32"}, "coop RT=2 C=32", "32 members: a whole XCD per tile, 64 granules per gather"),
    ((24, 12, 26624), {"GPFQ_COOP_RT": "2", "GPFQ_COOP_C": "4"}, "coop RT=2 C=4 waves=7 S=26 grid=48", "12 tiles: not a multiple of eight, no placement, device scope throughout"),
    ((128, 12, 26624), {"GPFQ_COOP_XCD_TILES": "0"}, "coop RT=2 C=4 waves=7 S=26 grid=256", "placement off: members spread over the XCDs, local publishing never offered"),
    ((16, 1, 26624), {"GPFQ_COOP_RT": "2", "GPFQ_COOP_C": "4"}, "coop RT=2 C=4", "ONE column: only the device-scope step"),
    ((16, 2, 26624), {"GPFQ_COOP_RT": "2", "GPFQ_COOP_C": "4"}, "coop RT=2 C=4", "two columns: the first plain publish is the last step"),
    ((16, 131, 20000), {"GPFQ_COOP_RT": "2", "GPFQ_COOP_C": "4"}, "coop RT=2 C=4", "three Q / idx history flushes, both parities of the exchange buffer many times over"),
    ((300, 8, 26624), {"GPFQ_COOP_RT": "2", "GPFQ_COOP_C": "4"}, "coop RT=2 C=4 waves=7 S=26 grid=256 rounds=3", "rounds: every launch establishes locality again; the last round partial (22 tiles: no placement)"),
]


@pytest.mark.parametrize("shape,env,plan_desc,why", CASES, ids=["%dx%dx%d_%s" % (c[0] + ("_".join(c[2].split()[1:3]),)) + "_%d" % i for i, c in enumerate(CASES)])
def test_lock_step_local_publishing_equals_oracle_and_device_scope(oracle_mod, monkeypatch, shape, env, plan_desc, why):
    from quantized_neural_nets_amd import _lib
    N, d, m = shape
    for k, v in {**LOCK, **env}.items():
        monkeypatch.setenv(k, v)
    desc = _lib.describe_plan(N, d, m)
    assert desc.startswith(plan_desc) and "pipe=1" not in desc, desc
    W, A, X = bw.synthetic_layer(N, d, m, 777 + N + d, first_layer=False)
    step = bw.layer_step(W)
    r = _run(W, A, X, m, 0, step=step)
    assert r["timeouts"] == []
    Q, idx, U = oracle_mod.quantization(W.numpy(), A.numpy(), X.numpy(), float(r["step"]), 8)
    assert np.array_equal(r["idx"].cpu().numpy().astype(np.int16), idx)
    assert np.array_equal(r["Q"].cpu().numpy().view(np.uint32), Q.view(np.uint32))
    assert np.array_equal(r["U"].cpu().numpy(), U)
    monkeypatch.setenv("GPFQ_COOP_LOCAL", "0")
    r2 = _run(W, A, X, m, 0, step=step)
    assert r2["timeouts"] == [] and torch.equal(r2["idx"], r["idx"]) and torch.equal(r2["U"], r["U"]) and torch.equal(r2["usq_seg"], r["usq_seg"])


@pytest.mark.parametrize("mode", ["soft", "hard", "stochastic"])
def test_lock_step_local_publishing_other_quantizers(oracle_mod, monkeypatch, mode):
    for k, v in LOCK.items():
        monkeypatch.setenv(k, v)
    N, d, m = 128, 14, 26624
    W, A, X = bw.synthetic_layer(N, d, m, 31337, first_layer=False)
    step = bw.layer_step(W)
    r = _run(W, A, X, m, 0, mode=mode, seed=99, step=step)
    assert r["timeouts"] == []
    st = _run(W, A, X, m, 1, mode=mode, seed=99, step=step)
    assert torch.equal(st["idx"], r["idx"]) and torch.equal(st["U"], r["U"])
    monkeypatch.setenv("GPFQ_COOP_LOCAL", "0")
    r2 = _run(W, A, X, m, 0, mode=mode, seed=99, step=step)
    assert torch.equal(r2["idx"], r["idx"]) and torch.equal(r2["U"], r["U"])


def test_lock_step_forced_timeout_still_reported_and_redone(monkeypatch):
    """A poll limit of zero: an exchange whose first look is not answered gives up, the status word is raised and the layer is
    redone on the streaming plan; either way the result is exact -- with local publishing offered (the switch must not mask a
    timed-out step 0, and a launch that has given up keeps whatever scope it had)."""
    from quantized_neural_nets_amd import _lib
    for k, v in LOCK.items():
        monkeypatch.setenv(k, v)
    N, d, m = 128, 6, 26624
    W, A, X = bw.synthetic_layer(N, d, m, 5, first_layer=False)
    step = bw.layer_step(W)
    good = _run(W, A, X, m, 1, step=step)
    monkeypatch.setenv("GPFQ_COOP_SPIN_LIMIT", "0")
    r = _run(W, A, X, m, 0, step=step)
    assert torch.equal(r["idx"], good["idx"]) and torch.equal(r["U"], good["U"])
    assert _lib.describe_plan(N, d, m).startswith("coop RT=2 C=4")
