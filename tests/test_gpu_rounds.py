"""(Round 4: AUTO gives many of these shapes to the pipelined kernels, tests/test_gpu_pipe.py; the cases below that are about
a particular LOCK-STEP kernel pin that family with GPFQ_COOP_PIPE=0 -- those kernels remain the choice wherever a tile has
more than 64 granules per gather, more than seven segments per member, one or two rows, or one row per group.)
Long rows with more rows than the chip holds at once: the cooperative plan in ROUNDS (one co-resident launch per block
of rows, the residual of the block in registers for all d columns) against the CPU oracle and against the streaming
plan, bit for bit -- including the gather of more than 64 granules (two members per lane) and the two-row 16-wave
variant that m = 803 840 (785 segments, 12.3 per member at 64 members) needs.  Shapes are the 1x1 convolutions of
ResNet-50 / EfficientNet-B1 / VGG-16 at their calibration batches (BASELINE.json configs 2-4) with d cut short."""
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
    assert r["timeouts"] == []
    return r


# (N, d, m), forced configuration ({} = what AUTO picks), the plan that must result, what the case is there for
CASES = [
    ((300, 24, 51200), {"GPFQ_COOP_RT": "4", "GPFQ_COOP_C": "8", "GPFQ_COOP_PIPEL": "0"}, "coop RT=4 C=8 waves=7 S=50 grid=256 rounds=3",
     "three rounds, the last one partial (44 rows)"),
    ((300, 24, 51200), {"GPFQ_COOP_RT": "2", "GPFQ_COOP_C": "4"}, "coop RT=2 C=4 waves=13 S=50 grid=256 rounds=3",
     "two rows x 13 sweep waves: the 16-wave variant"),
    ((70, 16, 201728), {"GPFQ_COOP_RT": "4", "GPFQ_COOP_C": "32"}, "coop RT=4 C=32 waves=7 S=197 grid=256 rounds=3",
     "128 granules: two gathered members per lane; the last tile has 2 valid rows"),
    ((21, 10, 803840), {"GPFQ_COOP_RT": "2"}, "coop RT=2 C=64 waves=13 S=785 grid=256 rounds=3", "16-wave variant AND 128 granules, odd row count"),
    ((300, 24, 51200), {"GPFQ_COOP_PIPE": "0"}, "coop RT=4 C=4 waves=13 S=50 grid=256 rounds=2",
     "four rows x 13 sweep waves, columns staged through LDS (global_load_lds); 256 rows per round, the last round partial"),
    ((70, 16, 201728), {"GPFQ_COOP_PIPE": "0"}, "coop RT=4 C=16 waves=13 S=197 grid=256 rounds=2", "the same variant with 16 members (64 granules)"),
    ((21, 10, 803840), {"GPFQ_COOP_PIPEL": "0"}, "coop RT=4 C=64 waves=13 S=785 grid=256 rounds=2",
     "the same variant with 64 members: 256 granules gathered four per lane; 16 rows per round, 5 in the last"),
    ((20, 10, 720384), {"GPFQ_COOP_RT": "2"}, "coop RT=2 C=64 waves=11 S=704 grid=256 rounds=3", "128 granules at 64 members (VGG-16 conv1 rows)"),
    ((20, 10, 720384), {"GPFQ_COOP_PIPEL": "0"}, "coop RT=4 C=64 waves=11 S=704 grid=256 rounds=2",
     "the LDS-staged 64-member kernel with fewer than 13 sweep waves (VGG-16 conv1 rows: 11 segments per member)"),
    ((9, 6, 530000), {"GPFQ_COOP_PIPEL": "0"}, "coop RT=4 C=64 waves=9 S=518 grid=192",
     "the same with 9 sweep waves, members of 8 and 9 segments, one round with a partial last tile"),
    ((70, 12, 263168), {"GPFQ_COOP_PIPE": "0"}, "coop RT=4 C=32 waves=9 S=257 grid=256 rounds=3", "four rows x 9 sweep waves, one step of look-ahead, 128 granules"),
    ((12, 6, 1440768), {}, "coop RT=2 C=128 waves=11 S=1407 grid=256 rounds=3",
     "two rows on 128 members with 11 sweep waves: 256 granules, eight gathered per lane in 16 lanes per row"),
    ((3, 4, 1000000), {}, "coop RT=2 C=128 waves=8 S=977 grid=256",
     "the same kernel with 8 sweep waves, one round, the second tile with one valid row"),
    ((12, 6, 1440768), {"GPFQ_COOP_RT": "1"}, "coop RT=1 C=128 waves=11 S=1407 grid=256 rounds=6",
     "128 members of one row (EfficientNet-B1's first conv at batch 1024): 64 lanes x 2 granules, two rows per round"),
    ((3, 4, 3212288), {"GPFQ_COOP_RT": "1"}, "coop RT=1 C=256 waves=13 S=3137 grid=256 rounds=3",
     "one row on the whole chip (EfficientNet-B1's 112 x 112 maps): 256 members, four gathered per lane, 13 sweep waves"),
    ((5, 4, 3212288), {"GPFQ_COOP_RT": "2", "GPFQ_COOP_C": "256", "plan": "3"}, "coop RT=2 C=256 waves=13 S=3137 grid=256 rounds=3",
     "two rows on the whole chip: 512 granules, eight gathered per lane in two batches; the last tile has one valid row"),
    ((9, 3, 3212288), {}, "coop RT=4 C=256 waves=13 S=3137 grid=256 rounds=3",
     "FOUR rows on the whole chip (round 3): 1024 granules, sixteen gathered per lane in four batches, columns through LDS; "
     "the last tile has one valid row"),
    ((6, 5, 1100000), {"GPFQ_COOP_RT": "4", "GPFQ_COOP_C": "256", "plan": "3"}, "coop RT=4 C=256 waves=5 S=1075 grid=256 rounds=2",
     "the same variant with five sweep waves per member and an idle upper half of the slot tree (1075 of 2048 slots)"),
    ((5, 4, 300000), {"GPFQ_COOP_RT": "1", "GPFQ_COOP_C": "256", "plan": "3"}, "coop RT=1 C=256 waves=2 S=293 grid=256 rounds=5",
     "the same variant with two sweep waves per member and an idle upper half of the slot tree (293 of 512 slots)"),
]


@pytest.mark.parametrize("shape,env,plan_desc,why", CASES, ids=["%s_S%d" % (c[2].split(" waves")[0].replace(" ", "_"), c[0][2] // 1024) for c in CASES])
def test_rounds_equal_oracle_and_streaming(oracle_mod, monkeypatch, shape, env, plan_desc, why):
    from quantized_neural_nets_amd import _lib
    N, d, m = shape
    env = dict(env)
    plan = int(env.pop("plan", "0"))                 # 3: a configuration AUTO would not pick (it models streaming cheaper)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    assert _lib.describe_plan(N, d, m, 1, plan).startswith(plan_desc), _lib.describe_plan(N, d, m, 1, plan)
    W, A, X = bw.synthetic_layer(N, d, m, 777 + N, first_layer=False)
    step = bw.layer_step(W)
    r = _run(W, A, X, m, plan, step=step)
    Q, idx, U = oracle_mod.quantization(W.numpy(), A.numpy(), X.numpy(), float(r["step"]), 8)
    assert np.array_equal(r["idx"].cpu().numpy().astype(np.int16), idx)
    assert np.array_equal(r["Q"].cpu().numpy(), Q)
    assert np.array_equal(r["U"].cpu().numpy(), U)
    st = _run(W, A, X, m, 1, step=step)
    assert torch.equal(st["idx"], r["idx"]) and torch.equal(st["U"], r["U"]) and torch.equal(st["usq_seg"], r["usq_seg"])
    ref = (r["U"].double() ** 2).sum(1)
    assert torch.allclose(r["usq_seg"].double().sum(1), ref, rtol=1e-5)       # the epilogue's rows land at their own offsets


@pytest.mark.parametrize("mode", ["soft", "hard", "stochastic"])
def test_rounds_other_quantizers_and_global_row_keys(oracle_mod, monkeypatch, mode):
    """The other three quantizers through rounds (the 16-wave two-row variant and the four-row one); the stochastic
    quantizer's Philox key is the GLOBAL row number, so the rows of a later round must not repeat the first round's
    draws: equality with the oracle (keyed by global rows) and with the streaming plan shows it."""
    from quantized_neural_nets_amd import _lib
    omode = {"soft": oracle_mod.MODE_SOFT, "hard": oracle_mod.MODE_HARD, "stochastic": oracle_mod.MODE_STOCHASTIC}[mode]
    lmode = {"soft": _lib.MODE_SOFT, "hard": _lib.MODE_HARD, "stochastic": _lib.MODE_STOCHASTIC}[mode]
    monkeypatch.setenv("GPFQ_COOP_PIPEL", "0")       # (the twelve-row family takes the first two shapes in one round: tests/test_gpu_pipel.py)
    for (N, d, m) in ((21, 6, 803840), (260, 8, 51200), (9, 3, 3212288)):   # (the last: four rows x 256 members, three rounds)
        # every (rows, waves) pair is instantiated for all four quantizers (the stochastic forms of the four-row 12-wave
        # and of the 256-member two-row kernel fit their register budgets since round 3): the plan does not depend on the
        # quantizer, and what is described is what launches
        desc = _lib.describe_plan(N, d, m, 1, 0, lmode)
        assert "rounds=" in desc and desc == _lib.describe_plan(N, d, m, 1, 0, _lib.MODE_MSQ), desc
        W, A, X = bw.synthetic_layer(N, d, m, 91 + N, first_layer=False)
        step = bw.layer_step(W)
        r = _run(W, A, X, m, 0, mode=mode, seed=4321, step=step)
        Q, idx, U = oracle_mod.quantization(W.numpy(), A.numpy(), X.numpy(), float(r["step"]), 8, mode=omode, lamb=0.05, seed=4321)
        assert np.array_equal(r["idx"].cpu().numpy().astype(np.int16), idx)
        assert np.array_equal(r["U"].cpu().numpy(), U)
        st = _run(W, A, X, m, 1, mode=mode, seed=4321, step=step)
        assert torch.equal(st["idx"], r["idx"]) and torch.equal(st["U"], r["U"])


def test_stochastic_forms_of_the_four_row_12_wave_and_the_256_member_two_row_kernels(oracle_mod, monkeypatch):
    """Round 3 instantiated the two variants that had no stochastic form (their Philox rounds did not fit the register
    budget before the rare blocks stopped keeping per-lane addresses live): four rows at 9..12 sweep waves
    (gpfq_coop_rt4_m3_w12) and two rows on 256 members (gpfq_coop_rt2_m3_w16o), forced here, against the oracle's Philox
    stream keyed by global rows and against the streaming plan."""
    from quantized_neural_nets_amd import _lib
    for (N, d, m), env, want in (((22, 7, 40000), {"GPFQ_COOP_RT": "4", "GPFQ_COOP_C": "4"}, "coop RT=4 C=4 waves=10 S=40"),
                                 ((5, 3, 3212288), {"GPFQ_COOP_RT": "2", "GPFQ_COOP_C": "256"}, "coop RT=2 C=256 waves=13 S=3137")):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        desc = _lib.describe_plan(N, d, m, 1, 3, _lib.MODE_STOCHASTIC)
        assert desc.startswith(want), desc
        W, A, X = bw.synthetic_layer(N, d, m, 303 + N, first_layer=False)
        step = bw.layer_step(W)
        r = _run(W, A, X, m, 3, mode="stochastic", seed=99, step=step)
        Q, idx, U = oracle_mod.quantization(W.numpy(), A.numpy(), X.numpy(), float(r["step"]), 8, mode=oracle_mod.MODE_STOCHASTIC,
                                            lamb=0.05, seed=99)
        assert np.array_equal(r["idx"].cpu().numpy().astype(np.int16), idx)
        assert np.array_equal(r["U"].cpu().numpy(), U)
        for k in env:
            monkeypatch.delenv(k)
        st = _run(W, A, X, m, 1, mode="stochastic", seed=99, step=step)
        assert torch.equal(st["idx"], r["idx"]) and torch.equal(st["U"], r["U"])


def test_rounds_int16_indices_and_row_shard():
    """8-bit alphabets store int16 indices (the round's index pointer advances in bytes), and quantizing a block of rows
    alone gives the slice of the full result whatever the round boundaries are (the neuron-shard property)."""
    N, d, m = 300, 12, 51200
    W, A, X = bw.synthetic_layer(N, d, m, 5150, first_layer=False)
    step = bw.layer_step(W) / 16.0
    full = _run(W, A, X, m, 0, K=128, step=step)
    assert full["idx"].dtype == torch.int16 and int(full["idx"].abs().max()) > 8
    st = _run(W, A, X, m, 1, K=128, step=step)
    assert torch.equal(st["idx"], full["idx"]) and torch.equal(st["U"], full["U"])
    part = _run(W[101:259].contiguous(), A, X, m, 0, K=128, step=step)
    assert torch.equal(part["idx"], full["idx"][101:259]) and torch.equal(part["U"], full["U"][101:259])


def test_rounds_stop_after_a_timeout_and_the_layer_is_redone(monkeypatch):
    """A spin limit of 0 makes the first round's exchange give up; the later rounds must return at once (status word
    raised), and the driver-level entry redoes the layer on the whole-row streaming plan: same bits as streaming."""
    from quantized_neural_nets_amd import StepAlgorithm as SA
    N, d, m = 300, 8, 51200
    W, A, X = bw.synthetic_layer(N, d, m, 616, first_layer=False)
    step = bw.layer_step(W)
    ref = _run(W, A, X, m, 1, step=step)
    monkeypatch.setenv("GPFQ_COOP_SPIN_LIMIT", "0")
    r = SA._quantize_layer_ex(W.to(DEV), A.to(DEV), X.to(DEV), m, 1.16 / 8, 8, 1, None, 0.05, 1, False,
                              torch.device(DEV), step_override=step, plan=0, compute_errors=False)
    torch.cuda.synchronize()
    assert r["timeouts"], "the forced timeout was not reported"
    assert torch.equal(r["idx"], ref["idx"]) and torch.equal(r["U"], ref["U"])


def test_first_poll_pause_changes_no_bit(monkeypatch):
    """The pause before the first poll of an exchange (gpfq_capi.hip first_poll_pause, by members; GPFQ_COOP_POLL_DELAY
    overrides it, units of 256 clocks, 0 .. 31) is timing only: none, the table's and the longest give the same bits --
    on the register-window kernel and on the LDS-staged one."""
    from quantized_neural_nets_amd import StepAlgorithm as SA, _lib
    monkeypatch.setenv("GPFQ_COOP_PIPE", "0")        # (the lock-step kernels: the pipelined ones have no such pause)
    for (N, d, m, want) in ((24, 6, 60000, "coop RT="), (300, 5, 51200, "coop RT=4 C=4 waves=13")):
        assert _lib.describe_plan(N, d, m).startswith(want), _lib.describe_plan(N, d, m)
        W, A, X = bw.synthetic_layer(N, d, m, 77 + N, first_layer=False)
        step = bw.layer_step(W)
        outs = []
        for delay in (None, "0", "31"):
            if delay is None:
                monkeypatch.delenv("GPFQ_COOP_POLL_DELAY", raising=False)
            else:
                monkeypatch.setenv("GPFQ_COOP_POLL_DELAY", delay)
            r = SA._quantize_layer_ex(W.to(DEV), A.to(DEV), X.to(DEV), m, 1.16 / 8, 8, 1, None, 0.05, 1, False,
                                      torch.device(DEV), step_override=step, plan=0, compute_errors=False)
            torch.cuda.synchronize()
            assert not r["timeouts"]
            outs.append(r)
        for r in outs[1:]:
            assert torch.equal(r["idx"], outs[0]["idx"]) and torch.equal(r["Q"], outs[0]["Q"]) and torch.equal(r["U"], outs[0]["U"])


@pytest.mark.parametrize("mode", ["msq", "soft", "stochastic"])
def test_depthwise_long_rows_one_row_per_group(oracle_mod, mode):
    """Depthwise convolutions (groups == out channels, one row per group, every group with its own 9 columns) with long
    rows: the cooperative kernel's grouped variant in rounds of as many groups as fit the chip -- EfficientNet-B1's
    features.2.0.block.1.0 at batch 1024 is 96 groups of 370 688 samples.  Against the oracle and the streaming plan,
    bit for bit; the stochastic quantizer's keys are the global rows (= groups)."""
    from quantized_neural_nets_amd import StepAlgorithm as SA, _lib
    omode = {"msq": oracle_mod.MODE_MSQ, "soft": oracle_mod.MODE_SOFT, "stochastic": oracle_mod.MODE_STOCHASTIC}[mode]
    for (G, dg, m, want) in ((19, 9, 370688, "coop RT=1 C=32 waves=12 S=362 grid=256 rounds=3 groups=19"),
                             (70, 5, 30000, "coop RT=1 C=4 waves=8 S=30 grid=256 rounds=2 groups=70")):
        assert _lib.describe_plan(G, dg, m, G).startswith(want), _lib.describe_plan(G, dg, m, G)
        W, A, X = bw.synthetic_layer(G, G * dg, m, 31 + G, first_layer=False)
        W = W[:, :dg].contiguous()
        reg = {"soft": "L1"}.get(mode)
        K = 2

        def run(plan):
            r = SA._quantize_layer_ex(W.to(DEV), A.to(DEV), X.to(DEV), m, 1.16 / K, K, 1, reg, 0.05, G, mode == "stochastic",
                                      torch.device(DEV), plan=plan, seed=99, compute_errors=False)
            torch.cuda.synchronize()
            assert r["timeouts"] == []
            return r
        r = run(0)
        o = oracle_mod.quantize_layer(W.numpy(), A.numpy(), X.numpy(), 1.16 / K, K, 1.0, reg, 0.05, G,
                                      **({"stochastic": True, "seed": 99} if mode == "stochastic" else {}))
        assert np.array_equal(r["idx"].cpu().numpy().astype(np.int16), o["idx"])
        assert np.array_equal(r["U"].cpu().numpy(), o["U"])
        st = run(1)
        assert torch.equal(st["idx"], r["idx"]) and torch.equal(st["U"], r["U"]) and torch.equal(st["usq_seg"], r["usq_seg"])
