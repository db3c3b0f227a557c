"""Parity of the HIP path (through the C ABI, via the Python mirror of the reference's operator surface)
against (1) the golden vectors made from the imported reference and (2) the CPU oracle.

Bar: alphabet indices bit-exact vs the reference fixtures; residual U within 1e-5 of the reference
(BASELINE.md 5); and, because the kernels and the oracle share one canonical reduction order, idx / Q / U
BIT-EXACT against the oracle on any seeded input.
"""
import json
import os

import numpy as np
import pytest
import torch

import golden_inputs as gi

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def qnn():
    import quantized_neural_nets_amd as q
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return q


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _run_layer(qnn, case, W, A, X, plan=0):
    from quantized_neural_nets_amd import StepAlgorithm, _lib
    K = 2 ** (case["bits"] - 1)
    if plan == 3:
        try:
            _lib.describe_plan(W.shape[0], W.shape[1], A.shape[0], case["groups"], 3)
        except _lib.GpfqError:
            pytest.skip("cooperative plan does not apply to this shape")
    r = StepAlgorithm._quantize_layer_ex(_t(W), _t(A), _t(X), A.shape[0], case["scalar"] / K, K,
                                         case["percentile"], case["reg"], case["lamb"], case["groups"], False,
                                         torch.device(DEV), plan=plan)
    torch.cuda.synchronize()
    return r


def test_quantizer_kernels_match_reference_vectors(qnn):
    from quantized_neural_nets_amd import StepAlgorithm as SA
    fx = np.load(os.path.join(gi.GOLDEN_DIR, "g1_quantizers.npz"))
    cfgs = json.loads(str(fx["meta"]))["configs"]
    for ci, c in enumerate(cfgs):
        x = _t(fx["x_%d" % ci])
        for name, fn in (("msq", SA._msq), ("soft", SA._soft_thresholding_msq), ("hard", SA._hard_thresholding_msq)):
            out = fn(c["step"], x, c["K"], c["lamb"]).cpu().numpy()
            ref = fx["%s_%d" % (name, ci)]
            assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), (name, ci)


def test_stochastic_quantizer_is_unbiased(qnn):
    from quantized_neural_nets_amd import StepAlgorithm as SA
    x = torch.full((200000,), 0.3, device=DEV)
    q = SA._stochastic_msq(0.25, x.clone(), 4, 0.0)
    assert set(torch.unique(q).cpu().tolist()) == {0.25, 0.5}
    assert abs(q.mean().item() - 0.3) < 2e-3
    q = SA._stochastic_msq(0.25, torch.tensor([5.0, -5.0], device=DEV), 4, 0.0)
    assert q.cpu().tolist() == [1.0, -1.0]


def test_stochastic_quantizer_against_reference_fixture(qnn, oracle_mod):
    """G7 (tools/make_golden.py gen_stochastic, from the reference's _stochastic_msq, step_algorithm.py:7-35):
    (i) where the answer does not depend on the draw the HIP quantizer returns the reference's bits; (ii) inside the
    alphabet its two possible answers are bitwise the reference's two, its draws are bitwise the oracle's Philox stream
    (so its round-down count IS the oracle's), and that count agrees with the reference's own frequency within a
    binomial bound (6 sigma of the difference of two samples)."""
    from quantized_neural_nets_amd import StepAlgorithm as SA
    from quantized_neural_nets_amd.step_algorithm import _elementwise
    from quantized_neural_nets_amd import _lib
    from test_oracle_golden import g7_expected_down_probability, g7_binomial_slack
    fx = np.load(os.path.join(gi.GOLDEN_DIR, "g7_stochastic.npz"))
    top = np.float32(1) - np.float32(2 ** -24)
    for ci, c in enumerate(json.loads(str(fx["meta"]))["configs"]):
        x, ref = fx["det_x_%d" % ci], fx["det_q_%d" % ci]
        for seed in (0, 123456789):
            out = SA._stochastic_msq(c["step"], _t(x.copy()), c["K"], 0.0, seed=seed, column=ci).cpu().numpy()
            assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), (ci, seed)
        x, lo, hi, ref_down = (fx["rnd_%s_%d" % (k, ci)] for k in ("x", "lo", "hi", "down_count"))
        n = c["draws"]
        for u, want in ((0.0, lo), (top, hi)):
            out = _elementwise(_lib.MODE_STOCHASTIC, c["step"], _t(x), c["K"], 0.0, _t(np.full(x.shape, u, np.float32))).cpu().numpy()
            assert np.array_equal(out.view(np.uint32), want.view(np.uint32)), (ci, u)
        p = g7_expected_down_probability(x, c["step"]).astype(np.float64)
        for j in range(0, len(x), 3):
            xs = torch.full((n,), float(x[j]), device=DEV, dtype=torch.float32)
            q = SA._stochastic_msq(c["step"], xs, c["K"], 0.0, seed=77 + ci, column=j, row_id0=0).cpu().numpy()
            un = oracle_mod.philox_uniform_vec(seed=77 + ci, row0=0, col=j, n=n)
            qo, _ = oracle_mod.quantizer_vec(oracle_mod.MODE_STOCHASTIC, c["step"], np.full(n, x[j], np.float32), c["K"], 0.0, uniform=un)
            assert np.array_equal(q.view(np.uint32), qo.view(np.uint32)), (ci, j)          # same stream, same bits
            down = int((q == lo[j]).sum())
            assert abs(down - int(ref_down[j])) <= np.sqrt(2.0) * g7_binomial_slack(n, p[j]), (ci, j, down, int(ref_down[j]))


@pytest.mark.parametrize("name", gi.available_cases())
@pytest.mark.parametrize("plan", [0, 1, 3])
def test_layer_against_reference_and_oracle(qnn, oracle_mod, name, plan):
    case, (W, A, X), fx, meta = gi.load_case(name)
    r = _run_layer(qnn, case, W, A, X, plan)
    K = 2 ** (case["bits"] - 1)
    assert np.float32(float(r["step"])) == fx["step"]
    idx = r["idx"].cpu().numpy()
    Q = r["Q"].cpu().numpy()
    U = r["U"].cpu().numpy()
    # (1) the reference's own outputs
    assert np.array_equal(idx.astype(np.int16), fx["idx"]), "alphabet indices differ from the reference"
    assert np.array_equal(Q, fx["Q"])
    assert np.abs(U - fx["U"]).max() <= 1e-5
    assert abs(float(r["quantize_error"]) - float(fx["quantize_error"])) <= 1e-4 * float(fx["quantize_error"])
    assert abs(float(r["relative_quantize_error"]) - float(fx["relative_quantize_error"])) <= 1e-4 * float(
        fx["relative_quantize_error"])
    if case["groups"] == 1:
        assert np.allclose(r["relative_adder"].cpu().numpy(), fx["relative_adder"], rtol=1e-4, atol=1e-6)
        assert tuple(r["quantize_adder"].shape) == (A.shape[0], W.shape[0])
    else:
        assert r["quantize_adder"] is None and r["relative_adder"] is None
    # (2) the oracle, bit for bit
    o = oracle_mod.quantize_layer(W, A, X, case["scalar"] / K, K, case["percentile"], case["reg"], case["lamb"],
                                  case["groups"])
    assert np.array_equal(idx.astype(np.int16), o["idx"])
    assert np.array_equal(Q.view(np.uint32), o["Q"].view(np.uint32))
    assert np.array_equal(U, o["U"])


RANDOM_SHAPES = [
    # N, d, m, groups, mode, bits
    (48, 40, 7168, 1, "msq", 4),      # resident, 7 waves
    (20, 24, 16384, 1, "soft", 2),    # resident, 16 waves (largest)
    (12, 30, 17000, 1, "msq", 4),     # stream, 17 segments, ragged tail
    (5, 12, 70000, 1, "hard", 3),     # stream, > 64 segments (second-level lanes wrap)
    (1024, 16, 300, 1, "msq", 4),     # many rows, single segment
    (1030, 8, 2100, 1, "msq", 4),     # stream RT=4 with a ragged last row tile (forced below)
    (64, 9, 1500, 64, "msq", 4),      # depthwise
    (36, 20, 4000, 3, "soft", 4),     # grouped
]


@pytest.mark.parametrize("shape", RANDOM_SHAPES)
@pytest.mark.parametrize("plan", [0, 1, 3])
def test_random_shapes_bit_exact_vs_oracle(qnn, oracle_mod, shape, plan):
    N, d, m, groups, mode, bits = shape
    reg = {"msq": None, "soft": "L1", "hard": "L0"}[mode]
    case = dict(name="rnd_%s" % "_".join(map(str, shape)), N=N, d=d, m=m, bits=bits, scalar=1.16, percentile=1.0,
                reg=reg, lamb=0.02, groups=groups, first_layer=False, zero_every=5, seed=3)
    W, A, X = gi.make_inputs(case)
    r = _run_layer(qnn, case, W, A, X, plan)
    K = 2 ** (bits - 1)
    o = oracle_mod.quantize_layer(W, A, X, 1.16 / K, K, 1.0, reg, 0.02, groups)
    assert np.float32(float(r["step"])) == o["step"]
    assert np.array_equal(r["idx"].cpu().numpy().astype(np.int16), o["idx"])
    assert np.array_equal(r["Q"].cpu().numpy().view(np.uint32), o["Q"].view(np.uint32))
    assert np.array_equal(r["U"].cpu().numpy(), o["U"])


@pytest.mark.parametrize("rt,c", [(1, 4), (1, 16), (2, 4), (2, 16), (4, 4), (4, 8), (4, 16)])
@pytest.mark.parametrize("mode", ["msq", "hard"])
def test_cooperative_configurations(qnn, oracle_mod, monkeypatch, rt, c, mode):
    """Forced (rows per workgroup, members per row) pairs of the cooperative plan, ragged row tiles included:
    every configuration reproduces the oracle bit for bit (the slot tree makes the order independent of C)."""
    N, d, m, bits = 11, 20, 40000, 4            # S = 40 segments; 11 rows -> ragged tiles for RT = 2, 4
    reg = {"msq": None, "hard": "L0"}[mode]
    case = dict(name="coop_%s" % mode, N=N, d=d, m=m, bits=bits, scalar=1.16, percentile=1.0, reg=reg, lamb=0.02,
                groups=1, first_layer=False, zero_every=6, seed=9)
    W, A, X = gi.make_inputs(case)
    monkeypatch.setenv("GPFQ_COOP_RT", str(rt))
    monkeypatch.setenv("GPFQ_COOP_C", str(c))
    from quantized_neural_nets_amd import _lib
    desc = _lib.describe_plan(N, d, m, 1, 3)
    assert desc.startswith("coop RT=%d C=%d" % (rt, c)), desc
    r = _run_layer(qnn, case, W, A, X, 3)
    _lib.check_status(DEV)
    o = oracle_mod.quantize_layer(W, A, X, 1.16 / 8, 8, 1.0, reg, 0.02, 1)
    assert np.array_equal(r["idx"].cpu().numpy().astype(np.int16), o["idx"])
    assert np.array_equal(r["U"].cpu().numpy(), o["U"])
    assert np.array_equal(r["Q"].cpu().numpy().view(np.uint32), o["Q"].view(np.uint32))


@pytest.mark.parametrize("d", [1, 2, 3, 4, 5, 6, 7, 13])
@pytest.mark.parametrize("kind", ["resident3", "resident10", "resident16", "coop"])
def test_every_tail_of_the_unrolled_column_loop(qnn, oracle_mod, monkeypatch, d, kind):
    """The register-resident kernels rotate their column buffers through a six-fold unrolled loop with the loads two
    steps ahead: every d mod 6 (and d = 1, 2, where the look-ahead re-reads the last column) must end on the right
    buffer.  One shape per kernel variant family (3 / 10 / 16 waves resident, cooperative)."""
    N, m, env = {"resident3": (5, 2100, {}), "resident10": (3, 10000, {}),
                 "resident16": (2, 16000, {}),
                 "coop": (6, 20000, {"GPFQ_COOP_RT": "2", "GPFQ_COOP_C": "4"})}[kind]
    case = dict(name="tail_%s_%d" % (kind, d), N=N, d=d, m=m, bits=4, scalar=1.16, percentile=1.0, reg="L1", lamb=0.01,
                groups=1, first_layer=False, zero_every=4, seed=20 + d)
    W, A, X = gi.make_inputs(case)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    from quantized_neural_nets_amd import _lib
    plan = 3 if kind == "coop" else 0
    desc = _lib.describe_plan(N, d, m, 1, plan)
    assert desc.startswith("coop RT=2 C=4" if kind == "coop" else "resident RT=1"), desc
    r = _run_layer(qnn, case, W, A, X, plan)
    _lib.check_status(DEV)
    o = oracle_mod.quantize_layer(W, A, X, 1.16 / 8, 8, 1.0, "L1", 0.01, 1)
    assert np.array_equal(r["idx"].cpu().numpy().astype(np.int16), o["idx"])
    assert np.array_equal(r["Q"].cpu().numpy().view(np.uint32), o["Q"].view(np.uint32))
    assert np.array_equal(r["U"].cpu().numpy(), o["U"])


def test_resident_plan_ragged_shape(qnn, oracle_mod):
    """The resident kernel (one row per workgroup, one wave per segment) on a shape with a ragged last segment and a
    row count that is no multiple of anything."""
    N, d, m = 19, 24, 5000                       # 5 segments
    case = dict(name="resrt", N=N, d=d, m=m, bits=4, scalar=1.16, percentile=1.0, reg="L1", lamb=0.01, groups=1,
                first_layer=False, zero_every=5, seed=4)
    W, A, X = gi.make_inputs(case)
    from quantized_neural_nets_amd import _lib
    assert _lib.describe_plan(N, d, m).startswith("resident RT=1")
    r = _run_layer(qnn, case, W, A, X, 0)
    o = oracle_mod.quantize_layer(W, A, X, 1.16 / 8, 8, 1.0, "L1", 0.01, 1)
    assert np.array_equal(r["idx"].cpu().numpy().astype(np.int16), o["idx"])
    assert np.array_equal(r["U"].cpu().numpy(), o["U"])


@pytest.mark.parametrize("rt,c", [(1, 2), (2, 4), (4, 2), (4, 8), (1, 16)])
def test_cooperative_streaming_configurations(qnn, oracle_mod, monkeypatch, rt, c):
    """Streaming plan with the columns of each row tile split over C workgroups (long rows, few of them)."""
    N, d, m = 9, 14, 140000                      # 137 segments -> 256 slots; ragged row tiles for RT = 2, 4
    case = dict(name="scoop", N=N, d=d, m=m, bits=4, scalar=1.16, percentile=1.0, reg=None, lamb=0.0, groups=1,
                first_layer=False, zero_every=5, seed=2)
    W, A, X = gi.make_inputs(case)
    monkeypatch.setenv("GPFQ_STREAM_RT", str(rt))
    monkeypatch.setenv("GPFQ_STREAM_C", str(c))
    from quantized_neural_nets_amd import _lib
    assert _lib.describe_plan(N, d, m, 1, 1).startswith("stream RT=%d C=%d" % (rt, c)), _lib.describe_plan(N, d, m, 1, 1)
    r = _run_layer(qnn, case, W, A, X, 1)
    _lib.check_status(DEV)
    o = oracle_mod.quantize_layer(W, A, X, 1.16 / 8, 8, 1.0, None, 0.0, 1)
    assert np.array_equal(r["idx"].cpu().numpy().astype(np.int16), o["idx"])
    assert np.array_equal(r["U"].cpu().numpy(), o["U"])


@pytest.mark.parametrize("N,expect", [(8, "coop RT=1 C=16"), (16, "coop RT="), (32, "coop RT=")])
def test_eight_gpu_shard_shapes_of_long_rows(qnn, oracle_mod, N, expect):
    """What one rank of an 8-GPU neuron shard sees for ResNet-50 layer1 / layer2.0 at batch 1024 (m = 93 184, a
    few rows), bit-exact against the oracle; plus the widest configuration (32 members per row tile), forced."""
    from quantized_neural_nets_amd import _lib
    d, m = 18, 93184
    case = dict(name="shard8", N=N, d=d, m=m, bits=4, scalar=1.16, percentile=1.0, reg=None, lamb=0.0, groups=1,
                first_layer=False, zero_every=7, seed=6)
    W, A, X = gi.make_inputs(case)
    assert _lib.describe_plan(N, d, m).startswith(expect), _lib.describe_plan(N, d, m)
    r = _run_layer(qnn, case, W, A, X, 0)
    _lib.check_status(DEV)
    o = oracle_mod.quantize_layer(W, A, X, 1.16 / 8, 8, 1.0, None, 0.0, 1)
    assert np.array_equal(r["idx"].cpu().numpy().astype(np.int16), o["idx"])
    assert np.array_equal(r["U"].cpu().numpy(), o["U"])
    if N == 8:
        os.environ["GPFQ_COOP_RT"], os.environ["GPFQ_COOP_C"] = "1", "32"
        try:
            assert _lib.describe_plan(N, d, m).startswith("coop RT=1 C=32")
            r2 = _run_layer(qnn, case, W, A, X, 0)
            _lib.check_status(DEV)
        finally:
            del os.environ["GPFQ_COOP_RT"], os.environ["GPFQ_COOP_C"]
        assert torch.equal(r2["idx"], r["idx"]) and torch.equal(r2["U"], r["U"])


@pytest.mark.parametrize("rt", [1, 2, 4])
def test_one_segment_rows(qnn, oracle_mod, monkeypatch, rt):
    """m <= 1024 (fully connected layers): a workgroup is one wave of the resident kernel's one-segment variant -- no LDS,
    no barrier -- with 1, 2 or 4 rows per wave: ragged last tile, history flush across 64-step blocks, grouped, all
    four quantizers (the stochastic one against the oracle's Philox stream)."""
    from quantized_neural_nets_amd import _lib
    # m <= 256 / <= 512: the variants that load and sweep one / two quarters of the segment (exact: the rest is zero padding)
    for (N, d, m, groups, reg) in [(13, 150, 700, 1, None), (9, 70, 1024, 3, "L0"), (7, 65, 333, 1, "L1"), (11, 90, 200, 1, None),
                                   (6, 70, 256, 2, None), (5, 66, 512, 1, "L0"), (8, 33, 257, 1, None), (4, 20, 32, 1, "L1")]:
        case = dict(name="wave", N=N, d=d, m=m, bits=4, scalar=1.16, percentile=1.0, reg=reg, lamb=0.02, groups=groups,
                    first_layer=False, zero_every=9, seed=12)
        W, A, X = gi.make_inputs(case)
        monkeypatch.setenv("GPFQ_RESIDENT_RT", str(rt))
        assert _lib.describe_plan(N, d, m, groups).startswith("resident RT=%d waves=1 S=1" % rt)
        r = _run_layer(qnn, case, W, A, X, 0)
        o = oracle_mod.quantize_layer(W, A, X, 1.16 / 8, 8, 1.0, reg, 0.02, groups)
        assert np.array_equal(r["idx"].cpu().numpy().astype(np.int16), o["idx"])
        assert np.array_equal(r["Q"].cpu().numpy().view(np.uint32), o["Q"].view(np.uint32))
        assert np.array_equal(r["U"].cpu().numpy(), o["U"])


def test_long_layer_crosses_history_blocks(qnn, oracle_mod):
    """d > 64 with d % 64 != 0: the Q / idx history is flushed every 64 columns and once more at the end."""
    case = dict(name="hist", N=6, d=201, m=1500, bits=3, scalar=1.16, percentile=1.0, reg=None, lamb=0.0, groups=1,
                first_layer=True, zero_every=0, seed=8)
    W, A, X = gi.make_inputs(case)
    for plan in (0, 3):
        r = _run_layer(qnn, case, W, A, X, plan)
        o = oracle_mod.quantize_layer(W, A, X, 1.16 / 4, 4, 1.0, None, 0.0, 1)
        assert np.array_equal(r["idx"].cpu().numpy().astype(np.int16), o["idx"])
        assert np.array_equal(r["Q"].cpu().numpy().view(np.uint32), o["Q"].view(np.uint32))
        assert np.array_equal(r["U"].cpu().numpy(), o["U"])


def test_quantization_in_place_with_initial_residual_and_views(qnn, oracle_mod):
    """StepAlgorithm._quantization mirrors step_algorithm.py:107-148: in place on Q and U, U may be non-zero,
    inputs may be strided group views (:236)."""
    from quantized_neural_nets_amd import StepAlgorithm as SA
    case, (W, A, X), fx, _ = gi.load_case("g4_groups2")
    step = float(fx["step"])
    g, N, d, m = 2, 16, 36, 120
    Wt, At, Xt = _t(W), _t(A), _t(X)
    Q = torch.zeros_like(Wt)
    U = torch.zeros(N, m, device=DEV)
    W3, Q3, U3 = Wt.view(g, -1, d), Q.view(g, -1, d), U.view(g, -1, m)
    A3, X3 = At.view(m, g, -1), Xt.view(m, g, -1)
    for i in range(g):
        SA._quantization(W3[i], Q3[i], U3[i], A3[:, i, :], X3[:, i, :], SA._msq, step, 8, 0.0)
    torch.cuda.synchronize()
    assert np.array_equal(Q.cpu().numpy(), fx["Q"]) and np.abs(U.cpu().numpy() - fx["U"]).max() <= 1e-5
    # split the column range in two calls, carrying U
    case, (W, A, X), fx, _ = gi.load_case("g2_16x64x96_msq_b4")
    Wt, At, Xt = _t(W), _t(A), _t(X)
    Q = torch.zeros_like(Wt)
    U = torch.zeros(16, 96, device=DEV)
    k = 23
    SA._quantization(Wt[:, :k], Q[:, :k], U, At[:, :k], Xt[:, :k], SA._msq, float(fx["step"]), 8, 0.0)
    SA._quantization(Wt[:, k:], Q[:, k:], U, At[:, k:], Xt[:, k:], SA._msq, float(fx["step"]), 8, 0.0)
    torch.cuda.synchronize()
    assert np.array_equal(Q.cpu().numpy(), fx["Q"]) and np.abs(U.cpu().numpy() - fx["U"]).max() <= 1e-5


def test_stochastic_loop_matches_oracle_bitwise(qnn, oracle_mod):
    from quantized_neural_nets_amd import StepAlgorithm as SA
    case, (W, A, X), fx, _ = gi.load_case("g2_16x64x96_msq_b4")
    r = SA._quantize_layer_ex(_t(W), _t(A), _t(X), 96, 1.16 / 8, 8, 1.0, None, 0.0, 1, True, torch.device(DEV), seed=77)
    torch.cuda.synchronize()
    Q, idx, U = oracle_mod.quantization(W, A, X, float(r["step"]), 8, mode=oracle_mod.MODE_STOCHASTIC, seed=77)
    assert np.array_equal(r["idx"].cpu().numpy().astype(np.int16), idx)
    assert np.array_equal(r["U"].cpu().numpy(), U)
    assert not np.array_equal(idx, fx["idx"])           # it really is stochastic
    assert np.abs(idx).max() <= 8
    # ONE generator behind the name: the standalone quantizer with the layer's seed and column = t is the loop's quantizer
    # at step t.  Column 0: u = w_0 * a_0, s = <u, x_0> / ||x_0||^2 in the canonical order (the oracle's cdot).
    x0, a0 = X[:, 0].astype(np.float32), A[:, 0].astype(np.float32)
    r0 = np.sqrt(oracle_mod.cdot(x0, x0), dtype=np.float32)
    n2 = np.float32(r0 * r0)
    s0 = np.array([np.float32(oracle_mod.cdot((np.float32(w) * a0).astype(np.float32), x0)) / n2 for w in W[:, 0]], dtype=np.float32)
    q0 = SA._stochastic_msq(float(r["step"]), torch.from_numpy(s0).to(DEV), 8, 0.0, seed=77, column=0, row_id0=0)
    assert np.array_equal(q0.cpu().numpy().view(np.uint32), r["Q"][:, 0].cpu().numpy().view(np.uint32))
    assert not torch.equal(q0, SA._stochastic_msq(float(r["step"]), torch.from_numpy(s0).to(DEV), 8, 0.0, seed=78))


def test_cpu_tensors_are_refused(qnn):
    from quantized_neural_nets_amd import StepAlgorithm as SA, _lib
    W = torch.zeros(4, 4)
    with pytest.raises(_lib.GpfqError):
        SA._quantize_layer(W, torch.zeros(8, 4), torch.zeros(8, 4), 8, 0.1, 8, 1, None, 0.1, 1, False, "cpu")


def test_c_abi_argument_errors(qnn):
    from quantized_neural_nets_amd import _lib
    import ctypes
    x = torch.zeros(8, device=DEV)
    rc = _lib.lib.gpfq_quantizer_f32(9, 0.1, ctypes.c_void_p(x.data_ptr()), 8, 4, 0.0, None,
                                     ctypes.c_void_p(x.data_ptr()), None, None)
    assert rc == -1 and b"bad argument" in _lib.lib.gpfq_last_error()
    with pytest.raises(_lib.GpfqError):
        _lib.describe_plan(64, 9, 100000, 1, _lib.PLAN_RESIDENT)


@pytest.mark.parametrize("shape", [(3, 3, 1_100_000, "msq"), (2, 2, 2_200_000, "soft"), (5, 2, 4_194_304, "msq")])
@pytest.mark.parametrize("plan", [0, 1, 4])
def test_rows_longer_than_a_million_samples(qnn, oracle_mod, shape, plan):
    """m > 1 048 576 (EfficientNet-B1's 112x112 1x1 convs see 3.2 M calibration rows at batch 1024): 1025..4096
    segments per row, i.e. 32 or 64 slots per lane in the second level of the canonical tree -- on the cooperative
    streaming plan, on whole-row streaming (the fallback plan), bit-exact against the oracle; one more segment is refused."""
    from quantized_neural_nets_amd import _lib
    N, d, m, mode = shape
    reg = {"msq": None, "soft": "L1"}[mode]
    case = dict(name="long_%d" % m, N=N, d=d, m=m, bits=4, scalar=1.16, percentile=1.0, reg=reg, lamb=0.02, groups=1,
                first_layer=False, zero_every=0, seed=5)
    W, A, X = gi.make_inputs(case)
    r = _run_layer(qnn, case, W, A, X, plan)
    o = oracle_mod.quantize_layer(W, A, X, 1.16 / 8, 8, 1.0, reg, 0.02, 1)
    assert np.array_equal(r["idx"].cpu().numpy().astype(np.int16), o["idx"])
    assert np.array_equal(r["U"].cpu().numpy(), o["U"])
    assert abs(float(r["quantize_error"]) - o["quantize_error"]) <= 1e-4 * o["quantize_error"]
    with pytest.raises(_lib.GpfqError):
        _lib.describe_plan(4, 4, 4_194_305)


@pytest.mark.parametrize("shape", [(40, 300, 3000), (12, 120, 40000), (9, 200, 700)])
def test_division_free_path_equals_the_division_path(qnn, monkeypatch, shape):
    """MSQ layers find the index from one multiplication and run the reference's two divisions only near a rounding
    boundary (gpfq_device.h quant_msq_from_dot); GPFQ_EXACT_DIVISIONS=1 runs the divisions on every step.  Same bits --
    on the resident, the cooperative and the one-segment kernels, and on a layer with zero columns (zero dot products
    always take the divisions)."""
    N, d, m = shape
    case = dict(name="fastdiv", N=N, d=d, m=m, bits=4, scalar=1.16, percentile=1.0, reg=None, lamb=0.0, groups=1,
                first_layer=False, zero_every=7, seed=33)
    W, A, X = gi.make_inputs(case)
    fast = _run_layer(qnn, case, W, A, X, 0)
    monkeypatch.setenv("GPFQ_EXACT_DIVISIONS", "1")
    exact = _run_layer(qnn, case, W, A, X, 0)
    assert torch.equal(fast["idx"], exact["idx"]) and torch.equal(fast["U"], exact["U"])
    assert torch.equal(fast["Q"].view(torch.int32), exact["Q"].view(torch.int32))
