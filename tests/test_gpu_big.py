"""GPU parity at headline scale against reference-held data that was NOT seed-searched (tests/golden/g6_*.npz,
tools/make_golden.py gen_big_case), for every kernel family, through the tie audit of tests/tie_audit.py; and the
one-call C entry point gpfq_quantize_layer_f32 driven through the ctypes stub of INTEGRATION.md section B verbatim."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import golden_inputs as gi
from test_oracle_golden import check_big_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.mark.parametrize("name", sorted(gi.BIG_CASES))
@pytest.mark.parametrize("plan", [0, 1, 3])
def test_big_case_against_reference_and_oracle(oracle_mod, name, plan):
    """Every G6 shape has m >= 2048, so the cooperative plan (3) applies to all of them: every cooperative variant --
    plain, in rounds, LDS-staged, 128 and 256 members of one row, grouped (depthwise) -- meets reference-held data here."""
    from quantized_neural_nets_amd import StepAlgorithm, _lib
    case, (W, A, X), fx, meta = gi.load_big_case(name)
    K = 2 ** (case["bits"] - 1)
    g = case["groups"]
    desc = _lib.describe_plan(case["N"], case["d"], case["m"], g, plan)
    r = StepAlgorithm._quantize_layer_ex(_t(W), _t(A), _t(X), case["m"], case["scalar"] / K, K, case["percentile"],
                                         case["reg"], case["lamb"], g, False, torch.device(DEV), plan=plan)
    torch.cuda.synchronize()
    got = dict(idx=r["idx"].cpu().numpy(), U=r["U"].cpu().numpy(), step=float(r["step"]),
               quantize_error=float(r["quantize_error"]), relative_quantize_error=float(r["relative_quantize_error"]),
               relative_adder=None if g > 1 else r["relative_adder"].cpu().numpy())
    rep = check_big_case(name, got, "HIP[%s]" % desc.split(" d=")[0])
    # and the oracle, bit for bit (the canonical order is shared)
    o = oracle_mod.quantize_layer(W, A, X, case["scalar"] / K, K, case["percentile"], case["reg"], case["lamb"], g)
    assert np.array_equal(got["idx"].astype(np.int16), o["idx"])
    assert np.array_equal(got["U"], o["U"])
    assert np.array_equal(r["Q"].cpu().numpy().view(np.uint32), o["Q"].view(np.uint32))


def _integration_stub():
    """The python block of INTEGRATION.md section B, executed as written (only the library path is filled in)."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"```python\n(# gpfq_binding\.py.*?)```", text, re.S)
    assert m, "INTEGRATION.md section B stub not found"
    src = m.group(1).replace("/path/to/quantized_neural_nets_amd/libgpfq_hip.so",
                             os.path.join(ROOT, "quantized_neural_nets_amd", "libgpfq_hip.so"))
    ns = {}
    exec(compile(src, "INTEGRATION.md#B", "exec"), ns)
    return ns


@pytest.mark.parametrize("name", ["g2_32x288x1024_msq_b4", "g2_8x48x5000_soft_b2", "g3_percentile95_hard",
                                  "g4_groups2", "g4_depthwise", "g6_256x1152x2048_msq_b4", "g6_64x576x66560_msq_b4"])
def test_one_call_entry_point_through_the_integration_stub(oracle_mod, name):
    """gpfq_quantize_layer_f32 -- the symbol INTEGRATION.md tells a maintainer of step_algorithm.py:212-237 to bind --
    run on the GPU exactly through that stub: indices / Q / U against the reference fixture and the oracle."""
    ns = _integration_stub()
    big = name.startswith("g6_")
    case, (W, A, X), fx, meta = (gi.load_big_case if big else gi.load_case)(name)
    K = 2 ** (case["bits"] - 1)
    mode = ns["MODE"]["L1"] if case["reg"] == "L1" else ns["MODE"]["L0"] if case["reg"] == "L0" else ns["MODE"]["msq"]
    Q, idx, U = ns["quantize_layer_native"](_t(W), _t(A), _t(X), float(fx["step"]), K, mode, float(case["lamb"]),
                                            case["groups"])
    torch.cuda.synchronize()
    o = oracle_mod.quantize_layer(W, A, X, case["scalar"] / K, K, case["percentile"], case["reg"], case["lamb"],
                                  case["groups"], step=float(fx["step"]))
    assert np.array_equal(idx.cpu().numpy().astype(np.int16), o["idx"])
    assert np.array_equal(U.cpu().numpy(), o["U"])
    assert np.array_equal(Q.cpu().numpy().view(np.uint32), o["Q"].view(np.uint32))
    if big:
        check_big_case(name, dict(idx=idx.cpu().numpy(), U=U.cpu().numpy(), step=float(fx["step"])), "C-ABI one-call")
    else:
        assert np.array_equal(idx.cpu().numpy().astype(np.int16), fx["idx"])
        assert np.array_equal(Q.cpu().numpy(), fx["Q"])
        assert np.abs(U.cpu().numpy() - fx["U"]).max() <= 1e-5


@pytest.mark.parametrize("name", ["g2_32x288x1024_msq_b4", "g2_24x96x2500_msq_b4", "g2_8x48x5000_soft_b2", "g3_percentile95_hard",
                                  "g4_groups4_soft", "g4_depthwise_hard", "g6_256x1152x2048_msq_b4", "g6_128x1152x26624_msq_b4"])
def test_torch_extension_operator(oracle_mod, name):
    """torch.ops.gpfq.quantize_layer (the thin PyTorch-ROCm extension over the C ABI): indices / Q / U bit-equal to the
    oracle and to the reference fixture; strided (m, D) views are taken as they are; the fused sum of squares matches U."""
    from quantized_neural_nets_amd import torch_ext  # noqa: F401
    big = name.startswith("g6_")
    case, (W, A, X), fx, meta = (gi.load_big_case if big else gi.load_case)(name)
    K = 2 ** (case["bits"] - 1)
    mode = 1 if case["reg"] == "L1" else 2 if case["reg"] == "L0" else 0
    Ad, Xd = _t(A), _t(X)
    if not big:                                         # a view with a leading dimension: columns [0, D) of a wider matrix
        wide = torch.zeros((A.shape[0], A.shape[1] + 5), device=DEV)
        wide[:, :A.shape[1]] = Ad
        Ad = wide[:, :A.shape[1]]
    Q, idx, U, usq = torch.ops.gpfq.quantize_layer(_t(W), Ad, Xd, float(fx["step"]), K, mode, float(case["lamb"]), case["groups"], 0, 0)
    torch.cuda.synchronize()
    o = oracle_mod.quantize_layer(W, A, X, case["scalar"] / K, K, case["percentile"], case["reg"], case["lamb"], case["groups"],
                                  step=float(fx["step"]))
    assert idx.dtype == torch.int8
    assert np.array_equal(idx.cpu().numpy().astype(np.int16), o["idx"])
    assert np.array_equal(U.cpu().numpy(), o["U"]) and np.array_equal(Q.cpu().numpy().view(np.uint32), o["Q"].view(np.uint32))
    assert torch.allclose(usq.double().sum(1), (U.double() ** 2).sum(1), rtol=1e-5)
    if big:
        check_big_case(name, dict(idx=idx.cpu().numpy(), U=U.cpu().numpy(), step=float(fx["step"])), "torch.ops.gpfq")
    else:
        assert np.array_equal(idx.cpu().numpy().astype(np.int16), fx["idx"]) and np.array_equal(Q.cpu().numpy(), fx["Q"])
    with pytest.raises(RuntimeError):
        torch.ops.gpfq.quantize_layer(_t(W), _t(A)[:, :-1], _t(X)[:, :-1], 0.1, K, mode, 0.0, case["groups"], 0, 0)
    q = torch.ops.gpfq.quantizer(_t(W), float(fx["step"]), K, 0, 0.0, None)
    qo, _ = oracle_mod.quantizer_vec(0, float(fx["step"]), W, K)
    assert np.array_equal(q.cpu().numpy().ravel().view(np.uint32), qo.view(np.uint32))
