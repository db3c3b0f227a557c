"""The resident kernels' PREFETCH AGENT (gpfq_loop_kernels.h resident_prefetch_agent, round 4): one extra wave per workgroup
that touches the lines of column t + K so that they sit in the XCD's L2 when the sweeps ask for them -- the work shared by
the workgroups of an XCD (1, 2 or 4 lines of every segment each).  It only loads into a register nobody reads: with it, without
it, at any distance, the outputs are the same bits -- and equal the CPU oracle."""
import numpy as np
import pytest
import torch

import bench_workload as bw

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(W, A, X, m, step):
    from quantized_neural_nets_amd import StepAlgorithm as SA
    r = SA._quantize_layer_ex(W.to(DEV), A.to(DEV), X.to(DEV), m, 1.16 / 8, 8, 1, None, 0.05, 1, False, torch.device(DEV),
                              step_override=step, compute_errors=False)
    torch.cuda.synchronize()
    return r


@pytest.mark.parametrize("shape,why", [
    ((256, 40, 7168), "256 workgroups of 7 sweep waves: one line of every segment per workgroup (32 per XCD)"),
    ((512, 24, 3072), "two rows per workgroup, three sweep waves"),
    ((200, 30, 5000), "200 workgroups: 25 per XCD, two lines of every segment each"),
    ((100, 30, 5000), "100 workgroups: 13 per XCD, four lines of every segment each"),
    ((40, 30, 5000), "40 workgroups: fewer than eight per XCD -- no agent"),
    ((40, 12, 15000), "15 sweep waves + the agent: the 16-wave variant full (40 workgroups: four lines each would need 120 lanes -- no agent)"),
    ((256, 12, 11000), "11 sweep waves + the agent in the 12-wave variant"),
    ((300, 3, 7168), "three columns: the agent's distance exceeds the layer"),
])
def test_prefetch_agent_changes_no_bit(oracle_mod, monkeypatch, shape, why):
    from quantized_neural_nets_amd import _lib
    N, d, m = shape
    assert _lib.describe_plan(N, d, m).startswith("resident"), _lib.describe_plan(N, d, m)
    W, A, X = bw.synthetic_layer(N, d, m, 31 + N, first_layer=False)
    step = bw.layer_step(W)
    ref = _run(W, A, X, m, step)                                   # default: the agent at its default distance
    Q, idx, U = oracle_mod.quantization(W.numpy(), A.numpy(), X.numpy(), float(ref["step"]), 8)
    assert np.array_equal(ref["idx"].cpu().numpy().astype(np.int16), idx) and np.array_equal(ref["U"].cpu().numpy(), U)
    for k in ("0", "1", "5"):
        monkeypatch.setenv("GPFQ_RESIDENT_PREFETCH", k)
        r = _run(W, A, X, m, step)
        assert torch.equal(r["idx"], ref["idx"]) and torch.equal(r["U"], ref["U"]) and torch.equal(r["Q"], ref["Q"]), k
        assert torch.equal(r["usq_seg"], ref["usq_seg"]), k
