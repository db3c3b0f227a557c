"""The N > 1 path on CPU: world_size-2 (and 3) gloo process groups drive quantized_neural_nets_amd.dist with
the per-rank computation supplied by the CPU oracle.  Checks that the gathered indices / rebuilt Q / error
metrics equal the unsharded result bit for bit (indices, Q) for the three partition kinds."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as td
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_run_rows(step, K, mode, lamb, m, dg, seed=0):
    import oracle

    def run_rows(W_loc, groups_loc, A_loc, lda, X_loc, ldx, row_id0):
        Nl = W_loc.shape[0]
        if Nl == 0:
            return (torch.zeros(0, dg), torch.zeros(0, dg, dtype=torch.int8), torch.zeros(0, m), torch.zeros(0, 1))
        Wn = W_loc.contiguous().numpy()
        An = np.ascontiguousarray(A_loc.numpy())
        Xn = np.ascontiguousarray(X_loc.numpy())
        Ng = Nl // groups_loc
        Q = np.zeros((Nl, dg), np.float32)
        idx = np.zeros((Nl, dg), np.int16)
        U = np.zeros((Nl, m), np.float32)
        for g in range(groups_loc):
            q, i, u = oracle.quantization(Wn[g * Ng:(g + 1) * Ng], An[:, g * dg:(g + 1) * dg], Xn[:, g * dg:(g + 1) * dg],
                                          step, K, mode=mode, lamb=lamb, nthreads=1, seed=seed,
                                          row_id0=int(row_id0) + g * Ng)      # as the kernels: key = row_id0 + local row
            Q[g * Ng:(g + 1) * Ng], idx[g * Ng:(g + 1) * Ng], U[g * Ng:(g + 1) * Ng] = q, i, u
        usq_seg = (torch.from_numpy(U).double() ** 2).sum(1, keepdim=True).float()   # one "segment" per row here
        return torch.from_numpy(Q), torch.from_numpy(idx.astype(np.int8)), torch.from_numpy(U), usq_seg
    return run_rows


def _worker(rank, world, port, case_name, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import golden_inputs as gi
        import oracle
        from quantized_neural_nets_amd import dist as qd
        case, (W, A, X), fx, _ = gi.load_case(case_name)
        K = 2 ** (case["bits"] - 1)
        groups, dg, m = case["groups"], case["d"], case["m"]
        mode = 1 if case["reg"] == "L1" else 2 if case["reg"] == "L0" else 0
        step = float(fx["step"])
        ctx = qd.enable()
        assert qd.active() is ctx and ctx.world == world
        # the hook bench.py hangs its events on: called around the layer-end all_gather of the index shards, in pairs
        tags = []
        ctx.event_hook = tags.append
        Wt, At, Xt = torch.from_numpy(W), torch.from_numpy(A), torch.from_numpy(X)
        Q, idx, U_loc, usq_seg, rows = qd.quantize_sharded(ctx, Wt, At, At.shape[1], Xt, Xt.shape[1], groups, dg, step, K,
                                                           mode, float(np.float32(case["lamb"])), torch.int8,
                                                           _oracle_run_rows(step, K, mode, case["lamb"], m, dg))
        assert tags == ["collective_begin", "collective_end"], tags      # ONE collective of indices per layer
        ctx.event_hook = None
        met = qd.sharded_error_metrics(ctx, Wt, At, usq_seg.double().sum(1), groups, rows, U_loc)
        # stochastic quantizer: the Philox keys are the GLOBAL row numbers, so the sharded draw equals the unsharded one
        # (this is what used to break for 1 < groups < world)
        Ngf = W.shape[0] // groups
        full = np.concatenate([oracle.quantization(W[g * Ngf:(g + 1) * Ngf], A[:, g * dg:(g + 1) * dg], X[:, g * dg:(g + 1) * dg],
                                                   step, K, mode=3, nthreads=1, seed=5, row_id0=g * Ngf)[1]
                               for g in range(groups)], 0)
        _, idx_s, _, _, _ = qd.quantize_sharded(ctx, Wt, At, At.shape[1], Xt, Xt.shape[1], groups, dg, step, K, 3, 0.0,
                                                torch.int8, _oracle_run_rows(step, K, 3, 0.0, m, dg, seed=5))
        assert np.array_equal(idx_s.numpy().astype(np.int16), full), "stochastic shard != unsharded draw"
        # every rank holds the full gathered result
        assert np.array_equal(idx.numpy().astype(np.int16), fx["idx"]), "gathered indices differ from the reference"
        assert np.array_equal(Q.numpy(), fx["Q"])
        assert np.array_equal(U_loc.numpy(), fx["U"][rows.numpy()])
        assert abs(float(met["quantize_error"]) - float(fx["quantize_error"])) <= 1e-4 * float(fx["quantize_error"])
        assert abs(float(met["relative_quantize_error"]) - float(fx["relative_quantize_error"])) <= 1e-4 * float(
            fx["relative_quantize_error"])
        if groups == 1:
            assert np.allclose(met["relative_adder"].numpy(), fx["relative_adder"], rtol=1e-4, atol=1e-6)
        open(os.path.join(out_dir, "ok_%d" % rank), "w").write("%d" % rows.numel())
    finally:
        td.destroy_process_group()


@pytest.mark.parametrize("case_name,world", [
    ("g2_64x147x512_msq_b4", 2),     # rows
    ("g2_16x64x96_hard_b4", 2),      # rows, hard-threshold index encoding
    ("g4_depthwise", 2),             # whole groups per rank
    ("g4_groups2", 3),               # rows inside groups (1 < groups < world), with an uneven split
    ("g3_m1", 2),
])
def test_sharded_equals_unsharded(tmp_path, case_name, world):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, case_name, str(tmp_path)), nprocs=world, join=True)
    got = sorted(os.listdir(tmp_path))
    assert got == ["ok_%d" % r for r in range(world)]
    import golden_inputs as gi
    assert sum(int(open(os.path.join(tmp_path, f)).read()) for f in got) == gi.CASES[case_name]["N"]
