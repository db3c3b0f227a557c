"""Rehearsal of the N > 1 path ON the GPU box: two processes share the one MI355X (gloo process group, the
collectives staged through the host), each quantizes its neuron shard with the HIP kernels, and the gathered
result must equal the reference fixture and the single-process result bit for bit.  RCCL itself needs one GPU
per rank and is exercised by the driver's multi-GPU bench; everything around the collective is covered here."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, names, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as td
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import golden_inputs as gi
        from quantized_neural_nets_amd import StepAlgorithm as SA, dist as qd
        dev = torch.device("cuda:0")
        qd.enable()
        for name in names:
            case, (W, A, X), fx, _ = gi.load_case(name)
            K = 2 ** (case["bits"] - 1)
            r = SA._quantize_layer_ex(torch.from_numpy(W).to(dev), torch.from_numpy(A).to(dev),
                                      torch.from_numpy(X).to(dev), A.shape[0], case["scalar"] / K, K, case["percentile"],
                                      case["reg"], case["lamb"], case["groups"], False, dev)
            torch.cuda.synchronize()
            assert np.array_equal(r["idx"].cpu().numpy().astype(np.int16), fx["idx"]), name
            assert np.array_equal(r["Q"].cpu().numpy(), fx["Q"]), name
            rows = r["rows"].cpu().numpy()
            assert np.array_equal(r["U"].cpu().numpy(), fx["U"][rows]), name
            assert abs(float(r["quantize_error"]) - float(fx["quantize_error"])) <= 1e-4 * float(fx["quantize_error"])
            assert abs(float(r["relative_quantize_error"]) - float(fx["relative_quantize_error"])) <= 1e-4 * float(
                fx["relative_quantize_error"])
        open(os.path.join(out_dir, "ok_%d" % rank), "w").write("ok")
    finally:
        td.destroy_process_group()


def test_two_ranks_on_one_gpu_match_the_reference(tmp_path):
    names = ["g2_64x147x512_msq_b4", "g2_24x96x2500_msq_b4", "g4_depthwise", "g2_16x64x96_hard_b4"]
    mp.spawn(_worker, args=(2, _free_port(), names, str(tmp_path)), nprocs=2, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok_0", "ok_1"]


def test_bench_runs_sharded(tmp_path):
    """bench.py's N = 2 code path end to end on a reduced workload (two layers, gloo, shared card)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1",
           "--warmup", "1", "--backend", "gloo", "--share-gpu", "--layers", "layer4.1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["value"] > 0 and rec["scaling"] == "strong"
    assert rec["roofline"]["kernel"].startswith("gpfq_")
    assert rec["output_check"]["mismatches"] == 0 and rec["output_check"]["layers"] == 1


def test_bench_capture_mode_runs():
    """bench.py --capture: column preparation by the driver's fused patch gather from synthetic feature maps."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--capture", "--steps", "1", "--warmup", "1", "--layers", "layer4.1",
           "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["config"]["column_prep"].startswith("fused patch gather") and rec["output_check"]["mismatches"] == 0
    assert rec["roofline"]["kernel"] == "gpfq_resident_rt2_m0_w8" and rec["roofline_l2"]["frac"] <= 1.0


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT torchrun (what the driver's scaling run does): bench.py starts
    torch.distributed.run itself as a child process before touching the GPU, relays the JSON line and the exit code."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--backend", "gloo",
           "--share-gpu", "--layers", "layer4.1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["value"] > 0 and rec["roofline"] is not None and rec["roofline_l2"] is not None
    assert rec["output_check"]["mismatches"] == 0
    # and a child that fails makes the parent fail (no silent rc 0)
    bad = subprocess.run(cmd + ["--plan", "9"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert bad.returncode != 0


def test_bench_emulated_world_is_rank_zero_of_the_sharded_run():
    """`bench.py --emulate-world 8` (the per-rank step of an 8-GPU run, measured on one GPU: rank 0's rows of every layer,
    full columns, no collective) quantizes exactly the rows `dist.partition` gives rank 0 -- its output and oracle checks
    run on those rows -- and marks its line as a projection."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--layers", "layer3.0", "--no-cpu-baseline",
           "--emulate-world", "8"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 1 and rec["emulated_world"] == 8 and "projection" in rec
    assert rec["output_check"]["mismatches"] == 0 and rec["output_check"]["weights"] == 32 * 2304       # 256 rows / 8
    assert rec["oracle_shape_check"]["mismatches"] == 0 and rec["oracle_shape_check"]["shapes"] == 1
    assert "coop RT=1 C=8" in out.stderr                   # the plan of a 32-row shard of layer3.0.conv2, not of the full layer
