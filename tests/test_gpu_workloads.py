"""BASELINE.json configs 1-4 as WHOLE workloads on the MI355X, each checked against the CPU oracle.

`bench.py --workload X --steps 1 --warmup 0` quantizes every layer of the model (reference shapes:
quantize_neural_net.py:136-193 visits every layer, :334-347 is the m rule) through StepAlgorithm._quantize_layer_ex exactly
as the driver does; after the run it compares, for every DISTINCT (N, d_g, m, groups) shape of the model, the first and last
rows x the first columns of the timed run's indices with oracle.quantization on the same inputs (`oracle_shape_check`),
and every layer's indices with a rerun on the streaming kernel family (`output_check`).  A mismatch makes bench.py exit 3.

ResNet-18 additionally runs with --force-shard: a ONE-rank `nccl` process group, so that every layer of a real model goes
through dist.quantize_sharded -> all_gather_into_tensor(int8) -> all_reduce on the GPU -- RCCL itself, not gloo.
ResNet-50 (54 layers; the inputs of the full layers are 10^2 GB of host randn) runs on its distinct shapes with the input
features of each layer cut to 96 per group: N, m and groups -- hence plan, kernel variant and rounds -- are the full
layer's."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*extra, timeout=850, shared_card=False, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline"] + list(extra)
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-4000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["output_check"]["mismatches"] == 0, rec["output_check"]
    assert rec["oracle_shape_check"]["mismatches"] == 0, rec["oracle_shape_check"]
    # (two ranks rehearsing on ONE card are two tenants: each sizes its cooperative grids for a whole chip, so a launch may find
    # its peers not resident, give up within its bound and be redone on the streaming plan -- correct results, reported)
    assert rec["cooperative_timeouts"] == 0 or shared_card
    return rec, out.stderr


def test_resnet18_all_layers_sharded_through_rccl_world_of_one():
    """config 1: ResNet-18, all 21 conv + fc layers, 4-bit, batch 256 -- and the collectives through RCCL."""
    rec, err = run_bench("--workload", "r18", "--force-shard", "--backend", "nccl")
    assert rec["config"]["layers"] == 21 and rec["config"]["weights"] == 11_678_912
    assert "one-rank nccl group" in rec["config"]["parallelism"]
    assert rec["oracle_shape_check"]["shapes"] == 12 and rec["oracle_shape_check"]["weights"] > 10_000
    assert rec["output_check"]["layers"] == 21


def test_vgg16_all_layers():
    """config 2: VGG-16, all 16 layers, 4-bit, batch 512 (720 384-sample rows, fc6 with 25 088 columns)."""
    rec, err = run_bench("--workload", "vgg16")
    assert rec["config"]["layers"] == 16 and rec["config"]["weights"] == 138_344_128
    assert rec["oracle_shape_check"]["shapes"] == 12
    assert rec["output_check"]["layers"] >= 14          # (the two 720 384-sample convs are not rerun on the streaming plan)


def test_efficientnet_b1_all_layers_sparse_gpfq():
    """config 4 (one GPU's worth): EfficientNet-B1, all 116 layers, 2-bit, L1 lambda 0.1, batch 1024: depthwise convs as
    grouped cooperative launches, squeeze-excite 1x1 convs on 1x1 maps, rows of 3.2 M samples on the whole chip."""
    rec, err = run_bench("--workload", "effnet_b1")
    assert rec["config"]["layers"] == 116 and "L1 lamb 0.1" in rec["config"]["workload"] and "2-bit" in rec["config"]["workload"]
    assert rec["oracle_shape_check"]["shapes"] == 55
    assert rec["output_check"]["layers"] >= 100


def test_resnet50_all_distinct_layer_shapes():
    """config 3 (one GPU's worth): every distinct layer shape of ResNet-50's 54 layers at batch 1024 -- the 1x1 convs
    that run in rounds on the twelve-row LDS-staged pipelined kernels (round 5; before: the four-row lock-step ones, which
    tests/test_gpu_rounds.py keeps covered) included -- with full N, m, groups and 96 input features."""
    rec, err = run_bench("--workload", "r50_all", "--distinct-shapes", "--max-cols", "96")
    assert rec["config"]["layers"] == 24
    assert rec["oracle_shape_check"]["shapes"] == rec["config"]["layers"]
    assert "gpfq_pipel_m0_w8" in rec["roofline"]["families"] and "gpfq_pipe_rg2_m0_w8s" in rec["roofline"]["families"]
    assert 0 < rec["roofline_bound"]["frac"] <= 1.0 and rec["roofline_bound"]["families"]["gpfq_pipel_m0_w8"]["roof"] == "vector ALU"


def test_step_with_a_timed_out_launch_is_redone_layer_by_layer():
    """bench.py queues the (independent) layers of a step back to back and reads the cooperative kernels' status word once per
    step; a step in which a launch gave up (here every cooperative one: spin limit 0) is redone with the read after every layer,
    where each timed-out layer is redone on the streaming plan (step_algorithm.py run_rows) -- the step's indices equal the
    streaming rerun's and the oracle's, the timeouts are counted.  --status-per-layer: the same result with the read per layer."""
    forced = {"GPFQ_COOP_SPIN_LIMIT": "0"}
    rec, err = run_bench("--max-cols", "24", shared_card=True, env_extra=forced)
    assert "once per step" in rec["config"]["status_read"]
    assert rec["cooperative_timeouts"] >= 1 + 8          # the step, then the eight cooperative layers of its redo
    assert rec["output_check"]["layers"] == 16 and rec["oracle_shape_check"]["shapes"] == 7
    rec2, err = run_bench("--max-cols", "24", "--status-per-layer", shared_card=True, env_extra=forced)
    assert "after every layer" in rec2["config"]["status_read"] and rec2["cooperative_timeouts"] == 8
    rec3, err = run_bench("--max-cols", "24")
    assert rec3["cooperative_timeouts"] == 0 and rec3["device_allocations_in_timed_region"]["num_alloc_retries"] == 0


def test_sharded_path_through_rccl_equals_unsharded(tmp_path):
    """dist.enable(force=True) in a world of one rank on the `nccl` backend: quantize_sharded's all_gather_into_tensor of
    the int8 indices and the all_reduce of the partial sums execute in RCCL on the GPU; the result must equal the
    unsharded one bit for bit -- rows partition, whole-groups partition (depthwise), error metrics."""
    code = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, torch, torch.distributed as td
import golden_inputs as gi
from quantized_neural_nets_amd import _lib
from quantized_neural_nets_amd import StepAlgorithm as SA, dist as qd
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
td.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert td.get_backend() == "nccl"
try:
    for name in ["g2_64x147x512_msq_b4", "g2_24x96x2500_msq_b4", "g4_depthwise", "g4_groups2", "g2_16x64x96_hard_b4"]:
        case, (W, A, X), fx, _ = gi.load_case(name)
        K = 2 ** (case["bits"] - 1)
        args = (torch.from_numpy(W).to(dev), torch.from_numpy(A).to(dev), torch.from_numpy(X).to(dev), A.shape[0],
                case["scalar"] / K, K, case["percentile"], case["reg"], case["lamb"], case["groups"], False, dev)
        qd.disable()
        plain = SA._quantize_layer_ex(*args)
        assert plain["rows"] is None
        ctx = qd.enable(force=True)
        assert qd.active() is not None and qd.active().world == 1
        # the device's host lock (scratch -> launch -> status read, _lib.exclusive) is released before the collective: a rank
        # blocked in the all_gather while holding it would deadlock a peer rank living in another thread on the same card
        held = []
        ctx.event_hook = lambda tag: held.append((tag, _lib.exclusive(dev)._is_owned()))
        sh = SA._quantize_layer_ex(*args)
        torch.cuda.synchronize()
        assert [t for t, _ in held] == ["collective_begin", "collective_end"] and not any(h for _, h in held), held
        assert sh["rows"] is not None and sh["rows"].numel() == W.shape[0]          # the sharded path ran
        assert torch.equal(sh["idx"], plain["idx"]) and torch.equal(sh["Q"], plain["Q"]) and torch.equal(sh["U"], plain["U"]), name
        assert np.array_equal(sh["idx"].cpu().numpy().astype(np.int16), fx["idx"]), name
        for k in ("quantize_error", "relative_quantize_error"):
            assert abs(float(sh[k]) - float(plain[k])) <= 1e-5 * abs(float(plain[k])), (name, k)
    print("rccl-ok")
finally:
    td.destroy_process_group()
''' % (ROOT, ROOT)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0 and "rccl-ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_bench_on_one_card_over_gloo_runs_the_whole_multi_rank_script():
    """`bench.py --gpus 2` end to end (the self-launch under torch.distributed.run, rank set-up, the sharded layers, the events
    around the all_gather, the max over ranks, the checks on every rank) with both ranks on THIS card and `gloo` for the
    collectives -- the multi-rank path of the script the driver starts on an 8-GPU node, shape-reduced (--max-cols).  Not a
    scaling number; RCCL itself runs in test_resnet18_all_layers_sharded_through_rccl_world_of_one."""
    rec, err = run_bench("--gpus", "2", "--share-gpu", "--backend", "gloo", "--max-cols", "24", shared_card=True)
    assert rec["n_gpus"] == 2 and rec["scaling"] == "strong" and "neuron-shard x2" in rec["config"]["parallelism"]
    assert rec["config"]["layers"] == 16 and rec["config"]["weights"] == sum(N * 24 for N in (64,) * 3 + (128,) * 4 + (256,) * 6 + (512,) * 3)
    assert [r["rank"] for r in rec["per_rank"]] == [0, 1]
    for r in rec["per_rank"]:
        assert r["prep_ms"] > 0 and r["loop_ms"] > 0 and r["collective_ms"] > 0 and r["wall_ms_per_step"] > 0
    assert rec["collective_ms_per_step"] > 0
    assert rec["expected"] is not None and "emulated_world2" in rec["expected"]["source"] and rec["expected"]["per_rank_ms_per_step"] > 0
    oc = rec["output_check"]
    assert oc["mismatches"] == 0 and oc["ranks"] == 2 and oc["mismatches_all_ranks"] == 0 and oc["weights_all_ranks"] == 2 * oc["weights"]
    assert rec["oracle_shape_check"]["mismatches"] == 0
