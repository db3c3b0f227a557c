#!/usr/bin/env python3
"""bench.py -- M weights quantized / sec of the GPFQ loop on the ResNet-50 3x3 conv layers at calibration
batch 1024 (BASELINE.json metric), synthetic activations/weights of the named layer shapes.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (one rank per GPU)

A "step" is one pass of the hot path over the whole workload: for each of the 16 layers, column preparation
(transpose + norms) + the GPFQ loop kernel (+ the layer-end RCCL all_gather of the int8 indices when N > 1).
Inputs are resident in HBM before the timed region.  With N > 1 the output neurons of every layer are sharded
across the ranks (strong scaling: total work fixed).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import bench_workload as bw  # noqa: E402

HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: 8 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(layers, budget_s=20.0):
    """The reference's per-step op sequence (torch ops, oracle/gpfq_oracle.py torch_restatement_quantization)
    and the C oracle, timed on this box's host cores on a bounded sample: the first columns of four of the
    workload's layer shapes, as many as fit in budget_s/4 seconds each (at least 8).  torch gets
    min(cores, 16) threads: with one thread per core of a 256-core host the reference's small ops crawl."""
    import oracle
    ncores = os.cpu_count() or 1
    nthreads = min(ncores, 16)
    torch.set_num_threads(nthreads)
    shapes = []
    seen = set()
    for name, N, d, m in layers:
        if (N, d, m) not in seen and (N, m) in ((512, 3072), (256, 7168), (128, 26624), (64, 93184)):
            seen.add((N, d, m))
            shapes.append((name, N, d, m))
    tot_w = tot_t = tot_w_c = tot_t_c = 0.0
    sample = []
    for name, N, d, m in shapes:
        cap = min(d, 512)
        W, A, X = bw.synthetic_layer(N, d, m, 4321, d_limit=cap)
        step = bw.layer_step(W)
        Q = torch.zeros_like(W)
        U = torch.zeros(N, m)
        stept = torch.tensor(step)
        cols, t0 = 0, time.perf_counter()
        while cols < cap and (cols < 8 or time.perf_counter() - t0 < budget_s / len(shapes)):
            oracle.torch_restatement_quantization(W[:, cols:cols + 4], Q[:, cols:cols + 4], U, A[:, cols:cols + 4],
                                                  X[:, cols:cols + 4], stept, 8)
            cols += 4
        dt = time.perf_counter() - t0
        tot_w += N * cols
        tot_t += dt
        ccap = min(cap, 128)
        t0 = time.perf_counter()
        oracle.quantization(W[:, :ccap].numpy(), A[:, :ccap].numpy(), X[:, :ccap].numpy(), step, 8, nthreads=nthreads)
        dtc = time.perf_counter() - t0
        cap = ccap
        tot_w_c += N * cap
        tot_t_c += dtc
        sample.append("%s first %d cols" % (name, cols))
        log("cpu baseline %-16s N=%d m=%d: torch-op restatement %d cols %.2fs (%.4f Mw/s); C oracle %d cols %.2fs (%.4f Mw/s)"
            % (name, N, m, cols, dt, N * cols / dt / 1e6, cap, dtc, N * cap / dtc / 1e6))
    return {"value": round(tot_w / tot_t / 1e6, 5), "unit": "M weights/s", "cores": nthreads, "kind": "port",
            "sample": "torch-op restatement of step_algorithm.py:140-148 on " + "; ".join(sample),
            "host_cores": ncores, "c_oracle_value": round(tot_w_c / tot_t_c / 1e6, 5)}


def kernel_name(desc):
    """The template instantiation a plan description launches (quantized_neural_nets_amd/csrc launch_slab):
    the names rocprofv3 reports."""
    w = desc.split()
    kv = dict(x.split("=") for x in w[1:] if "=" in x)
    rt, waves = int(kv["RT"]), int(kv["waves"])
    if w[0] == "resident":
        if int(kv.get("S", "0")) == 1:
            return "gpfq_wave_kernel<"
        return "gpfq_resident_kernel<0, %d>" % (8 if waves <= 8 else 12 if waves <= 12 else 16)
    if w[0] == "coop":
        if rt == 1:
            return "gpfq_coop_kernel<1, 0, 12, 2>"
        if rt == 2 or waves <= 8:
            return "gpfq_coop_kernel<%d, 0, %d, 2>" % (rt, 8 if waves <= 8 else 12)
        return "gpfq_coop_kernel<4, 0, 12, 1>"
    return "gpfq_stream_kernel<%d, true" % rt


def pmc_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the newest committed rocprofv3 PMC summary (tools/pmc_traffic.py:
    separate FETCH_SIZE / WRITE_SIZE passes of this same command, gfx950 FETCH_SIZE correction applied)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")))
    if not files:
        return None, None
    data = json.load(open(files[-1]))
    for name, v in data.get("kernels", {}).items():
        if kernel in name:
            return v["hbm_bytes_per_launch"], os.path.relpath(files[-1], ROOT)
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=None, help="calibration batch (default: the workload's named batch; 1024 for the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--plan", type=int, default=0, help="0 auto, 1 stream, 2 resident (debug)")
    ap.add_argument("--layers", default=None, help="substring filter on layer names (debug)")
    ap.add_argument("--workload", default="r50_3x3", choices=sorted(bw.WORKLOADS),
                    help="r50_3x3 = the headline config (sixteen 3x3 convs at batch 1024); secondary: r50_all_convs (all 53 "
                         "conv layers), r18 (ResNet-18 at batch 256), vgg16 (VGG-16 at batch 512)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses cuda:0")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from quantized_neural_nets_amd import StepAlgorithm, _lib, dist as qdist
    import torch.distributed as td
    if world > 1:
        if args.backend == "nccl":
            td.init_process_group("nccl", device_id=dev)
        else:
            td.init_process_group(args.backend)
        qdist.enable()
    StepAlgorithm.plan = args.plan
    # host threads for the input generation: torchrun pins OMP_NUM_THREADS to 1 per rank; share the cores instead
    torch.set_num_threads(max(1, min(32, (os.cpu_count() or 1) // max(world, 1))))

    layer_fn, named_batch, workload_desc = bw.WORKLOADS[args.workload]
    if args.batch is None:
        args.batch = named_batch
    layers = layer_fn(args.batch)
    if args.layers:
        layers = [l for l in layers if args.layers in l[0]]
    total_weights = sum(N * d for _, N, d, _ in layers)
    alg_bytes = {name: bw.algorithmic_bytes(N, d, m) for name, N, d, m in layers}

    # ---- synthetic inputs, generated on the host (identical bits on every rank), resident in HBM
    t0 = time.perf_counter()
    data = []
    for li, (name, N, d, m) in enumerate(layers):
        W, A, X = bw.synthetic_layer(N, d, m, 1234 + li, first_layer=False)
        step = bw.layer_step(W)
        data.append((name, W.to(dev), A.to(dev), X.to(dev), step, m))
        del W, A, X
    torch.cuda.synchronize()
    if rank == 0:
        log("inputs: %d layers, %.3f M weights, generated in %.1fs" % (len(layers), total_weights / 1e6,
                                                                       time.perf_counter() - t0))

    events = []          # (layer name, tag, event) recorded by the hook inside the timed region
    cur = {"name": None, "on": False}

    def hook(tag, shape):
        if cur["on"]:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            events.append((cur["name"], tag, ev))

    StepAlgorithm.event_hook = hook

    def one_step():
        for name, W, A, X, step, m in data:
            cur["name"] = name
            StepAlgorithm._quantize_layer_ex(W, A, X, m, 1.16 / 8, 8, 1, None, 0.1, 1, False, dev,
                                             compute_errors=False, step_override=step)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            td.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    fence()
    cur["on"] = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    fence()
    elapsed = time.perf_counter() - t0
    cur["on"] = False
    _lib.check_status(dev)          # a cooperative kernel that gave up waiting for a peer invalidates the run
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
        td.all_reduce(tmax, op=td.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- per-launch durations of the loop kernel from the events recorded on the launch stream
    per_layer = {}
    assert len(events) % 3 == 0
    for i in range(0, len(events), 3):
        (n0, tg0, e0), (n1, tg1, e1), (n2, tg2, e2) = events[i], events[i + 1], events[i + 2]
        assert (tg0, tg1, tg2) == ("prepare_begin", "loop_begin", "loop_end") and n0 == n1 == n2
        rec = per_layer.setdefault(n0, {"prep_ms": 0.0, "loop_ms": 0.0, "n": 0})
        rec["prep_ms"] += e0.elapsed_time(e1)
        rec["loop_ms"] += e1.elapsed_time(e2)
        rec["n"] += 1
    fam = {}
    for name, N, d, m in layers:
        rec = per_layer.get(name)
        if not rec:
            continue
        Nl = N
        if world > 1:
            kind_, chunk = qdist.partition(N, 1, world)
            a, b = qdist.local_range(kind_, chunk, N, 1, rank)
            Nl = b - a
        desc = _lib.describe_plan(max(Nl, 1), d, m)
        kind = kernel_name(desc)
        f = fam.setdefault(kind, {"ms": 0.0, "bytes": 0.0, "launches": 0})
        f["ms"] += rec["loop_ms"]
        f["bytes"] += bw.algorithmic_bytes(Nl, d, m) * rec["n"]
        f["launches"] += rec["n"]
        if rank == 0:
            lm, pm = rec["loop_ms"] / rec["n"], rec["prep_ms"] / rec["n"]
            log("%-16s N=%4d d=%5d m=%6d %-24s loop %8.3f ms (%.3f us/col, %6.0f GB/s alg, %5.1f%% of 8 TB/s)  prep %7.3f ms"
                % (name, N, d, m, " ".join(desc.split()[:3]), lm, lm * 1e3 / d, bw.algorithmic_bytes(Nl, d, m) / lm / 1e6,
                   bw.algorithmic_bytes(Nl, d, m) / lm / 1e6 / HBM_PEAK_GBPS * 100, pm))

    if rank == 0:
        dom = max(fam, key=lambda k: fam[k]["ms"]) if fam else None
        roofline = None
        if dom:
            f = fam[dom]
            achieved = f["bytes"] / (f["ms"] * 1e-3) / 1e9
            traffic, tsrc = pmc_traffic(dom)
            roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1),
                        "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                        "traffic": traffic, "traffic_source": tsrc, "launches": f["launches"],
                        "avg_launch_ms": round(f["ms"] / f["launches"], 4),
                        "alg_bytes_per_launch": round(f["bytes"] / f["launches"]),
                        "families": {k: {"ms_total": round(v["ms"], 3), "launches": v["launches"],
                                         "achieved_GBps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)}
                                     for k, v in fam.items()}}
        out = {
            "metric": "M weights quantized/sec (GPFQ loop), %s, calib batch %d"
                      % ("ResNet-50 conv layers" if args.workload.startswith("r50") else workload_desc.split(" all")[0] + " layers", args.batch),
            "value": round(total_weights * args.steps / elapsed / 1e6, 4),
            "unit": "M weights/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload_desc + ", calibration batch %d, 4-bit (K=8), scalar 1.16, "
                                   "retain_rate 0.25" % args.batch,
                       "layers": len(layers), "weights": total_weights,
                       "algorithmic_bytes": sum(alg_bytes.values()),
                       "parallelism": "neuron-shard x%d + all_gather(int8 idx)" % world if world > 1 else "single GPU"},
            "roofline_whole_job_frac": round(sum(alg_bytes.values()) * args.steps / elapsed / 1e9 / HBM_PEAK_GBPS / max(world, 1), 4),
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline and args.workload == "r50_3x3":
            out["cpu_baseline"] = cpu_baseline(layers)
        print(json.dumps(out), flush=True)
    if world > 1:
        td.destroy_process_group()


if __name__ == "__main__":
    main()
