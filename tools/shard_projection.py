#!/usr/bin/env python3
"""Projection of the neuron-sharded (N-GPU) headline step from ONE GPU: every layer of the workload is timed with the
rows one rank would own at world sizes 1, 2, 4 and 8 (full d, full columns -- every rank prepares all of them), and
the per-rank step time is the sum over the layers plus a fixed allowance per layer for the int8 all_gather.

    python tools/shard_projection.py [workload] > profiles/rNN_shard_projection[_workload].json    (on the GPU box)

workload: r50_3x3 (default, the headline) or any other key of tests/bench_workload.py WORKLOADS (r50_all, ...).

This is a PROJECTION: no multi-GPU hardware run is behind it (the build pod has one GPU; the 1/2/4/8-GPU scaling run
is the driver's).  What it measures is real -- the single-GPU kernels on the shard shapes -- what it assumes is that
the ranks do not disturb each other and that an all_gather of <= 295 KB per rank costs ALLGATHER_US.
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench_workload as bw  # noqa: E402
from quantized_neural_nets_amd import StepAlgorithm, _lib  # noqa: E402

ALLGATHER_US = 40.0        # small-message RCCL all_gather over xGMI, latency-bound (assumption, not measured here)


def time_layer(W, A, X, m, step, reps=3):
    dev = W.device
    best = None
    for _ in range(reps):
        ev = []

        def hook(tag, shape):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            ev.append(e)
        StepAlgorithm._quantize_layer_ex(W, A, X, m, 1.16 / 8, 8, 1, None, 0.1, 1, False, dev, compute_errors=False,
                                         step_override=step, event_hook=hook)
        torch.cuda.synchronize()
        t = (ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]))
        if best is None or t[0] + t[1] < best[0] + best[1]:
            best = t
    return best


def main():
    dev = torch.device("cuda:0")
    workload = sys.argv[1] if len(sys.argv) > 1 else "r50_3x3"
    fn, batch, wdesc = bw.WORKLOADS[workload][:3]
    layers = [l[:5] for l in bw.normalize_layers(fn(batch))]
    if any(g != 1 for *_, g in layers):
        sys.exit("grouped layers shard by groups, not by rows: not covered by this projection")
    layers = [l[:4] for l in layers]
    shapes = {}
    for name, N, d, m in layers:
        shapes.setdefault((N, d, m), []).append(name)
    worlds = (1, 2, 4, 8)
    per_shape = {}
    for (N, d, m), names in shapes.items():
        W, A, X = bw.synthetic_layer(N, d, m, 4242)
        step = bw.layer_step(W)
        Wd, Ad, Xd = W.to(dev), A.to(dev), X.to(dev)
        rec = {"layers": names, "N": N, "d": d, "m": m, "shards": {}}
        for k in worlds:
            rows = -(-N // k)
            prep_ms, loop_ms = time_layer(Wd[:rows].contiguous(), Ad, Xd, m, step)
            rec["shards"][str(k)] = {"rows": rows, "plan": _lib.describe_plan(rows, d, m).split(" d=")[0], "prep_ms": round(prep_ms, 4),
                                     "loop_ms": round(loop_ms, 4), "us_per_column": round(loop_ms * 1e3 / d, 4)}
            print("N=%d d=%d m=%d x%d: rows %d %-34s loop %.3f ms (%.3f us/col) prep %.3f ms" % (
                N, d, m, k, rows, rec["shards"][str(k)]["plan"], loop_ms, loop_ms * 1e3 / d, prep_ms), file=sys.stderr, flush=True)
        per_shape["%dx%dx%d" % (N, d, m)] = rec
        del Wd, Ad, Xd
    total_w = sum(N * d for _, N, d, _ in layers)
    proj = {}
    for k in worlds:
        ms = 0.0
        for (N, d, m), names in shapes.items():
            s = per_shape["%dx%dx%d" % (N, d, m)]["shards"][str(k)]
            ms += len(names) * (s["prep_ms"] + s["loop_ms"] + (ALLGATHER_US * 1e-3 if k > 1 else 0.0))
        proj[str(k)] = {"ms_per_step": round(ms, 3), "M_weights_per_s": round(total_w / ms / 1e3, 1)}
    for k in worlds:
        proj[str(k)]["speedup_vs_1"] = round(proj["1"]["ms_per_step"] / proj[str(k)]["ms_per_step"], 3)
    out = {"what": "PROJECTION from single-GPU timings of the per-rank shard shapes; NO multi-GPU (N > 1) hardware run exists in this round",
           "workload": "%s: %s" % (workload, wdesc),
           "assumptions": {"allgather_us_per_layer": ALLGATHER_US, "ranks_do_not_interfere": True,
                           "every_rank_prepares_all_columns": True},
           "kernel_source_sha256": _lib.kernel_source_digest(), "projected": proj, "per_shape": per_shape,
           "reading": "a layer costs d steps of a latency chain whatever the number of rows, so fewer rows per GPU help only where "
                      "they allow a cheaper configuration; the north star's >= 6x at 8 GPUs assumed a bandwidth-bound single GPU"}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
