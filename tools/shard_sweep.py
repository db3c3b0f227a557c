#!/usr/bin/env python3
"""Sweep of the cooperative configurations (rows per workgroup RT x members C) on the per-rank shard shapes of a workload:
for every distinct (N, m) of the layers and world sizes 1, 2, 4, 8 (rows = ceil(N / world)), the loop kernel is timed on
DLIMIT columns with what AUTO picks and with every forced (RT, C) pair the library accepts, plus the resident plan where
it applies.  One JSON line per timing; tools/shard_sweep_report.py (or any reader) compares AUTO with the best.

    python tools/shard_sweep.py [workload] > gpurun_out/shard_sweep.jsonl        (GPU box)
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


def clear_env():
    for k in list(os.environ):
        if k.startswith("GPFQ_") and k != "GPFQ_LIB_OVERRIDE":
            del os.environ[k]


def time_cfg(W, A, X, m, step, plan, reps=3):
    dev = W.device
    best = 1e9
    for _ in range(reps):
        ev = []

        def hook(tag, shape):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            ev.append(e)
        StepAlgorithm._quantize_layer_ex(W, A, X, m, 1.16 / 8, 8, 1, None, 0.1, 1, False, dev, compute_errors=False,
                                         step_override=step, plan=plan, event_hook=hook)
        torch.cuda.synchronize()
        best = min(best, ev[1].elapsed_time(ev[2]))
    _lib.check_status(dev)
    return best


def main():
    dev = torch.device("cuda:0")
    workload = sys.argv[1] if len(sys.argv) > 1 else "r50_all"
    dl = int(os.environ.get("DLIMIT", "96"))
    fn, batch = bw.WORKLOADS[workload][:2]
    layers = [l[:5] for l in bw.normalize_layers(fn(batch))]
    shapes = sorted({(N, m) for _, N, d, m, g in layers if g == 1})
    for (N, m) in shapes:
        d = min(dl, max(dg for _, n, dg, mm, g in layers if (n, mm) == (N, m) and g == 1))
        W, A, X = bw.synthetic_layer(N, d, m, 99)
        step = bw.layer_step(W)
        Ad, Xd = A.to(dev), X.to(dev)
        S = -(-m // 1024)
        for world in (1, 2, 4, 8):
            rows = -(-N // world)
            Wd = W[:rows].contiguous().to(dev)
            cfgs = [("auto", 0, {})]
            if S <= 16:
                cfgs += [("resident rt%d" % rt, 2, {"GPFQ_RESIDENT_RT": str(rt)}) for rt in (1, 2, 4)]
            if S > 1:
                for rt in (1, 2, 4):
                    for c in (2, 4, 8, 16, 32, 64, 128, 256):
                        if c <= S:
                            cfgs.append(("coop rt%d c%d" % (rt, c), 3, {"GPFQ_COOP_RT": str(rt), "GPFQ_COOP_C": str(c)}))
            seen_desc = set()
            for tag, plan, env in cfgs:
                clear_env()
                os.environ.update(env)
                try:
                    desc = _lib.describe_plan(rows, d, m, 1, plan).split(" d=")[0]
                except _lib.GpfqError:
                    continue
                if tag != "auto" and (desc in seen_desc or not desc.startswith(tag.split()[0])):
                    continue
                if tag.startswith("coop") and ("RT=%s " % env["GPFQ_COOP_RT"] not in desc or "C=%s " % env["GPFQ_COOP_C"] not in desc):
                    continue
                if tag.startswith("resident") and "RT=%s " % env["GPFQ_RESIDENT_RT"] not in desc:
                    continue
                seen_desc.add(desc) if tag != "auto" else None
                try:
                    ms = time_cfg(Wd, Ad, Xd, m, step, plan)
                except _lib.GpfqError as e:
                    continue
                rec = {"N": N, "m": m, "S": S, "world": world, "rows": rows, "cfg": tag, "plan": desc, "d": d,
                       "us_per_col": round(ms * 1e3 / d, 4)}
                print(json.dumps(rec), flush=True)
            clear_env()
        del Ad, Xd


if __name__ == "__main__":
    main()
