#!/usr/bin/env python3
"""Scratch: the pause before the first poll of a cooperative exchange (GPFQ_COOP_POLL_DELAY, units of 256 clocks) by
configuration: loop time per column.   python tools/scratch/poll_delay_sweep.py [N,d,m[,ENV=VAL...] ...]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench_workload as bw
from quantized_neural_nets_amd import StepAlgorithm, _lib
dev = torch.device("cuda:0")
delays = [int(v) for v in os.environ.get("DELAYS", "0,1,2,3,4,6,8,10,12,16").split(",")]
cases = sys.argv[1:] or [
    "64,256,93184", "128,256,26624", "256,256,26624", "128,256,93184",            # headline: 2x8, 2x4, 4x4, 4x8
    "512,256,13312", "256,192,51200", "64,128,201728", "128,96,185344",            # 4x2, 4x4 (13 waves), 4x16
    "32,96,263168", "16,64,803840", "16,64,720384", "8,32,1440768", "4,16,3212288", "8,16,3212288"]   # 4x32, 4x64, 4x64, 2x128, lh
for a in cases:
    parts = a.split(",")
    N, d, m = (int(v) for v in parts[:3])
    env = dict(kv.split("=") for kv in parts[3:])
    os.environ.update(env)
    W, A, X = bw.synthetic_layer(N, d, m, 7, d_limit=d)
    step = bw.layer_step(W)
    Wd, Ad, Xd = W.to(dev), A.to(dev), X.to(dev)
    del A, X
    out = []
    for dl in delays:
        os.environ["GPFQ_COOP_POLL_DELAY"] = str(dl)
        best = 1e9
        for it in range(3):
            ev = []
            def hook(tag, shape):
                if tag in ("loop_begin", "loop_end"):
                    e = torch.cuda.Event(enable_timing=True); e.record(); ev.append(e)
            StepAlgorithm._quantize_layer_ex(Wd, Ad, Xd, m, 1.16 / 8, 8, 1, None, 0.1, 1, False, dev, compute_errors=False,
                                             step_override=step, event_hook=hook)
            torch.cuda.synchronize()
            best = min(best, ev[0].elapsed_time(ev[1]))
        out.append(best * 1e3 / d)
    os.environ.pop("GPFQ_COOP_POLL_DELAY")
    for k in env: os.environ.pop(k)
    bi = min(range(len(out)), key=lambda i: out[i])
    print("%-28s %-40s" % (a, _lib.describe_plan(N, d, m)[:40]) + " ".join("%d:%.3f%s" % (dl, v, "*" if i == bi else "") for i, (dl, v) in enumerate(zip(delays, out))), flush=True)
    del Wd, Ad, Xd
    torch.cuda.empty_cache()
