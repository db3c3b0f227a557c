#!/usr/bin/env python3
"""Scratch: rows per wave of the one-segment resident kernels (GPFQ_RESIDENT_RT = 1 / 2 / 4) by row count and samples:
loop time per column.   python tools/scratch/one_segment_rt.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench_workload as bw
from quantized_neural_nets_amd import StepAlgorithm, _lib
dev = torch.device("cuda:0")
d = 1024
for m in (256, 512, 1024):
    for N in (512, 1000, 1280, 2048, 3072, 4096, 8192):
        W, A, X = bw.synthetic_layer(N, d, m, 7, d_limit=d)
        step = bw.layer_step(W)
        Wd, Ad, Xd = W.to(dev), A.to(dev), X.to(dev)
        out = []
        for rt in ("1", "2", "4"):
            os.environ["GPFQ_RESIDENT_RT"] = rt
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
        os.environ.pop("GPFQ_RESIDENT_RT")
        print("m=%5d N=%5d  RT=1 %.3f  RT=2 %.3f  RT=4 %.3f us/col   auto: %s" % (m, N, out[0], out[1], out[2], _lib.describe_plan(N, d, m)[:24]), flush=True)
