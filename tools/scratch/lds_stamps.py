#!/usr/bin/env python3
"""Diagnostic: per-wave phase cycles of the LDS-staged four-row cooperative kernel (needs the stamps patch of
tools/scratch/lds_stamps.patch applied and `make -C quantized_neural_nets_amd/csrc stamps`).
   GPFQ_LIB_OVERRIDE=.../libgpfq_hip_stamps.so python tools/scratch/lds_stamps.py N,d,m ..."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench_workload as bw
from quantized_neural_nets_amd import StepAlgorithm, _lib
names = ["top", "vmcnt", "sweep", "tree+dma", "barrier1", "sloads/reducer", "barrier2", "-"]
dev = torch.device("cuda:0")
for a in sys.argv[1:]:
    if "=" in a:
        k, v = a.split("="); os.environ[k] = v; continue
    N, d, m = (int(v) for v in a.split(","))
    W, A, X = bw.synthetic_layer(N, d, m, 99, d_limit=d)
    step = bw.layer_step(W)
    for it in range(2):
        StepAlgorithm._quantize_layer_ex(W.to(dev), A.to(dev), X.to(dev), m, 1.16 / 8, 8, 1, None, 0.1, 1, False, dev,
                                         compute_errors=False, step_override=step)
    torch.cuda.synchronize()
    scr = _lib.scratch(dev)
    dbg = scr[96 * 1024 + 64: 96 * 1024 + 64 + 32 * 64].view(torch.int64).cpu().tolist()
    print(a, _lib.describe_plan(N, d, m))
    print("  wave " + " ".join("%9s" % n for n in names[:7]) + "   arrives at barrier 1")
    for w in list(range(14)) + list(range(16, 30)):
        v = [x / d for x in dbg[8 * w: 8 * w + 8]]
        print("  %4d " % w + " ".join("%9.0f" % x for x in v[:7]) + "   %9.0f" % sum(v[:4]))
