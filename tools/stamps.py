#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the cooperative kernel (needs `make -C quantized_neural_nets_amd/csrc stamps`).
   GPFQ_LIB_OVERRIDE=.../libgpfq_hip_stamps.so python tools/stamps.py N,d,m [ENV=..]"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench_workload as bw
from quantized_neural_nets_amd import StepAlgorithm, _lib
names = ["top (q from LDS, last quarter of the loads)", "wait+sweep", "lane tree + LDS + 3/4 of the loads", "barrier1", "(reducer) enter", "(reducer) slot tree + exchange + quantizer", "leave", "barrier2"]
dev = torch.device("cuda:0")
for a in sys.argv[1:]:
    if "=" in a:
        k, v = a.split("="); os.environ[k] = v; continue
    N, d, m = (int(v) for v in a.split(","))
    dl = min(d, 512)
    W, A, X = bw.synthetic_layer(N, d, m, 99, d_limit=dl)
    step = bw.layer_step(W)
    loop_ms = []
    def hook(tag, shape):                        # events around the loop kernel: cycles / time = the clock it ran at
        if tag in ("loop_begin", "loop_end"):
            e = torch.cuda.Event(enable_timing=True); e.record(); loop_ms.append(e)
    for it in range(2):
        del loop_ms[:]
        StepAlgorithm._quantize_layer_ex(W.to(dev), A.to(dev), X.to(dev), m, 1.16 / 8, 8, 1, None, 0.1, 1, False, dev,
                                         compute_errors=False, step_override=step, event_hook=hook)
    torch.cuda.synchronize()
    ms = loop_ms[0].elapsed_time(loop_ms[1])
    scr = _lib.scratch(dev)
    dbg = scr[96 * 1024 + 64: 96 * 1024 + 64 + 128].view(torch.int64).cpu().tolist()
    desc = _lib.describe_plan(N, dl, m)
    print(a, desc, "loop %.3f ms = %.3f us/step" % (ms, ms * 1e3 / dl))
    if desc.startswith("resident"):
        names = ["loop bookkeeping, last quarter of the loads, history", "wait for column t (vmcnt)", "sweeps", "first quarter of the loads, lane trees, LDS write, second quarter",
                 "barrier", "third quarter, LDS read, slot tree, divisions, quantizer, readlanes", "-", "-"]
    who = (("wave0", 0), ("lastwave", 8))
    if "pipe=1" in desc:
        # pipelined cooperative kernels: cycles per PHASE (four phases per step), reducer wave and sweep wave 0 of workgroup 0
        dl = 4 * dl
        who = (("reducer", 0), ("sweep0", 8))
        who = (("gatherer", 0), ("sweep0", 8))
        rn = ["barrier", "requested gather lands", "re-polls", "next request", "tree + quantizer + q to LDS", "-", "PUBLISHER barrier", "PUBLISHER slot tree + publish"]
        sn = ["barrier", "q from LDS + column wait", "sweep", "lane tree + LDS", "column / weight requests", "-", "-", "-"]
    if "pipel=1" in desc:
        # twelve rows in three groups, columns through LDS: cycles per PHASE (three per step), reducer wave and sweep wave 0 of workgroup 0
        dl = 3 * min(d, 512)
        who = (("reducer", 0), ("sweep0", 8))
        rn = ["barrier", "slot tree + publish", "requested gather lands", "re-polls", "tree + quantizer + q to LDS", "pause + next request", "-", "-"]
        sn = ["barrier", "q from LDS + column wait + DMA", "sweep (2 pairs, LDS reads)", "lane tree + LDS", "weight requests", "-", "-", "-"]
    for w, off in who:
        if "pipe=1" in desc or "pipel=1" in desc:
            names = rn if off == 0 else sn
        tot = sum(dbg[off:off + 8])
        print("  %-8s total %.0f cyc/step (clock %.2f GHz):" % (w, tot / dl, tot / dl / (ms * 1e3 / dl) / 1e3), "  ".join("%s %.0f" % (names[i], dbg[off + i] / dl) for i in range(8)))
