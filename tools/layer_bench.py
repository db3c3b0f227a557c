#!/usr/bin/env python3
"""Time the loop kernel of single layer shapes under different plan settings (GPU box only).
  python tools/layer_bench.py "64,576,93184" "plan=0" "plan=0,GPFQ_COOP_RT=2,GPFQ_COOP_WGS_PER_CU=2" ...
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench_workload as bw  # noqa: E402
from quantized_neural_nets_amd import StepAlgorithm, _lib  # noqa: E402


def main():
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:] if "=" not in a]
    configs = [a for a in sys.argv[1:] if "=" in a] or ["plan=0"]
    dev = torch.device("cuda:0")
    for (N, d, m) in shapes:
        dl = min(d, int(os.environ.get("DLIMIT", "512")))
        W, A, X = bw.synthetic_layer(N, d, m, 99, d_limit=dl)
        step = bw.layer_step(W)
        W, A, X = W.to(dev), A.to(dev), X.to(dev)
        ref = None
        for cfg in configs:
            kv = dict(x.split("=") for x in cfg.split(","))
            plan = int(kv.pop("plan", "0"))
            for k in list(os.environ):
                if k.startswith("GPFQ_") and k != "GPFQ_LIB_OVERRIDE":
                    del os.environ[k]
            os.environ.update(kv)
            ev = []

            def hook(tag, shape):
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                ev.append(e)
            try:
                desc = _lib.describe_plan(N, dl, m, 1, plan)
                best = 1e9
                for it in range(4):
                    del ev[:]
                    r = StepAlgorithm._quantize_layer_ex(W, A, X, m, 1.16 / 8, 8, 1, None, 0.1, 1, False, dev,
                                                         compute_errors=False, step_override=step, plan=plan,
                                                         event_hook=hook)
                    torch.cuda.synchronize()
                    best = min(best, ev[1].elapsed_time(ev[2]))
                _lib.check_status(dev)
                same = ""
                if ref is None:
                    ref = r["idx"].clone()
                else:
                    same = "same" if torch.equal(ref, r["idx"]) else "DIFFERENT"
                print("N=%d d=%d m=%d %-44s %-40s %8.3f ms  %6.3f us/col %s" % (N, dl, m, cfg, desc, best, best * 1e3 / dl, same),
                      flush=True)
            except _lib.GpfqError as e:
                print("N=%d m=%d %-44s ERROR %s" % (N, m, cfg, e), flush=True)


if __name__ == "__main__":
    main()
