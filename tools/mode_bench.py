#!/usr/bin/env python3
"""Time the loop kernel of single layer shapes under the four quantizers (GPU box only):
  python tools/mode_bench.py "112,672,51200" "320,1920,13312" ...      (first DLIMIT columns, default 256)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench_workload as bw  # noqa: E402
from quantized_neural_nets_amd import StepAlgorithm, _lib  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    for a in sys.argv[1:]:
        N, d, m = (int(v) for v in a.split(","))
        dl = min(d, int(os.environ.get("DLIMIT", "256")))
        W, A, X = bw.synthetic_layer(N, d, m, 99, d_limit=dl)
        step = bw.layer_step(W, K=2)
        W, A, X = W.to(dev), A.to(dev), X.to(dev)
        for name, reg, stoch in (("msq", None, False), ("soft", "L1", False), ("hard", "L0", False), ("stochastic", None, True)):
            ev = []

            def hook(tag, shape):
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                ev.append(e)
            best = 1e9
            for it in range(4):
                del ev[:]
                StepAlgorithm._quantize_layer_ex(W, A, X, m, 1.16 / 2, 2, 1, reg, 0.1, 1, stoch, dev, compute_errors=False,
                                                 step_override=step, event_hook=hook, seed=5)
                torch.cuda.synchronize()
                best = min(best, ev[1].elapsed_time(ev[2]))
            mode = {"msq": 0, "soft": 1, "hard": 2, "stochastic": 3}[name]
            print("N=%d d=%d m=%d %-10s %-52s %8.3f ms %7.3f us/col" % (N, dl, m, name, _lib.describe_plan(N, dl, m, 1, 0, mode), best, best * 1e3 / dl), flush=True)


if __name__ == "__main__":
    main()
