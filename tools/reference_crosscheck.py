#!/usr/bin/env python3
"""SURVEY 8(d)'s cross-check, build container only: the TRUE reference loop (StepAlgorithm._quantization imported from
/root/reference/src/step_algorithm.py:107-148) and the torch-op restatement that bench.py's cpu_baseline times on the GPU
box (oracle/gpfq_oracle.py torch_restatement_quantization) on the SAME inputs, the same columns and the same thread count,
alternating, best of `reps`: the two timings must agree within ~10 %, or the baseline quoted next to the GPU numbers is not
the reference's cost.  Also checks that both produce the same Q and U (they are the same op sequence).

    python tools/reference_crosscheck.py [--threads T] [--reps R] > profiles/r04_reference_crosscheck.txt
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import make_golden as mg
    import oracle
    if not os.path.isdir(mg.REF_SRC):
        sys.exit("reference not present: this tool only runs in the build container")
    SA = mg.import_reference_step_algorithm()
    torch.set_num_threads(args.threads)
    print("host: %d cores, torch %s, %d threads, best of %d alternating runs" % (os.cpu_count(), torch.__version__, args.threads, args.reps))
    print("%-34s %10s %12s %12s %8s" % ("shape (ResNet-50 3x3 @ batch 1024)", "columns", "reference s", "restatement s", "ratio"))
    worst = 0.0
    tot_ref = tot_res = tot_w = 0.0
    for (N, d, m, steps) in [(512, 4608, 3072, 256), (256, 2304, 7168, 256), (128, 1152, 26624, 128), (64, 576, 93184, 64)]:
        g = torch.Generator().manual_seed(1234)
        W = torch.randn(N, steps, generator=g) * (2.0 / d) ** 0.5
        pre = torch.randn(m, steps, generator=g)
        A = torch.relu(pre)
        X = torch.relu(pre + 0.05 * torch.randn(m, steps, generator=g))
        step = torch.tensor(1.16 / 8) * W.abs().max(dim=1).values.mean()
        best_ref = best_res = 1e30
        for _ in range(args.reps):
            Q1, U1 = torch.zeros_like(W), torch.zeros(N, m)
            t0 = time.perf_counter()
            mg.quiet(SA._quantization, W, Q1, U1, A, X, SA._msq, step, 8, 0.0)
            best_ref = min(best_ref, time.perf_counter() - t0)
            Q2, U2 = torch.zeros_like(W), torch.zeros(N, m)
            t0 = time.perf_counter()
            oracle.torch_restatement_quantization(W, Q2, U2, A, X, step, 8)
            best_res = min(best_res, time.perf_counter() - t0)
        assert torch.equal(Q1, Q2) and torch.equal(U1, U2), "the restatement is not the reference's op sequence"
        ratio = best_res / best_ref
        worst = max(worst, abs(ratio - 1.0))
        tot_ref += best_ref; tot_res += best_res; tot_w += N * steps
        print("N=%-4d d=%-5d m=%-6d %14d %12.3f %12.3f %8.3f   (%.4f / %.4f M weights/s; Q and U bit-equal)" % (
            N, d, m, steps, best_ref, best_res, ratio, N * steps / best_ref / 1e6, N * steps / best_res / 1e6))
    print("all four: reference %.4f M weights/s, restatement %.4f M weights/s, ratio %.3f; worst per-shape deviation %.1f %%  -> %s" % (
        tot_w / tot_ref / 1e6, tot_w / tot_res / 1e6, tot_res / tot_ref, worst * 100, "AGREE (within 10 %)" if worst <= 0.10 else "DISAGREE"))


if __name__ == "__main__":
    main()
