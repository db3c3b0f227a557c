#!/usr/bin/env python3
"""Randomized soak of the loop kernels against the CPU oracle (GPU box only; test infrastructure, not product).
Random (N, d, m, groups, mode, bits, plan) cases -- small enough for the oracle -- must agree bit for bit in idx, Q, U.
    python tools/soak.py [cases=200] [seed=1]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_inputs as gi  # noqa: E402
from oracle import gpfq_oracle as oracle  # noqa: E402
from quantized_neural_nets_amd import StepAlgorithm, _lib  # noqa: E402


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    kinds = {}
    t0 = time.time()
    bad = 0
    for ci in range(ncases):
        fam = rng.choice(["wave", "resident", "coop", "coop_rows", "rounds", "stream", "grouped", "depthwise", "chip_wide", "pipe", "pipe", "lock", "pipel", "pipel"])
        groups = 1
        if fam == "wave":                           # one-segment rows: the resident kernel's one-wave variant, 1 / 2 / 4 rows per wave
            N, m = int(rng.integers(1, 300)), int(rng.integers(1, 1025))
            os.environ["GPFQ_RESIDENT_RT"] = str(int(rng.choice([1, 2, 4])))
        elif fam == "resident":
            N, m = int(rng.integers(1, 40)), int(rng.integers(1025, 16385))
            # rows per workgroup: forced, so that the 2- and 4-row kernels (chosen on their own only from 512 rows on) and
            # their ragged last tiles are covered at sizes the oracle finishes quickly
            os.environ["GPFQ_RESIDENT_RT"] = str(int(rng.choice([1, 2, 4])))
        elif fam == "coop":
            N, m = int(rng.integers(1, 24)), int(rng.integers(16385, 60000))
        elif fam == "coop_rows":                    # enough rows for the 2- and 4-row cooperative variants
            N, m = int(rng.integers(24, 300)), int(rng.integers(16385, 100000))
        elif fam == "rounds":                       # long rows, a forced (rows, members) pair: more tiles than fit one launch
            N, m = int(rng.integers(20, 200)), int(rng.integers(30000, 400000))
            rt = int(rng.choice([1, 2, 4]))
            os.environ["GPFQ_COOP_RT"] = str(rt)
            # (every member count a variant gathers: up to 256 for one and two rows, 64 and 256 -- not 128 -- for four)
            os.environ["GPFQ_COOP_C"] = str(int(rng.choice([c for c in (8, 16, 32, 64, 128, 256) if rt < 4 or c != 128])))
            if int(os.environ["GPFQ_COOP_C"]) >= 128:   # rows long enough for that many members (one segment each at least)
                m = int(rng.integers(270000, 700000))
        elif fam == "chip_wide":                    # 256 granules, four gathered per lane: four rows x 64 members (columns
            if rng.integers(0, 2):                  # staged through LDS), or one row on 256 members; AUTO picks both
                N, m = int(rng.integers(5, 40)), int(rng.integers(786433, 830000))
            else:
                N, m = int(rng.integers(1, 5)), int(rng.integers(2700000, 3300000))
        elif fam == "pipe":                         # the pipelined cooperative kernels (round 4): four groups of one row or of a pair,
            N, m = int(rng.integers(4, 300)), int(rng.integers(16385, 260000))   # any member count they take, both reducer arrangements
            os.environ["GPFQ_COOP_PIPE"] = "1"
            os.environ["GPFQ_COOP_RT"] = str(int(rng.choice([4, 8])))
            if rng.integers(0, 3) == 0:
                os.environ["GPFQ_PIPE_LOCAL"] = "0"  # device-scope publishing throughout
            if rng.integers(0, 2):
                os.environ["GPFQ_COOP_C"] = str(int(rng.choice([2, 4, 8, 16, 32, 64])))
        elif fam == "pipel":                        # twelve rows in three groups, columns through LDS (round 5): every member count,
            N = int(rng.integers(1, 120))           # ragged last tiles, rounds
            c = int(rng.choice([2, 4, 8, 16, 32, 64, 128]))
            segs = int(rng.integers(max(c, 2), 7 * c + 1))                        # 1 .. 7 segments per member
            m = 1024 * (segs - 1) + int(rng.integers(1, 1025))
            os.environ["GPFQ_COOP_PIPEL"] = "1"
            os.environ["GPFQ_COOP_C"] = str(c)
        elif fam == "lock":                         # the lock-step kernels where AUTO would pipeline: tile counts that are multiples
            rt = int(rng.choice([1, 2, 4]))         # of eight (members of a tile placed on one XCD), and some that are not
            N = rt * 8 * int(rng.integers(1, 5)) - (int(rng.integers(0, rt * 8)) if rng.integers(0, 4) == 0 else 0)
            m = int(rng.integers(16385, 120000))
            os.environ["GPFQ_COOP_PIPE"] = "0"
            os.environ["GPFQ_COOP_RT"] = str(rt)
            os.environ["GPFQ_COOP_C"] = str(int(rng.choice([2, 4, 8, 16, 32])))
        elif fam == "stream":
            N, m = int(rng.integers(1, 12)), int(rng.integers(16385, 90000))
        elif fam == "depthwise":                    # one long row per group: the cooperative one-row kernel's grouped variant
            groups = int(rng.integers(2, 60))
            N, m = groups, int(rng.integers(17000, 200000))
        else:
            groups = int(rng.choice([2, 3, 4]))
            N, m = groups * int(rng.integers(1, 8)), int(rng.integers(1, 6000))
        d = int(rng.integers(1, 5 if fam == "chip_wide" else 7 if fam == "rounds" else 10 if fam == "depthwise" else 12 if fam in ("coop_rows", "pipe", "lock") else
                             (6 if m > 300000 else 12) if fam == "pipel" else 40))
        bits = int(rng.choice([2, 3, 4]))
        reg = [None, "L1", "L0"][int(rng.integers(0, 3))]
        plan = 1 if fam == "stream" else 3 if fam == "pipel" else 0      # (3: the cooperative family asked for, short rows too)
        case = dict(name="soak%d" % ci, N=N, d=d, m=m, bits=bits, scalar=1.16, percentile=1.0, reg=reg, lamb=0.02,
                    groups=groups, first_layer=bool(rng.integers(0, 2)), zero_every=int(rng.choice([0, 3, 7])),
                    seed=int(rng.integers(0, 1 << 30)))
        W, A, X = gi.make_inputs(case)
        K = 2 ** (bits - 1)
        r = StepAlgorithm._quantize_layer_ex(torch.from_numpy(W).to(dev), torch.from_numpy(A).to(dev),
                                             torch.from_numpy(X).to(dev), m, 1.16 / K, K, 1.0, reg, 0.02, groups, False,
                                             dev, compute_errors=False, plan=plan)
        torch.cuda.synchronize()
        _lib.check_status(dev)
        os.environ.pop("GPFQ_RESIDENT_RT", None)
        o = oracle.quantize_layer(W, A, X, 1.16 / K, K, 1.0, reg, 0.02, groups)
        full = _lib.describe_plan(N, d, m, groups, plan)
        desc = (full.split(" S=")[0] if full.startswith("coop") else full.split()[0]) + (
            "+groups" if "groups=" in full else "+rounds" if "rounds=" in full else "") + ("+pipe" if "pipe=1" in full else "+pipel" if "pipel=1" in full else "")
        kinds[desc] = kinds.get(desc, 0) + 1
        for k in ("GPFQ_COOP_RT", "GPFQ_COOP_C", "GPFQ_COOP_PIPE", "GPFQ_PIPE_LOCAL", "GPFQ_COOP_PIPEL"):
            os.environ.pop(k, None)
        ok = (np.array_equal(r["idx"].cpu().numpy().astype(np.int16), o["idx"])
              and np.array_equal(r["Q"].cpu().numpy().view(np.uint32), o["Q"].view(np.uint32))
              and np.array_equal(r["U"].cpu().numpy(), o["U"])
              and torch.allclose(r["usq_seg"].double().sum(1), (r["U"].double() ** 2).sum(1), rtol=1e-5, atol=1e-30))
        if not ok:
            bad += 1
            print("MISMATCH", case, desc, flush=True)
        if ci % 25 == 24:
            print("%d cases, %d mismatches, %.0f s, plans %s" % (ci + 1, bad, time.time() - t0, kinds), flush=True)
    print("done: %d cases, %d mismatches, plans %s" % (ncases, bad, kinds))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
