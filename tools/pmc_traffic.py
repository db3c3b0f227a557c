#!/usr/bin/env python3
"""Turn the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel-trace only) of
`bench.py --steps 1 --warmup 1` into per-kernel HBM-side bytes per launch.

  python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> > profiles/rNN_pmc_traffic.json

Correction (MI355X_MICROARCH.md, HBM section): on gfx950 FETCH_SIZE counts 64 B per 128-B request, i.e. reads
exactly half of a wide coalesced stream (checked here on gpfq_transpose_pad_kernel, whose byte count is known:
it reads 2*m*d*4 B and writes 2*m_pad*d*4 B) -> bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.  Infinity-Cache
hits are included in the counters (they sit on the fabric side of the L2s), so this is an upper bound of DRAM traffic.
"""
import collections
import csv
import json
import sys


def agg(path):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        d[r["Kernel_Name"]][0] += 1
        d[r["Kernel_Name"]][1] += float(r["Counter_Value"])
    return d


def main():
    f, w = agg(sys.argv[1]), agg(sys.argv[2])
    out = {}
    for k in f:
        if "gpfq" not in k:
            continue
        n = f[k][0]
        fetch_kb = f[k][1] / n
        write_kb = w[k][1] / w[k][0] if k in w else 0.0
        out[k] = {"launches": n, "FETCH_SIZE_KB_avg": round(fetch_kb, 1), "WRITE_SIZE_KB_avg": round(write_kb, 1),
                  "hbm_bytes_per_launch": int((2 * fetch_kb + write_kb) * 1024)}
    import importlib.util
    import os
    # the digest of the kernel sources these counters were collected on (bench.py refuses a summary of other sources);
    # the module is loaded by path so that neither torch nor the built library is needed here
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("_digest", os.path.join(root, "quantized_neural_nets_amd", "_digest.py"))
    dg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dg)
    json.dump({"command": "python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-output-check",
               "source_sha256": dg.kernel_source_digest(),
               "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes (gfx950 FETCH_SIZE correction)", "kernels": out},
              sys.stdout, indent=1)


if __name__ == "__main__":
    main()
