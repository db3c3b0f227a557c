#!/usr/bin/env python3
"""Turn the rocprofv3 PMC passes of tools/profile_counters.sh (one counter_collection.csv per pass, each a separate run
of `bench.py --steps 1 --warmup 1`) into per-kernel counter averages per launch.

  python tools/pmc_counters.py <pass1 counter_collection.csv> <pass2 ...> ... > profiles/rNN_pmc_counters.json

Per kernel name: launches, and for every counter the average value per launch (rocprofv3 reports one value per
dispatch, summed over the chip's counter instances); the same again per (grid, workgroup) shape of the kernel, because
one template instantiation serves layers of different row length.  Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES,
SQ_BUSY_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count QUAD-cycles (4 shader clocks); SQ_INSTS_* count wave instructions.
The summary is stamped with the digest of the kernel sources it was collected on.
"""
import collections
import csv
import importlib.util
import json
import os
import sys


def main():
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    per_shape = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for path in sys.argv[1:]:
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"]
            if "gpfq" not in k:
                continue
            c, v = r["Counter_Name"], float(r["Counter_Value"])
            e = per[k][c]
            e[0] += 1
            e[1] += v
            shape = "grid=%s wg=%s" % (r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?"))
            e = per_shape[(k, shape)][c]
            e[0] += 1
            e[1] += v
    out = {}
    for k, cs in per.items():
        out[k] = {"launches": max(n for n, _ in cs.values()),
                  "per_launch": {c: round(s / n, 2) for c, (n, s) in sorted(cs.items())},
                  "shapes": {}}
    for (k, shape), cs in per_shape.items():
        out[k]["shapes"][shape] = {"launches": max(n for n, _ in cs.values()),
                                   "per_launch": {c: round(s / n, 2) for c, (n, s) in sorted(cs.items())}}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("_digest", os.path.join(root, "quantized_neural_nets_amd", "_digest.py"))
    dg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dg)
    json.dump({"command": "python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-output-check (one rocprofv3 --pmc pass per counter group, kernel-trace only)",
               "source_sha256": dg.kernel_source_digest(),
               "units": "SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_*: quad-cycles summed over waves (or SEs for BUSY); SQ_INSTS_*: wave instructions; TCC_* / TCP_*: requests (128-B lines unless the counter name says otherwise)",
               "kernels": out}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
