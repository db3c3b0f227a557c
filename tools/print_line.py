#!/usr/bin/env python3
"""One-line summary of bench.py JSON lines:  python tools/print_line.py file.json [...]"""
import json
import sys

for path in sys.argv[1:]:
    j = json.loads(open(path).read().strip().splitlines()[-1])
    print("%-52s %9.2f %s  %8.3f ms/step  kernel %s frac %.3f traffic %s  l2 frac %s  output mismatches %s" % (
        path, j["value"], j["unit"], j["ms_per_step"], j["roofline"].get("kernel"), j["roofline"].get("frac") or 0,
        j["roofline"].get("traffic"), (j.get("roofline_l2") or {}).get("frac"), (j.get("output_check") or {}).get("mismatches")))
