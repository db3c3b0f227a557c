#!/usr/bin/env python3
"""Projection of the neuron-sharded step from a shard sweep (tools/shard_sweep.py): for every layer of a workload and world
sizes 1, 2, 4, 8 the per-rank loop time = d x (microseconds per column measured on the rank's rows, AUTO's plan; and the
best forced configuration next to it), plus the column preparation every rank repeats in full (bytes at the measured
one-pass rate) and an allowance per layer for the int8 all_gather.

    python tools/shard_sweep_report.py gpurun_out/shard_sweep_r50_all.jsonl r50_all > profiles/r03_shard_projection_r50_all.json
"""
import collections
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench_workload as bw  # noqa: E402

ALLGATHER_US = 40.0          # small-message RCCL all_gather over xGMI (assumption; the one-rank RCCL path is tested, not timed)
PREP_TBPS = 4.5              # one-pass column preparation, measured (tools/prep_bench.py: 4.1-5.9 TB/s by shape)
LAUNCH_US = 12.0             # per layer: launches + the status read behind a cooperative launch


def main():
    path, workload = sys.argv[1], sys.argv[2]
    rows = [json.loads(l) for l in open(path)]
    by = collections.defaultdict(list)
    for r in rows:
        by[(r["N"], r["m"], r["world"])].append(r)
    fn, batch, desc = bw.WORKLOADS[workload][:3]
    layers = [l[:5] for l in bw.normalize_layers(fn(batch))]
    out = {"what": "PROJECTION from single-GPU timings of the per-rank shard shapes (tools/shard_sweep.py, %d timings); no "
                   "multi-GPU hardware run is behind it" % len(rows),
           "workload": "%s: %s" % (workload, desc),
           "assumptions": {"allgather_us_per_layer": ALLGATHER_US, "prep_TBps": PREP_TBPS, "launch_us_per_layer": LAUNCH_US,
                           "every_rank_prepares_all_columns": True, "ranks_do_not_interfere": True},
           "projected": {}, "per_layer": []}
    tot = {w: {"auto": 0.0, "best": 0.0, "prep": 0.0} for w in (1, 2, 4, 8)}
    for name, N, d, m, g in layers:
        if g != 1:
            sys.exit("grouped layers are not covered")
        mp = -(-m // 1024) * 1024
        prep_ms = 2 * (m + mp) * d * 4 / (PREP_TBPS * 1e9)
        rec = {"layer": name, "N": N, "d": d, "m": m, "prep_ms": round(prep_ms, 3), "worlds": {}}
        for w in (1, 2, 4, 8):
            grp = by[(N, m, w)]
            auto = [r for r in grp if r["cfg"] == "auto"][0]
            best = min(grp, key=lambda r: r["us_per_col"])
            a_ms, b_ms = auto["us_per_col"] * d * 1e-3, min(best["us_per_col"], auto["us_per_col"]) * d * 1e-3
            extra = (LAUNCH_US + (ALLGATHER_US if w > 1 else 0.0)) * 1e-3
            tot[w]["auto"] += a_ms + prep_ms + extra
            tot[w]["best"] += b_ms + prep_ms + extra
            tot[w]["prep"] += prep_ms
            rec["worlds"][str(w)] = {"rows": auto["rows"], "plan": auto["plan"], "us_per_col": auto["us_per_col"],
                                     "loop_ms": round(a_ms, 3), "best_plan": best["plan"], "best_us_per_col": best["us_per_col"]}
        out["per_layer"].append(rec)
    weights = sum(N * d for _, N, d, _, _ in layers)
    for w in (1, 2, 4, 8):
        out["projected"][str(w)] = {"ms_per_step": round(tot[w]["auto"], 2), "ms_per_step_best_config": round(tot[w]["best"], 2),
                                    "prep_ms": round(tot[w]["prep"], 2),
                                    "M_weights_per_s": round(weights / tot[w]["auto"] / 1e3, 1),
                                    "speedup_vs_1": round(tot[1]["auto"] / tot[w]["auto"], 2)}
    # the floor: what 8 GPUs could reach if the replicated preparation were free
    out["floor"] = {"loop_only_speedup_at_8": round((tot[1]["auto"] - tot[1]["prep"]) / (tot[8]["auto"] - tot[8]["prep"]), 2),
                    "reading": "a layer costs d sequential steps whatever the number of rows; a step is a latency chain (sweep, lane "
                               "tree, barrier, slot tree, quantizer: 0.44 us at 3 segments, 0.58 at 7) plus, where a row is split "
                               "over workgroups, one granule exchange (0.65-0.85 us fabric round trip + gather); fewer rows per GPU "
                               "only remove ROUNDS (layers whose rows do not fit the chip at once) and allow one- instead of two- "
                               "or four-row tiles (0.1-0.5 us per step)"}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
