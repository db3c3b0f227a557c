#!/usr/bin/env python3
"""Column preparation on its own: the one-pass kernel (gpfq_prepare_columns_ws_f32; GPFQ_PREP_TC forces 32 / 64 columns
per workgroup) against the two-pass path (GPFQ_NO_FUSED_PREP=1), per layer shape, in TB/s of moved bytes
(2 matrices x (m x D read + m_pad x D written) x 4 B).   python tools/prep_bench.py [r50_3x3|effnet_b1|...]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench_workload as bw
from quantized_neural_nets_amd import _lib

dev = torch.device("cuda:0")
L = _lib.lib
wl = sys.argv[1] if len(sys.argv) > 1 else "r50_3x3"
layers = bw.normalize_layers(bw.WORKLOADS[wl][0](bw.WORKLOADS[wl][1]))
seen = set()
tot = {}
for name, N, dg, m, g in (l[:5] for l in layers):
    D = dg * g
    if (m, D) in seen:
        continue
    seen.add((m, D))
    mp = L.gpfq_padded_m(m)
    A = torch.randn(m, D, device=dev)
    X = torch.randn(m, D, device=dev)
    AT = torch.empty(D, mp, device=dev)
    XT = torch.empty(D, mp, device=dev)
    nrm = torch.empty(2 * D, device=dev)
    part = torch.empty(max(int(L.gpfq_prepare_ws_bytes(D, m)), 4) // 4, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    st = _lib.current_stream_ptr(dev)
    by = 2 * (m * D + mp * D) * 4
    row = "%-24s m=%8d D=%5d %7.1f MB:" % (name, m, D, by / 1e6)
    configs = (("two-pass", {"GPFQ_NO_FUSED_PREP": "1"}), ("tc64", {"GPFQ_PREP_TC": "64"}), ("tc32", {"GPFQ_PREP_TC": "32"}), ("auto", {}))
    if os.environ.get("PREP_CONFIGS"):              # "tag:K=V,K=V;tag:..." replaces the default set; outputs are compared with the first one's
        configs = tuple((c.split(":")[0], dict(kv.split("=") for kv in c.split(":")[1].split(",") if kv)) for c in os.environ["PREP_CONFIGS"].split(";"))
    ref = None
    for tag, env in configs:
        for k in [k for k in os.environ if k.startswith("GPFQ_") and k != "GPFQ_LIB_OVERRIDE"]:
            os.environ.pop(k, None)
        os.environ.update(env)
        def run():
            _lib.check(L.gpfq_prepare_columns_ws_f32(p(A), D, p(X), D, m, D, p(AT), p(XT), p(nrm), mp, p(part), part.numel() * 4, st))
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 5
        e0.record()
        for _ in range(n):
            run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        tot[tag] = tot.get(tag, 0.0) + ms
        same = ""
        if ref is None:
            ref = (AT.clone(), XT.clone(), nrm.clone())
        elif not (torch.equal(ref[0], AT) and torch.equal(ref[1], XT) and torch.equal(ref[2], nrm)):
            same = " DIFFERENT"
        row += "  %s %.3f ms %.2f TB/s%s" % (tag, ms, by / ms / 1e9, same)
    print(row, flush=True)
    del A, X, AT, XT, ref
print("total ms:", {k: round(v, 3) for k, v in tot.items()})
