import os, sys, ctypes
import numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import golden_inputs as gi, oracle
from quantized_neural_nets_amd import StepAlgorithm as SA, _lib
dev = torch.device("cuda:0")
cases = [(16, 8, 40000, 4, 16), (16, 8, 40000, 2, 8), (16, 8, 40000, 2, 16), (16, 8, 40000, 1, 16), (16, 8, 40000, 4, 8),
         (8, 5, 3072, 4, 2), (8, 5, 3072, 2, 2), (8, 5, 3072, 1, 2), (8, 5, 6144, 2, 2), (8, 5, 6144, 2, 4), (8, 5, 12288, 2, 2), (8, 5, 12288, 2, 4)]
for (N, d, m, rt, c) in cases:
    os.environ["GPFQ_COOP_RT"] = str(rt); os.environ["GPFQ_COOP_C"] = str(c)
    case = dict(name="dbg", N=N, d=d, m=m, bits=4, scalar=1.16, percentile=1.0, reg=None, lamb=0.0, groups=1, first_layer=False, zero_every=0, seed=1)
    W, A, X = gi.make_inputs(case)
    try:
        desc = _lib.describe_plan(N, d, m, 1, 3)
    except Exception as e:
        print(N, d, m, rt, c, "no plan", e); continue
    r = SA._quantize_layer_ex(torch.from_numpy(W).to(dev), torch.from_numpy(A).to(dev), torch.from_numpy(X).to(dev), m, 1.16/8, 8, 1.0, None, 0.0, 1, False, dev, plan=3, compute_errors=False, check_status=False)
    torch.cuda.synchronize()
    scr = _lib.scratch(dev)
    st = scr[96*1024: 96*1024+64].view(torch.int32).cpu().tolist()
    o = oracle.quantize_layer(W, A, X, 1.16/8, 8)
    idx = r["idx"].cpu().numpy().astype(np.int16)
    bad = (idx != o["idx"])
    print(N, d, m, desc, "| status", st[:4], "| idx ok" if not bad.any() else "| idx BAD first bad col per row %s" % [int(np.argmax(b)) if b.any() else -1 for b in bad],
          "| U ok" if np.array_equal(r["U"].cpu().numpy(), o["U"]) else "| U BAD", flush=True)
    _lib.status_ok(dev)
