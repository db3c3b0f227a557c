import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import golden_inputs as gi, oracle
from quantized_neural_nets_amd import StepAlgorithm as SA, _lib
dev = torch.device("cuda:0")
for (N, d, m, rt, c) in [(4, 5, 3072, 2, 2), (4, 5, 3072, 2, 2), (4, 5, 6144, 2, 2), (6, 5, 3072, 2, 2), (4, 5, 3072, 1, 2), (2, 5, 3072, 1, 2), (16, 8, 40000, 1, 16), (16, 8, 40000, 1, 8), (32, 8, 40000, 1, 8)]:
    os.environ["GPFQ_COOP_RT"] = str(rt); os.environ["GPFQ_COOP_C"] = str(c)
    case = dict(name="dbg", N=N, d=d, m=m, bits=4, scalar=1.16, percentile=1.0, reg=None, lamb=0.0, groups=1, first_layer=False, zero_every=0, seed=1)
    W, A, X = gi.make_inputs(case)
    desc = _lib.describe_plan(N, d, m, 1, 3)
    r = SA._quantize_layer_ex(torch.from_numpy(W).to(dev), torch.from_numpy(A).to(dev), torch.from_numpy(X).to(dev), m, 1.16/8, 8, 1.0, None, 0.0, 1, False, dev, plan=3, compute_errors=False, check_status=False)
    torch.cuda.synchronize()
    o = oracle.quantize_layer(W, A, X, 1.16/8, 8)
    idx = r["idx"].cpu().numpy().astype(np.int16)
    U = r["U"].cpu().numpy()
    bad = (idx != o["idx"])
    print(N, d, m, desc, "first bad col per row", [int(np.argmax(b)) if b.any() else -1 for b in bad], "U nan rows", np.isnan(U).any(1).astype(int).tolist(), flush=True)
    if bad.any():
        i = int(np.argmax(bad.any(1)))
        print("  row", i, "idx", idx[i].tolist(), "want", o["idx"][i].tolist(), "Q", r["Q"].cpu().numpy()[i].tolist())
    scr = _lib.scratch(dev); print("  status", scr[96*1024: 96*1024+64].view(torch.int32).cpu().tolist()); scr[96*1024:96*1024+64].zero_()
