"""diag: what does hipLaunchCooperativeKernel (GPFQ_COOP_LAUNCH_API=1) cost per launch against the plain launch?
One-column cooperative layers launched back to back without status reads; GPU time between stream events and host wall."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
import bench_workload as bw
from quantized_neural_nets_amd import StepAlgorithm as SA, _lib
dev = torch.device("cuda:0")
for (N, d, m) in ((64, 1, 93184), (128, 1, 26624)):
    W, A, X = bw.synthetic_layer(N, d, m, 3, first_layer=False)
    step = bw.layer_step(W)
    W, A, X = W.to(dev), A.to(dev), X.to(dev)
    for api in ("0", "1", "0", "1"):
        os.environ["GPFQ_COOP_LAUNCH_API"] = api
        n = 300
        try:
            for _ in range(20):
                SA._quantize_layer_ex(W, A, X, m, 1.16 / 8, 8, 1, None, 0.1, 1, False, dev, compute_errors=False, step_override=step, check_status=False)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter(); e0.record()
            for _ in range(n):
                SA._quantize_layer_ex(W, A, X, m, 1.16 / 8, 8, 1, None, 0.1, 1, False, dev, compute_errors=False, step_override=step, check_status=False)
            e1.record(); torch.cuda.synchronize(); t1 = time.perf_counter()
            _lib.check_status(dev)
            print("%s api=%s: %.2f us per layer call on the GPU timeline, %.2f us host wall (%s)" % ((N, d, m), api, e0.elapsed_time(e1) * 1e3 / n, (t1 - t0) * 1e6 / n, _lib.describe_plan(N, d, m)))
        except Exception as ex:
            print((N, d, m), "api=%s FAILED: %s" % (api, ex))
