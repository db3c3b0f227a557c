"""diag: is quantize_network() on resnet50 @ batch 32 reproducible run to run (fused, fused, unfold)?  per-layer digests of A, X, idx"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import quantized_neural_nets_amd.quantize_neural_net as qnn_mod
from quantized_neural_nets_amd import QuantizeNeuralNet, StepAlgorithm, arch
from quantized_neural_nets_amd.main import SyntheticLoader
from quantized_neural_nets_amd.step_algorithm import PreparedColumns
DEV = "cuda:0"
torch.backends.cudnn.deterministic = True
torch.backends.cudnn.benchmark = False

def digest(t):
    v = t.contiguous().view(torch.int32).to(torch.int64)
    return int((v * (torch.arange(v.numel(), device=v.device).view(v.shape) % 1000003 + 1)).sum().item())

def run(fused, model_name="resnet50", batch=32, seed=0):
    torch.manual_seed(seed); np.random.seed(seed)
    model = arch.ARCHITECTURES[model_name]().to(DEV).eval()
    q = QuantizeNeuralNet(model, model_name, batch, SyntheticLoader(batch, 224, seed + 1), 4, 4, [], 1.16, 1.16, 1, 1, None, 0.1, 0.25, False, torch.device(DEV))
    real = StepAlgorithm._quantize_layer_ex
    recs = []
    def checked(W, A, X, m, *a, **kw):
        res = real(W, A, X, m, *a, **kw)
        Am = A.matrix() if isinstance(A, PreparedColumns) else A
        Xm = X.matrix() if isinstance(X, PreparedColumns) else X
        recs.append((tuple(W.shape), int(m), digest(Am), digest(Xm), digest(res["idx"].to(torch.int32).float()), float(res["step"])))
        return res
    qnn_mod.FUSED_CAPTURE = fused
    StepAlgorithm._quantize_layer_ex = checked
    import io, contextlib
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            q.quantize_network()
    finally:
        StepAlgorithm._quantize_layer_ex = real
        qnn_mod.FUSED_CAPTURE = True
    return recs

a, b, c = run(True), run(True), run(False)
for name, x, y in (("fused vs fused", a, b), ("fused vs unfold", a, c)):
    for i, (u, v) in enumerate(zip(x, y)):
        if u != v:
            print(name, "first difference at layer", i, "shape", u[0], "m", u[1], "A equal", u[2] == v[2], "X equal", u[3] == v[3], "idx equal", u[4] == v[4], "step equal", u[5] == v[5])
            break
    else:
        print(name, "identical over", len(x), "layers")
