#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING THE REFERENCE in the build container.

Container-only: /root/reference does not exist on the GPU box and nothing at test/bench time runs this.
The fixtures hold data only (expected outputs + the digest of the regenerable inputs); no reference source.

  python tools/make_golden.py            # all loop cases (G1-G4) + driver case (G5) + G6 + G7
  python tools/make_golden.py --bench    # also time the true reference loop on the ResNet-50 3x3 shapes

What is called (reference paths relative to /root/reference/src):
  StepAlgorithm._msq/_soft_thresholding_msq/_hard_thresholding_msq   step_algorithm.py:38-104   -> g1_quantizers.npz
  StepAlgorithm._quantize_layer                                     step_algorithm.py:151-249  -> g2/g3/g4 *.npz
  StepAlgorithm._quantization (per group, for the residual U)       step_algorithm.py:107-148
  QuantizeNeuralNet(...).quantize_network()                         quantize_neural_net.py:32-214 -> g5_driver.npz
  StepAlgorithm._quantize_layer at headline scale, NO seed search   step_algorithm.py:151-249  -> g6_*.npz (idx + U digests)
  StepAlgorithm._stochastic_msq (draw-independent points + frequencies) step_algorithm.py:7-35  -> g7_stochastic.npz
"""
import argparse
import io
import json
import os
import sys
import time
import types
import contextlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF_SRC = "/root/reference/src"

import golden_inputs as gi  # noqa: E402


def import_reference_step_algorithm():
    sys.path.insert(0, REF_SRC)
    try:
        import step_algorithm  # the reference's own module
    finally:
        sys.path.pop(0)
    return step_algorithm.StepAlgorithm


def provenance():
    return dict(torch=torch.__version__, numpy=np.__version__,
                blas="mkl" if torch.backends.mkl.is_available() else "other",
                threads=torch.get_num_threads(), generated="tools/make_golden.py (reference imported in container)")


def quiet(fn, *a, **k):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(buf):
        return fn(*a, **k)


# ------------------------------------------------------------------------------------------------ G1
def gen_quantizers(SA):
    rng = np.random.default_rng(42)
    kat = np.array([-1, -.625, -.375, -.125, -.1, -0.0, 0.0, .1, .125, .374, .375, .625, .7, 5], np.float32)
    out = {}
    cfgs = []
    for ci, (step, K, lamb) in enumerate([(0.25, 2, 0.1), (0.0371, 8, 0.01), (0.113, 1, 0.05), (0.5, 128, 0.3)]):
        x = np.concatenate([kat, (rng.standard_normal(400) * step * K * 0.7).astype(np.float32),
                            (np.arange(-40, 41, dtype=np.float32) * np.float32(step) * np.float32(0.5)),
                            np.array([lamb, -lamb, np.nextafter(np.float32(lamb), np.float32(1)),
                                      -np.nextafter(np.float32(lamb), np.float32(1))], np.float32)]).astype(np.float32)
        xt = torch.from_numpy(x)
        st = torch.tensor(step, dtype=torch.float32)   # the reference passes a 0-dim fp32 tensor (step*rad)
        out["x_%d" % ci] = x
        out["msq_%d" % ci] = SA._msq(st, xt.clone(), K, lamb).numpy()
        out["soft_%d" % ci] = SA._soft_thresholding_msq(st, xt.clone(), K, lamb).numpy()
        out["hard_%d" % ci] = SA._hard_thresholding_msq(st, xt.clone(), K, lamb).numpy()
        cfgs.append(dict(step=float(np.float32(step)), K=K, lamb=lamb))
    out["meta"] = np.array(json.dumps(dict(configs=cfgs, provenance=provenance())))
    np.savez_compressed(os.path.join(gi.GOLDEN_DIR, "g1_quantizers.npz"), **out)
    print("g1_quantizers: %d configs" % len(cfgs))


# ------------------------------------------------------------------------------------------------ G7
def gen_stochastic(SA):
    """The reference's _stochastic_msq (step_algorithm.py:7-35) as far as a random quantizer can be pinned:
    (i)  arguments whose answer does not depend on the draw -- on the alphabet grid (p = 1: bernoulli(1) always rounds
         down, to the argument itself), beyond the alphabet (|x| >= step*K: both neighbours clip to +-step*K), +-0 --
         outputs kept bitwise;
    (ii) for 64 arguments inside the alphabet, BOTH values the reference ever returns (bitwise: step*floor(x/step) and
         step*(floor(x/step)+1)) and how often it returned the lower one over n seeded draws (torch.bernoulli on
         torch's global CPU generator: the stream itself cannot be reproduced anywhere else, its frequencies can)."""
    out, cfgs = {}, []
    n = 1 << 17
    for ci, (step, K) in enumerate([(0.25, 2), (0.0371, 8), (0.113, 1), (0.5, 128)]):
        rng = np.random.default_rng(4242 + ci)
        stf = np.float32(step)
        st = torch.tensor(step, dtype=torch.float32)
        # (i) draw-independent arguments
        grid = (np.arange(-K - 3, K + 4, dtype=np.float32) * stf).astype(np.float32)        # z integer: p == 1
        grid = grid[(grid / stf) == np.floor(grid / stf)]                                   # (keep those whose fp32 quotient IS an integer)
        beyond = np.concatenate([(np.float32(K) + rng.uniform(0.01, 6.0, 40).astype(np.float32)) * stf,
                                 -(np.float32(K) + rng.uniform(0.01, 6.0, 40).astype(np.float32)) * stf,
                                 np.array([5.0, -5.0, 1000.0, -1000.0], np.float32) * max(1.0, float(K) * step)]).astype(np.float32)
        det = np.concatenate([grid, beyond, np.array([0.0, -0.0], np.float32)]).astype(np.float32)
        torch.manual_seed(1000 + ci)
        r1 = SA._stochastic_msq(st, torch.from_numpy(det.copy()), K, 0.0).numpy().copy()
        torch.manual_seed(2000 + ci)
        r2 = SA._stochastic_msq(st, torch.from_numpy(det.copy()), K, 0.0).numpy().copy()
        assert np.array_equal(r1.view(np.uint32), r2.view(np.uint32)), "a 'deterministic' argument depends on the draw"
        out["det_x_%d" % ci], out["det_q_%d" % ci] = det, r1
        # (ii) 64 arguments strictly inside the alphabet, off the grid
        z = rng.uniform(-K, K, 64).astype(np.float32)
        z[:8] = np.floor(z[:8]) + np.array([0.001, 0.01, 0.1, 0.25, 0.5, 0.75, 0.9, 0.999], np.float32)
        x = (z * stf).astype(np.float32)
        zz = x / stf
        x = x[(zz != np.floor(zz)) & (np.floor(zz) >= -K) & (np.floor(zz) + 1 <= K)]
        torch.manual_seed(3000 + ci)
        big = torch.from_numpy(np.repeat(x[:, None], n, axis=1).copy())          # (len(x), n): every argument n times
        q = SA._stochastic_msq(st, big.reshape(-1), K, 0.0).reshape(len(x), n).numpy()
        lo, hi = q.min(axis=1), q.max(axis=1)
        assert np.all((q == lo[:, None]) | (q == hi[:, None])) and np.all(lo < hi)
        out["rnd_x_%d" % ci] = x
        out["rnd_lo_%d" % ci], out["rnd_hi_%d" % ci] = lo.astype(np.float32), hi.astype(np.float32)
        out["rnd_down_count_%d" % ci] = (q == lo[:, None]).sum(axis=1).astype(np.int64)
        cfgs.append(dict(step=float(stf), K=K, draws=n, torch_seed=3000 + ci))
    out["meta"] = np.array(json.dumps(dict(configs=cfgs, provenance=provenance())))
    np.savez_compressed(os.path.join(gi.GOLDEN_DIR, "g7_stochastic.npz"), **out)
    print("g7_stochastic: %d configs, %d draws per argument" % (len(cfgs), n))


# ------------------------------------------------------------------------------------------ G2/G3/G4
def fp64_margin(case, W, A, X, Q, step):
    """Replay the recurrence in float64 along the reference's own Q and return the smallest distance of
    the rounding argument to a decision boundary (in alphabet-index units)."""
    N, d, g = case["N"], case["d"], case["groups"]
    K = 2 ** (case["bits"] - 1)
    lamb = case["lamb"]
    reg = case["reg"]
    Ng = N // g
    step = float(step)
    best = np.inf
    for gi_ in range(g):
        Wg = W[gi_ * Ng:(gi_ + 1) * Ng].astype(np.float64)
        Qg = Q[gi_ * Ng:(gi_ + 1) * Ng].astype(np.float64)
        Ag = A[:, gi_ * d:(gi_ + 1) * d].astype(np.float64)
        Xg = X[:, gi_ * d:(gi_ + 1) * d].astype(np.float64)
        U = np.zeros((Ng, A.shape[0]))
        for t in range(d):
            U += np.outer(Wg[:, t], Ag[:, t])
            nrm = float(Xg[:, t] @ Xg[:, t])
            if nrm > 0:
                s = (U @ Xg[:, t]) / nrm
                if reg == "L1":
                    y = np.sign(s) * np.maximum(np.abs(s) - lamb, 0)
                elif reg == "L0":
                    best = min(best, float(np.min(np.abs(np.abs(s) - lamb))) / step)
                    y = np.sign(s) * np.maximum(np.abs(s) - lamb, 0)
                else:
                    y = s
                z = y / step + 0.5
                f = np.floor(z)
                frac = z - f
                nb = np.where(frac < 0.5, f - 1, f + 1)
                dist = np.minimum(frac, 1 - frac)
                real = np.minimum(np.abs(f), K) != np.minimum(np.abs(nb), K)
                if np.any(real):
                    best = min(best, float(np.min(dist[real])))
            U -= np.outer(Qg[:, t], Xg[:, t])
    return best


def run_reference_layer(SA, case, W, A, X):
    N, d, m, g = case["N"], case["d"], case["m"], case["groups"]
    K = 2 ** (case["bits"] - 1)
    step_base = case["scalar"] / K          # quantize_neural_net.py:92-93
    Wt, At, Xt = torch.from_numpy(W.copy()), torch.from_numpy(A.copy()), torch.from_numpy(X.copy())
    dev = torch.device("cpu")
    Q, qe, rqe, adder, radder = quiet(SA._quantize_layer, Wt, At, Xt, m, step_base, K, case["percentile"],
                                      case["reg"], case["lamb"], g, False, dev)
    # step exactly as step_algorithm.py:191-192
    rad = torch.quantile(torch.abs(Wt), case["percentile"], axis=1).mean()
    step = step_base * rad - case["lamb"] / K if case["reg"] == 'L0' else step_base * rad
    # residual U per group through the reference's own inner loop
    quantizer = (SA._soft_thresholding_msq if case["reg"] == 'L1' else
                 SA._hard_thresholding_msq if case["reg"] == 'L0' else SA._msq)
    Q2 = torch.zeros_like(Wt)
    U = torch.zeros(N, m)
    Ng = N // g
    A3, X3 = At.view(m, g, -1), Xt.view(m, g, -1)
    for i in range(g):
        quiet(SA._quantization, Wt[i * Ng:(i + 1) * Ng], Q2[i * Ng:(i + 1) * Ng], U[i * Ng:(i + 1) * Ng],
              A3[:, i, :], X3[:, i, :], quantizer, step, K, case["lamb"])
    Q = Q.reshape(N, d)
    assert torch.equal(Q, Q2), "reference _quantize_layer and per-group _quantization disagree"
    if g == 1:
        assert torch.equal(adder, U.T)
    res = dict(Q=Q.numpy().copy(), U=U.numpy().copy(), step=np.float32(step.item()),
               quantize_error=np.float32(float(qe)), relative_quantize_error=np.float32(float(rqe)))
    if radder is not None:
        res["relative_adder"] = radder.numpy().copy()
    return res


def index_of(case, Q, step):
    """Alphabet index implied by the reference's Q (the reference stores no integers)."""
    K = 2 ** (case["bits"] - 1)
    step = np.float32(step)
    if case["reg"] == 'L0':
        lam = np.float32(case["lamb"])
        mag = (np.abs(Q) - lam) / step
        k = np.rint(mag).astype(np.int32)
        idx = np.where(Q == 0, 0, np.sign(Q).astype(np.int32) * (k + 1))
        rebuilt = np.where(idx == 0, np.float32(0), np.sign(idx).astype(np.float32) *
                           (lam + step * (np.abs(idx) - 1).astype(np.float32)))
    else:
        idx = np.rint(Q / step).astype(np.int32)
        rebuilt = (np.sign(idx).astype(np.float32) * step) * np.abs(idx).astype(np.float32)
    assert np.array_equal(rebuilt.astype(np.float32), Q.astype(np.float32)), "index reconstruction is not exact"
    assert np.all(np.abs(idx) <= K + (1 if case["reg"] == 'L0' else 0))
    return idx.astype(np.int16)


def gen_loop_case(SA, name):
    case = gi.CASES[name]
    nweights = case["N"] * case["d"]
    want = 1e-4 if nweights <= 20000 else 2e-5
    for off in range(200):
        W, A, X = gi.make_inputs(case, off)
        res = run_reference_layer(SA, case, W, A, X)
        margin = fp64_margin(case, W, A, X, res["Q"], res["step"])
        if margin > want:
            break
    else:
        raise RuntimeError("no seed with margin > %g for %s" % (want, name))
    idx = index_of(case, res["Q"], res["step"])
    meta = dict(case=case, seed_offset=off, inputs_sha256=gi.inputs_digest(W, A, X), margin=margin,
                provenance=provenance())
    np.savez_compressed(os.path.join(gi.GOLDEN_DIR, name + ".npz"), meta=np.array(json.dumps(meta)),
                        idx=idx, **res)
    print("%-28s seed_offset=%d margin=%.2e step=%.6g levels=%d |U|max=%.3g" % (
        name, off, margin, res["step"], len(np.unique(idx)), np.abs(res["U"]).max()))


# ------------------------------------------------------------------------------------------------ G6
def gen_big_case(SA, name):
    """Headline-scale case: ONE run of the reference on seed offset 0 -- no seed search, no margin requirement.
    Stores int8 indices, the step, the error outputs and per-row digests of U (gi.row_checksums) instead of Q / U."""
    case = gi.BIG_CASES[name]
    W, A, X = gi.make_inputs(case, 0)
    N, d, m = case["N"], case["d"], case["m"]
    K = 2 ** (case["bits"] - 1)
    step_base = case["scalar"] / K
    Wt, At, Xt = torch.from_numpy(W.copy()), torch.from_numpy(A.copy()), torch.from_numpy(X.copy())
    t0 = time.time()
    if case["groups"] == 1:
        Q, qe, rqe, adder, radder = quiet(SA._quantize_layer, Wt, At, Xt, m, step_base, K, case["percentile"],
                                          case["reg"], case["lamb"], 1, False, torch.device("cpu"))
        U = adder.T.contiguous().numpy()            # quantize_adder = U.T  (step_algorithm.py:216)
    else:
        # grouped layers: the reference returns no residual (step_algorithm.py:226-237); U comes from its own inner
        # loop run per group, after checking that loop reproduces _quantize_layer's Q (run_reference_layer)
        res = run_reference_layer(SA, case, W, A, X)
        Q, qe, rqe, U = torch.from_numpy(res["Q"]), res["quantize_error"], res["relative_quantize_error"], res["U"]
        radder = None
    dt = time.time() - t0
    rad = torch.quantile(torch.abs(Wt), case["percentile"], axis=1).mean()
    step = step_base * rad - case["lamb"] / K if case["reg"] == 'L0' else step_base * rad
    Q = Q.numpy().reshape(N, d)
    idx = index_of(case, Q, step.item())
    assert np.abs(idx).max() <= 127
    margin = fp64_margin(case, W, A, X, Q, step.item())
    meta = dict(case=case, seed_offset=0, seed_search=False, inputs_sha256=gi.inputs_digest(W, A, X), margin=margin,
                reference_seconds=dt, provenance=provenance())
    np.savez_compressed(os.path.join(gi.GOLDEN_DIR, name + ".npz"), meta=np.array(json.dumps(meta)),
                        idx=idx.astype(np.int8), step=np.float32(step.item()),
                        quantize_error=np.float32(float(qe)), relative_quantize_error=np.float32(float(rqe)),
                        **({} if radder is None else {"relative_adder": radder.numpy().copy()}), **gi.row_checksums(U))
    print("%-30s NO seed search: fp64 margin=%.2e step=%.6g levels=%d ref %.1fs (%.4f Mw/s, %d threads)" % (
        name, margin, step.item(), len(np.unique(idx)), dt, N * d / dt / 1e6, torch.get_num_threads()))


# ------------------------------------------------------------------------------------------------ G5
def import_reference_driver():
    """quantize_neural_net.py imports torchvision (unused) and utils.py imports torchvision block classes
    only to whitelist them; torchvision is not installed, so give sys.modules empty placeholders."""
    import torch.nn as nn
    names = {"torchvision": [], "torchvision.models": [],
             "torchvision.models.resnet": ["BasicBlock", "Bottleneck", "ResNet"],
             "torchvision.models.googlenet": ["BasicConv2d", "Inception", "InceptionAux"],
             "torchvision.models.efficientnet": ["Conv2dNormActivation", "SqueezeExcitation", "MBConv"],
             "torchvision.models.mobilenetv2": ["InvertedResidual"]}
    for mod, classes in names.items():
        if mod not in sys.modules:
            mm = types.ModuleType(mod)
            for c in classes:
                setattr(mm, c, type(c, (nn.Module,), {}))
            sys.modules[mod] = mm
    sys.path.insert(0, REF_SRC)
    try:
        import quantize_neural_net
    finally:
        sys.path.pop(0)
    return quantize_neural_net


def gen_driver():
    qnn = import_reference_driver()
    out = {}
    metas = []
    for ci, cfg in enumerate([dict(bits=4, reg=None, lamb=0.1, retain_rate=0.25),
                              dict(bits=2, reg='L1', lamb=0.02, retain_rate=0.5),
                              dict(bits=3, reg=None, lamb=0.1, retain_rate=1)]):
        rng = np.random.default_rng(777 + ci)
        net = gi.toy_net(rng)
        B = 6
        nlayers = 5
        batches = gi.toy_batches(rng, B, nlayers)
        np.random.seed(11 + ci)
        torch.manual_seed(11 + ci)
        quant = qnn.QuantizeNeuralNet(net, "toy", B, batches, mlp_bits=cfg["bits"], cnn_bits=cfg["bits"],
                                      ignore_layers=[], mlp_alphabet_scalar=1.16, cnn_alphabet_scalar=1.16,
                                      mlp_percentile=1, cnn_percentile=1, reg=cfg["reg"], lamb=cfg["lamb"],
                                      retain_rate=cfg["retain_rate"], stochastic_quantization=False,
                                      device=torch.device("cpu"))
        qnet = quiet(quant.quantize_network)
        for li, layer in enumerate(quant.quantized_network_layers):
            out["c%d_layer%d_weight" % (ci, li)] = layer.weight.detach().numpy().copy()
        metas.append(dict(cfg=cfg, net_seed=777 + ci, np_seed=11 + ci, batch=B, nlayers=nlayers))
        assert len(quant.quantized_network_layers) == nlayers
        del qnet
    out["meta"] = np.array(json.dumps(dict(configs=metas, provenance=provenance())))
    np.savez_compressed(os.path.join(gi.GOLDEN_DIR, "g5_driver.npz"), **out)
    print("g5_driver: %d configs" % len(metas))


# --------------------------------------------------------------------------------------------- bench
def bench_reference(SA):
    """Time the TRUE reference inner loop on the ResNet-50 3x3 shapes (first `steps` columns)."""
    torch.set_num_threads(os.cpu_count())
    for (N, d, m, steps) in [(512, 4608, 3072, 256), (256, 2304, 7168, 256), (128, 1152, 26624, 128), (64, 576, 93184, 64)]:
        g = torch.Generator().manual_seed(1234)
        W = torch.randn(N, steps, generator=g) * (2.0 / d) ** 0.5
        pre = torch.randn(m, steps, generator=g)
        A = torch.relu(pre)
        X = torch.relu(pre + 0.05 * torch.randn(m, steps, generator=g))
        Q = torch.zeros_like(W)
        U = torch.zeros(N, m)
        step = torch.tensor(1.16 / 8) * W.abs().max(dim=1).values.mean()
        t0 = time.time()
        quiet(SA._quantization, W, Q, U, A, X, SA._msq, step, 8, 0.0)
        dt = time.time() - t0
        print("ref loop N=%d d=%d m=%d steps=%d: %.2fs  %.4f Mw/s (%d threads)" % (
            N, d, m, steps, dt, N * steps / dt / 1e6, torch.get_num_threads()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bench", action="store_true")
    ap.add_argument("--only", default=None, help="substring filter on case names")
    ap.add_argument("--skip-driver", action="store_true")
    ap.add_argument("--skip-big", action="store_true", help="skip the headline-scale G6 cases (minutes of reference time)")
    args = ap.parse_args()
    if not os.path.isdir(REF_SRC):
        sys.exit("reference not present: this tool only runs in the build container")
    os.makedirs(gi.GOLDEN_DIR, exist_ok=True)
    torch.set_num_threads(4)
    SA = import_reference_step_algorithm()
    if args.only is None:
        gen_quantizers(SA)
    if args.only is None or args.only in "g7_stochastic":
        gen_stochastic(SA)
    for name in gi.CASES:
        if args.only is None or args.only in name:
            gen_loop_case(SA, name)
    for name in gi.BIG_CASES:
        if (args.only is None and not args.skip_big) or (args.only is not None and args.only in name):
            gen_big_case(SA, name)
    if not args.skip_driver and args.only is None:
        gen_driver()
    if args.bench:
        bench_reference(SA)


if __name__ == "__main__":
    main()
