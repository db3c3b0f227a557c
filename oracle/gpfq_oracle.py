"""ctypes front end of oracle/gpfq_oracle.c (+ a torch-op restatement used as the CPU baseline).

TEST INFRASTRUCTURE ONLY -- see the header of gpfq_oracle.c.  Functions cite the reference lines
(/root/reference/src/...) they restate.  Nothing here reads /root/reference at run time.
"""
import ctypes
import os
import subprocess

import numpy as np

MODE_MSQ, MODE_SOFT, MODE_HARD, MODE_STOCHASTIC = 0, 1, 2, 3

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgpfq_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i16p = ctypes.POINTER(ctypes.c_int16)
_i32p = ctypes.POINTER(ctypes.c_int)


def build(force=False):
    """Compile the C oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "gpfq_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libgpfq_oracle.so"])
    return _SO


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        lib = ctypes.CDLL(_SO)
        lib.gpfq_oracle_quantization.restype = ctypes.c_int
        lib.gpfq_oracle_quantization.argtypes = [
            _f32p, ctypes.c_long, _f32p, ctypes.c_long, _f32p, ctypes.c_long,
            _f32p, ctypes.c_long, _f32p, ctypes.c_long,
            ctypes.c_long, ctypes.c_long, ctypes.c_long, ctypes.c_float, ctypes.c_int, ctypes.c_int,
            ctypes.c_float, ctypes.c_uint64, ctypes.c_uint64, _i16p, ctypes.c_long, ctypes.c_int]
        lib.gpfq_oracle_quantize_groups.restype = ctypes.c_int
        lib.gpfq_oracle_quantize_groups.argtypes = [
            _f32p, _f32p, _f32p, _f32p, _f32p, ctypes.c_long, ctypes.c_long, ctypes.c_long, ctypes.c_int,
            ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_uint64, _i16p, ctypes.c_int]
        lib.gpfq_oracle_quantizer_vec.restype = None
        lib.gpfq_oracle_quantizer_vec.argtypes = [
            ctypes.c_int, ctypes.c_float, _f32p, ctypes.c_long, ctypes.c_int, ctypes.c_float, _f32p, _f32p, _i32p]
        lib.gpfq_oracle_cdot.restype = ctypes.c_float
        lib.gpfq_oracle_cdot.argtypes = [_f32p, _f32p, ctypes.c_long]
        lib.gpfq_oracle_philox_uniform.restype = ctypes.c_float
        lib.gpfq_oracle_philox_uniform.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64]
        lib.gpfq_oracle_philox_uniform_vec.restype = None
        lib.gpfq_oracle_philox_uniform_vec.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_long, _f32p]
        lib.gpfq_oracle_max_threads.restype = ctypes.c_int
        _lib = lib
    return _lib


def _p(a):
    return a.ctypes.data_as(_f32p)


def _c32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def max_threads():
    return int(_load().gpfq_oracle_max_threads())


def quantizer_vec(mode, step, x, K, lamb=0.0, uniform=None):
    """Elementwise quantizer (step_algorithm.py:38-56 msq, :84-104 soft, :59-81 hard, :7-35 stochastic).

    Returns (q float32[n], idx int32[n])."""
    x = _c32(x).ravel()
    out = np.empty_like(x)
    idx = np.empty(x.shape, dtype=np.int32)
    un = None
    if uniform is not None:
        un = _c32(uniform).ravel()
    _load().gpfq_oracle_quantizer_vec(int(mode), np.float32(step), _p(x), x.size, int(K), np.float32(lamb),
                                      _p(un) if un is not None else None, _p(out),
                                      idx.ctypes.data_as(_i32p))
    return out, idx


def philox_uniform_vec(seed, row0, col, n):
    """The stochastic quantizer's draws U[0,1) for rows row0 .. row0+n-1 at one column (Philox4x32-10 keyed by
    (seed, row, column): the counter-based stand-in for torch.bernoulli's global stream, step_algorithm.py:28)."""
    out = np.empty((int(n),), np.float32)
    _load().gpfq_oracle_philox_uniform_vec(int(seed), int(row0), int(col), int(n), _p(out))
    return out


def cdot(u, x):
    """Canonical fp32 dot product (see gpfq_oracle.c header)."""
    u = _c32(u).ravel()
    x = _c32(x).ravel()
    assert u.size == x.size
    mp = ((u.size + 1023) // 1024) * 1024
    up = np.zeros(mp, np.float32); up[:u.size] = u
    xp = np.zeros(mp, np.float32); xp[:x.size] = x
    return np.float32(_load().gpfq_oracle_cdot(_p(up), _p(xp), mp // 1024))


def quantization(W, A, X, step, K, mode=MODE_MSQ, lamb=0.0, U0=None, seed=0, row_id0=0, nthreads=None):
    """GPFQ loop on one group (step_algorithm.py:107-148).  W [N,d], A/X [m,d] (any strides).

    Returns (Q float32[N,d], idx int16[N,d], U float32[N,m])."""
    W = _c32(W)
    A = _c32(A)
    X = _c32(X)
    N, d = W.shape
    m = A.shape[0]
    assert A.shape == (m, d) and X.shape == (m, d)
    Q = np.zeros((N, d), np.float32)
    idx = np.zeros((N, d), np.int16)
    U = np.zeros((N, m), np.float32) if U0 is None else _c32(U0).copy()
    nt = nthreads or max_threads()
    rc = _load().gpfq_oracle_quantization(
        _p(W), d, _p(Q), d, _p(U), m, _p(A), d, _p(X), d, N, d, m, np.float32(step), int(K), int(mode),
        np.float32(lamb), int(seed), int(row_id0), idx.ctypes.data_as(_i16p), d, int(nt))
    if rc:
        raise MemoryError("gpfq_oracle_quantization failed rc=%d" % rc)
    return Q, idx, U


def alphabet_step(W, step_size, boundary_idx, percentile, reg, lamb):
    """step_algorithm.py:191-192 with the same torch CPU ops the reference uses (fp32 0-dim tensor)."""
    import torch
    Wt = torch.from_numpy(_c32(W))
    rad = torch.quantile(torch.abs(Wt), percentile, axis=1).mean()
    step = step_size * rad - lamb / boundary_idx if reg == 'L0' else step_size * rad
    return np.float32(step.item())


def quantize_layer(W, A, X, step_size, boundary_idx, percentile=1.0, reg=None, lamb=0.0, groups=1,
                   stochastic=False, seed=0, nthreads=None, step=None):
    """StepAlgorithm._quantize_layer (step_algorithm.py:151-249) on numpy arrays.

    W [N,d_g], A/X [m, groups*d_g].  Returns dict(Q, idx, U, step, quantize_error, relative_quantize_error,
    relative_adder (None for groups > 1))."""
    W = _c32(W)
    A = _c32(A)
    X = _c32(X)
    N, dg = W.shape
    m = A.shape[0]
    assert A.shape == (m, groups * dg) and X.shape == A.shape
    if step is None:
        step = alphabet_step(W, step_size, boundary_idx, percentile, reg, lamb)
    mode = MODE_SOFT if reg == 'L1' else MODE_HARD if reg == 'L0' else MODE_STOCHASTIC if stochastic else MODE_MSQ
    Q = np.zeros((N, dg), np.float32)
    idx = np.zeros((N, dg), np.int16)
    U = np.zeros((N, m), np.float32)
    nt = nthreads or max_threads()
    rc = _load().gpfq_oracle_quantize_groups(_p(W), _p(Q), _p(U), _p(A), _p(X), N, dg, m, int(groups),
                                             np.float32(step), int(boundary_idx), mode, np.float32(lamb),
                                             int(seed), idx.ctypes.data_as(_i16p), int(nt))
    if rc:
        raise RuntimeError("gpfq_oracle_quantize_groups failed rc=%d" % rc)
    out = dict(Q=Q, idx=idx, U=U, step=np.float32(step))
    # error metrics, step_algorithm.py:216-219 (groups == 1) / :239-243 (mean over groups)
    if groups == 1:
        AW = A.astype(np.float64) @ W.astype(np.float64).T
        out["quantize_error"] = float(np.linalg.norm(U.astype(np.float64)))
        out["relative_quantize_error"] = out["quantize_error"] / float(np.linalg.norm(AW))
        out["relative_adder"] = (np.linalg.norm(U.astype(np.float64), axis=1) /
                                 (np.linalg.norm(AW, axis=0) + 1e-5)).astype(np.float32)
    else:
        Ng = N // groups
        qe = rqe = 0.0
        for g in range(groups):
            Ug = U[g * Ng:(g + 1) * Ng].astype(np.float64)
            AW = A[:, g * dg:(g + 1) * dg].astype(np.float64) @ W[g * Ng:(g + 1) * Ng].astype(np.float64).T
            qe += float(np.linalg.norm(Ug))
            rqe += float(np.linalg.norm(Ug)) / float(np.linalg.norm(AW))
        out["quantize_error"] = qe / groups
        out["relative_quantize_error"] = rqe / groups
        out["relative_adder"] = None
    return out


def torch_restatement_quantization(W, Q, U, A, X, step, K, steps=None):
    """The reference's op sequence for the msq loop written with the same torch ops
    (step_algorithm.py:140-148 + :56), used ONLY as the timed CPU baseline in bench.py: it reproduces
    what the reference spends per step on the host (outer-product temporaries, strided column reads,
    BLAS gemv, ~15 small ops).  In place on Q and U (torch CPU tensors)."""
    import torch
    d = W.shape[1] if steps is None else min(steps, W.shape[1])
    Kt = None
    for t in range(d):
        U += W[:, t].unsqueeze(1) * A[:, t].unsqueeze(0)
        nrm = torch.linalg.norm(X[:, t], 2) ** 2
        if nrm > 0:
            s = U.matmul(X[:, t]) / nrm
        else:
            s = torch.zeros_like(U[:, 0])
        if Kt is None:
            Kt = torch.ones_like(s) * K
        Q[:, t] = torch.sign(s) * step * torch.minimum(torch.abs(torch.floor(s / step + 0.5)), Kt)
        U -= Q[:, t].unsqueeze(1) * X[:, t].unsqueeze(0)
    return d
