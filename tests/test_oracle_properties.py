"""Property tests of the CPU oracle (hypothesis): size-independent facts of the GPFQ recurrence that the HIP
kernels are then held to bit for bit (tests/test_gpu_*.py compare HIP with this oracle)."""
import numpy as np
from hypothesis import given, settings, strategies as st

import oracle


def _inputs(seed, N, d, m, zero_cols):
    rng = np.random.default_rng(seed)
    W = (rng.standard_normal((N, d)) * np.sqrt(2.0 / d)).astype(np.float32)
    pre = rng.standard_normal((m, d)).astype(np.float32)
    A = np.maximum(pre, 0)
    X = np.maximum(pre + np.float32(0.05) * rng.standard_normal((m, d)).astype(np.float32), 0).astype(np.float32)
    for c in zero_cols:
        X[:, c % d] = 0.0
    return W, A, X


@settings(max_examples=25, deadline=None)
@given(seed=st.integers(0, 2 ** 20), N=st.integers(1, 12), d=st.integers(1, 24), m=st.integers(1, 2300),
       bits=st.sampled_from([2, 3, 4]), mode=st.sampled_from([0, 1, 2]),
       zero_cols=st.lists(st.integers(0, 23), max_size=3))
def test_recurrence_properties(seed, N, d, m, bits, mode, zero_cols):
    oracle.build()
    K = 2 ** (bits - 1)
    lamb = 0.03
    W, A, X = _inputs(seed, N, d, m, zero_cols)
    step = np.float32(1.16 / K * np.abs(W).max(1).mean())
    Q, idx, U = oracle.quantization(W, A, X, step, K, mode=mode, lamb=lamb)
    # the alphabet is respected
    assert np.abs(idx).max() <= K + (1 if mode == 2 else 0)
    lam32 = np.float32(lamb)
    if mode == 2:
        want = np.where(idx == 0, np.float32(0), np.sign(idx).astype(np.float32) * (lam32 + step * (np.abs(idx) - 1).astype(np.float32)))
    else:
        want = (np.sign(idx).astype(np.float32) * step) * np.abs(idx).astype(np.float32)
    assert np.array_equal(want.astype(np.float32), Q)
    # a column of the quantized-net input that is identically zero quantizes to zero (step_algorithm.py:143-146)
    for c in set(z % d for z in zero_cols):
        assert not idx[:, c].any()
    # rows are independent: any subset equals the slice of the full run
    a, b = sorted(np.random.default_rng(seed + 1).integers(0, N + 1, 2))
    if b > a:
        Qs, idxs, Us = oracle.quantization(W[a:b], A, X, step, K, mode=mode, lamb=lamb)
        assert np.array_equal(idxs, idx[a:b]) and np.array_equal(Us, U[a:b])
    # the residual is what the recurrence says it is: U = W A^T - Q X^T (fp64 check of the fp32 accumulation)
    ref = W.astype(np.float64) @ A.astype(np.float64).T - Q.astype(np.float64) @ X.astype(np.float64).T
    assert np.abs(U - ref).max() <= 1e-4 * (1.0 + np.abs(ref).max())
    # columns can be processed in two calls carrying U (the in-place surface of _quantization)
    if d >= 2:
        k = d // 2
        Q1, i1, U1 = oracle.quantization(W[:, :k], A[:, :k], X[:, :k], step, K, mode=mode, lamb=lamb)
        Q2, i2, U2 = oracle.quantization(W[:, k:], A[:, k:], X[:, k:], step, K, mode=mode, lamb=lamb, U0=U1)
        assert np.array_equal(np.concatenate([i1, i2], 1), idx) and np.array_equal(U2, U)


@settings(max_examples=30, deadline=None)
@given(seed=st.integers(0, 2 ** 20), n=st.integers(1, 5000))
def test_canonical_dot_is_close_to_float64_and_padding_invariant(seed, n):
    oracle.build()
    rng = np.random.default_rng(seed)
    u = rng.standard_normal(n).astype(np.float32)
    x = rng.standard_normal(n).astype(np.float32)
    got = float(oracle.cdot(u, x))
    ref = float(u.astype(np.float64) @ x.astype(np.float64))
    assert abs(got - ref) <= 2e-6 * float(np.abs(u.astype(np.float64) * x).sum()) + 1e-30
    # appending zeros (more padding, possibly more segments and slots) never changes a bit
    pad = int(rng.integers(1, 3000))
    got2 = float(oracle.cdot(np.concatenate([u, np.zeros(pad, np.float32)]), np.concatenate([x, np.zeros(pad, np.float32)])))
    if ((n + 1023) // 1024) == ((n + pad + 1023) // 1024):
        assert got2 == got
