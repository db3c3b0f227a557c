"""Tie audit: what an alphabet-index mismatch between two implementations of the GPFQ loop means.

The reference leaves the order of ONE reduction per step to its BLAS (U.matmul(x), step_algorithm.py:144); this
build uses a fixed canonical order.  Two correct implementations can therefore disagree on an index only where
the rounding argument of step_algorithm.py:56 / :78-81 / :103-104 lands on a decision boundary to within the
rounding noise of that dot product -- and from that column on the row follows a different greedy path
(step_algorithm.py:148 feeds q back into U), so only a row's FIRST divergence is meaningful (SURVEY.md 7,
hard part 1).

audit() therefore, for every row whose indices differ:
  * finds the first diverging column t;
  * replays the row's residual up to t in fp32 exactly as step_algorithm.py:141/:148 do (elementwise, order-free:
    every implementation holds these very bits while the indices agree);
  * recomputes <u, x_t> / ||x_t||^2 in float64 and measures the distance of the rounding argument to the
    boundary that separates the two indices, in alphabet-index units;
  * accepts the row only if the two indices are neighbours across that boundary and the distance is < tol.
Everything else is a real mismatch.
"""
import numpy as np


def alphabet_value(idx, step, lamb, reg):
    """fp32 alphabet value of an index, with the operations of step_algorithm.py:56 / :81."""
    step = np.float32(step)
    k = np.asarray(idx).astype(np.float32)
    sg = np.sign(k).astype(np.float32)
    if reg == "L0":
        mag = np.float32(lamb) + step * (np.abs(k) - np.float32(1))
        return np.where(k == 0, np.float32(0), sg * mag).astype(np.float32)
    return ((sg * step) * np.abs(k)).astype(np.float32)


def _level(idx, reg):
    """signed value of floor(z) that produces this index (unclipped)"""
    if reg == "L0":
        return idx - 1 if idx > 0 else idx + 1
    return idx


def boundary_margin(s64, step, K, lamb, reg, ka, kb):
    """Distance (alphabet-index units) from the float64 rounding argument to the decision boundary between the
    indices ka and kb; inf if they are not neighbours across a single boundary."""
    step = float(step)
    ka, kb = int(ka), int(kb)
    if abs(ka - kb) != 1:
        return float("inf")
    if reg == "L0":
        if ka == 0 or kb == 0:                      # the threshold |s| > lamb (F.threshold, step_algorithm.py:78)
            return abs(abs(s64) - lamb) / step
        y = np.sign(s64) * max(abs(s64) - lamb, 0.0)
    elif reg == "L1":
        y = np.sign(s64) * max(abs(s64) - lamb, 0.0)
    else:
        y = s64
    z = y / step + 0.5
    return abs(z - max(_level(ka, reg), _level(kb, reg)))


def replay_row_fp32(w, q, A, X, t):
    """Residual of one row after the updates of columns 0..t-1 plus the `+ w_t a_t` of column t, in fp32 with the
    reference's operation order (mul, add; mul, sub).  A, X: (m, d) column blocks of the row's group."""
    u = np.zeros(A.shape[0], np.float32)
    for s in range(t):
        u += np.float32(w[s]) * A[:, s]
        u -= np.float32(q[s]) * X[:, s]
    u += np.float32(w[t]) * A[:, t]
    return u


def audit(case, W, A, X, step, idx_ref, idx_test, tol=1e-5):
    """Compare two index matrices of one layer.  Returns a report dict; report["unexplained"] lists the rows whose
    first divergence is NOT a tie within tol.  `case` carries bits / reg / lamb / groups."""
    idx_ref = np.asarray(idx_ref).astype(np.int32)
    idx_test = np.asarray(idx_test).astype(np.int32)
    assert idx_ref.shape == idx_test.shape == W.shape
    N, d = W.shape
    g = case["groups"]
    Ng = N // g
    K = 2 ** (case["bits"] - 1)
    reg, lamb = case["reg"], float(case["lamb"] or 0.0)
    diff = idx_ref != idx_test
    rows = np.nonzero(diff.any(1))[0]
    ties, unexplained = [], []
    for i in rows:
        t = int(np.argmax(diff[i]))
        gi_ = i // Ng
        Ag, Xg = A[:, gi_ * d:(gi_ + 1) * d], X[:, gi_ * d:(gi_ + 1) * d]
        q = alphabet_value(idx_ref[i, :t], step, lamb, reg)
        u = replay_row_fp32(W[i], q, Ag, Xg, t)
        x = Xg[:, t].astype(np.float64)
        n2 = float(x @ x)
        s64 = float(u.astype(np.float64) @ x) / n2 if n2 > 0 else 0.0
        mg = boundary_margin(s64, step, K, lamb, reg, idx_ref[i, t], idx_test[i, t]) if n2 > 0 else float("inf")
        rec = dict(row=int(i), col=t, ref=int(idx_ref[i, t]), test=int(idx_test[i, t]), margin=float(mg),
                   later_differences=int(diff[i].sum()) - 1)
        (ties if mg < tol else unexplained).append(rec)
    return dict(weights=int(N * d), rows_compared=int(N), rows_diverged=int(len(rows)), ties=ties,
                unexplained=unexplained, agreeing_rows=np.nonzero(~diff.any(1))[0])


def assert_parity(case, W, A, X, step, idx_ref, idx_test, tol=1e-5, what="indices"):
    """Bit-exact, or every diverging row's first divergence is a tie within tol.  Returns the report."""
    rep = audit(case, W, A, X, step, idx_ref, idx_test, tol)
    if rep["unexplained"]:
        raise AssertionError("%s: %d row(s) diverge away from any decision boundary (tol %g): %s" % (
            what, len(rep["unexplained"]), tol, rep["unexplained"][:5]))
    return rep
