"""Output-neuron sharding of one layer across the GPUs of a node (new in this build; the reference is
single-device, main.py:55).

Rows of W / Q / U are independent in the GPFQ recurrence (step_algorithm.py:141-148 is row-wise; the
gemv is N separate dot products), so each rank quantizes a disjoint set of output neurons against the
SAME activations and alphabet step, and one RCCL all_gather per layer (torch.distributed backend
"nccl" = RCCL over xGMI) exchanges the int8 alphabet indices; Q is rebuilt locally from the indices.
The kernels' reduction order depends only on m, never on the number of local rows, so the gathered
result is bitwise equal to the single-GPU result.  U stays sharded; the error metrics are rebuilt from
per-rank partial sums (one tiny all_reduce).

Nothing here touches the GPU directly: the per-rank computation is injected (`run_rows`), which is how
the world_size-2 gloo tests drive this module on CPU.
"""
import torch
import torch.distributed as td

_ctx = None


class ShardContext:
    def __init__(self, group=None, gather_residual=False, force=False):
        self.group = group
        self.rank = td.get_rank(group)
        self.world = td.get_world_size(group)
        self.gather_residual = gather_residual
        self.force = force
        self.event_hook = None      # optional callable(tag): "collective_begin" / "collective_end" around the index all_gather


def enable(group=None, gather_residual=False, force=False):
    """Turn on neuron sharding for every following StepAlgorithm._quantize_layer call.
    torch.distributed must be initialised (one process per GPU).
    force: take the sharded path -- partition, all_gather of the indices, all_reduce of the partial sums -- even in a
    world of ONE rank, where it is an identity: the way to drive the collectives through RCCL on a single GPU."""
    global _ctx
    if not td.is_available() or not td.is_initialized():
        raise RuntimeError("torch.distributed is not initialised")
    _ctx = ShardContext(group, gather_residual, force)
    return _ctx


def disable():
    global _ctx
    _ctx = None


def active():
    return _ctx if (_ctx is not None and (_ctx.world > 1 or _ctx.force)) else None


def _ceil_div(a, b):
    return (a + b - 1) // b


def partition(N, groups, world):
    """How the N output neurons of a layer are split.  Returns (kind, chunk):
       'rows'           groups == 1: rank r owns rows [r*chunk, (r+1)*chunk)
       'groups'         groups >= world: rank r owns whole groups [r*chunk, (r+1)*chunk)
       'rows_in_groups' 1 < groups < world: rank r owns rows [r*chunk, (r+1)*chunk) of EVERY group"""
    if groups == 1:
        return "rows", _ceil_div(N, world)
    if groups >= world:
        return "groups", _ceil_div(groups, world)
    return "rows_in_groups", _ceil_div(N // groups, world)


def local_range(kind, chunk, N, groups, rank):
    """(start, end) in the unit of `kind` (rows, groups, rows inside a group), clipped."""
    total = N if kind == "rows" else groups if kind == "groups" else N // groups
    a = min(rank * chunk, total)
    b = min((rank + 1) * chunk, total)
    return a, b


def rebuild_q(idx, step, K, mode, lamb):
    """Alphabet value from its index with the kernel's own fp32 operations (msq/soft/stochastic:
    (sign*step)*|k|, step_algorithm.py:56; hard: sign*(lamb + step*k), step_algorithm.py:81)."""
    k = idx.to(torch.float32)
    step_t = torch.tensor(step, dtype=torch.float32, device=idx.device)
    sg = torch.sign(k)
    if mode == 2:
        lam = torch.tensor(lamb, dtype=torch.float32, device=idx.device)
        mag = lam + step_t * (torch.abs(k) - 1.0)
        return torch.where(k == 0, torch.zeros_like(k), sg * mag)
    return (sg * step_t) * torch.abs(k)


def _all_gather_rows(ctx, local, chunk_rows):
    """all_gather of equally sized row blocks (local is padded to chunk_rows rows)."""
    hook = getattr(ctx, "event_hook", None)
    if hook:
        hook("collective_begin")
    try:
        return _all_gather_rows_impl(ctx, local, chunk_rows)
    finally:
        if hook:
            hook("collective_end")


def _all_gather_rows_impl(ctx, local, chunk_rows):
    d = local.shape[1]
    if local.shape[0] < chunk_rows:
        pad = torch.zeros((chunk_rows - local.shape[0], d), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], 0)
    local = local.contiguous()
    if local.is_cuda and td.get_backend(ctx.group) == "gloo":
        # rehearsal set-up (several ranks sharing one card, gloo): stage the collective through the host
        host = torch.empty((ctx.world * chunk_rows, d), dtype=local.dtype)
        td.all_gather_into_tensor(host, local.cpu(), group=ctx.group)
        return host.to(local.device)
    out = torch.empty((ctx.world * chunk_rows, d), dtype=local.dtype, device=local.device)
    td.all_gather_into_tensor(out, local, group=ctx.group)
    return out


def _col_slice(M, lo, hi):
    """columns [lo, hi) of a layer input: an (m, D) matrix or a PreparedColumns (which slices rows of its T)"""
    return M.cols(lo, hi) if hasattr(M, "cols") else M[:, lo:hi]


def quantize_sharded(ctx, W, A, lda, X, ldx, groups, dg, step, K, mode, lamb, idx_dtype, run_rows):
    """Quantize this rank's neurons and gather the indices of all ranks.
    run_rows(W_loc, groups_loc, A_loc, lda, X_loc, ldx, row_id0) -> (Q_loc, idx_loc, U_loc, usq_seg_loc); row_id0
    is the global number of local row 0: the stochastic quantizer's Philox key is (seed, global row, column), so
    every launch covers ONE contiguous block of global rows.
    Returns (Q (N, dg), idx (N, dg), U_loc, usq_seg_loc, rows) with rows = the global row numbers of U_loc."""
    N = W.shape[0]
    Ng = N // groups
    kind, chunk = partition(N, groups, ctx.world)
    a, b = local_range(kind, chunk, N, groups, ctx.rank)
    dev = W.device
    if kind == "rows":
        rows = torch.arange(a, b, device=dev)
        _, idx_loc, U_loc, usq_loc = run_rows(W[a:b], 1, A, lda, X, ldx, a)
        gathered = _all_gather_rows(ctx, idx_loc, chunk)[:N]
    elif kind == "groups":
        rows = torch.arange(a * Ng, b * Ng, device=dev)
        A_loc = _col_slice(A, a * dg, max(b, a) * dg if b > a else a * dg)
        X_loc = _col_slice(X, a * dg, max(b, a) * dg if b > a else a * dg)
        _, idx_loc, U_loc, usq_loc = run_rows(W[a * Ng:b * Ng], max(b - a, 1), A_loc, lda, X_loc, ldx, a * Ng)
        gathered = _all_gather_rows(ctx, idx_loc, chunk * Ng)[:N]
    else:
        nl = b - a
        rows = (torch.arange(groups, device=dev)[:, None] * Ng + torch.arange(a, b, device=dev)[None, :]).reshape(-1)
        # one launch per group (1 < groups < world <= 8: a handful): rows [a, b) of group g are the contiguous global
        # rows g*Ng + a ..., which keeps the Philox keys of the stochastic quantizer equal to the single-GPU ones
        W3 = W.view(groups, Ng, dg)
        parts = [run_rows(W3[g, a:b].contiguous(), 1, _col_slice(A, g * dg, (g + 1) * dg), lda,
                          _col_slice(X, g * dg, (g + 1) * dg), ldx, g * Ng + a) for g in range(groups)]
        idx_loc = torch.cat([p[1] for p in parts], 0)
        U_loc = torch.cat([p[2] for p in parts], 0)
        usq_loc = torch.cat([p[3] for p in parts], 0)
        blk = torch.zeros((groups, chunk, dg), dtype=idx_loc.dtype, device=dev)
        blk[:, :nl] = idx_loc.view(groups, nl, dg)
        g = _all_gather_rows(ctx, blk.view(groups * chunk, dg), groups * chunk)
        g = g.view(ctx.world, groups, chunk, dg).permute(1, 0, 2, 3).reshape(groups, ctx.world * chunk, dg)
        gathered = g[:, :Ng].reshape(N, dg)
    gathered = gathered.contiguous()
    Q = rebuild_q(gathered, step, K, mode, lamb)
    return Q, gathered, U_loc, usq_loc, rows


def sharded_error_metrics(ctx, W, A, usq_loc, groups, rows, U_loc=None):
    """step_algorithm.py:216-219 / :239-243 from per-rank partial sums of squares.  usq_loc: ||U[i, :]||^2 of the
    local rows (float64, from the loop kernel's epilogue); U_loc is only needed for gather_residual."""
    if hasattr(A, "matrix"):
        A = A.matrix()
    N, dg = W.shape
    mm = A.shape[0]
    dev = W.device
    usq = torch.zeros((N,), dtype=torch.float32, device=dev)
    asq = torch.zeros((N,), dtype=torch.float32, device=dev)
    if rows.numel() > 0:
        usq[rows] = usq_loc.float()
        W_loc = W[rows]
        if groups == 1:
            AW = A @ W_loc.T                                           # (m, n_loc)
            asq[rows] = (AW.double() ** 2).sum(0).float()
        else:
            Ng = N // groups
            gid = torch.div(rows, Ng, rounding_mode='floor')
            A3 = A.reshape(mm, groups, dg)
            for g in torch.unique(gid).tolist():
                sel = rows[gid == g]
                AW = A3[:, g, :] @ W[sel].T
                asq[sel] = (AW.double() ** 2).sum(0).float()
    both = torch.stack([usq, asq])
    if both.is_cuda and td.get_backend(ctx.group) == "gloo":
        host = both.cpu()
        td.all_reduce(host, group=ctx.group)
        both = host.to(both.device)
    else:
        td.all_reduce(both, group=ctx.group)
    usq, asq = both[0], both[1]
    if groups == 1:
        quantize_error = usq.sum().sqrt()
        relative_quantize_error = quantize_error / asq.sum().sqrt()
        relative_adder = usq.sqrt() / (asq.sqrt() + 1e-5)
        quantize_adder = None
        if ctx.gather_residual:
            chunk = partition(N, 1, ctx.world)[1]
            quantize_adder = _all_gather_rows(ctx, U_loc, chunk)[:N].T
    else:
        Ng = N // groups
        un = usq.view(groups, Ng).sum(1).sqrt()
        an = asq.view(groups, Ng).sum(1).sqrt()
        quantize_error = un.sum() / groups
        relative_quantize_error = (un / an).sum() / groups
        quantize_adder = None
        relative_adder = None
    return dict(quantize_error=quantize_error, relative_quantize_error=relative_quantize_error,
                quantize_adder=quantize_adder, relative_adder=relative_adder)
