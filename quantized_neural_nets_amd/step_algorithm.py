"""StepAlgorithm -- MI355X counterpart of the reference's operator surface (src/step_algorithm.py).

Same names, argument order and return values as the reference class (it is used unbound, without an
instance: quantize_neural_net.py:150, :180), so the driver and user code call it unchanged:

    StepAlgorithm._quantize_layer(W, analog_layer_input, quantized_layer_input, m, step_size, boundary_idx,
                                  percentile, reg, lamb, groups, stochastic_quantization, device)
        -> (Q, quantize_error, relative_quantize_error, quantize_adder, relative_adder)   step_algorithm.py:151-249
    StepAlgorithm._quantization(W, Q, U, analog_layer_input, quantized_layer_input, quantizer,
                                step_size, boundary_idx, lamb) -> None (in place on Q, U)      step_algorithm.py:107-148
    StepAlgorithm._msq / _soft_thresholding_msq / _hard_thresholding_msq / _stochastic_msq     step_algorithm.py:7-104

All arithmetic of the loop runs in the HIP kernels behind include/gpfq.h (ctypes, _lib.py).  torch is used
for device memory, the current stream, the alphabet-radius statistic (:191) and the error-metric GEMM
(:216-219).  Tensors must live on the GPU; there is no CPU path.
"""
import ctypes

import torch

from . import _lib
from . import dist as _dist

_MODE_BY_NAME = {"_msq": _lib.MODE_MSQ, "_soft_thresholding_msq": _lib.MODE_SOFT,
                 "_hard_thresholding_msq": _lib.MODE_HARD, "_stochastic_msq": _lib.MODE_STOCHASTIC}


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _rows_view(t, name):
    """Row-major 2-D view with unit column stride (copies only if the view cannot be expressed with a
    leading dimension).  Returns (tensor, leading dimension in elements)."""
    _lib.require_gpu_tensor(t, name)
    if t.dim() != 2:
        raise _lib.GpfqError("%s must be 2-D" % name)
    if t.shape[1] > 1 and t.stride(1) != 1:
        t = t.contiguous()
    if t.shape[0] > 1 and t.stride(0) < t.shape[1]:
        t = t.contiguous()
    ld = t.stride(0) if t.shape[0] > 1 else max(t.shape[1], 1)
    return t, max(ld, t.shape[1], 1)


class PreparedColumns:
    """A layer input already in the kernels' column layout: T is (D, m_pad) fp32 on the GPU, row f = input
    feature f, column k = calibration sample k, zero for k >= m (include/gpfq.h gpfq_prepare_columns_f32).
    Produced by the fused conv capture (gpfq_gather_patches_f32); accepted by StepAlgorithm._quantize_layer in
    place of the (m, D) matrix, whose `.shape` / `.matrix()` it still offers."""

    def __init__(self, T, m):
        self.T, self.m = T, int(m)

    @property
    def shape(self):
        return (self.m, self.T.shape[0])

    @property
    def device(self):
        return self.T.device

    def matrix(self):
        return self.T[:, :self.m].T

    def cols(self, lo, hi):
        return PreparedColumns(self.T[lo:hi], self.m)


def _idx_dtype(K):
    return (torch.int8, 1) if K <= 126 else (torch.int16, 2)


def _elementwise(mode, step_size, x, boundary_idx, lamb, uniform=None):
    _lib.require_gpu_tensor(x, "x")
    xc = x.contiguous()
    out = torch.empty_like(xc)
    un = None
    if uniform is not None:
        un = uniform.contiguous()
    _lib.check(_lib.lib.gpfq_quantizer_f32(mode, float(step_size), _ptr(xc), xc.numel(), int(boundary_idx),
                                           float(lamb if lamb is not None else 0.0),
                                           _ptr(un) if un is not None else None, _ptr(out), None,
                                           _lib.current_stream_ptr(x.device)))
    return out.view(x.shape)


class _SeedCounter:
    """Default seeds for the stochastic quantizer when the caller passes none (the reference's 12-argument
    surface has no seed argument): one process-wide counter, so that successive layers draw independent streams.
    Callers that need reproducibility or re-entrancy (QuantizeNeuralNet does) pass `seed=` explicitly."""

    def __init__(self):
        import threading
        self._lock = threading.Lock()
        self._next = 0

    def take(self):
        with self._lock:
            v = self._next
            self._next += 1
            return v


_default_seeds = _SeedCounter()


class StepAlgorithm:
    # There is NO mutable class state: the kernel family, the stochastic seed and the profiling hook are per-call
    # arguments of _quantize_layer_ex / _quantization (plan=, seed=, event_hook=), and results are returned, never
    # parked on the class.

    def _stochastic_msq(step_size, x, boundary_idx, lamb, seed=None, column=0, row_id0=0):
        '''Stochastic rounding to the alphabet, clipped to boundary_idx (step_algorithm.py:7-35).
        Like the reference it overwrites and returns x.  The draws come from the SAME counter-based generator the loop
        kernels use -- element i takes U(seed, row row_id0 + i, column) -- so that this name means one stream: with the
        layer's seed and column = t it is exactly the loop's quantizer at step t (tests/test_gpu_parity.py).  seed None:
        the next value of the process-wide counter.  The reference draws from torch's CPU generator instead, which no GPU
        path can reproduce: distribution parity only.'''
        _lib.require_gpu_tensor(x, "x")
        if seed is None:
            seed = _default_seeds.take()
        un = torch.empty((x.numel(),), device=x.device, dtype=torch.float32)
        _lib.check(_lib.lib.gpfq_philox_uniform_f32(int(seed), int(row_id0), int(column), un.numel(), _ptr(un),
                                                    _lib.current_stream_ptr(x.device)))
        x.copy_(_elementwise(_lib.MODE_STOCHASTIC, step_size, x, boundary_idx, lamb, un.view(x.shape)))
        return x

    def _msq(step_size, x, boundary_idx, lamb):
        '''Nearest element of the uniform symmetric alphabet (step_algorithm.py:38-56).'''
        return _elementwise(_lib.MODE_MSQ, step_size, x, boundary_idx, lamb)

    def _hard_thresholding_msq(step_size, x, boundary_idx, lamb):
        '''Hard-thresholding quantizer, reg == 'L0' (step_algorithm.py:59-81).'''
        return _elementwise(_lib.MODE_HARD, step_size, x, boundary_idx, lamb)

    def _soft_thresholding_msq(step_size, x, boundary_idx, lamb):
        '''Soft-thresholding quantizer, reg == 'L1' (step_algorithm.py:84-104).'''
        return _elementwise(_lib.MODE_SOFT, step_size, x, boundary_idx, lamb)

    def _quantization(W, Q, U, analog_layer_input, quantized_layer_input, quantizer,
                      step_size, boundary_idx, lamb, plan=None, seed=None):
        '''The GPFQ loop over the columns of one group, in place on Q and U (step_algorithm.py:107-148).

        W, Q : (N, d) weights / quantized weights        U : (N, m) residual, read and updated
        analog_layer_input, quantized_layer_input : (m, d), may be strided views (:236)
        quantizer : one of the four StepAlgorithm quantizers (selects the fused epilogue)
        plan : kernel family (include/gpfq.h GPFQ_PLAN_*; None = auto)   seed : stochastic quantizer's Philox key
        '''
        mode = _MODE_BY_NAME.get(getattr(quantizer, "__name__", None))
        if mode is None:
            raise _lib.GpfqError("quantizer must be one of StepAlgorithm._msq/_soft_thresholding_msq/"
                                 "_hard_thresholding_msq/_stochastic_msq")
        Wv, ldw = _rows_view(W, "W")
        A, lda = _rows_view(analog_layer_input, "analog_layer_input")
        X, ldx = _rows_view(quantized_layer_input, "quantized_layer_input")
        N, d = Wv.shape
        m = A.shape[0]
        if A.shape != (m, d) or X.shape != (m, d) or tuple(Q.shape) != (N, d) or tuple(U.shape) != (N, m):
            raise _lib.GpfqError("shape mismatch: W %s Q %s U %s A %s X %s" % (
                tuple(W.shape), tuple(Q.shape), tuple(U.shape), tuple(A.shape), tuple(X.shape)))
        Qv, ldq = _rows_view(Q, "Q")
        Uv, ldu = _rows_view(U, "U")
        dev = W.device
        mp = _lib.lib.gpfq_padded_m(m)
        AT = torch.empty((max(d, 1), mp), device=dev, dtype=torch.float32)
        XT = torch.empty((max(d, 1), mp), device=dev, dtype=torch.float32)
        nrm2 = torch.empty((2 * max(d, 1),), device=dev, dtype=torch.float32)      # {norm, reciprocal} per column (ABI 3)
        st = _lib.current_stream_ptr(dev)
        part = torch.empty((max(int(_lib.lib.gpfq_prepare_ws_bytes(d, m)), 4) // 4,), device=dev, dtype=torch.float32)
        _lib.check(_lib.lib.gpfq_prepare_columns_ws_f32(_ptr(A), lda, _ptr(X), ldx, m, d, _ptr(AT), _ptr(XT),
                                                        _ptr(nrm2), mp, _ptr(part), part.numel() * 4, st))
        if seed is None:
            seed = _default_seeds.take() if mode == _lib.MODE_STOCHASTIC else 0
        plan = _lib.PLAN_AUTO if plan is None else int(plan)
        U0 = None

        def launch(pl, scr):
            _lib.check(_lib.lib.gpfq_quantization_f32(
                _ptr(Wv), ldw, _ptr(Qv), ldq, _ptr(Uv), ldu, 1, _ptr(AT), _ptr(XT), _ptr(nrm2), N, d, m, mp,
                float(step_size), int(boundary_idx), mode, float(lamb if lamb is not None else 0.0), int(seed), 0,
                None, 0, 1, None, pl, _ptr(scr), scr.numel(), st))
            return bool(_lib.lib.gpfq_last_launch_used_exchange())

        if plan != _lib.PLAN_STREAM_ROWS and N > 0 and d > 0:
            # this surface updates U in place, so a launch that gives up waiting for a peer workgroup has already
            # spoilt its input: keep a copy whenever the launch MAY wait on other workgroups.  An initial residual always
            # streams (gpfq_capi.hip run_loop), and the library's own plan description -- the same choose_plan, the same
            # GPFQ_* overrides -- says whether that streaming plan splits rows over workgroups ("C="); its only fallback
            # is to whole rows, which waits for nobody.  Whole-row plans (depthwise layers: one call per group,
            # step_algorithm.py:235-237) therefore pay no N x m copy per call.
            asked = _lib.PLAN_STREAM if plan == _lib.PLAN_AUTO else plan
            try:
                may_wait = " C=" in _lib.describe_plan(N, d, m, 1, asked, mode)
            except _lib.GpfqError:
                may_wait = True
            if may_wait:
                U0 = Uv.clone()
        # scratch -> launch -> status read under the device's lock: the scratch (granules, status words) is one per device
        with _lib.exclusive(dev):
            scr = _lib.scratch(dev)
            if launch(plan, scr) and not _lib.status_ok(dev):
                # never hand back what a timed-out launch left behind: redo on the plan that waits for nobody
                if U0 is None:
                    raise _lib.GpfqError("a cooperative launch timed out and no copy of the initial residual was kept")
                Uv.copy_(U0)
                launch(_lib.PLAN_STREAM_ROWS, scr)
        if Qv.data_ptr() != Q.data_ptr():
            Q.copy_(Qv)
        if Uv.data_ptr() != U.data_ptr():
            U.copy_(Uv)

    def prepare_columns(layer_input):
        '''An (m, D) layer input in the kernels' column layout (PreparedColumns: (D, m_pad), zero padded), on the current
        stream.  The analog input of a layer does not depend on the layers quantized before it (only the quantized
        network's input does), so a caller may prepare it AHEAD -- on a side stream, while the previous layer's loop
        runs -- and hand it to _quantize_layer(_ex) in place of the matrix; the quantized input is then transposed (and
        its column norms taken) when it exists.'''
        if isinstance(layer_input, PreparedColumns):
            return layer_input
        A, lda = _rows_view(layer_input, "layer_input")
        m, D = A.shape
        mp = _lib.lib.gpfq_padded_m(m)
        T = torch.empty((max(D, 1), mp), device=A.device, dtype=torch.float32)
        if D == 0:
            return PreparedColumns(T[:0], m)
        _lib.check(_lib.lib.gpfq_prepare_columns_ws_f32(_ptr(A), lda, None, 0, m, D, _ptr(T), None, None, mp,
                                                        None, 0, _lib.current_stream_ptr(A.device)))
        return PreparedColumns(T, m)

    def _alphabet_step(W, step_size, boundary_idx, percentile, reg, lamb):
        '''rad = mean over neurons of the per-neuron |w| quantile; step = step_size*rad (- lamb/K for L0)
        (step_algorithm.py:191-192).  The per-row statistic is computed on the GPU; the mean of the N row
        values and the scalar arithmetic use the same torch CPU ops as the reference's CPU path, so the
        fp32 step is bit-identical to it.  Returns a 0-dim fp32 CPU tensor.'''
        N, d = W.shape
        if percentile == 1:
            Wv, ldw = _rows_view(W, "W")
            rowstat = torch.empty((N,), device=W.device, dtype=torch.float32)
            _lib.check(_lib.lib.gpfq_row_absmax_f32(_ptr(Wv), ldw, N, d, _ptr(rowstat),
                                                    _lib.current_stream_ptr(W.device)))
        else:
            rowstat = torch.quantile(torch.abs(W), percentile, axis=1)
        rad = rowstat.cpu().mean()
        return step_size * rad - lamb / boundary_idx if reg == 'L0' else step_size * rad

    def _quantize_layer_ex(W, analog_layer_input, quantized_layer_input, m, step_size, boundary_idx, percentile,
                           reg, lamb, groups, stochastic_quantization, device, compute_errors=True,
                           step_override=None, plan=None, seed=None, event_hook=None, check_status=True):
        '''_quantize_layer with the extra outputs the native path produces.  Per-call options (nothing is read
        from or left on the class): plan = kernel family (GPFQ_PLAN_*, None = auto); seed = Philox key of the
        stochastic quantizer (None = next value of a process-wide counter); event_hook = callable(tag, shape)
        invoked around the column preparation and the loop kernel (bench.py); check_status=False defers the status
        read of the plans that can time out to the caller (_lib.check_status) -- only for callers that neither
        consume nor forward the outputs before that.  Returns a dict with
        Q (N, d_g) fp32, idx (N, d_g) int8/int16 alphabet indices, U (local rows of the residual), step,
        and, if compute_errors, quantize_error / relative_quantize_error / quantize_adder / relative_adder.
        When neuron sharding is enabled (dist.enable) W's rows are split over the ranks, the index shards are
        all-gathered (one RCCL all_gather per layer) and Q is rebuilt locally; U stays sharded.'''
        _lib.require_gpu_tensor(W, "W")
        if W.dim() != 2:
            raise _lib.GpfqError("W must be (N, d)")
        W = W.contiguous()
        N, dg = W.shape
        # either input may already be in the kernels' column layout (PreparedColumns): both from the fused conv capture, or
        # the analog one alone, prepared ahead of time on another stream (StepAlgorithm.prepare_columns)
        a_prep = isinstance(analog_layer_input, PreparedColumns)
        x_prep = isinstance(quantized_layer_input, PreparedColumns)
        if a_prep:
            A, lda = analog_layer_input, 0
            _lib.require_gpu_tensor(A.T, "analog_layer_input")
        else:
            _lib.require_gpu_tensor(analog_layer_input, "analog_layer_input")
            A, lda = _rows_view(analog_layer_input, "analog_layer_input")
        if x_prep:
            X, ldx = quantized_layer_input, 0
            _lib.require_gpu_tensor(X.T, "quantized_layer_input")
        else:
            _lib.require_gpu_tensor(quantized_layer_input, "quantized_layer_input")
            X, ldx = _rows_view(quantized_layer_input, "quantized_layer_input")
        mm = A.shape[0]
        if tuple(A.shape) != (mm, groups * dg) or tuple(X.shape) != tuple(A.shape):
            raise _lib.GpfqError("layer inputs must be (m, groups*d): got %s / %s for W %s groups %d" % (
                tuple(A.shape), tuple(X.shape), tuple(W.shape), groups))
        if N % groups != 0:
            raise _lib.GpfqError("out_channels must be divisible by groups")
        K = int(boundary_idx)
        lamb_f = float(lamb if lamb is not None else 0.0)
        if step_override is not None:
            step_t = torch.tensor(step_override, dtype=torch.float32)
        else:
            step_t = StepAlgorithm._alphabet_step(W, step_size, K, percentile, reg, lamb_f)
        step = float(step_t)
        if reg == 'L1':
            mode = _lib.MODE_SOFT
        elif reg == 'L0':
            mode = _lib.MODE_HARD
        else:
            mode = _lib.MODE_STOCHASTIC if stochastic_quantization else _lib.MODE_MSQ
        if seed is None:
            seed = _default_seeds.take() if mode == _lib.MODE_STOCHASTIC else 0
        seed = int(seed)
        plan = _lib.PLAN_AUTO if plan is None else int(plan)
        idx_dtype, idx_bytes = _idx_dtype(K)
        dev = W.device
        st = _lib.current_stream_ptr(dev)

        def run_rows(W_loc, groups_loc, A_loc, lda_loc, X_loc, ldx_loc, row_id0):
            """all groups of a (sub)layer in one launch; returns Q, idx, U for those rows"""
            Nl = W_loc.shape[0]
            Q = torch.empty((Nl, dg), device=dev, dtype=torch.float32)
            idx = torch.empty((Nl, dg), device=dev, dtype=idx_dtype)
            U = torch.empty((Nl, mm), device=dev, dtype=torch.float32)
            mp = _lib.lib.gpfq_padded_m(mm)
            # per-segment sums of squares of the final residual, written by the loop kernel's last step (f2 epilogue)
            usq_seg = torch.empty((Nl, mp // 1024), device=dev, dtype=torch.float32)
            if Nl == 0 or dg == 0:
                Q.zero_(); idx.zero_(); U.zero_(); usq_seg.zero_()
                return Q, idx, U, usq_seg
            # the two halves of gpfq_quantize_layer_f32, called separately so that a profiler hook can
            # bracket the column preparation and the loop kernel with events on the current stream
            D = groups_loc * dg
            nrm2 = torch.empty((2 * D,), device=dev, dtype=torch.float32)          # {norm, reciprocal} per column (ABI 3)
            hook = event_hook
            if hook:
                hook("prepare_begin", (Nl, dg, mm, groups_loc))
            def as_columns(M, name):
                T = M.T
                if not (T.is_contiguous() and tuple(T.shape) == (D, mp)):
                    raise _lib.GpfqError("%s: PreparedColumns must be contiguous (D, m_pad)" % name)
                return T
            AT = as_columns(A_loc, "analog_layer_input") if a_prep else torch.empty((D, mp), device=dev, dtype=torch.float32)
            XT = as_columns(X_loc, "quantized_layer_input") if x_prep else torch.empty((D, mp), device=dev, dtype=torch.float32)
            if x_prep:
                # columns came out of the capture kernel already transposed and padded: only the norms are missing
                _lib.check(_lib.lib.gpfq_column_norms_f32(_ptr(XT), D, mm, mp, _ptr(nrm2), st))
            if not (a_prep and x_prep):
                # one pass: transpose + pad of the matrices that still need it, with the canonical column norms of X carried
                # along (`part`: one partial sum per column and 1024-sample segment); a prepared input is skipped (NULL)
                part = torch.empty((max(int(_lib.lib.gpfq_prepare_ws_bytes(D, mm)), 4) // 4,), device=dev, dtype=torch.float32)
                _lib.check(_lib.lib.gpfq_prepare_columns_ws_f32(
                    None if a_prep else _ptr(A_loc), lda_loc, None if x_prep else _ptr(X_loc), ldx_loc, mm, D,
                    _ptr(AT), _ptr(XT), _ptr(nrm2), mp, _ptr(part), part.numel() * 4, st))
            if hook:
                hook("loop_begin", (Nl, dg, mm, groups_loc))
            def launch(pl, scr):
                _lib.check(_lib.lib.gpfq_quantize_groups_prepared_f32(
                    _ptr(W_loc), _ptr(Q), _ptr(U), _ptr(AT), _ptr(XT), _ptr(nrm2), Nl, dg, mm, mp, groups_loc,
                    step, K, mode, lamb_f, seed, int(row_id0), _ptr(idx), idx_bytes, _ptr(usq_seg), pl,
                    _ptr(scr), scr.numel(), st))
                return bool(_lib.lib.gpfq_last_launch_used_exchange())

            # scratch -> launch -> status read (-> redo) under the device's lock (_lib.exclusive): the scratch area, its
            # granules and its status words are one per device, and two host threads on two streams must neither run two
            # cooperative grids on them at once nor read each other's timeout.  The lock covers THIS launch only -- not the
            # column preparation above, and never a collective: with neuron sharding the caller (dist.quantize_sharded) goes
            # on to an all_gather, and a rank that held the device's lock while blocked in it would deadlock against a peer
            # rank living in another thread of the same process, on the same card, waiting for that lock.
            with _lib.exclusive(dev):
                scr = _lib.scratch(dev)
                waits = launch(plan, scr)
                if hook:
                    hook("loop_end", (Nl, dg, mm, groups_loc))
                # A plan whose workgroups wait for each other can give up (bounded spins: another process on the card can
                # break co-residency).  Its outputs are not handed on -- to the caller, to the all_gather -- before the
                # status word has been read, and a timed-out layer is redone on the plan that waits for nobody.  The other
                # plans (resident, wave, whole-row streaming) cannot time out and cost no synchronisation here.
                if waits and check_status and not _lib.status_ok(dev):
                    timeouts.append((Nl, dg, mm))
                    launch(_lib.PLAN_STREAM_ROWS, scr)
            return Q, idx, U, usq_seg

        timeouts = []
        shard = _dist.active()
        if shard is None:
            Q, idx, U, usq_seg = run_rows(W, groups, A, lda, X, ldx, 0)
            rows = None
        else:
            Q, idx, U, usq_seg, rows = _dist.quantize_sharded(shard, W, A, lda, X, ldx, groups, dg, step, K, mode,
                                                              lamb_f, idx_dtype, run_rows)
        out = dict(Q=Q, idx=idx, U=U, usq_seg=usq_seg, step=step_t, rows=rows, timeouts=timeouts)
        if compute_errors:
            out.update(StepAlgorithm._error_metrics(W, A, usq_seg, groups, rows, shard, U))
        return out

    def _error_metrics(W, A, usq_seg, groups, rows, shard, U):
        '''step_algorithm.py:216-219 (groups == 1) and :239-243 (mean over groups of per-group norms).
        ||U[i, :]||^2 comes out of the loop kernel (usq_seg: its last step leaves one sum of squares per 1024-element
        segment; the few segments of a row are added here in float64), so U itself is not read again; A @ W.T is one
        MFMA GEMM (torch.matmul -> hipBLASLt), computed once instead of the reference's twice.'''
        if isinstance(A, PreparedColumns):
            A = A.matrix()
        N, dg = W.shape
        mm = A.shape[0]
        usq = usq_seg.double().sum(1)                                      # (rows,)  ||U[i, :]||^2
        if shard is not None:
            return _dist.sharded_error_metrics(shard, W, A, usq, groups, rows, U)
        if groups == 1:
            quantize_adder = U.T
            AW = A @ W.T
            relative_adder = (usq.sqrt() / (torch.linalg.norm(AW, axis=0).double() + 1e-5)).float()
            quantize_error = usq.sum().sqrt().float()
            relative_quantize_error = quantize_error / torch.linalg.norm(AW, ord='fro')
        else:
            Ng = N // groups
            un = usq.view(groups, Ng).sum(1).sqrt().float()
            A3 = A.reshape(mm, groups, dg).permute(1, 0, 2)                # (g, m, d_g)
            AW = torch.bmm(A3, W.view(groups, Ng, dg).transpose(1, 2))      # (g, m, Ng)
            an = torch.linalg.norm(AW.reshape(groups, -1), dim=1)
            quantize_error = un.sum() / groups
            relative_quantize_error = (un / an).sum() / groups
            quantize_adder = None
            relative_adder = None
        return dict(quantize_error=quantize_error, relative_quantize_error=relative_quantize_error,
                    quantize_adder=quantize_adder, relative_adder=relative_adder)

    def _quantize_layer(W, analog_layer_input, quantized_layer_input, m,
                        step_size, boundary_idx, percentile,
                        reg, lamb, groups, stochastic_quantization, device):
        '''Quantize one layer, all output neurons in parallel (step_algorithm.py:151-249).

        W : (N, d_g) layer weights (conv kernels already flattened by the caller)
        analog_layer_input, quantized_layer_input : (m, groups*d_g)
        step_size : alphabet step before scaling by the layer radius; boundary_idx : K = 2**(bits-1)
        Returns (Q, quantize_error, relative_quantize_error, quantize_adder, relative_adder) like the reference.
        '''
        print(f'The number of groups: {groups}\n')
        r = StepAlgorithm._quantize_layer_ex(W, analog_layer_input, quantized_layer_input, m, step_size,
                                             boundary_idx, percentile, reg, lamb, groups,
                                             stochastic_quantization, device)
        return r["Q"], r["quantize_error"], r["relative_quantize_error"], r["quantize_adder"], r["relative_adder"]
