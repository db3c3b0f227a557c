/*
 * gpfq.h -- C ABI of the MI355X-native GPFQ (greedy path-following quantization) hot path.
 *
 * This is the drop-in boundary: a plain C interface (device pointers + sizes + a hipStream_t passed as
 * void*), with no torch types.  Every entry point names the reference interface it replaces
 * (reference = YixuanSeanZhou/Quantized_Neural_Nets, file:line under src/).  The Python mirror of the
 * reference's operator surface (quantized_neural_nets_amd/step_algorithm.py) binds these with ctypes;
 * INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - all pointers are DEVICE pointers (hipMalloc / torch CUDA tensors) unless the name ends in _host;
 *   - all matrices are row-major fp32 with an explicit leading dimension in ELEMENTS;
 *   - every call is asynchronous on `stream` (a hipStream_t; NULL = the legacy default stream);
 *   - return value 0 = ok, negative = error; gpfq_last_error() gives the message for this thread;
 *   - nothing is allocated inside a call: the caller provides the workspace (gpfq_workspace_bytes).
 */
#ifndef GPFQ_H
#define GPFQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPFQ_ABI_VERSION 3   /* 2: usq_seg outputs, GPFQ_PLAN_STREAM_ROWS, gpfq_last_launch_used_exchange
                                3: nrm2 holds {norm, reciprocal} pairs: 2 * D floats */

/* quantizer selection, step_algorithm.py:198-208 */
enum {
    GPFQ_MODE_MSQ = 0,        /* StepAlgorithm._msq                    step_algorithm.py:38-56   */
    GPFQ_MODE_SOFT = 1,       /* StepAlgorithm._soft_thresholding_msq  step_algorithm.py:84-104  (reg == 'L1') */
    GPFQ_MODE_HARD = 2,       /* StepAlgorithm._hard_thresholding_msq  step_algorithm.py:59-81   (reg == 'L0') */
    GPFQ_MODE_STOCHASTIC = 3  /* StepAlgorithm._stochastic_msq         step_algorithm.py:7-35    (counter-based RNG) */
};

/* kernel family selection (GPFQ_PLAN_AUTO picks from (N, m) -- see DESIGN.md "Kernel plans") */
enum {
    GPFQ_PLAN_AUTO = 0,
    GPFQ_PLAN_STREAM = 1,     /* residual U streamed through HBM/L2 every step (any size)        */
    GPFQ_PLAN_RESIDENT = 2,   /* residual U resident in registers for the whole column loop      */
    GPFQ_PLAN_COOP = 3,       /* resident, each row split by columns over C co-operating workgroups; more rows than
                                 the chip holds at once run as one launch per block of rows on the caller's stream */
    GPFQ_PLAN_STREAM_ROWS = 4 /* streamed, whole rows per workgroup: never waits for another workgroup (the
                                 fallback after GPFQ_ERR_TIMEOUT; slower than the other plans on few long rows) */
};

/* error codes */
enum {
    GPFQ_OK = 0,
    GPFQ_ERR_ARG = -1,        /* bad shape / pointer / enum                                       */
    GPFQ_ERR_WORKSPACE = -2,  /* workspace too small                                              */
    GPFQ_ERR_HIP = -3,        /* a HIP runtime call failed (message has hipGetErrorString)        */
    GPFQ_ERR_UNSUPPORTED = -4,/* plan cannot run this shape (e.g. forced RESIDENT with huge m)    */
    GPFQ_ERR_TIMEOUT = -5     /* a cooperative kernel gave up waiting for a peer (gpfq_read_status) */
};

int gpfq_abi_version(void);
const char* gpfq_last_error(void);

/* m rounded up to the segment size (1024) the kernels and the canonical reduction use */
int64_t gpfq_padded_m(int64_t m);

/*
 * Bytes of device workspace gpfq_quantize_layer_f32 needs for a layer: the scratch area below, then the
 * transposed+padded activation columns AT, XT [groups*d_g][m_pad] and the column norms [groups*d_g].
 */
size_t gpfq_workspace_bytes(int64_t N, int64_t d_g, int64_t m, int groups);

/*
 * Scratch area of the cooperative plan (exchange granules + status words).  256-byte aligned, ZEROED ONCE
 * by the caller after allocation; the library re-zeroes the granules before every launch and leaves the
 * status words alone until gpfq_read_status reads them.  Without scratch the cooperative plan is not used.
 */
size_t gpfq_scratch_bytes(void);

/*
 * Synchronises `stream` and copies the 4 status words to the host: [0] != 0 means a cooperative kernel
 * timed out waiting for a peer workgroup ([1..3] = column, row tile, member) and its outputs are invalid.
 * Returns GPFQ_ERR_TIMEOUT in that case (and clears the words), else 0.
 */
int gpfq_read_status(void* scratch, int* status_host4, void* stream);

/*
 * 1 if the last loop launch made from this host thread used a plan whose workgroups wait for each other (the
 * cooperative plan, or the streaming plan with a row's columns split over workgroups), i.e. if its outputs are
 * valid only once gpfq_read_status has returned 0; 0 for the plans that cannot time out.  Lets a caller skip the
 * synchronising status read after the other launches.
 */
int gpfq_last_launch_used_exchange(void);

/*
 * Diagnostics of the cooperative plans (no reference counterpart: the reference has no workgroups that wait for each other).
 * gpfq_spin_limit_word: the bound the kernels count their polls against, 256 * min(GPFQ_COOP_SPIN_LIMIT, 2^24 - 2) -- the
 *   kernels test `(spins += 256) > word + pause` (pause < 32), so the word must leave the 32-bit counter room to EXCEED it.
 * gpfq_coop_launch_api_active: 1 if the next cooperative grid on the current device is launched through
 *   hipLaunchCooperativeKernel (the runtime then refuses a grid that cannot be co-resident instead of letting it spin to its
 *   bound): always with GPFQ_COOP_LAUNCH_API=1, never with -1, and by default from the first timeout gpfq_read_status has
 *   reported on that device in this process (a shared card), for the rest of the process.
 * gpfq_clear_contention: forget that timeout for the current device (a host that knows the other tenant has left).
 */
unsigned gpfq_spin_limit_word(void);
int gpfq_coop_launch_api_active(void);
void gpfq_clear_contention(void);

/*
 * Column preparation: AT[t][k] = A[k][t], XT[t][k] = X[k][t] for k < m, zero for m <= k < m_pad, and
 * nrm2[2t] = ||X[:, t]||_2 ** 2 exactly as step_algorithm.py:142 spells it (sqrt of the sum of squares,
 * squared) with the canonical reduction order, nrm2[2t + 1] = 1 / nrm2[2t] (0 for a zero column; the loop kernels use
 * it to find the alphabet index without a division on their critical path, and fall back to the reference's two
 * divisions whenever the result could depend on them).  Replaces the strided column reads of
 * step_algorithm.py:141-148 (analog_layer_input[:, t], quantized_layer_input[:, t]).
 *   A, X   [m][lda / ldx]   D = number of columns used (groups * d_g)
 *   AT, XT [D][m_pad]       nrm2 [2 * D]  (pairs; a group's slice starts at 2 * g * d_g)
 * One of A, X may be NULL: that matrix is skipped (its outputs are not touched; nrm2 belongs to X) -- the analog columns
 * do not depend on the quantized layers before them and can be prepared ahead, on another stream.
 */
int gpfq_prepare_columns_f32(const float* A, int64_t lda, const float* X, int64_t ldx, int64_t m, int64_t D,
                             float* AT, float* XT, float* nrm2, int64_t m_pad, void* stream);

/*
 * The same in ONE pass over A and X: the transposing kernel also carries every column's canonical chains of x * x, so
 * that XT is not read again for the norms (bit-identical nrm2).  ws: gpfq_prepare_ws_bytes(D, m) bytes of device memory
 * (one partial sum per column and 1024-sample segment), 4-byte aligned; with ws == NULL (or too small) this IS
 * gpfq_prepare_columns_f32.
 */
size_t gpfq_prepare_ws_bytes(int64_t D, int64_t m);
int gpfq_prepare_columns_ws_f32(const float* A, int64_t lda, const float* X, int64_t ldx, int64_t m, int64_t D,
                                float* AT, float* XT, float* nrm2, int64_t m_pad, void* ws, size_t ws_bytes, void* stream);

/*
 * The GPFQ loop on one group, in place on Q and U -- replaces
 *   StepAlgorithm._quantization(W, Q, U, analog_layer_input, quantized_layer_input, quantizer,
 *                               step_size, boundary_idx, lamb)            step_algorithm.py:107-148
 * with the columns already prepared (gpfq_prepare_columns_f32).
 *   W   [N][ldw]  read only                 Q   [N][ldq]  written (alphabet values, fp32)
 *   U   [N][ldu]  residual: read as the initial value if u_has_init != 0 (else taken as 0), written at the end
 *   AT, XT [d][m_pad], nrm2 [2 * d]          idx [N][ldi]  optional alphabet indices (may be NULL)
 *   idx_bytes 1 (int8, needs K <= 126) or 2 (int16)
 *   index encoding: msq / soft / stochastic -> k in [-K, K], Q = sign(k)*step*|k|;
 *                   hard -> 0 or +-(k+1), k in [0, K], Q = +-(lamb + step*k)
 *   seed, row_id0: key of the counter-based generator of GPFQ_MODE_STOCHASTIC (row_id0 = global index of row 0)
 *   usq_seg [N][m_pad/1024]  optional (may be NULL): sum of squares of every 1024-element segment of the FINAL
 *                   residual, accumulated in the kernel's last step -- the caller sums a row's segments to get
 *                   ||U[i,:]||^2 for the error metrics of step_algorithm.py:216-219 / :239-243 without a pass over U
 *   plan: GPFQ_PLAN_*       scratch: gpfq_scratch_bytes() bytes or NULL (then the cooperative plan is not used)
 */
int gpfq_quantization_f32(const float* W, int64_t ldw, float* Q, int64_t ldq, float* U, int64_t ldu,
                          int u_has_init, const float* AT, const float* XT, const float* nrm2,
                          int64_t N, int64_t d, int64_t m, int64_t m_pad,
                          float step, int K, int mode, float lamb, uint64_t seed, uint64_t row_id0,
                          void* idx, int64_t ldi, int idx_bytes, float* usq_seg, int plan, void* scratch,
                          size_t scratch_bytes, void* stream);

/*
 * One whole layer (all groups in one launch) -- the native part of
 *   StepAlgorithm._quantize_layer(W, analog_layer_input, quantized_layer_input, m, step_size, boundary_idx,
 *                                 percentile, reg, lamb, groups, stochastic_quantization, device)
 *                                                                          step_algorithm.py:151-249
 * i.e. lines :194-196 (Q, U allocation is the caller's), :212-214 (groups == 1) and :221-237 (grouped loop).
 * `step` is the final alphabet step (step_algorithm.py:191-192), computed by the caller.
 *   W, Q [N][d_g] contiguous     U [N][m] contiguous (written; initial residual is 0)
 *   A, X [m][lda / ldx], group g uses columns [g*d_g, (g+1)*d_g) and rows [g*N/groups, (g+1)*N/groups) of W
 *   idx  [N][d_g] optional       workspace >= gpfq_workspace_bytes(N, d_g, m, groups), 256-byte aligned
 *   row_id0: global index of row 0 (keys the stochastic generator when a layer is sharded by rows)
 */
int gpfq_quantize_layer_f32(const float* W, const float* A, int64_t lda, const float* X, int64_t ldx,
                            int64_t N, int64_t d_g, int64_t m, int groups,
                            float step, int K, int mode, float lamb, uint64_t seed, uint64_t row_id0,
                            float* Q, void* idx, int idx_bytes, float* U, float* usq_seg,
                            void* workspace, size_t workspace_bytes, int plan, void* stream);

/*
 * The second half of gpfq_quantize_layer_f32 on its own: all groups of a layer in one launch, on columns
 * already prepared by gpfq_prepare_columns_f32 with D = groups*d_g (AT, XT [groups*d_g][m_pad], nrm2 pairs
 * [2*groups*d_g]).  Lets a caller time / overlap the column preparation and the loop separately.
 */
int gpfq_quantize_groups_prepared_f32(const float* W, float* Q, float* U, const float* AT, const float* XT,
                                      const float* nrm2, int64_t N, int64_t d_g, int64_t m, int64_t m_pad,
                                      int groups, float step, int K, int mode, float lamb, uint64_t seed,
                                      uint64_t row_id0, void* idx, int idx_bytes, float* usq_seg, int plan,
                                      void* scratch, size_t scratch_bytes, void* stream);

/*
 * Elementwise quantizer on a device vector (the four quantizers as standalone ops, for known-answer
 * tests): out[i] = quantizer(step, x[i], K, lamb).  step_algorithm.py:7-104.
 * uniform: per-element U[0,1) draws for GPFQ_MODE_STOCHASTIC (may be NULL for the other modes).
 */
int gpfq_quantizer_f32(int mode, float step, const float* x, int64_t n, int K, float lamb,
                       const float* uniform, float* out, int32_t* idx, void* stream);

/*
 * The uniform draws GPFQ_MODE_STOCHASTIC takes inside the loop kernels, on their own: out[i] = U[0,1) of the
 * counter-based generator at (seed, row row_id0 + i, column) -- pass them as `uniform` to gpfq_quantizer_f32 to quantize
 * the projections of one column exactly as the loop does at that step (step_algorithm.py:27-35 draws from torch's global
 * CPU stream instead, which no GPU path can reproduce).
 */
int gpfq_philox_uniform_f32(uint64_t seed, uint64_t row_id0, uint64_t column, int64_t n, float* out, void* stream);

/*
 * nrm2[2t] = ||column t||_2 ** 2 and nrm2[2t + 1] = its reciprocal (nrm2 [2 * D]), of columns that are already in the
 * prepared layout XT [D][m_pad] (step_algorithm.py:142), canonical reduction order -- the norm half of gpfq_prepare_columns_f32.
 */
int gpfq_column_norms_f32(const float* XT, int64_t D, int64_t m, int64_t m_pad, float* nrm2, void* stream);

/*
 * Fused activation capture for Conv2d layers -- replaces SaveInputConv2d.__call__'s unfold / transpose /
 * reshape / index (quantize_neural_net.py:334-347) AND the transpose of gpfq_prepare_columns_f32: the sampled
 * kernel-sized patches of an NCHW feature map are written directly in the prepared column layout.
 *   x            [B][C][H][W] contiguous fp32
 *   patch_index  [m] int64, the reference's rand_indices: b*L + l with L = Lh*Lw blocks per image; blocks lie
 *                on a grid whose stride is the kernel size (nn.Unfold(kernel, dilation, padding, kernel), :320)
 *   outT         [C*kh*kw][m_pad]: row f = (c, i, j) channel-major, column k = patch k; zero for k >= m
 */
int gpfq_gather_patches_f32(const float* x, int64_t B, int64_t C, int64_t H, int64_t W, int kh, int kw, int pad_h,
                            int pad_w, int dil_h, int dil_w, const int64_t* patch_index, int64_t m, float* outT,
                            int64_t m_pad, void* stream);

/*
 * Row statistic for the alphabet radius, percentile == 1 only: rowmax[i] = max_j |W[i][j]|
 * (torch.quantile(|W|, 1, axis=1), step_algorithm.py:191).
 */
int gpfq_row_absmax_f32(const float* W, int64_t ldw, int64_t N, int64_t d, float* rowmax, void* stream);

/* Writes a one-line description of the plan `plan` resolves to for this shape (GPFQ_PLAN_AUTO: the plan AUTO would
 * pick) with the quantizer GPFQ_MODE_MSQ; returns the plan id or a negative error. */
int gpfq_describe_plan(int64_t N, int64_t d_g, int64_t m, int groups, int plan, char* buf, size_t buf_bytes);
/* The same for a given quantizer: the plan depends on `mode` where a kernel variant is not instantiated for every
 * quantizer (two cooperative variants have no GPFQ_MODE_STOCHASTIC form), so that the description is always the
 * plan that launches. */
int gpfq_describe_plan_mode(int64_t N, int64_t d_g, int64_t m, int groups, int plan, int mode, char* buf,
                            size_t buf_bytes);

#ifdef __cplusplus
}
#endif
#endif /* GPFQ_H */
