/*
 * gpfq_oracle.c -- CPU restatement of the reference GPFQ per-layer quantization loop.
 *
 * THIS FILE IS TEST INFRASTRUCTURE.  It is the checker the HIP path is compared with;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product (quantized_neural_nets_amd/) never links, imports or calls it.
 *
 * What it restates (reference = YixuanSeanZhou/Quantized_Neural_Nets, paths under src/):
 *   gpfq_oracle_msq / _soft / _hard / _stochastic  <- step_algorithm.py:38-56, 84-104, 59-81, 7-35
 *   gpfq_oracle_quantization                       <- step_algorithm.py:107-148 (loop :140-148)
 *   gpfq_oracle_quantize_groups                    <- step_algorithm.py:212-247 (groups==1 and grouped loop)
 *
 * Parity pin: checked against the golden vectors in tests/golden/ (npz files), which were produced by
 * importing the reference's own step_algorithm.py in the build container (tools/make_golden.py).
 * Alphabet indices must match exactly; the residual U within 1e-5 (tests/test_oracle_golden.py).
 *
 * Arithmetic (all fp32, every operation individually rounded, compiled with -ffp-contract=off):
 *   step_algorithm.py:141   u[k] = u[k] + (w_t * a_t[k])          mul, then add
 *   step_algorithm.py:142   nrm  = sqrtf(cdot(x_t, x_t)); nrm = nrm * nrm   ("norm(x,2) ** 2")
 *   step_algorithm.py:143-6 s    = nrm > 0 ? cdot(u, x_t) / nrm : 0
 *   step_algorithm.py:147   q    = quantizer(step, s, K, lamb)
 *   step_algorithm.py:148   u[k] = u[k] - (q * x_t[k])             mul, then sub
 *
 * The reference evaluates the two reductions (linalg.norm, U.matmul(x)) with whatever order its BLAS /
 * ATen build uses; that order is not specified, so this restatement fixes ONE canonical order, cdot(),
 * shared bit-for-bit with the HIP kernels (see DESIGN.md "Canonical reduction order"):
 *   - the vector is zero-padded to a multiple of 1024 and cut into segments of 1024 elements;
 *   - inside segment s, "lane" l (0..63) owns the 16 elements 1024 s + 256 c + 4 l + j (c, j = 0..3) and
 *     accumulates them in that order with a fused multiply-add chain starting from +0.0f;
 *   - the 64 lane sums are added by a balanced pairwise tree over the lane index (tree64);
 *   - the S segment sums are placed into P = 2^ceil(log2 S) slots, segment s at slot floor(s*P/S) (the
 *     other slots hold +0.0f), and the slots are added by a balanced pairwise tree over the slot index.
 *     (Any aligned power-of-two block of slots holds S/C +- 1 segments, which is what lets C workgroups
 *     each pre-reduce a block of one row and still reproduce this order bit for bit.)
 * Zero padding is an exact no-op for every step of that definition.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define GPFQ_SEG 1024

enum { GPFQ_MODE_MSQ = 0, GPFQ_MODE_SOFT = 1, GPFQ_MODE_HARD = 2, GPFQ_MODE_STOCHASTIC = 3 };

/* torch.sign: (0 < x) - (x < 0); sign(-0.0) == +0.0 */
static inline float sgnf(float x) { return (float)((0.0f < x) - (x < 0.0f)); }

/* ---- quantizers -------------------------------------------------------------------------------- */

/* step_algorithm.py:56  sign(x) * step * min(|floor(x/step + 0.5)|, K) */
float gpfq_oracle_msq(float step, float x, int K, float lamb, int* idx)
{
    (void)lamb;
    float z = x / step;
    z = z + 0.5f;
    float r = fminf(fabsf(floorf(z)), (float)K);
    float sg = sgnf(x);
    if (idx) *idx = (int)(sg * r);
    return (sg * step) * r;
}

/* step_algorithm.py:103-104  soft threshold, then msq */
float gpfq_oracle_soft(float step, float x, int K, float lamb, int* idx)
{
    float y = sgnf(x) * fmaxf(fabsf(x) - lamb, 0.0f);
    return gpfq_oracle_msq(step, y, K, lamb, idx);
}

/* step_algorithm.py:78-81  hard threshold; alphabet {0} U +-(lamb + step*k), k = 0..K.
 * idx encoding: 0, or sign * (k + 1). */
float gpfq_oracle_hard(float step, float x, int K, float lamb, int* idx)
{
    float ax = fabsf(x);
    float x1 = (ax > lamb ? ax : 0.0f) * sgnf(x);            /* F.threshold(|x|, lamb, 0) * sign(x) */
    float s1 = sgnf(x1);
    float y = s1 * fmaxf(fabsf(x1) - lamb, 0.0f);
    float z = y / step;
    z = z + 0.5f;
    float rv = fminf(fabsf(floorf(z)), (float)K);
    float mask = (fabsf(x1) > lamb) ? 1.0f : 0.0f;
    float mag = lamb + step * rv;
    if (idx) *idx = (mask != 0.0f) ? (int)(s1 * (rv + 1.0f)) : 0;
    return (s1 * mag) * mask;
}

/* Philox4x32-10, one block per (row, column) -> uniform in [0,1).  The reference draws from torch's
 * global generator (torch.bernoulli, step_algorithm.py:28); that stream cannot be reproduced outside
 * torch, so the stochastic mode is pinned HIP-vs-oracle bit-exactly and against the reference only in
 * distribution. */
static inline uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
float gpfq_oracle_philox_uniform(uint64_t seed, uint64_t row, uint64_t col)
{
    uint32_t c0 = (uint32_t)col, c1 = (uint32_t)(col >> 32), c2 = (uint32_t)row, c3 = (uint32_t)(row >> 32);
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = mulhi32(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = mulhi32(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return (float)(c0 >> 8) * (1.0f / 16777216.0f);
}

/* the draws of rows row0 .. row0 + n - 1 at one column (what the loop's stochastic quantizer consumes at that step) */
void gpfq_oracle_philox_uniform_vec(uint64_t seed, uint64_t row0, uint64_t col, long n, float* out)
{
    for (long i = 0; i < n; ++i) out[i] = gpfq_oracle_philox_uniform(seed, row0 + (uint64_t)i, col);
}

/* step_algorithm.py:27-35  p = 1 - x/step + floor(x/step); Bernoulli(p) -> round down, else up; clip */
float gpfq_oracle_stochastic(float step, float x, int K, float uniform, int* idx)
{
    float z = x / step;
    float fl = floorf(z);
    float p = (1.0f - z) + fl;
    float lev = (uniform < p) ? fl : (fl + 1.0f);
    float q = step * lev;
    if (fabsf(q) > step * (float)K) {
        float sg = sgnf(q);
        q = (sg * step) * (float)K;
        lev = sg * (float)K;
    }
    if (idx) *idx = (int)lev;
    return q;
}

/* ---- canonical reduction ----------------------------------------------------------------------- */

static inline float tree64(float* v)
{
    for (int off = 1; off < 64; off <<= 1)
        for (int l = 0; l < 64; l += 2 * off) v[l] = v[l] + v[l + off];
    return v[0];
}

/* second level: segment sums -> P slots -> pairwise tree */
static float slot_tree(const float* seg, long S)
{
    long P = 1;
    while (P < S) P <<= 1;
    float* slot = (float*)malloc((size_t)P * sizeof(float));
    for (long i = 0; i < P; ++i) slot[i] = 0.0f;
    for (long s = 0; s < S; ++s) slot[(s * P) / S] = seg[s];
    for (long off = 1; off < P; off <<= 1)
        for (long l = 0; l < P; l += 2 * off) slot[l] = slot[l] + slot[l + off];
    float r = slot[0];
    free(slot);
    return r;
}

/* u, x: length S*1024 (zero padded) */
float gpfq_oracle_cdot(const float* u, const float* x, long S)
{
    float segbuf[64];
    float* seg = S <= 64 ? segbuf : (float*)malloc((size_t)S * sizeof(float));
    for (long s = 0; s < S; ++s) {
        float acc[64];
        const float* us = u + s * GPFQ_SEG;
        const float* xs = x + s * GPFQ_SEG;
        for (int l = 0; l < 64; ++l) acc[l] = 0.0f;
        for (int c = 0; c < 4; ++c)
            for (int l = 0; l < 64; ++l) {
                const float* up = us + 256 * c + 4 * l;
                const float* xp = xs + 256 * c + 4 * l;
                float a = acc[l];
                a = fmaf(up[0], xp[0], a);
                a = fmaf(up[1], xp[1], a);
                a = fmaf(up[2], xp[2], a);
                a = fmaf(up[3], xp[3], a);
                acc[l] = a;
            }
        seg[s] = tree64(acc);
    }
    float r = slot_tree(seg, S);
    if (seg != segbuf) free(seg);
    return r;
}

/* ---- the loop ---------------------------------------------------------------------------------- */

static long pad_seg(long m) { return ((m + GPFQ_SEG - 1) / GPFQ_SEG) * GPFQ_SEG; }

/* one row, whole t loop.  AT/XT: [d][mp] transposed+padded columns; nrm2[d]. */
static void run_row(const float* w, long d, long m, long mp, const float* AT, const float* XT, const float* nrm2,
                    float* u_io /* m */, float* q_out, long q_stride, int16_t* idx_out, long idx_stride,
                    float step, int K, int mode, float lamb, uint64_t seed, uint64_t row_id, float* u /* mp scratch */)
{
    long S = mp / GPFQ_SEG;
    memcpy(u, u_io, (size_t)m * sizeof(float));
    for (long k = m; k < mp; ++k) u[k] = 0.0f;
    for (long t = 0; t < d; ++t) {
        const float* a = AT + t * mp;
        const float* x = XT + t * mp;
        float wt = w[t];
        for (long k = 0; k < mp; ++k) {
            float p = wt * a[k];
            u[k] = u[k] + p;
        }
        float s = 0.0f;
        if (nrm2[t] > 0.0f) s = gpfq_oracle_cdot(u, x, S) / nrm2[t];
        int id = 0;
        float q;
        switch (mode) {
        case GPFQ_MODE_SOFT: q = gpfq_oracle_soft(step, s, K, lamb, &id); break;
        case GPFQ_MODE_HARD: q = gpfq_oracle_hard(step, s, K, lamb, &id); break;
        case GPFQ_MODE_STOCHASTIC:
            q = gpfq_oracle_stochastic(step, s, K, gpfq_oracle_philox_uniform(seed, row_id, (uint64_t)t), &id);
            break;
        default: q = gpfq_oracle_msq(step, s, K, lamb, &id); break;
        }
        q_out[t * q_stride] = q;
        if (idx_out) idx_out[t * idx_stride] = (int16_t)id;
        for (long k = 0; k < mp; ++k) {
            float p = q * x[k];
            u[k] = u[k] - p;
        }
    }
    memcpy(u_io, u, (size_t)m * sizeof(float));
}

/*
 * In-place GPFQ loop on one group (mirror of StepAlgorithm._quantization, step_algorithm.py:107-148).
 *   W  [N][ldw]  read      Q [N][ldq] written      U [N][ldu] read (initial residual) and written
 *   A, X [m][lda/ldx]: column t of this group is A[k*lda + t]
 *   idx (optional) [N][ldi] int16 alphabet indices
 *   row_id0: global row number of row 0 (only used to key the stochastic generator)
 * returns 0, or -1 on allocation failure.
 */
int gpfq_oracle_quantization(const float* W, long ldw, float* Q, long ldq, float* U, long ldu,
                             const float* A, long lda, const float* X, long ldx,
                             long N, long d, long m, float step, int K, int mode, float lamb,
                             uint64_t seed, uint64_t row_id0, int16_t* idx, long ldi, int nthreads)
{
    long mp = pad_seg(m > 0 ? m : 1);
    float* AT = (float*)calloc((size_t)(d > 0 ? d : 1) * mp, sizeof(float));
    float* XT = (float*)calloc((size_t)(d > 0 ? d : 1) * mp, sizeof(float));
    float* nrm2 = (float*)calloc((size_t)(d > 0 ? d : 1), sizeof(float));
    if (!AT || !XT || !nrm2) { free(AT); free(XT); free(nrm2); return -1; }
    if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(static)
#endif
    for (long t = 0; t < d; ++t) {
        float* at = AT + t * mp;
        float* xt = XT + t * mp;
        for (long k = 0; k < m; ++k) { at[k] = A[k * lda + t]; xt[k] = X[k * ldx + t]; }
        float r = sqrtf(gpfq_oracle_cdot(xt, xt, mp / GPFQ_SEG));
        nrm2[t] = r * r;
    }
    int fail = 0;
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
    {
        float* u = (float*)malloc((size_t)mp * sizeof(float));
        if (!u) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
            fail = 1;
        }
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (long i = 0; i < N; ++i) {
            if (!u) continue;
            run_row(W + i * ldw, d, m, mp, AT, XT, nrm2, U + i * ldu, Q + i * ldq, 1,
                    idx ? idx + i * ldi : NULL, 1, step, K, mode, lamb, seed, row_id0 + (uint64_t)i, u);
        }
        free(u);
    }
    free(AT); free(XT); free(nrm2);
    return fail ? -1 : 0;
}

/*
 * All groups of one layer (step_algorithm.py:212-214 for groups == 1, :221-237 for groups > 1).
 *   W, Q [N][d_g] contiguous; U [N][m] contiguous; A, X [m][groups*d_g] contiguous;
 *   group i owns rows [i*N/groups, (i+1)*N/groups) and columns [i*d_g, (i+1)*d_g) of A and X.
 */
int gpfq_oracle_quantize_groups(const float* W, float* Q, float* U, const float* A, const float* X,
                                long N, long d_g, long m, int groups, float step, int K, int mode, float lamb,
                                uint64_t seed, int16_t* idx, int nthreads)
{
    if (groups < 1 || N % groups != 0) return -2;
    long Ng = N / groups;
    long ld = (long)groups * d_g;
    for (int g = 0; g < groups; ++g) {
        int rc = gpfq_oracle_quantization(W + (long)g * Ng * d_g, d_g, Q + (long)g * Ng * d_g, d_g,
                                          U + (long)g * Ng * m, m, A + (long)g * d_g, ld, X + (long)g * d_g, ld,
                                          Ng, d_g, m, step, K, mode, lamb, seed, (uint64_t)g * Ng,
                                          idx ? idx + (long)g * Ng * d_g : NULL, d_g, nthreads);
        if (rc) return rc;
    }
    return 0;
}

/* elementwise quantizer on a vector (known-answer tests) */
void gpfq_oracle_quantizer_vec(int mode, float step, const float* x, long n, int K, float lamb,
                               const float* uniform, float* out, int* idx)
{
    for (long i = 0; i < n; ++i) {
        int id = 0;
        float q;
        switch (mode) {
        case GPFQ_MODE_SOFT: q = gpfq_oracle_soft(step, x[i], K, lamb, &id); break;
        case GPFQ_MODE_HARD: q = gpfq_oracle_hard(step, x[i], K, lamb, &id); break;
        case GPFQ_MODE_STOCHASTIC: q = gpfq_oracle_stochastic(step, x[i], K, uniform ? uniform[i] : 0.5f, &id); break;
        default: q = gpfq_oracle_msq(step, x[i], K, lamb, &id); break;
        }
        out[i] = q;
        if (idx) idx[i] = id;
    }
}

int gpfq_oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
