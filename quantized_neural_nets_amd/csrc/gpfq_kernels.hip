// gpfq_kernels.hip -- hand-written gfx950 kernels of the GPFQ hot path and the C ABI of include/gpfq.h.
//
// Path (reference = YixuanSeanZhou/Quantized_Neural_Nets, src/):
//   StepAlgorithm._quantization   step_algorithm.py:107-148   -> gpfq_resident_kernel / gpfq_stream_kernel
//   quantizers                    step_algorithm.py:7-104     -> gpfq_device.h quant_*
//   column reads [:, t], norm     step_algorithm.py:141-144   -> gpfq_transpose_pad_kernel, gpfq_colnorm_kernel
//
// One launch runs the WHOLE column loop of a layer (all groups): rows of the residual U are independent,
// so a workgroup that owns a set of output neurons needs no inter-workgroup synchronisation.  Per step a
// workgroup makes ONE pass over its rows, fusing  u -= q_{t-1} x_{t-1};  u += w_t a_t;  <u, x_t>.
// Compile with -ffp-contract=off (see gpfq_device.h).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>

#include "../../include/gpfq.h"
#include "gpfq_device.h"

namespace gpfq {

struct LoopParams {
    const float* W; int64_t ldw;
    float* Q; int64_t ldq;
    float* U; int64_t ldu; int u_has_init;
    const float* AT; const float* XT; const float* nrm2;
    int64_t Ng;        // rows per group
    int64_t d;         // columns per group
    int64_t m; int64_t m_pad; int S;
    QuantCfg qc;
    uint64_t row_id0;
    void* idx; int64_t ldi; int idx_bytes;
};

__device__ __forceinline__ void store_q(const LoopParams& p, int64_t grow, int64_t t, float q, int id)
{
    p.Q[grow * p.ldq + t] = q;
    if (p.idx) {
        if (p.idx_bytes == 1) reinterpret_cast<int8_t*>(p.idx)[grow * p.ldi + t] = (int8_t)id;
        else reinterpret_cast<int16_t*>(p.idx)[grow * p.ldi + t] = (int16_t)id;
    }
}

template <bool VEC>
__device__ __forceinline__ void load_u16(float (&u)[16], const float* __restrict__ Urow, int64_t kbase, int64_t m)
{
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int64_t k0 = kbase + 256 * c;
        if (VEC && k0 + 3 < m) {
            float4 v = *reinterpret_cast<const float4*>(Urow + k0);
            u[4 * c + 0] = v.x; u[4 * c + 1] = v.y; u[4 * c + 2] = v.z; u[4 * c + 3] = v.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) u[4 * c + j] = (k0 + j < m) ? Urow[k0 + j] : 0.0f;
        }
    }
}

template <bool VEC>
__device__ __forceinline__ void store_u16(const float (&u)[16], float* __restrict__ Urow, int64_t kbase, int64_t m)
{
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int64_t k0 = kbase + 256 * c;
        if (VEC && k0 + 3 < m) {
            *reinterpret_cast<float4*>(Urow + k0) = make_float4(u[4 * c + 0], u[4 * c + 1], u[4 * c + 2], u[4 * c + 3]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (k0 + j < m) Urow[k0 + j] = u[4 * c + j];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Resident plan: the residual row lives in registers for the whole column loop.  One wave per canonical
// segment (blockDim.x = 64*S, S <= 16), RT rows per workgroup sharing the activation registers.
// HBM/L2 traffic per step: only x_{t+1}, a_{t+1} (prefetched behind the reduction of step t).
// ------------------------------------------------------------------------------------------------
template <int RT, bool VEC>
__global__ void __launch_bounds__(1024) gpfq_resident_kernel(LoopParams p)
{
    extern __shared__ float smem[];                 // [2][RT][S] segment sums, double buffered by step parity
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int S = p.S;
    const int P = pow2_ceil(S);                     // S <= 16 here: one slot per lane
    const SlotMap smap = make_slot_map(S, P, 0, 1, lane, P);
    const int g = blockIdx.y;
    const int64_t row0 = (int64_t)blockIdx.x * RT;  // row inside the group
    const int64_t colbase = ((int64_t)g * p.d) * p.m_pad + (int64_t)wave * kSeg + 4 * lane;
    const float* __restrict__ ATp = p.AT + colbase;
    const float* __restrict__ XTp = p.XT + colbase;
    const float* __restrict__ nrm = p.nrm2 + (int64_t)g * p.d;
    const int64_t kbase = (int64_t)wave * kSeg + 4 * lane;

    int64_t grow[RT];
    bool valid[RT];
    float u[RT][16];
    const float* __restrict__ wrow[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        valid[r] = (row0 + r) < p.Ng;
        grow[r] = (int64_t)g * p.Ng + (valid[r] ? row0 + r : p.Ng - 1);
        wrow[r] = p.W + grow[r] * p.ldw;
        if (p.u_has_init) load_u16<VEC>(u[r], p.U + grow[r] * p.ldu, kbase, p.m);
        else {
#pragma unroll
            for (int e = 0; e < 16; ++e) u[r][e] = 0.0f;
        }
    }
    float qprev[RT], wcur[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) { qprev[r] = 0.0f; wcur[r] = wrow[r][0]; }
    float n2cur = nrm[0];

    float xa[16], xb[16], aa[16];
    load16(xa, XTp);
    load16(aa, ATp);

    auto body = [&](int64_t t, float (&xc)[16], float (&xo)[16]) {
        // xc = x_t, xo = x_{t-1} (dead after the sweep, then receives x_{t+1})
        float acc[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r)
            acc[r] = (t > 0) ? sweep16<true>(u[r], xo, aa, xc, qprev[r], wcur[r])
                             : sweep16<false>(u[r], xo, aa, xc, 0.0f, wcur[r]);
        const bool more = t + 1 < p.d;
        float wn[RT], n2n = 0.0f;
        if (more) {                                  // prefetch behind the reduction
            load16(xo, XTp + (t + 1) * p.m_pad);
            load16(aa, ATp + (t + 1) * p.m_pad);
#pragma unroll
            for (int r = 0; r < RT; ++r) wn[r] = wrow[r][t + 1];
            n2n = nrm[t + 1];
        }
        float* seg = smem + (size_t)(t & 1) * RT * S;
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            float sg = wave_tree64(acc[r]);
            if (lane == 0) seg[r * S + wave] = sg;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            float tot = combine_slots(seg + r * S, smap, 1, P, S - 1);
            float s = (n2cur > 0.0f) ? tot / n2cur : 0.0f;
            int id;
            float q = quantize(p.qc, s, p.row_id0 + (uint64_t)grow[r], (uint64_t)t, id);
            qprev[r] = q;
            if (threadIdx.x == 0 && valid[r]) store_q(p, grow[r], t, q, id);
        }
        if (more) {
#pragma unroll
            for (int r = 0; r < RT; ++r) wcur[r] = wn[r];
            n2cur = n2n;
        }
    };

    int64_t t = 0;
    for (; t + 1 < p.d; t += 2) { body(t, xa, xb); body(t + 1, xb, xa); }
    if (t < p.d) body(t, xa, xb);

    // pending subtraction of the last step, then write the residual (step_algorithm.py:148)
#pragma unroll
    for (int r = 0; r < RT; ++r) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float xl = (p.d & 1) ? xa[e] : xb[e];
            float pq = qprev[r] * xl;
            u[r][e] = u[r][e] - pq;
        }
    }
#pragma unroll
    for (int r = 0; r < RT; ++r)
        if (valid[r]) store_u16<VEC>(u[r], p.U + grow[r] * p.ldu, kbase, p.m);
}

// ------------------------------------------------------------------------------------------------
// Streaming plan: any (N, m).  The residual rows stay in the caller's U (HBM / L2 / Infinity Cache) and
// are read and written once per step; wave w owns segments w, w+NW, ... of the workgroup's RT rows.
// ------------------------------------------------------------------------------------------------
template <int RT, bool VEC>
__global__ void __launch_bounds__(512) gpfq_stream_kernel(LoopParams p)
{
    extern __shared__ float smem[];                 // [2][RT][S]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int NW = blockDim.x >> 6;
    const int S = p.S;
    const int P = pow2_ceil(S);
    const int per = P > 64 ? P / 64 : 1, nl = P > 64 ? 64 : P;
    const SlotMap smap = make_slot_map(S, P, 0, per, lane, nl);
    const int g = blockIdx.y;
    const int64_t row0 = (int64_t)blockIdx.x * RT;
    const float* __restrict__ ATg = p.AT + ((int64_t)g * p.d) * p.m_pad + 4 * lane;
    const float* __restrict__ XTg = p.XT + ((int64_t)g * p.d) * p.m_pad + 4 * lane;
    const float* __restrict__ nrm = p.nrm2 + (int64_t)g * p.d;

    int64_t grow[RT];
    bool valid[RT];
    float qprev[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        valid[r] = (row0 + r) < p.Ng;
        grow[r] = (int64_t)g * p.Ng + (valid[r] ? row0 + r : p.Ng - 1);
        qprev[r] = 0.0f;
    }

    for (int64_t t = 0; t <= p.d; ++t) {
        const bool last = (t == p.d);               // extra pass: only the pending subtraction
        const bool first = (t == 0);
        float w[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) w[r] = last ? 0.0f : p.W[grow[r] * p.ldw + t];
        float* seg = smem + (size_t)(t & 1) * RT * S;
        for (int s = wave; s < S; s += NW) {
            const int64_t kbase = (int64_t)s * kSeg + 4 * lane;
            float xp[16], xc[16], ac[16];
            if (!first) load16(xp, XTg + (t - 1) * p.m_pad + (int64_t)s * kSeg);
            if (!last) {
                load16(xc, XTg + t * p.m_pad + (int64_t)s * kSeg);
                load16(ac, ATg + t * p.m_pad + (int64_t)s * kSeg);
            }
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                float u[16];
                float* Urow = p.U + grow[r] * p.ldu;
                if (first && !p.u_has_init) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) u[e] = 0.0f;
                } else {
                    load_u16<VEC>(u, Urow, kbase, p.m);
                }
                if (last) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) { float pq = qprev[r] * xp[e]; u[e] = u[e] - pq; }
                } else {
                    float acc = first ? sweep16<false>(u, xc, ac, xc, 0.0f, w[r])
                                      : sweep16<true>(u, xp, ac, xc, qprev[r], w[r]);
                    float sg = wave_tree64(acc);
                    if (lane == 0) seg[r * S + s] = sg;
                }
                if (valid[r] || RT == 1) store_u16<VEC>(u, Urow, kbase, p.m);
            }
        }
        if (last) break;
        __syncthreads();
        const float n2 = nrm[t];
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            float tot = combine_slots(seg + r * S, smap, per, nl, S - 1);
            float sv = (n2 > 0.0f) ? tot / n2 : 0.0f;
            int id;
            float q = quantize(p.qc, sv, p.row_id0 + (uint64_t)grow[r], (uint64_t)t, id);
            qprev[r] = q;
            if (threadIdx.x == 0 && valid[r]) store_q(p, grow[r], t, q, id);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Column preparation
// ------------------------------------------------------------------------------------------------
// out[t][k] = in[k][t] for k < m, 0 for m <= k < m_pad.  blockIdx.z selects A or X.  64x64 tiles via LDS.
__global__ void __launch_bounds__(256) gpfq_transpose_pad_kernel(const float* __restrict__ A, int64_t lda,
                                                                 const float* __restrict__ X, int64_t ldx,
                                                                 int64_t m, int64_t D, float* __restrict__ AT,
                                                                 float* __restrict__ XT, int64_t m_pad)
{
    __shared__ float tile[64][65];
    const float* __restrict__ in = blockIdx.z ? X : A;
    const int64_t ld = blockIdx.z ? ldx : lda;
    float* __restrict__ out = blockIdx.z ? XT : AT;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t k0 = (int64_t)blockIdx.x * 64, t0 = (int64_t)blockIdx.y * 64;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int64_t k = k0 + ty + 4 * i, t = t0 + tx;
        tile[ty + 4 * i][tx] = (k < m && t < D) ? in[k * ld + t] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int64_t t = t0 + ty + 4 * i, k = k0 + tx;
        if (t < D) out[t * m_pad + k] = tile[tx][ty + 4 * i];
    }
}

// nrm2[t] = (sqrt(cdot(x_t, x_t)))^2, canonical order.  One 256-thread workgroup per column; dynamic LDS S floats.
__global__ void __launch_bounds__(256) gpfq_colnorm_kernel(const float* __restrict__ XT, int64_t m_pad, int S,
                                                           float* __restrict__ nrm2)
{
    extern __shared__ float seg[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float* __restrict__ x = XT + (int64_t)blockIdx.x * m_pad + 4 * lane;
    for (int s = wave; s < S; s += 4) {
        float xv[16];
        load16(xv, x + (int64_t)s * kSeg);
        float acc = 0.0f;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc = __builtin_fmaf(xv[e], xv[e], acc);
        float sg = wave_tree64(acc);
        if (lane == 0) seg[s] = sg;
    }
    __syncthreads();
    if (wave == 0) {
        const int P = pow2_ceil(S);
        const int per = P > 64 ? P / 64 : 1, nl = P > 64 ? 64 : P;
        const SlotMap smap = make_slot_map(S, P, 0, per, lane, nl);
        float tot = combine_slots(seg, smap, per, nl, S - 1);
        float r = sqrtf(tot);
        if (lane == 0) nrm2[blockIdx.x] = r * r;
    }
}

__global__ void gpfq_quantizer_kernel(int mode, float step, const float* __restrict__ x, int64_t n, float Kf,
                                      float lamb, const float* __restrict__ uniform, float* __restrict__ out,
                                      int32_t* __restrict__ idx)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int id = 0;
    float q;
    switch (mode) {
    case MODE_SOFT: q = quant_soft(step, x[i], Kf, lamb, id); break;
    case MODE_HARD: q = quant_hard(step, x[i], Kf, lamb, id); break;
    case MODE_STOCHASTIC: q = quant_stochastic(step, x[i], Kf, uniform ? uniform[i] : 0.5f, id); break;
    default: q = quant_msq(step, x[i], Kf, id); break;
    }
    out[i] = q;
    if (idx) idx[i] = id;
}

// rowmax[i] = max_j |W[i][j]|  (max is exact, any order)
__global__ void __launch_bounds__(256) gpfq_row_absmax_kernel(const float* __restrict__ W, int64_t ldw, int64_t d,
                                                              float* __restrict__ rowmax)
{
    __shared__ float part[4];
    const float* __restrict__ w = W + (int64_t)blockIdx.x * ldw;
    float v = 0.0f;
    for (int64_t j = threadIdx.x; j < d; j += blockDim.x) v = fmaxf(v, fabsf(w[j]));
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) rowmax[blockIdx.x] = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
}

}  // namespace gpfq

// ================================================================================================
// C ABI
// ================================================================================================
namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg)
{
    g_err = msg;
    return code;
}

int hip_fail(hipError_t e, const char* what)
{
    return fail(GPFQ_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

constexpr int kMaxResidentSegments = 16;

struct Plan {
    int kind;      // GPFQ_PLAN_STREAM / GPFQ_PLAN_RESIDENT
    int RT;        // rows per workgroup
    int waves;     // waves per workgroup
    int S;         // segments per row
};

int choose_plan(int64_t Ng, int64_t m_pad, int requested, Plan* out)
{
    Plan pl;
    if (m_pad / gpfq::kSeg > 1024) return fail(GPFQ_ERR_UNSUPPORTED, "m > 1048576 calibration rows is not supported");
    pl.S = (int)(m_pad / gpfq::kSeg);
    if (requested == GPFQ_PLAN_RESIDENT && pl.S > kMaxResidentSegments)
        return fail(GPFQ_ERR_UNSUPPORTED, "resident plan needs m_pad <= 16384");
    if (requested == GPFQ_PLAN_RESIDENT || (requested == GPFQ_PLAN_AUTO && pl.S <= kMaxResidentSegments)) {
        pl.kind = GPFQ_PLAN_RESIDENT;
        pl.RT = 1;
        pl.waves = pl.S;
    } else if (requested == GPFQ_PLAN_STREAM || requested == GPFQ_PLAN_AUTO) {
        pl.kind = GPFQ_PLAN_STREAM;
        pl.RT = Ng >= 1024 ? 4 : (Ng >= 512 ? 2 : 1);
        pl.waves = pl.S < 8 ? pl.S : 8;
    } else {
        return fail(GPFQ_ERR_ARG, "unknown plan id");
    }
    *out = pl;
    return GPFQ_OK;
}

template <int RT>
hipError_t launch_loop(const Plan& pl, const gpfq::LoopParams& p, int groups, bool vec, hipStream_t st)
{
    dim3 grid((unsigned)((p.Ng + RT - 1) / RT), (unsigned)groups, 1);
    dim3 block((unsigned)(64 * pl.waves), 1, 1);
    size_t shm = sizeof(float) * 2 * RT * (size_t)p.S;
    if (pl.kind == GPFQ_PLAN_RESIDENT) {
        if (vec) hipLaunchKernelGGL((gpfq::gpfq_resident_kernel<RT, true>), grid, block, shm, st, p);
        else hipLaunchKernelGGL((gpfq::gpfq_resident_kernel<RT, false>), grid, block, shm, st, p);
    } else {
        if (vec) hipLaunchKernelGGL((gpfq::gpfq_stream_kernel<RT, true>), grid, block, shm, st, p);
        else hipLaunchKernelGGL((gpfq::gpfq_stream_kernel<RT, false>), grid, block, shm, st, p);
    }
    return hipGetLastError();
}

int run_loop(gpfq::LoopParams p, int groups, int plan, hipStream_t st)
{
    if (p.Ng <= 0 || p.d <= 0 || groups <= 0) return GPFQ_OK;   // nothing to do
    Plan pl;
    int rc = choose_plan(p.Ng, p.m_pad, plan, &pl);
    if (rc) return rc;
    if (groups > 65535) return fail(GPFQ_ERR_UNSUPPORTED, "groups > 65535");
    p.S = pl.S;
    const bool vec = ((p.ldu & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.U) & 15) == 0);
    hipError_t e;
    switch (pl.RT) {
    case 4: e = launch_loop<4>(pl, p, groups, vec, st); break;
    case 2: e = launch_loop<2>(pl, p, groups, vec, st); break;
    default: e = launch_loop<1>(pl, p, groups, vec, st); break;
    }
    if (e != hipSuccess) return hip_fail(e, "GPFQ loop launch");
    return GPFQ_OK;
}

int check_mode(int mode, int K, int idx_bytes, const void* idx)
{
    if (mode < 0 || mode > 3) return fail(GPFQ_ERR_ARG, "mode must be 0..3");
    if (K < 1) return fail(GPFQ_ERR_ARG, "boundary index K must be >= 1");
    if (idx) {
        if (idx_bytes != 1 && idx_bytes != 2) return fail(GPFQ_ERR_ARG, "idx_bytes must be 1 or 2");
        if (idx_bytes == 1 && K > 126) return fail(GPFQ_ERR_ARG, "int8 indices need K <= 126; use idx_bytes = 2");
        if (K > 32766) return fail(GPFQ_ERR_ARG, "K too large for int16 indices");
    }
    return GPFQ_OK;
}

}  // namespace

extern "C" {

int gpfq_abi_version(void) { return GPFQ_ABI_VERSION; }

const char* gpfq_last_error(void) { return g_err.c_str(); }

int64_t gpfq_padded_m(int64_t m)
{
    if (m < 1) m = 1;
    return ((m + gpfq::kSeg - 1) / gpfq::kSeg) * gpfq::kSeg;
}

size_t gpfq_workspace_bytes(int64_t N, int64_t d_g, int64_t m, int groups)
{
    (void)N;
    if (d_g < 0 || m < 0 || groups < 1) return 0;
    const size_t D = (size_t)d_g * (size_t)groups;
    const size_t mp = (size_t)gpfq_padded_m(m);
    size_t cols = D * mp * sizeof(float);
    size_t nrm = ((D * sizeof(float) + 255) / 256) * 256;
    return 2 * cols + nrm + 256;
}

int gpfq_prepare_columns_f32(const float* A, int64_t lda, const float* X, int64_t ldx, int64_t m, int64_t D,
                             float* AT, float* XT, float* nrm2, int64_t m_pad, void* stream)
{
    if (!A || !X || !AT || !XT || !nrm2) return fail(GPFQ_ERR_ARG, "null pointer");
    if (m < 0 || D < 0 || lda < D || ldx < D) return fail(GPFQ_ERR_ARG, "bad shape (need lda, ldx >= D)");
    if (m_pad != gpfq_padded_m(m)) return fail(GPFQ_ERR_ARG, "m_pad must equal gpfq_padded_m(m)");
    if ((reinterpret_cast<uintptr_t>(AT) & 15) || (reinterpret_cast<uintptr_t>(XT) & 15))
        return fail(GPFQ_ERR_ARG, "AT / XT must be 16-byte aligned");
    if (D == 0) return GPFQ_OK;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)(m_pad / 64), (unsigned)((D + 63) / 64), 2);
    if (grid.y > 65535) return fail(GPFQ_ERR_UNSUPPORTED, "too many columns");
    hipLaunchKernelGGL(gpfq::gpfq_transpose_pad_kernel, grid, dim3(256), 0, st, A, lda, X, ldx, m, D, AT, XT, m_pad);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "transpose launch");
    const int S = (int)(m_pad / gpfq::kSeg);
    hipLaunchKernelGGL(gpfq::gpfq_colnorm_kernel, dim3((unsigned)D), dim3(256), sizeof(float) * (size_t)S, st, XT, m_pad,
                       S, nrm2);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "colnorm launch");
    return GPFQ_OK;
}

int gpfq_quantization_f32(const float* W, int64_t ldw, float* Q, int64_t ldq, float* U, int64_t ldu,
                          int u_has_init, const float* AT, const float* XT, const float* nrm2,
                          int64_t N, int64_t d, int64_t m, int64_t m_pad,
                          float step, int K, int mode, float lamb, uint64_t seed, uint64_t row_id0,
                          void* idx, int64_t ldi, int idx_bytes, int plan, void* stream)
{
    if (!W || !Q || !U || !AT || !XT || !nrm2) return fail(GPFQ_ERR_ARG, "null pointer");
    if (N < 0 || d < 0 || m < 0 || ldw < d || ldq < d || ldu < m || (idx && ldi < d))
        return fail(GPFQ_ERR_ARG, "bad shape / leading dimension");
    if (m_pad != gpfq_padded_m(m)) return fail(GPFQ_ERR_ARG, "m_pad must equal gpfq_padded_m(m)");
    int rc = check_mode(mode, K, idx_bytes, idx);
    if (rc) return rc;
    gpfq::LoopParams p;
    p.W = W; p.ldw = ldw; p.Q = Q; p.ldq = ldq; p.U = U; p.ldu = ldu; p.u_has_init = u_has_init;
    p.AT = AT; p.XT = XT; p.nrm2 = nrm2; p.Ng = N; p.d = d; p.m = m; p.m_pad = m_pad; p.S = 0;
    p.qc.step = step; p.qc.Kf = (float)K; p.qc.lamb = lamb; p.qc.mode = mode; p.qc.seed = seed;
    p.row_id0 = row_id0; p.idx = idx; p.ldi = ldi; p.idx_bytes = idx_bytes;
    return run_loop(p, 1, plan, (hipStream_t)stream);
}

int gpfq_quantize_groups_prepared_f32(const float* W, float* Q, float* U, const float* AT, const float* XT,
                                      const float* nrm2, int64_t N, int64_t d_g, int64_t m, int64_t m_pad,
                                      int groups, float step, int K, int mode, float lamb, uint64_t seed,
                                      uint64_t row_id0, void* idx, int idx_bytes, int plan, void* stream)
{
    if (!W || !Q || !U || !AT || !XT || !nrm2) return fail(GPFQ_ERR_ARG, "null pointer");
    if (groups < 1 || N < 0 || d_g < 0 || m < 0) return fail(GPFQ_ERR_ARG, "bad shape");
    if (N % groups != 0) return fail(GPFQ_ERR_ARG, "N must be divisible by groups");
    if (m_pad != gpfq_padded_m(m)) return fail(GPFQ_ERR_ARG, "m_pad must equal gpfq_padded_m(m)");
    int rc = check_mode(mode, K, idx_bytes, idx);
    if (rc) return rc;
    gpfq::LoopParams p;
    p.W = W; p.ldw = d_g; p.Q = Q; p.ldq = d_g; p.U = U; p.ldu = m; p.u_has_init = 0;
    p.AT = AT; p.XT = XT; p.nrm2 = nrm2; p.Ng = N / groups; p.d = d_g; p.m = m; p.m_pad = m_pad; p.S = 0;
    p.qc.step = step; p.qc.Kf = (float)K; p.qc.lamb = lamb; p.qc.mode = mode; p.qc.seed = seed;
    p.row_id0 = row_id0; p.idx = idx; p.ldi = d_g; p.idx_bytes = idx_bytes;
    return run_loop(p, groups, plan, (hipStream_t)stream);
}

int gpfq_quantize_layer_f32(const float* W, const float* A, int64_t lda, const float* X, int64_t ldx,
                            int64_t N, int64_t d_g, int64_t m, int groups,
                            float step, int K, int mode, float lamb, uint64_t seed, uint64_t row_id0,
                            float* Q, void* idx, int idx_bytes, float* U,
                            void* workspace, size_t workspace_bytes, int plan, void* stream)
{
    if (!W || !A || !X || !Q || !U || !workspace) return fail(GPFQ_ERR_ARG, "null pointer");
    if (groups < 1 || N < 0 || d_g < 0 || m < 0) return fail(GPFQ_ERR_ARG, "bad shape");
    if (N % groups != 0) return fail(GPFQ_ERR_ARG, "N must be divisible by groups");
    const int64_t D = d_g * (int64_t)groups;
    if (lda < D || ldx < D) return fail(GPFQ_ERR_ARG, "A / X need groups*d_g columns");
    int rcm = check_mode(mode, K, idx_bytes, idx);
    if (rcm) return rcm;
    if (workspace_bytes < gpfq_workspace_bytes(N, d_g, m, groups))
        return fail(GPFQ_ERR_WORKSPACE, "workspace smaller than gpfq_workspace_bytes()");
    if (reinterpret_cast<uintptr_t>(workspace) & 255) return fail(GPFQ_ERR_ARG, "workspace must be 256-byte aligned");
    const int64_t mp = gpfq_padded_m(m);
    char* ws = static_cast<char*>(workspace);
    float* AT = reinterpret_cast<float*>(ws);
    float* XT = reinterpret_cast<float*>(ws + (size_t)D * mp * sizeof(float));
    float* nrm2 = reinterpret_cast<float*>(ws + 2 * (size_t)D * mp * sizeof(float));
    int rc = gpfq_prepare_columns_f32(A, lda, X, ldx, m, D, AT, XT, nrm2, mp, stream);
    if (rc) return rc;
    return gpfq_quantize_groups_prepared_f32(W, Q, U, AT, XT, nrm2, N, d_g, m, mp, groups, step, K, mode, lamb, seed,
                                             row_id0, idx, idx_bytes, plan, stream);
}

int gpfq_quantizer_f32(int mode, float step, const float* x, int64_t n, int K, float lamb,
                       const float* uniform, float* out, int32_t* idx, void* stream)
{
    if (!x || !out) return fail(GPFQ_ERR_ARG, "null pointer");
    if (mode < 0 || mode > 3 || K < 1 || n < 0) return fail(GPFQ_ERR_ARG, "bad argument");
    if (n == 0) return GPFQ_OK;
    hipLaunchKernelGGL(gpfq::gpfq_quantizer_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, mode, step, x, n, (float)K, lamb, uniform, out, idx);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "quantizer launch");
    return GPFQ_OK;
}

int gpfq_row_absmax_f32(const float* W, int64_t ldw, int64_t N, int64_t d, float* rowmax, void* stream)
{
    if (!W || !rowmax) return fail(GPFQ_ERR_ARG, "null pointer");
    if (N < 0 || d < 0 || ldw < d) return fail(GPFQ_ERR_ARG, "bad shape");
    if (N == 0) return GPFQ_OK;
    hipLaunchKernelGGL(gpfq::gpfq_row_absmax_kernel, dim3((unsigned)N), dim3(256), 0, (hipStream_t)stream, W, ldw, d,
                       rowmax);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "row_absmax launch");
    return GPFQ_OK;
}

int gpfq_describe_plan(int64_t N, int64_t d_g, int64_t m, int groups, int plan, char* buf, size_t buf_bytes)
{
    if (groups < 1 || N % groups != 0) return fail(GPFQ_ERR_ARG, "bad groups");
    Plan pl;
    int rc = choose_plan(N / groups, gpfq_padded_m(m), plan, &pl);
    if (rc) return rc;
    if (buf && buf_bytes)
        snprintf(buf, buf_bytes, "%s RT=%d waves=%d S=%d grid=(%lld,%d) d=%lld",
                 pl.kind == GPFQ_PLAN_RESIDENT ? "resident" : "stream", pl.RT, pl.waves, pl.S,
                 (long long)((N / groups + pl.RT - 1) / pl.RT), groups, (long long)d_g);
    return pl.kind;
}

}  // extern "C"
