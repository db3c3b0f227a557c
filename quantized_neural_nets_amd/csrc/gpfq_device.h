// gpfq_device.h -- device-side building blocks shared by the GPFQ kernels (gfx950 / CDNA4, wave64).
//
// Arithmetic contract (DESIGN.md "Canonical arithmetic"): every fp32 operation of the reference's loop
// (step_algorithm.py:141-148 and the quantizers :7-104) is performed as an individually rounded fp32
// operation in the reference's order; the file is compiled with -ffp-contract=off so that a*b+c is never
// fused, and the only fused multiply-adds are the explicit __builtin_fmaf chains of the canonical dot product.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gpfq {

constexpr int kSeg = 1024;   // elements per canonical segment = 64 lanes x 16 elements
constexpr int kWave = 64;

enum { MODE_MSQ = 0, MODE_SOFT = 1, MODE_HARD = 2, MODE_STOCHASTIC = 3 };

// Balanced pairwise tree over the 64 lanes, result in every lane.  The xor butterfly evaluates, in every
// lane, exactly the tree ((v0+v1)+(v2+v3))+... of the oracle (fp add is commutative, so both operand
// orders of a level give the same bits).
__device__ __forceinline__ float wave_tree64(float v)
{
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) v = v + __shfl_xor(v, off, kWave);
    return v;
}

// torch.sign: (0 < x) - (x < 0)
__device__ __forceinline__ float sgnf(float x) { return (float)((0.0f < x) - (x < 0.0f)); }

// step_algorithm.py:56
__device__ __forceinline__ float quant_msq(float step, float x, float Kf, int& idx)
{
    float z = x / step;
    z = z + 0.5f;
    float r = fminf(fabsf(floorf(z)), Kf);
    float sg = sgnf(x);
    idx = (int)(sg * r);
    return (sg * step) * r;
}

// step_algorithm.py:103-104
__device__ __forceinline__ float quant_soft(float step, float x, float Kf, float lamb, int& idx)
{
    float y = sgnf(x) * fmaxf(fabsf(x) - lamb, 0.0f);
    return quant_msq(step, y, Kf, idx);
}

// step_algorithm.py:78-81
__device__ __forceinline__ float quant_hard(float step, float x, float Kf, float lamb, int& idx)
{
    float ax = fabsf(x);
    float x1 = (ax > lamb ? ax : 0.0f) * sgnf(x);
    float s1 = sgnf(x1);
    float y = s1 * fmaxf(fabsf(x1) - lamb, 0.0f);
    float z = y / step;
    z = z + 0.5f;
    float rv = fminf(fabsf(floorf(z)), Kf);
    float mask = (fabsf(x1) > lamb) ? 1.0f : 0.0f;
    float mag = lamb + step * rv;
    idx = (mask != 0.0f) ? (int)(s1 * (rv + 1.0f)) : 0;
    return (s1 * mag) * mask;
}

// Philox4x32-10 keyed by seed, one block per (row, column) -> uniform [0,1) with 24 bits
__device__ __forceinline__ float philox_uniform(uint64_t seed, uint64_t row, uint64_t col)
{
    uint32_t c0 = (uint32_t)col, c1 = (uint32_t)(col >> 32), c2 = (uint32_t)row, c3 = (uint32_t)(row >> 32);
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return (float)(c0 >> 8) * (1.0f / 16777216.0f);
}

// step_algorithm.py:27-35
__device__ __forceinline__ float quant_stochastic(float step, float x, float Kf, float uniform, int& idx)
{
    float z = x / step;
    float fl = floorf(z);
    float p = (1.0f - z) + fl;
    float lev = (uniform < p) ? fl : (fl + 1.0f);
    float q = step * lev;
    if (fabsf(q) > step * Kf) {
        float sg = sgnf(q);
        q = (sg * step) * Kf;
        lev = sg * Kf;
    }
    idx = (int)lev;
    return q;
}

struct QuantCfg {
    float step;
    float Kf;
    float lamb;
    int mode;
    uint64_t seed;
};

__device__ __forceinline__ float quantize(const QuantCfg& c, float s, uint64_t row_id, uint64_t col, int& idx)
{
    switch (c.mode) {
    case MODE_SOFT: return quant_soft(c.step, s, c.Kf, c.lamb, idx);
    case MODE_HARD: return quant_hard(c.step, s, c.Kf, c.lamb, idx);
    case MODE_STOCHASTIC: return quant_stochastic(c.step, s, c.Kf, philox_uniform(c.seed, row_id, col), idx);
    default: return quant_msq(c.step, s, c.Kf, idx);
    }
}

// One canonical segment of one row: the fused residual update of step t
//   u <- (u - q_{t-1} * x_{t-1}) + w_t * a_t        (step_algorithm.py:148 of step t-1, :141 of step t)
// and this lane's 16-element fma chain of <u, x_t> (step_algorithm.py:144).  Element order inside the
// lane: e = 4*c + j  <->  k = 1024*s + 256*c + 4*lane + j.
template <bool SUB>
__device__ __forceinline__ float sweep16(float (&u)[16], const float (&xp)[16], const float (&a)[16],
                                         const float (&x)[16], float qprev, float w)
{
    float acc = 0.0f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        float uu = u[e];
        if (SUB) {
            float p = qprev * xp[e];
            uu = uu - p;
        }
        float pa = w * a[e];
        uu = uu + pa;
        u[e] = uu;
        acc = __builtin_fmaf(uu, x[e], acc);
    }
    return acc;
}

__device__ __forceinline__ void load16(float (&dst)[16], const float* __restrict__ p /* + 4*lane applied */)
{
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float4 v = *reinterpret_cast<const float4*>(p + 256 * c);
        dst[4 * c + 0] = v.x; dst[4 * c + 1] = v.y; dst[4 * c + 2] = v.z; dst[4 * c + 3] = v.w;
    }
}

}  // namespace gpfq
