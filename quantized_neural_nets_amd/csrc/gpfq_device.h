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

// ---- lane reductions ---------------------------------------------------------------------------
// DPP controls (gfx9 encoding): quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E, row_half_mirror = 0x141,
// row_mirror = 0x140.  hipcc folds the move into v_add_f32_dpp.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// v[l] + v[l ^ 16] / v[l ^ 32] through gfx950's v_permlane16_swap / v_permlane32_swap.  The instruction
// swaps IN PLACE between its two registers: rows {1,3} (resp. lanes 32-63) of the first with rows {0,2}
// (resp. lanes 0-31) of the second.  Starting from two copies of v, the first ends up holding the even
// (low) part in every row and the second the odd (high) part, so their sum is the butterfly level.
// Written as inline asm because hipcc (ROCm 7.2) returns the first output for BOTH results of
// __builtin_amdgcn_permlane{16,32}_swap (measured on gfx950: tools/scratch/dpp_probe.hip).  The s_nop
// covers the VALU-write -> permlane-read hazard, which hipcc does not pad inside an asm statement.
__device__ __forceinline__ float xor16_add(float v)
{
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float xor32_add(float v)
{
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}

// Balanced pairwise tree over the 64 lanes, ((v0+v1)+(v2+v3))+... exactly the oracle's tree64 (every level adds the two
// halves of an aligned block; fp add is commutative, so the mirror forms give the same bits as an xor butterfly), total
// delivered in lane 63 only, all six levels as DPP adds (no permlane swap, no extra moves).  Needs EXEC = all lanes.
// after the four in-row levels every lane of a row holds its row sum; row_bcast15 adds lane 15 of the previous
// row into rows 1 and 3 (r1+r0, r3+r2), row_bcast31 adds lane 31 (= r1+r0) into rows 2 and 3, so row 3 ends with
// (r3+r2)+(r1+r0): the canonical tree up to operand order.  Disabled rows add the +0.0f of `old`.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov_rows(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_tree64_lane63(float v)
{
    v = v + dpp_mov<0xB1>(v);
    v = v + dpp_mov<0x4E>(v);
    v = v + dpp_mov<0x141>(v);
    v = v + dpp_mov<0x140>(v);
    v = v + dpp_mov_rows<0x142, 0xa>(v);   // row_bcast15 into rows 1, 3
    v = v + dpp_mov_rows<0x143, 0xc>(v);   // row_bcast31 into rows 2, 3
    return v;
}

// The lane trees of RT = 2 or 4 rows at once.  The four in-row levels run per row (afterwards every lane of a 16-lane
// row holds that row's sum: A0..A3 for the first residual row, B0..B3 for the second, ...); the two cross-row levels
// are SHARED: v_permlane16_swap(a, b) leaves a = [A0 B0 A2 B2], b = [A1 B1 A3 B3], so ONE add gives
// [A0+A1, B0+B1, A2+A3, B2+B3] -- level five of both trees -- and v_permlane32_swap against a copy (two rows) or against
// the other pair (four rows) sets up level six the same way.  Same additions, same tree, operands swapped at most
// (fp add is commutative): bit-identical to wave_tree64_lane63 per row, in 13 instead of 20 (22 instead of 40)
// instructions -- and a wave issues in order, so on the one-wave-per-SIMD kernels an instruction saved is ~5 cycles
// off the step.  Result: the total of residual row r in every lane of lane row r (RT = 2: also in lane row r + 2).
__device__ __forceinline__ float tree16_all(float v)
{
    v = v + dpp_mov<0xB1>(v);
    v = v + dpp_mov<0x4E>(v);
    v = v + dpp_mov<0x141>(v);
    v = v + dpp_mov<0x140>(v);
    return v;
}
__device__ __forceinline__ float swap16_add(float a, float b)
{
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float swap32_add(float a, float b)
{
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}
template <int RT>
__device__ __forceinline__ float wave_tree64_rows(const float (&acc)[RT])
{
    static_assert(RT == 2 || RT == 4, "one row: wave_tree64_lane63");
    const float s01 = swap16_add(tree16_all(acc[0]), tree16_all(acc[1]));
    if constexpr (RT == 2) {
        return swap32_add(s01, s01);
    } else {
        const float s23 = swap16_add(tree16_all(acc[2]), tree16_all(acc[3]));
        return swap32_add(s01, s23);
    }
}

// The same tree over the first nl lanes only (nl a power of two, wave-uniform): the upper levels, which
// would add the +0.0f of idle lanes, are skipped (adding +0.0f is exact).  Result valid in lanes < nl.
__device__ __forceinline__ float wave_tree_n(float v, int nl)
{
    if (nl > 1) v = v + dpp_mov<0xB1>(v);
    if (nl > 2) v = v + dpp_mov<0x4E>(v);
    if (nl > 4) v = v + dpp_mov<0x141>(v);
    if (nl > 8) v = v + dpp_mov<0x140>(v);
    if (nl > 16) v = xor16_add(v);
    if (nl > 32) v = xor32_add(v);
    return v;
}

// The same tree over one row of 16 lanes whose lanes >= nl hold +0.0f, without the level tests: the extra levels add
// +0.0f, which is exact for every value except -0.0f -- and a segment sum is never -0.0f (its fma chain starts from
// +0.0f, and +0.0f + (-0.0f) = +0.0f).  Result in lanes < nl (in fact in all 16).
__device__ __forceinline__ float wave_tree16_zero_padded(float v)
{
    v = v + dpp_mov<0xB1>(v);
    v = v + dpp_mov<0x4E>(v);
    v = v + dpp_mov<0x141>(v);
    v = v + dpp_mov<0x140>(v);
    return v;
}

// ---- canonical second level: segment sums -> slots -> pairwise tree -----------------------------
// A row of S segments uses P = 2^ceil(log2 S) slots; segment s sits in slot floor(s*P/S).  A lane owns
// `per` consecutive slots starting at slot_base + lane*per; SlotMap remembers which of them are occupied
// and the first segment among them (segments appear in slot order), computed once per kernel.
struct SlotMap {
    int s0;
    unsigned mask;
    unsigned mask_hi;      // slots 32..63 of a lane's block (only blocks of 64 slots per lane use it: rows of 2049..4096 segments)
};

__device__ __forceinline__ SlotMap make_slot_map(int S, int P, int slot_base, int per, int lane, int nl)
{
    SlotMap mp;
    mp.s0 = (slot_base * S + P - 1) / P;      // idle lanes point at the block's first segment (a valid index)
    mp.mask = 0u;
    mp.mask_hi = 0u;
    if (lane < nl) {
        const int first = slot_base + lane * per;
        int lo = (first * S + P - 1) / P;
        int hi = ((first + per) * S + P - 1) / P;
        if (hi > S) hi = S;
        mp.s0 = lo;
        for (int sg = lo; sg < hi; ++sg) {
            const int bit = (int)(((int64_t)sg * P) / S) - first;
            if (bit < 32) mp.mask |= 1u << bit;
            else mp.mask_hi |= 1u << (bit - 32);
        }
    }
    return mp;
}

template <int PER>
__device__ __forceinline__ float block_tree(const float* seg, const SlotMap mp, int s_last)
{
    float v[PER];
    int sidx = mp.s0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const bool occ = (mp.mask >> k) & 1u;
        const int si = sidx < s_last ? sidx : s_last;
        const float val = seg[si];
        v[k] = occ ? val : 0.0f;
        sidx += occ ? 1 : 0;
    }
#pragma unroll
    for (int w = 1; w < PER; w <<= 1)
#pragma unroll
        for (int i = 0; i < PER; i += 2 * w) v[i] = v[i] + v[i + w];
    return v[0];
}

// The same balanced tree over PER = 32 or 64 consecutive slots as a tree of 16-slot blocks, one block at a time (no more
// registers than block_tree<16>): a balanced pairwise tree over an aligned block IS the tree of its aligned halves.
// Rows of more than 1024 segments (m > 1 048 576) only.
template <int PER>
__device__ __forceinline__ float block_tree_seq(const float* seg, const SlotMap mp, int s_last)
{
    static_assert(PER == 32 || PER == 64, "two or four blocks of 16 slots");
    float part[PER / 16];
    int s0 = mp.s0;
#pragma unroll
    for (int b = 0; b < PER / 16; ++b) {
        SlotMap sub;
        sub.mask = ((b < 2 ? mp.mask : mp.mask_hi) >> (16 * (b & 1))) & 0xffffu;
        sub.mask_hi = 0u;
        sub.s0 = s0;
        part[b] = block_tree<16>(seg, sub, s_last);
        s0 += __builtin_popcount(sub.mask);           // segments sit in slot order
    }
    if (PER == 32) return part[0] + part[1];
    return (part[0] + part[1]) + (part[2] + part[3]);
}

// tree over the slot block [slot_base, slot_base + per*nl) of one row; seg = the row's segment sums
// (indexed by segment number relative to seg_base); result wave-uniform.
// BIG: also handle 32 / 64 slots per lane (rows of 1025..4096 segments); the four-row streaming kernel is compiled
// without (its register budget is full, and the host never gives it such rows).
template <bool BIG>
__device__ __forceinline__ float combine_slots(const float* seg, const SlotMap mp, int per, int nl, int s_last)
{
    float v;
    switch (per) {
    case 2: v = block_tree<2>(seg, mp, s_last); break;
    case 4: v = block_tree<4>(seg, mp, s_last); break;
    case 8: v = block_tree<8>(seg, mp, s_last); break;
    case 16: v = block_tree<16>(seg, mp, s_last); break;
    case 32: if constexpr (BIG) v = block_tree_seq<32>(seg, mp, s_last); else v = 0.0f; break;
    case 64: if constexpr (BIG) v = block_tree_seq<64>(seg, mp, s_last); else v = 0.0f; break;
    default: v = block_tree<1>(seg, mp, s_last); break;
    }
    v = wave_tree_n(v, nl);
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

__device__ __forceinline__ int pow2_ceil(int n)
{
    int p = 1;
    while (p < n) p <<= 1;
    return p;
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

// The same quantizer from the DOT PRODUCT v = <u, x_t>, without the two divisions whenever the answer cannot depend on
// them -- in TWELVE vector instructions instead of about fifty.  (A resident workgroup runs one wave per SIMD, and a wave
// issues in order: every instruction of the step is on its critical path, whatever the data dependences say.)
// step_algorithm.py:145-146 computes  z = fl(fl(v / n2) / step)  and rounds  floor(fl(z + 0.5));  with the per-column
// in2 = fl(1 / n2) (column preparation) and inv_step = fl(1 / step) (host),  r = fl(fl(v * in2) * inv_step)  is within 4
// roundings of the real quotient and z within 2:  |fl(z + 0.5) - fl(r + 0.5)| < 0.7 * 2^-20 * |r| + 2^-24.  With  f = y - floor(y)  (exact), y = fl(r + 0.5):
//   * |r| <= K + 4:  if f keeps more than (K + 4) * 2^-18 from 0 and from 1 -- |f - 0.5| < thr = 0.5 - (K + 4) * 2^-18,
//     one compare -- the reference's floor is floor(y);
//   * |r| >  K + 4:  both floors are beyond K and the clip min(|floor|, K) returns K for both, whatever f is.
// Everything behind the floor is the reference's own arithmetic: magnitude step * min(|floor|, K) ((+-1 * step) * rm has
// the same magnitude), sign of s = fl(v / n2) = sign of v as long as the quotient cannot underflow to zero (|r| >= 2^-60
// and step >= 2^-40, checked by the host), index (int)(+-rm).  v == 0, NaN, Inf, |r| >= 2^23 fail one of the two compares.
// Otherwise the caller runs the divisions: the function returns false and q / idx are not to be used (about one value
// in 10^4 at 4 bits).  The host switches the path off (inv_step = NaN) for K > 1024.
// n2 == 0: in2 == 0 and v == +0 (a zero column): the divisions run (and return 0, as the reference's guard does).
// Checked against the division form on 4.8 * 10^9 random and boundary-hugging arguments, 0 mismatches, and the check
// does find mismatches with a tolerance of 2^-23 (tests/csrc/msq_fast_check.c; tests/test_host_logic.py runs a short pass).
// Returns whether every lane whose answer is USED may keep it (wave-uniform; Kf, step, thr are uniform values; `unused`
// is the mask of the lanes nobody reads -- they hold zeros, and a zero dot product never passes).
__device__ __forceinline__ bool quant_msq_from_dot(float v, float in2, float inv_step, float step, float Kf, float thr,
                                                   unsigned long long unused, float& q, int& idx)
{
    const float r = (v * in2) * inv_step;
    const float y = r + 0.5f;
    const float fl = floorf(y);
    const float d = (y - fl) - 0.5f;
    float rm;
    asm("v_min_f32 %0, |%1|, %2" : "=v"(rm) : "v"(fl), "s"(Kf));       // (fminf would canonicalize both operands first)
    q = __builtin_copysignf(step * rm, v);
    idx = (int)__builtin_copysignf(rm, v);
    // the two compare masks straight into the scalar unit (a bool per lane would be materialised and compared again)
    const unsigned long long ok = __builtin_amdgcn_ballot_w64(__builtin_fabsf(d) < thr) &
                                  __builtin_amdgcn_ballot_w64(__builtin_fabsf(r) >= 0x1p-60f);
    return (ok | unused) == __builtin_amdgcn_read_exec();
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

// Read-only kernel inputs that every lane reads at the same address (a weight, a column norm) go through the
// scalar cache: a pointer into the constant address space makes LLVM select s_load_dword.  Left as a plain global
// pointer they become VECTOR loads as soon as the kernel also stores to global memory (no-clobber cannot be
// proven), and a dword load at the tail of the in-order vector-memory counter makes its consumer wait for every
// column load issued before it.
typedef const float __attribute__((address_space(4))) kfloat;
typedef const char __attribute__((address_space(4))) kchar;
__device__ __forceinline__ const kfloat* as_scalar(const float* p)
{
    return reinterpret_cast<const kfloat*>(reinterpret_cast<uintptr_t>(p));
}
// scalar load at a 32-bit BYTE offset from a uniform base: s_load_dword sdst, sbase, soffset -- no 64-bit address
// arithmetic (a[i] with a signed int costs six scalar instructions per address, and a wave issues them in order)
__device__ __forceinline__ float sload(const kfloat* base, unsigned byte_off)
{
    return *reinterpret_cast<const kfloat*>(reinterpret_cast<const kchar*>(base) + byte_off);
}

__device__ __forceinline__ void load16(float (&dst)[16], const float* __restrict__ p /* + 4*lane applied */)
{
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float4 v = *reinterpret_cast<const float4*>(p + 256 * c);
        dst[4 * c + 0] = v.x; dst[4 * c + 1] = v.y; dst[4 * c + 2] = v.z; dst[4 * c + 3] = v.w;
    }
}

// ---- the column window: physical registers the compiler never sees --------------------------------------------
// The register-resident kernels keep five column buffers (x_{t-1}, x_t, x_{t+1} rotating through three, a_t, a_{t+1}
// through two; 16 fp32 per lane each) with the loads of column t+2 in flight across a whole step.  A register whose
// load is in flight must not be read, copied, spilt or reused by anything -- which cannot be promised for a value the
// compiler allocates (it answers register pressure with live-range splits, i.e. copies).  So the five buffers are
// not C++ values at all: they are the TOP registers of the kernel's budget, v[WB ...], touched only by the asm
// statements below -- and so are the residual rows themselves (16 registers per row behind the 80 of the columns),
// which are updated in place for d steps: as C++ values every unrolled step would get a fresh set of them, and copies.  The kernel is declared with amdgpu_num_vgpr(WB), which makes every register from WB up RESERVED
// for the compiler (it allocates nothing there -- by construction, not by luck), and one empty asm with the last
// register of the budget on its clobber list makes the kernel descriptor allocate the whole budget.
// tools/check_async_loads.py verifies on the ISA of every build that no instruction outside these asm statements
// names a window register and that every sweep is preceded by a wait that covers its columns.
// All statements are `asm volatile`: they keep their program order among themselves, which is the only ordering the
// window needs (loads -> wait -> sweep); everything the compiler sees (residual rows, sums) is ordinary data flow.
//
// B = first register of a buffer, Q = quarter (4 of the 16 elements: one 16-byte load per lane).
// NQ = quarters of the 1024-element segment that hold data (one-segment rows of m <= 256 / 512 samples: 1 / 2): the
// others are zero padding -- never loaded, never swept (x = a = 0 leaves u = 0 and the dot product untouched, exactly)
template <int B, int Q, int NQ = 4>
__device__ __forceinline__ void win_load4(const float* base, unsigned lane_off)
{
    if constexpr (Q < NQ)
        asm volatile("global_load_dwordx4 v[%c2+%c3:%c2+%c3+3], %0, %1 offset:%c4"
                     :: "v"(lane_off), "s"(base), "n"(B), "n"(4 * Q), "n"(1024 * Q));
}
template <int B, int NQ = 4>
__device__ __forceinline__ void win_load16(const float* base, unsigned lane_off)
{
    win_load4<B, 0, NQ>(base, lane_off);
    win_load4<B, 1, NQ>(base, lane_off);
    win_load4<B, 2, NQ>(base, lane_off);
    win_load4<B, 3, NQ>(base, lane_off);
}
template <int B>
__device__ __forceinline__ void win_zero16()
{
#define GPFQ_Z(e) "v_mov_b32 v[%c0+" #e "], 0\n\t"
    asm volatile(GPFQ_Z(0) GPFQ_Z(1) GPFQ_Z(2) GPFQ_Z(3) GPFQ_Z(4) GPFQ_Z(5) GPFQ_Z(6) GPFQ_Z(7) GPFQ_Z(8) GPFQ_Z(9) GPFQ_Z(10)
                 GPFQ_Z(11) GPFQ_Z(12) GPFQ_Z(13) GPFQ_Z(14) GPFQ_Z(15) "s_nop 0" :: "n"(B));
#undef GPFQ_Z
}
// wait until at most N of this wave's vector-memory operations are outstanding (they retire in order)
template <int N>
__device__ __forceinline__ void win_wait()
{
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}
// sweep16 of one residual row against window buffers: U = the row's 16 residual registers (in the window too: the
// residual is updated in place for d steps, and a value the compiler allocates would be a fresh set of registers --
// and sixteen copies -- per unrolled step), XP = x_{t-1}, A = a_t, X = x_t.  The same five individually rounded
// fp32 operations per element, in the same order, as sweep16<true>() above:
//   p = q*xp; u = u - p; pa = w*a; u = u + pa; acc = fma(u, x, acc)           (step_algorithm.py:148, :141, :144)
// Ends with s_nop 1: the consumer of acc is a DPP add, and hipcc pads nothing for a producer inside an asm statement.
// Packed forms for the elementwise part (two elements per instruction; each half is an individually rounded fp32
// operation, and u + (-p) is u - p exactly): what hipcc itself emits for sweep16 -- 48 instructions per row instead
// of 80, which matters because a wave issues one vector instruction per 4 cycles however idle its SIMD is.
// {q, w} travel as ONE register pair; op_sel broadcasts its low half (q) or its high half (w) to both lanes of the pack.
typedef float v2f __attribute__((ext_vector_type(2)));
template <int U, int XP, int A, int X, int NQ = 4>
__device__ __forceinline__ float win_sweep16(float q, float w)
{
    float acc;
    v2f t0, t1;
    const v2f qw = {q, w};
#define GPFQ_S(e, e1)                                                                                  \
    "v_pk_mul_f32 %1, %3, v[%c5+" #e ":%c5+" #e1 "] op_sel_hi:[0,1]\n\t"                                \
    "v_pk_mul_f32 %2, %3, v[%c6+" #e ":%c6+" #e1 "] op_sel:[1,0]\n\t"                                   \
    "v_pk_add_f32 v[%c4+" #e ":%c4+" #e1 "], v[%c4+" #e ":%c4+" #e1 "], %1 neg_lo:[0,1] neg_hi:[0,1]\n\t" \
    "v_pk_add_f32 v[%c4+" #e ":%c4+" #e1 "], v[%c4+" #e ":%c4+" #e1 "], %2\n\t"                           \
    "v_fmac_f32 %0, v[%c4+" #e "], v[%c7+" #e "]\n\t"                                                   \
    "v_fmac_f32 %0, v[%c4+" #e1 "], v[%c7+" #e1 "]\n\t"
    if constexpr (NQ == 1)
        asm volatile("v_mov_b32 %0, 0\n\t" GPFQ_S(0, 1) GPFQ_S(2, 3) "s_nop 1"
                     : "=&v"(acc), "=&v"(t0), "=&v"(t1) : "v"(qw), "n"(U), "n"(XP), "n"(A), "n"(X));
    else if constexpr (NQ == 2)
        asm volatile("v_mov_b32 %0, 0\n\t" GPFQ_S(0, 1) GPFQ_S(2, 3) GPFQ_S(4, 5) GPFQ_S(6, 7) "s_nop 1"
                     : "=&v"(acc), "=&v"(t0), "=&v"(t1) : "v"(qw), "n"(U), "n"(XP), "n"(A), "n"(X));
    else
    asm volatile("v_mov_b32 %0, 0\n\t"
                 GPFQ_S(0, 1) GPFQ_S(2, 3) GPFQ_S(4, 5) GPFQ_S(6, 7) GPFQ_S(8, 9) GPFQ_S(10, 11) GPFQ_S(12, 13) GPFQ_S(14, 15)
                 "s_nop 1"
                 : "=&v"(acc), "=&v"(t0), "=&v"(t1)
                 : "v"(qw), "n"(U), "n"(XP), "n"(A), "n"(X));
#undef GPFQ_S
    return acc;
}
// The sweep of TWO residual rows at once, their registers interleaved (UP + 2k = element k of the first row, UP + 2k + 1
// of the second): every instruction serves one element of BOTH rows, so the dot products' fma chains pack as well --
// v_pk_fma_f32, each half a fused multiply-add of its own row -- and a pair of rows costs 80 instructions instead of 96.
// The same five individually rounded operations per element and row, in the same order; {q0, q1} and {w0, w1} travel as
// register pairs, op_sel broadcasts element k of the column pair register to both halves.
#define GPFQ_SP(u0, u1, kb, kb1, sel)                                                                                \
    "v_pk_mul_f32 %1, %3, v[%c6+" #kb ":%c6+" #kb1 "] op_sel:[0," #sel "] op_sel_hi:[1," #sel "]\n\t"                 \
    "v_pk_mul_f32 %2, %4, v[%c7+" #kb ":%c7+" #kb1 "] op_sel:[0," #sel "] op_sel_hi:[1," #sel "]\n\t"                 \
    "v_pk_add_f32 v[%c5+" #u0 ":%c5+" #u1 "], v[%c5+" #u0 ":%c5+" #u1 "], %1 neg_lo:[0,1] neg_hi:[0,1]\n\t"           \
    "v_pk_add_f32 v[%c5+" #u0 ":%c5+" #u1 "], v[%c5+" #u0 ":%c5+" #u1 "], %2\n\t"                                     \
    "v_pk_fma_f32 %0, v[%c5+" #u0 ":%c5+" #u1 "], v[%c8+" #kb ":%c8+" #kb1 "], %0 op_sel:[0," #sel ",0] op_sel_hi:[1," #sel ",1]\n\t"
template <int UP, int XP, int A, int X, int NQ = 4>
__device__ __forceinline__ v2f win_sweep16_pair(float q0, float q1, float w0, float w1)
{
    v2f acc = {0.0f, 0.0f};
    v2f t0, t1;
    const v2f qq = {q0, q1}, ww = {w0, w1};
    if constexpr (NQ == 1)
        asm volatile(GPFQ_SP(0, 1, 0, 1, 0) GPFQ_SP(2, 3, 0, 1, 1) GPFQ_SP(4, 5, 2, 3, 0) GPFQ_SP(6, 7, 2, 3, 1) "s_nop 1"
                     : "+v"(acc), "=&v"(t0), "=&v"(t1) : "v"(qq), "v"(ww), "n"(UP), "n"(XP), "n"(A), "n"(X));
    else if constexpr (NQ == 2)
        asm volatile(GPFQ_SP(0, 1, 0, 1, 0) GPFQ_SP(2, 3, 0, 1, 1) GPFQ_SP(4, 5, 2, 3, 0) GPFQ_SP(6, 7, 2, 3, 1) GPFQ_SP(8, 9, 4, 5, 0) GPFQ_SP(10, 11, 4, 5, 1) GPFQ_SP(12, 13, 6, 7, 0) GPFQ_SP(14, 15, 6, 7, 1) "s_nop 1"
                     : "+v"(acc), "=&v"(t0), "=&v"(t1) : "v"(qq), "v"(ww), "n"(UP), "n"(XP), "n"(A), "n"(X));
    else
        asm volatile(GPFQ_SP(0, 1, 0, 1, 0) GPFQ_SP(2, 3, 0, 1, 1) GPFQ_SP(4, 5, 2, 3, 0) GPFQ_SP(6, 7, 2, 3, 1) GPFQ_SP(8, 9, 4, 5, 0) GPFQ_SP(10, 11, 4, 5, 1) GPFQ_SP(12, 13, 6, 7, 0) GPFQ_SP(14, 15, 6, 7, 1) GPFQ_SP(16, 17, 8, 9, 0) GPFQ_SP(18, 19, 8, 9, 1) GPFQ_SP(20, 21, 10, 11, 0) GPFQ_SP(22, 23, 10, 11, 1) GPFQ_SP(24, 25, 12, 13, 0) GPFQ_SP(26, 27, 12, 13, 1) GPFQ_SP(28, 29, 14, 15, 0) GPFQ_SP(30, 31, 14, 15, 1) "s_nop 1"
                     : "+v"(acc), "=&v"(t0), "=&v"(t1) : "v"(qq), "v"(ww), "n"(UP), "n"(XP), "n"(A), "n"(X));
    return acc;
}
#undef GPFQ_SP
// One quarter (four elements per lane) of the pair sweep with the COLUMNS IN ORDINARY REGISTERS -- read back from LDS by
// the variant that stages its columns there (coop_lds_body: four rows at 13 waves leave no room for a register window).
// UPC = first window register of the quarter of the pair: UP + 8 c.  xp / a / x: elements (0, 1) and (2, 3) as pairs.
#define GPFQ_SL(u0, u1, xp, a, x, sel)                                                                              \
    "v_pk_mul_f32 %1, %3, " xp " op_sel:[0," #sel "] op_sel_hi:[1," #sel "]\n\t"                                     \
    "v_pk_mul_f32 %2, %4, " a " op_sel:[0," #sel "] op_sel_hi:[1," #sel "]\n\t"                                      \
    "v_pk_add_f32 v[%c11+" #u0 ":%c11+" #u1 "], v[%c11+" #u0 ":%c11+" #u1 "], %1 neg_lo:[0,1] neg_hi:[0,1]\n\t"       \
    "v_pk_add_f32 v[%c11+" #u0 ":%c11+" #u1 "], v[%c11+" #u0 ":%c11+" #u1 "], %2\n\t"                                 \
    "v_pk_fma_f32 %0, v[%c11+" #u0 ":%c11+" #u1 "], " x ", %0 op_sel:[0," #sel ",0] op_sel_hi:[1," #sel ",1]\n\t"
template <int UPC>
__device__ __forceinline__ void win_sweep4_pair_lds(v2f& acc, v2f qq, v2f ww, v2f xp01, v2f xp23, v2f a01, v2f a23, v2f x01, v2f x23)
{
    v2f t0, t1;
    asm volatile(GPFQ_SL(0, 1, "%5", "%7", "%9", 0) GPFQ_SL(2, 3, "%5", "%7", "%9", 1)
                 GPFQ_SL(4, 5, "%6", "%8", "%10", 0) GPFQ_SL(6, 7, "%6", "%8", "%10", 1) "s_nop 0"
                 : "+v"(acc), "=&v"(t0), "=&v"(t1)
                 : "v"(qq), "v"(ww), "v"(xp01), "v"(xp23), "v"(a01), "v"(a23), "v"(x01), "v"(x23), "n"(UPC));
}
#undef GPFQ_SL
// ... and the pending subtraction of the last step for such a quarter: u = u - q * x_{d-1}, both rows of the pair
#define GPFQ_FL(u0, u1, x, sel)                                                                                     \
    "v_pk_mul_f32 %0, %1, " x " op_sel:[0," #sel "] op_sel_hi:[1," #sel "]\n\t"                                      \
    "v_pk_add_f32 v[%c4+" #u0 ":%c4+" #u1 "], v[%c4+" #u0 ":%c4+" #u1 "], %0 neg_lo:[0,1] neg_hi:[0,1]\n\t"
template <int UPC>
__device__ __forceinline__ void win_final_sub4_pair_lds(v2f qq, v2f x01, v2f x23)
{
    v2f t0;
    asm volatile(GPFQ_FL(0, 1, "%2", 0) GPFQ_FL(2, 3, "%2", 1) GPFQ_FL(4, 5, "%3", 0) GPFQ_FL(6, 7, "%3", 1) "s_nop 0"
                 : "=&v"(t0) : "v"(qq), "v"(x01), "v"(x23), "n"(UPC));
}
#undef GPFQ_FL
// the pending subtraction of the last step, in place: u = u - q * x_{d-1}                          (step_algorithm.py:148)
template <int U, int XL>
__device__ __forceinline__ void win_final_sub16(float q)
{
    float t;
#define GPFQ_F(e) "v_mul_f32 %0, %1, v[%c3+" #e "]\n\tv_sub_f32 v[%c2+" #e "], v[%c2+" #e "], %0\n\t"
    asm volatile(GPFQ_F(0) GPFQ_F(1) GPFQ_F(2) GPFQ_F(3) GPFQ_F(4) GPFQ_F(5) GPFQ_F(6) GPFQ_F(7) GPFQ_F(8) GPFQ_F(9) GPFQ_F(10)
                 GPFQ_F(11) GPFQ_F(12) GPFQ_F(13) GPFQ_F(14) GPFQ_F(15) "s_nop 0"
                 : "=&v"(t) : "v"(q), "n"(U), "n"(XL));
#undef GPFQ_F
}
// four consecutive window registers into ordinary ones
template <int B>
__device__ __forceinline__ void win_read4(float (&d)[4])
{
    asm volatile("v_mov_b32 %0, v[%c4+0]\n\tv_mov_b32 %1, v[%c4+1]\n\tv_mov_b32 %2, v[%c4+2]\n\tv_mov_b32 %3, v[%c4+3]\n\ts_nop 0"
                 : "=v"(d[0]), "=v"(d[1]), "=v"(d[2]), "=v"(d[3]) : "n"(B));
}
// the same two helpers for ONE row of an interleaved pair (its registers are B, B + 2, B + 4, ...)
template <int U, int XL>
__device__ __forceinline__ void win_final_sub16_s2(float q)
{
    float t;
#define GPFQ_F(e, u) "v_mul_f32 %0, %1, v[%c3+" #e "]\n\tv_sub_f32 v[%c2+" #u "], v[%c2+" #u "], %0\n\t"
    asm volatile(GPFQ_F(0, 0) GPFQ_F(1, 2) GPFQ_F(2, 4) GPFQ_F(3, 6) GPFQ_F(4, 8) GPFQ_F(5, 10) GPFQ_F(6, 12) GPFQ_F(7, 14)
                 GPFQ_F(8, 16) GPFQ_F(9, 18) GPFQ_F(10, 20) GPFQ_F(11, 22) GPFQ_F(12, 24) GPFQ_F(13, 26) GPFQ_F(14, 28)
                 GPFQ_F(15, 30) "s_nop 0"
                 : "=&v"(t) : "v"(q), "n"(U), "n"(XL));
#undef GPFQ_F
}
template <int B>
__device__ __forceinline__ void win_read4_s2(float (&d)[4])
{
    asm volatile("v_mov_b32 %0, v[%c4+0]\n\tv_mov_b32 %1, v[%c4+2]\n\tv_mov_b32 %2, v[%c4+4]\n\tv_mov_b32 %3, v[%c4+6]\n\ts_nop 0"
                 : "=v"(d[0]), "=v"(d[1]), "=v"(d[2]), "=v"(d[3]) : "n"(B));
}

// The lane number, recomputed where it is used and opaque to the compiler.  For the rare blocks of a kernel (the Q / idx
// flush every 64 steps, the epilogue): everything they derive from the lane number -- per-lane 64-bit store addresses for
// every row -- is then defined inside the block instead of being hoisted above the column loop and kept live across it
// (24 VGPRs in the 64-register variants, which went to scratch), and the loop's own copy of the lane number need not
// stay live for them either.  (v_mbcnt counts the bits of its MASK operand, here all ones, below the lane: the lane's
// position, whatever EXEC is.)
__device__ __forceinline__ int fresh_lane_id()
{
    int ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    return ln;
}

// A wave-uniform pointer as an SGPR pair, for the "s" base operand of the window loads.  Two HAZARDS of the gfx940 family
// that the compiler pads for its own instructions but not across the boundary of an asm statement, which it does not
// look into -- both met in round 3, when a change elsewhere in the kernels moved the scheduling:
//   * an SGPR written by a VALU instruction (v_readfirstlane) may not be read by a VMEM instruction within the next 5
//     wait states: with the readfirstlane a builtin, one build scheduled `v_readfirstlane_b32 s49, v5` directly in front
//     of the asm `global_load_dwordx4 ..., s[48:49]`; the load went out with a stale s49 and the kernel faulted;
//   * a VGPR written by a VALU instruction may not be read by v_readfirstlane / v_readlane in the very next wait state:
//     with the readfirstlanes inside an asm statement and the 64-bit add that forms the pointer right in front of it,
//     the low half came back stale (fault addresses of the form 0x7e1b00000000).
// So the readfirstlanes, one wait state in front of them and five behind them are ONE asm statement (prologue only:
// inside the loop the pointers advance by scalar adds, which have neither hazard), and tools/check_async_loads.py
// rejects any build whose ISA shows either pattern around an asm statement.
__device__ __forceinline__ const float* uniform_ptr(const float* p)
{
    const uintptr_t v = reinterpret_cast<uintptr_t>(p);
    const unsigned vlo = (unsigned)v, vhi = (unsigned)(v >> 32);
    unsigned lo, hi;
    asm volatile("s_nop 1\n\tv_readfirstlane_b32 %0, %2\n\tv_readfirstlane_b32 %1, %3\n\ts_nop 4" : "=s"(lo), "=s"(hi) : "v"(vlo), "v"(vhi));
    return reinterpret_cast<const float*>(((uintptr_t)hi << 32) | lo);
}

}  // namespace gpfq
