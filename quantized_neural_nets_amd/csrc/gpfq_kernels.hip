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
#include <stdlib.h>
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
// Slab plans (resident / cooperative): the residual lives in registers for the whole column loop.
//
// A workgroup owns an RT x (n_own segments) slab of U: RT rows sharing the activation registers, one wave
// per canonical segment.  Per step: every wave sweeps its segment (fused update + fma chain, sweep16) and
// reduces the 64 lane chains (wave_tree64); wave 0 then finishes the canonical slot tree, divides by the
// column norm, quantizes (the RT rows in RT different lanes) and hands q back through LDS.
//
//   resident (COOP = false): one workgroup holds whole rows (S <= 16 segments).  HBM/L2 traffic per step is
//     only x_{t+1}, a_{t+1}, prefetched behind the reduction of step t.
//   cooperative (COOP = true): a row is split by columns over C workgroups ("members"), needed when rows are
//     too long for one workgroup's registers or too few to fill the chip.  Each member reduces its own
//     aligned block of the slot tree, publishes RT partial sums as 8-byte {value, epoch} granules (one
//     write-through store each: the data is the flag), gathers the C*RT <= 64 granules of its row tile with
//     one load per lane per poll, finishes the tree over the C members and quantizes.  Every member computes
//     the same bits, so nothing else is exchanged.  Placement-independent: correctness needs only that all
//     workgroups are resident (the host sizes the grid from the occupancy query); spins are bounded and a
//     timeout raises the status word instead of hanging.
// ------------------------------------------------------------------------------------------------
struct SlabParams {
    const float* W; float* Q; float* U; void* idx;
    const float* AT; const float* XT; const float* nrm2;
    unsigned long long* xbuf; int* status;
    int64_t ldw, ldq, ldu, ldi, m, m_pad;
    int Ng, d, S, C, tiles, idx_bytes, vec;
    float step, Kf, lamb;
    unsigned spin_limit;
    uint64_t seed, row_id0;
};

template <int MODE>
__device__ __forceinline__ float quantize_mode(const SlabParams& p, float s, uint64_t row, uint64_t col, int& id)
{
    if (MODE == MODE_SOFT) return quant_soft(p.step, s, p.Kf, p.lamb, id);
    if (MODE == MODE_HARD) return quant_hard(p.step, s, p.Kf, p.lamb, id);
    if (MODE == MODE_STOCHASTIC) return quant_stochastic(p.step, s, p.Kf, philox_uniform(p.seed, row, col), id);
    return quant_msq(p.step, s, p.Kf, id);
}

// The reducer's serial section of one step (one wave, EXEC full): finish this workgroup's block of the slot tree
// for all RT rows at once (lane = row*nl + slot), exchange with the other members when cooperative, divide by the
// column norm, quantize the RT rows in RT lanes, hand q back through LDS and write Q / idx.
template <int RT, int MODE, bool COOP>
__device__ __forceinline__ void reducer_section(const SlabParams& p, const float* seg, float* qs, const SlotMap smap,
                                        int NW, int nl, int rlane, int lane, int tile, int c, int C, int par,
                                        int t, float n2cur, int row0, int64_t grow0, int seg_lo)
{
    // this workgroup's block of the slot tree for all RT rows at once: blocks of nl lanes
    float v;
    {
        const int rr = rlane < RT ? rlane : 0;
        const float val = seg[rr * NW + (smap.s0 - seg_lo)];
        v = ((smap.mask & 1u) && rlane < RT) ? val : 0.0f;
        v = wave_tree_n(v, nl);
    }
    bool timed_out = false;
    int blk = nl;                            // lanes r*blk .. r*blk+blk-1 hold row r's value
    if (COOP) {
        const unsigned epoch = (unsigned)t + 1u;
        unsigned long long* xb_ = p.xbuf + ((size_t)(tile * 2 + par) * C) * RT;
        if ((lane % nl) == 0 && rlane < RT)
            __hip_atomic_store(xb_ + (size_t)c * RT + rlane,
                               ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(v),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // gather: lane = r*C + member
        const bool want = lane < RT * C;
        const unsigned long long* src = xb_ + (want ? (size_t)(lane % C) * RT + (lane / C) : 0);
        unsigned long long gv = 0;
        unsigned spins = 0;
        for (;;) {
            gv = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool ok = !want || ((unsigned)(gv >> 32) == epoch);
            if (__all(ok)) break;
            if (++spins > p.spin_limit) { timed_out = true; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        v = want ? __uint_as_float((unsigned)gv) : 0.0f;
        v = wave_tree_n(v, C);               // upper levels of the slot tree, per aligned block of C lanes
        blk = C;
    }
    const int gr_ = lane / blk;              // row of this lane
    const bool lead = (lane % blk == 0) && gr_ < RT;
    const float sarg = (n2cur > 0.0f) ? v / n2cur : 0.0f;
    const bool rvalid = lead && (row0 + gr_ < p.Ng);
    const int64_t growl = grow0 + (rvalid ? gr_ : 0);
    int id;
    const float q = quantize_mode<MODE>(p, sarg, p.row_id0 + (uint64_t)growl, (uint64_t)t, id);
    if (lead) qs[par * (RT + 1) + gr_] = q;
    // Q / idx leave through a 64-step history in LDS and one coalesced store per row every 64 steps: a
    // store per step would queue behind the sweep waves' column loads in the vector-memory pipe and hold up
    // the reducer's arrival at the second barrier.
    float* hist = qs + 2 * (RT + 1);                 // [RT][64] values, then [RT][64] indices (as int bits)
    if (lead) {
        hist[gr_ * 64 + (t & 63)] = q;
        hist[(RT + gr_) * 64 + (t & 63)] = __int_as_float(id);
    }
    if ((t & 63) == 63 || t + 1 == p.d) {
        const int t0 = t & ~63;
        const int n = t - t0 + 1;                    // steps in this history block
        if (c == 0 && lane < n) {
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                if (row0 + r < p.Ng) {
                    const int64_t gw = grow0 + r;
                    p.Q[gw * p.ldq + t0 + lane] = hist[r * 64 + lane];
                    if (p.idx) {
                        const int iv = __float_as_int(hist[(RT + r) * 64 + lane]);
                        if (p.idx_bytes == 1) reinterpret_cast<int8_t*>(p.idx)[gw * p.ldi + t0 + lane] = (int8_t)iv;
                        else reinterpret_cast<int16_t*>(p.idx)[gw * p.ldi + t0 + lane] = (int16_t)iv;
                    }
                }
            }
        }
    }
    if (COOP && lane == 0) {
        qs[par * (RT + 1) + RT] = timed_out ? 1.0f : 0.0f;
        if (timed_out) {
            atomicExch(p.status, 1);
            p.status[1] = t; p.status[2] = tile; p.status[3] = c;
        }
    }
}

// MAXW = most waves per workgroup the instantiation may be launched with; it sets the register budget
// (16 waves -> 128 VGPRs, 12 -> 168, 8 -> 256): more rows per workgroup need the roomier variants.
template <int RT, int MODE, bool COOP, int MAXW>
__global__ void __launch_bounds__(64 * MAXW) gpfq_slab_kernel(const SlabParams p)
{
    extern __shared__ float smem[];                 // seg[2][RT][NW], then qs[2][RT + 1]
    const int NW = blockDim.x >> 6;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int S = p.S, C = COOP ? p.C : 1;
    const int P = pow2_ceil(S);
    int tile, c, g;
    if (COOP) {
        g = 0;
        // keep the members of one row tile on one XCD when the tile count allows it (blocks b and b+8 share an
        // XCD under round-robin dispatch; speed only, never correctness)
        if ((p.tiles & 7) == 0) {
            const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
            tile = (j / C) * 8 + xcd;
            c = j % C;
        } else {
            tile = blockIdx.x / C;
            c = blockIdx.x % C;
        }
    } else {
        tile = blockIdx.x; c = 0; g = blockIdx.y;
    }
    const int seg_lo = (c * S + C - 1) / C, seg_hi = ((c + 1) * S + C - 1) / C;
    const int n_own = seg_hi - seg_lo;              // <= max_own
    const int max_own = (S + C - 1) / C;            // sweep waves of the fullest member
    // The reducer (slot tree, exchange, quantizer) is a wave of its own when the launch has one to spare
    // (NW == max_own + 1): its serial section is then not delayed by its own column loads; otherwise wave 0
    // doubles as the reducer.
    const int rwave = NW > max_own ? max_own : 0;
    const bool active = wave < n_own;
    const int myseg = seg_lo + (active ? wave : 0);
    const int nl = P / C;                           // slots of this workgroup's block; RT*nl <= 64 (host guarantees it)
    // wave 0 reduces all RT rows at once: lane = r*nl + slot
    const SlotMap smap = make_slot_map(S, P, c * nl, 1, lane % nl, nl);
    const int rlane = lane / nl;                    // row whose slot this lane holds (>= RT: idle)

    float* segs = smem;                             // [2][RT][NW]
    float* qs = smem + 2 * RT * NW;                 // [2][RT + 1] (last = abort flag), then the Q / idx history [2*RT][64]

    const int row0 = tile * RT;                     // row inside the group
    const int64_t grow0 = (int64_t)g * p.Ng + row0; // global row of this tile's first row
    const int64_t kbase = (int64_t)myseg * kSeg + 4 * lane;
    const float* __restrict__ acol = p.AT + (int64_t)g * p.d * p.m_pad + kbase;
    const float* __restrict__ xcol = p.XT + (int64_t)g * p.d * p.m_pad + kbase;
    const float* __restrict__ nrm = p.nrm2 + (int64_t)g * p.d;

    float u[RT][16];
    const float* __restrict__ wrow[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const int64_t gr = grow0 + ((row0 + r < p.Ng) ? r : (p.Ng - 1 - row0));
        wrow[r] = p.W + gr * p.ldw;
#pragma unroll
        for (int e = 0; e < 16; ++e) u[r][e] = 0.0f;   // a non-zero initial residual is the streaming plan's job
    }
    float qprev[RT], wcur[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) { qprev[r] = 0.0f; wcur[r] = wrow[r][0]; }
    float n2cur = nrm[0];

    // xc = x_t, xo = x_{t-1} (all zero at t = 0, where q_{-1} = 0), aa = a_t
    float xc[16], xo[16], aa[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) { xc[e] = 0.0f; xo[e] = 0.0f; aa[e] = 0.0f; }
    if (active) { load16(xc, xcol); load16(aa, acol); }

#ifdef GPFQ_STAMPS
    // diagnostic build only: cycles per phase, summed by wave 0 (and the last wave) of block 0 into status[16..]
    unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stamp_prev = 0;
#define GPFQ_STAMP(i)                                                                                   \
    {                                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        unsigned long long now_;                                                                        \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");                      \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        stamp_sum[i] += now_ - stamp_prev;                                                              \
        stamp_prev = now_;                                                                              \
    }
#else
#define GPFQ_STAMP(i)
#endif
    bool dead = false;
    int t = 0;
#ifdef GPFQ_STAMPS
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
#endif
    while (t < p.d && !dead) {
        GPFQ_STAMP(0)
        const int par = t & 1;
        const bool more = t + 1 < p.d;
        float* seg = segs + par * RT * NW;
        float wn[RT], n2n = 0.0f;
#pragma unroll
        for (int r = 0; r < RT; ++r) wn[r] = 0.0f;
        if (active) {
            float acc[RT];
#pragma unroll
            for (int r = 0; r < RT; ++r) acc[r] = sweep16<true>(u[r], xo, aa, xc, qprev[r], wcur[r]);
            GPFQ_STAMP(1)
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const float sg = wave_tree64_lane63(acc[r]);
                if (lane == 63) seg[r * NW + wave] = sg;
            }
            GPFQ_STAMP(2)
        }
        if (more) {
#pragma unroll
            for (int r = 0; r < RT; ++r) wn[r] = wrow[r][t + 1];
            n2n = nrm[t + 1];
        }
        __syncthreads();
        GPFQ_STAMP(3)
        // Next column's loads, issued OFF the critical path (the sweep waves idle until the reducer is done) and
        // landing in the registers the sweep just finished with.  The opaque asm keeps LLVM from hoisting them
        // (they depend on nothing here): hoisted above the sweep they need 32 more VGPRs and a copy that waits
        // for them before the barrier, and their issue stalls (~60-180 cycles each) sit on the critical path.
        auto issue_loads = [&]() {
#pragma unroll
            for (int e = 0; e < 16; ++e) xo[e] = xc[e];
            int64_t adv = more ? p.m_pad : 0;        // the last step re-reads its own column: no join copy
            asm volatile("" : "+s"(adv)::"memory");
            xcol += adv;
            acol += adv;
            load16(xc, xcol);
            load16(aa, acol);
        };
        if (active && wave != rwave) issue_loads();
        if (wave == rwave) {
            GPFQ_STAMP(4)
            reducer_section<RT, MODE, COOP>(p, seg, qs, smap, NW, nl, rlane, lane, tile, c, C, par, t, n2cur, row0, grow0,
                                            seg_lo);
            GPFQ_STAMP(5)
        }
        if (active && wave == rwave) issue_loads();
        GPFQ_STAMP(6)
        __syncthreads();
        GPFQ_STAMP(7)
#pragma unroll
        for (int r = 0; r < RT; ++r) qprev[r] = qs[par * (RT + 1) + r];
        if (COOP) dead = qs[par * (RT + 1) + RT] != 0.0f;
#pragma unroll
        for (int r = 0; r < RT; ++r) wcur[r] = wn[r];
        n2cur = n2n;
        ++t;
    }
#ifdef GPFQ_STAMPS
    if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && (wave == rwave || wave == (rwave == 0 ? NW - 1 : 0)) && p.status) {
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(p.status + 16) + (wave == rwave ? 0 : 8);
        for (int i = 0; i < 8; ++i) dbg[i] = stamp_sum[i];
    }
#endif
    if (dead || !active) return;

    // pending subtraction of the last step, then write the residual (step_algorithm.py:148)
#pragma unroll
    for (int r = 0; r < RT; ++r) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float pq = qprev[r] * xo[e];       // xo = x_{d-1} after the last rotation
            u[r][e] = u[r][e] - pq;
        }
        if (row0 + r < p.Ng) {
            float* Urow = p.U + (grow0 + r) * p.ldu;
            if (p.vec) store_u16<true>(u[r], Urow, kbase, p.m);
            else store_u16<false>(u[r], Urow, kbase, p.m);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Streaming plan: any (N, m).  The residual rows stay in the caller's U (HBM / L2 / Infinity Cache) and
// are read and written once per step; wave w owns segments w, w+NW, ... of the workgroup's RT rows.
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// One-segment rows (m <= 1024: every fully connected layer, 1x1 convs on 1x1 maps, small depthwise maps):
// the whole row lives in ONE wave, so the step needs no LDS and no barrier at all -- sweep, lane tree,
// v_readlane, quantize, next step.  A workgroup is just four independent waves; Q / idx are kept 64 steps in
// registers (lane l holds step t0 + l) and leave as one coalesced store per row.
// ------------------------------------------------------------------------------------------------
template <int RT, int MODE>
__global__ void __launch_bounds__(256) gpfq_wave_kernel(const SlabParams p)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = blockIdx.y;
    const int row0 = (blockIdx.x * 4 + wave) * RT;  // first row of this wave's tile, inside the group
    if (row0 >= p.Ng) return;                       // whole wave leaves: nothing is shared between waves
    const int64_t grow0 = (int64_t)g * p.Ng + row0;
    const int64_t kbase = 4 * lane;
    const float* __restrict__ acol = p.AT + (int64_t)g * p.d * p.m_pad + kbase;
    const float* __restrict__ xcol = p.XT + (int64_t)g * p.d * p.m_pad + kbase;
    const float* __restrict__ nrm = p.nrm2 + (int64_t)g * p.d;

    float u[RT][16], xc[16], xo[16], aa[16];
    const float* __restrict__ wrow[RT];
    float qprev[RT], wcur[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        wrow[r] = p.W + (grow0 + ((row0 + r < p.Ng) ? r : 0)) * p.ldw;   // rows past the end duplicate the first
#pragma unroll
        for (int e = 0; e < 16; ++e) u[r][e] = 0.0f;
        qprev[r] = 0.0f;
        wcur[r] = wrow[r][0];
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) xo[e] = 0.0f;
    load16(xc, xcol);
    load16(aa, acol);
    float n2cur = nrm[0];
    float qh = 0.0f;                                // Q / idx history: lane = (step % 64) * ... one register per row
    float qhist[RT];
    int ihist[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) { qhist[r] = 0.0f; ihist[r] = 0; }
    (void)qh;
    for (int t = 0; t < p.d; ++t) {
        const bool more = t + 1 < p.d;
        float acc[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) acc[r] = sweep16<true>(u[r], xo, aa, xc, qprev[r], wcur[r]);
        // next column: issued right behind the sweep (its latency hides under the reduction and the quantizer);
        // the opaque asm keeps it from being hoisted above the sweep, unconditional so that no join copy is needed
#pragma unroll
        for (int e = 0; e < 16; ++e) xo[e] = xc[e];
        int64_t adv = more ? p.m_pad : 0;
        asm volatile("" : "+s"(adv) : "v"(acc[0]));
        xcol += adv;
        acol += adv;
        load16(xc, xcol);
        load16(aa, acol);
        float wn[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) wn[r] = more ? wrow[r][t + 1] : 0.0f;
        const float n2n = more ? nrm[t + 1] : 0.0f;
        // the RT row totals, row r parked in lane r, then ONE quantizer evaluation for all rows
        float v = 0.0f;
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const float sg = wave_tree64_lane63(acc[r]);
            const float tot = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sg), 63));
            if (lane == r) v = tot;
        }
        const float sarg = (n2cur > 0.0f) ? v / n2cur : 0.0f;
        int id;
        const float q = quantize_mode<MODE>(p, sarg, p.row_id0 + (uint64_t)(grow0 + (lane < RT ? lane : 0)), (uint64_t)t, id);
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            qprev[r] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, q), r));
            const int idr = __builtin_amdgcn_readlane(id, r);
            if (lane == (t & 63)) { qhist[r] = qprev[r]; ihist[r] = idr; }
        }
        if ((t & 63) == 63 || !more) {
            const int t0 = t & ~63;
            if (lane <= t - t0) {
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    if (row0 + r < p.Ng) {
                        const int64_t gw = grow0 + r;
                        p.Q[gw * p.ldq + t0 + lane] = qhist[r];
                        if (p.idx) {
                            if (p.idx_bytes == 1) reinterpret_cast<int8_t*>(p.idx)[gw * p.ldi + t0 + lane] = (int8_t)ihist[r];
                            else reinterpret_cast<int16_t*>(p.idx)[gw * p.ldi + t0 + lane] = (int16_t)ihist[r];
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RT; ++r) wcur[r] = wn[r];
        n2cur = n2n;
    }
#pragma unroll
    for (int r = 0; r < RT; ++r) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float pq = qprev[r] * xo[e];       // xo = x_{d-1} after the last rotation
            u[r][e] = u[r][e] - pq;
        }
        if (row0 + r < p.Ng) {
            float* Urow = p.U + (grow0 + r) * p.ldu;
            if (p.vec) store_u16<true>(u[r], Urow, kbase, p.m);
            else store_u16<false>(u[r], Urow, kbase, p.m);
        }
    }
}

struct StreamCoop {
    int C;                              // members per row tile (1 = every workgroup owns whole rows)
    int tiles;                          // row tiles
    unsigned long long* xbuf;           // exchange granules [tiles][2][C][RT] (C > 1 only)
    int* status;
    unsigned spin_limit;
};

// COOP = false: workgroup (blockIdx.x, blockIdx.y = group) owns RT whole rows.
// COOP = true : groups == 1; a row tile's columns are split over C workgroups (block -> (tile, member) as in the
//               slab kernel), each streams its segment range and the per-row partial sums are exchanged per step
//               with the same granule protocol -- so that few long rows still fill every CU and RT rows share each
//               column load.
template <int RT, bool VEC, bool COOP>
__global__ void __launch_bounds__(512) gpfq_stream_kernel(LoopParams p, StreamCoop sc)
{
    extern __shared__ float smem[];                 // seg[2][RT][n_max] | qs[2][RT+1]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int NW = blockDim.x >> 6;
    const int S = p.S, C = COOP ? sc.C : 1;
    const int P = pow2_ceil(S);
    int tile, c, g;
    if (COOP) {
        g = 0;
        if ((sc.tiles & 7) == 0) {
            const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
            tile = (j / C) * 8 + xcd;
            c = j % C;
        } else {
            tile = blockIdx.x / C;
            c = blockIdx.x % C;
        }
    } else {
        tile = blockIdx.x; c = 0; g = blockIdx.y;
    }
    const int seg_lo = (c * S + C - 1) / C, seg_hi = ((c + 1) * S + C - 1) / C;
    const int n_max = (S + C - 1) / C;              // LDS row length (segments of the fullest member)
    const int bslots = P / C;                       // this member's aligned block of the slot tree
    const int per = bslots > 64 ? bslots / 64 : 1, nl = bslots > 64 ? 64 : bslots;
    const SlotMap smap = make_slot_map(S, P, c * bslots, per, lane, nl);
    float* segs = smem;
    float* qs = smem + 2 * RT * n_max;

    const int64_t row0 = (int64_t)tile * RT;
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

    bool dead = false;
    for (int64_t t = 0; t <= p.d && !dead; ++t) {
        const bool last = (t == p.d);               // extra pass: only the pending subtraction
        const bool first = (t == 0);
        const int par = (int)(t & 1);
        float w[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) w[r] = last ? 0.0f : p.W[grow[r] * p.ldw + t];
        float* seg = segs + (size_t)par * RT * n_max;
        for (int s = seg_lo + wave; s < seg_hi; s += NW) {
            const int64_t kbase = (int64_t)s * kSeg + 4 * lane;
            float xp[16], xc[16], ac[16];
            if (!first) load16(xp, XTg + (t - 1) * p.m_pad + (int64_t)s * kSeg);
            if (!last) {
                load16(xc, XTg + t * p.m_pad + (int64_t)s * kSeg);
                load16(ac, ATg + t * p.m_pad + (int64_t)s * kSeg);
            }
            // all RT residual rows are requested before any of them is used: one round trip per segment, not RT
            float u[RT][16];
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                if (first && !p.u_has_init) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) u[r][e] = 0.0f;
                } else {
                    load_u16<VEC>(u[r], p.U + grow[r] * p.ldu, kbase, p.m);
                }
            }
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                if (last) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) { float pq = qprev[r] * xp[e]; u[r][e] = u[r][e] - pq; }
                } else {
                    float acc = first ? sweep16<false>(u[r], xc, ac, xc, 0.0f, w[r])
                                      : sweep16<true>(u[r], xp, ac, xc, qprev[r], w[r]);
                    float sg = wave_tree64_lane63(acc);
                    if (lane == 63) seg[r * n_max + (s - seg_lo)] = sg;
                }
                if (valid[r] || RT == 1) store_u16<VEC>(u[r], p.U + grow[r] * p.ldu, kbase, p.m);
            }
        }
        if (last) break;
        __syncthreads();
        if (wave == 0) {
            // this member's block of the slot tree, row r's value parked in lane r
            float mine = 0.0f;
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const float pr = combine_slots(seg + r * n_max - seg_lo, smap, per, nl, seg_hi - 1);
                if (lane == r) mine = pr;
            }
            float v = mine;
            int blk = 1;
            bool timed_out = false;
            if (COOP) {
                const unsigned epoch = (unsigned)t + 1u;
                unsigned long long* xb_ = sc.xbuf + ((size_t)(tile * 2 + par) * C) * RT;
                if (lane < RT)
                    __hip_atomic_store(xb_ + (size_t)c * RT + lane,
                                       ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(mine),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool want = lane < RT * C;    // gather: lane = r*C + member
                const unsigned long long* src = xb_ + (want ? (size_t)(lane % C) * RT + (lane / C) : 0);
                unsigned long long gv = 0;
                unsigned spins = 0;
                for (;;) {
                    gv = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const bool ok = !want || ((unsigned)(gv >> 32) == epoch);
                    if (__all(ok)) break;
                    if (++spins > sc.spin_limit) { timed_out = true; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
                v = want ? __uint_as_float((unsigned)gv) : 0.0f;
                v = wave_tree_n(v, C);
                blk = C;
            }
            const int gr_ = lane / blk;
            const bool lead = (lane % blk == 0) && gr_ < RT;
            const float n2 = nrm[t];
            const float sv = (n2 > 0.0f) ? v / n2 : 0.0f;
            const bool rvalid = lead && (row0 + gr_ < p.Ng);
            const int64_t growl = (int64_t)g * p.Ng + (rvalid ? row0 + gr_ : p.Ng - 1);
            int id;
            const float q = quantize(p.qc, sv, p.row_id0 + (uint64_t)growl, (uint64_t)t, id);
            if (lead) qs[par * (RT + 1) + gr_] = q;
            if (rvalid && c == 0) store_q(p, growl, t, q, id);
            if (lane == 0) {
                qs[par * (RT + 1) + RT] = timed_out ? 1.0f : 0.0f;
                if (timed_out) {
                    atomicExch(sc.status, 1);
                    sc.status[1] = (int)t; sc.status[2] = tile; sc.status[3] = c;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RT; ++r) qprev[r] = qs[par * (RT + 1) + r];
        if (COOP) dead = qs[par * (RT + 1) + RT] != 0.0f;
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
        float sg = wave_tree64_lane63(acc);
        if (lane == 63) seg[s] = sg;
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

// Activation capture for Conv2d layers, fused: the sampled kernel-sized patches of an NCHW feature map go
// straight into the transposed, zero-padded column layout the loop kernels read (row f = feature (c, i, j),
// channel-major; column k = sampled patch k).  Patches sit on a grid whose stride is the KERNEL SIZE, as the
// reference's nn.Unfold(kernel_size, dilation, padding, kernel_size) does (quantize_neural_net.py:320).
// Replaces unfold + transpose + reshape + index (quantize_neural_net.py:334-347) and gpfq_transpose_pad_kernel.
// blockDim = 256: 64 patches x 4 feature lanes; grid = (m_pad / 64, ceil(D / 64)).
__global__ void __launch_bounds__(256) gpfq_gather_patches_kernel(const float* __restrict__ x, int C, int H, int W,
                                                                  int kh, int kw, int ph, int pw, int dh, int dw,
                                                                  int Lw, int64_t L, const int64_t* __restrict__ patch,
                                                                  int64_t m, float* __restrict__ outT, int64_t m_pad, int D)
{
    const int kx = threadIdx.x & 63, fy = threadIdx.x >> 6;
    const int64_t k = (int64_t)blockIdx.x * 64 + kx;
    const bool live = k < m;
    int64_t b = 0;
    int y0 = 0, x0 = 0;
    if (live) {
        const int64_t pi = patch[k];
        b = pi / L;
        const int l = (int)(pi - b * L);
        y0 = (l / Lw) * kh - ph;
        x0 = (l % Lw) * kw - pw;
    }
    const float* __restrict__ img = x + b * (int64_t)C * H * W;
    const int f_end = min(D, (int)(blockIdx.y + 1) * 64);
    for (int f = blockIdx.y * 64 + fy; f < f_end; f += 4) {
        const int c = f / (kh * kw), r = f - c * (kh * kw);
        const int yy = y0 + (r / kw) * dh, xx = x0 + (r % kw) * dw;
        float v = 0.0f;
        if (live && yy >= 0 && yy < H && xx >= 0 && xx < W) v = img[((int64_t)c * H + yy) * W + xx];
        outT[(int64_t)f * m_pad + k] = v;
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
constexpr size_t kScratchBytes = 128 * 1024;        // [0, 96 KiB) exchange granules, [96 KiB, ...) status words
constexpr size_t kScratchStatusOffset = 96 * 1024;

struct Plan {
    int kind;      // GPFQ_PLAN_STREAM / GPFQ_PLAN_RESIDENT / GPFQ_PLAN_COOP
    int RT;        // rows per workgroup
    int waves;     // waves per workgroup
    int S;         // segments per row
    int C;         // coop: members per row tile
    int tiles;     // coop: row tiles
};

int device_cu_count()
{
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cus[dev] == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        cus[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return cus[dev];
}

int env_int(const char* name, int dflt)
{
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

int floor_pow2(int64_t n)
{
    int p = 1;
    while ((int64_t)p * 2 <= n) p *= 2;
    return p;
}

int slab_max_waves(bool coop, int RT);

// Cost model of one column step (microseconds), fitted to MI355X measurements (tools/layer_bench.py): a fixed
// latency (barriers, reductions, quantizer; plus the granule exchange when cooperative), RT sweeps issued by a
// wave that owns its SIMD slot, and the per-CU column traffic / issue contention that grows with the waves on a CU.
double slab_step_cost(int RT, int waves, int C)
{
    // gathering from 16 / 32 / 64 members costs 0.9 / 2.3 / 4 us more than from <= 8 (measured 0.9 and 2.3)
    return 0.5 + 0.285 * RT + 0.134 * waves + (C >= 64 ? 4.0 : C >= 32 ? 2.3 : C >= 16 ? 0.9 : 0.0);
}

// Cooperative configuration: cheapest modelled step among the (RT, C) pairs whose grid is co-resident.
// Depends on (Ng, S, CU count) only -- never on the data.
bool choose_coop(int64_t Ng, int S, int cus, Plan* pl)
{
    const int force_rt = env_int("GPFQ_COOP_RT", 0), force_c = env_int("GPFQ_COOP_C", 0);
    const int wgs_per_cu = env_int("GPFQ_COOP_WGS_PER_CU", 1) > 1 ? 2 : 1;
    const int capacity = cus * wgs_per_cu;
    double best = 1e30;
    bool found = false;
    for (int RT = 4; RT >= 1; RT >>= 1) {
        if (force_rt && RT != force_rt) continue;
        const int64_t tiles = (Ng + RT - 1) / RT;
        if (tiles > capacity) continue;
        for (int C = 64 / RT; C >= 2; C >>= 1) {
            if (force_c && C != force_c) continue;
            if (C > S || tiles * C > capacity) continue;
            const int NW = (S + C - 1) / C;
            if (NW > slab_max_waves(true, RT)) continue;
            const int wgs = (int)tiles * C;
            const int per_cu = (wgs + cus - 1) / cus;
            if ((per_cu * NW + 3) / 4 > 4) continue;
            // rounds of work if the grid does not cover the chip are not modelled: fewer workgroups than CUs
            // simply leave CUs idle, which costs nothing per step
            double cost = slab_step_cost(RT, per_cu * NW, C);
            if (RT == 4 && NW > 8) cost += 1.2;      // the 12-wave RT=4 variant spills registers
            if (!found || cost < best - 1e-9) {
                found = true;
                best = cost;
                pl->kind = GPFQ_PLAN_COOP; pl->RT = RT; pl->C = C; pl->tiles = (int)tiles; pl->waves = NW; pl->S = S;
            }
        }
    }
    return found;
}

// Streaming configuration.  Whole rows per workgroup when there are enough rows to fill the chip; otherwise
// (groups == 1, scratch available) the rows' columns are split over C workgroups so that tiles*C covers the CUs
// and RT rows share every column load.  Depends on (Ng, S, groups, CU count) only.
void choose_stream(int64_t Ng, int S, int groups, bool allow_coop, Plan* pl)
{
    const int cus = device_cu_count();
    pl->kind = GPFQ_PLAN_STREAM;
    pl->S = S;
    pl->C = 1;
    pl->RT = Ng >= 1024 ? 4 : (Ng >= 512 ? 2 : 1);
    pl->tiles = (int)((Ng + pl->RT - 1) / pl->RT);
    pl->waves = S < 8 ? S : 8;
    if (!allow_coop || groups != 1 || env_int("GPFQ_COOP_DISABLE", 0)) return;
    const int force_c = env_int("GPFQ_STREAM_C", 0), force_rt = env_int("GPFQ_STREAM_RT", 0);
    // Measured (tools/layer_bench.py, N = 64..512, m = 201 728..803 840): with all RT residual rows requested up
    // front, four rows per workgroup and enough members to keep one workgroup per CU beat whole rows (N = 256,
    // m = 803 840: 398 -> 339 us per column; N = 512, m = 201 728: 195 -> 170): fewer column bytes per U byte.
    // So: most workgroups first (up to one per CU), then most rows per workgroup.
    // Exception: a residual that fits the 256-MB Infinity Cache with whole rows already covering the chip streams
    // faster as it is (N = 256, m = 51 200: 15.5 vs 19.1 us).
    int best_rt = pl->RT, best_c = 1;
    int best_score = (pl->tiles >= cus ? cus : pl->tiles) * 8 + pl->RT;
    const bool cache_resident = (double)Ng * S * 4096.0 <= 200e6;
    if (cache_resident && pl->tiles >= cus && !(force_c && force_rt)) return;
    for (int RT = 4; RT >= 1; RT >>= 1) {
        if (force_rt && RT != force_rt) continue;
        const int64_t tiles = (Ng + RT - 1) / RT;
        if (tiles > cus) continue;
        for (int C = 64 / RT; C >= 2; C >>= 1) {
            if (force_c && C != force_c) continue;
            if (tiles * C > cus || S / C < 8) continue;           // every member keeps >= 8 segments (one per wave)
            const int score = (int)tiles * C * 8 + RT;
            if (score > best_score || (force_c && force_rt)) { best_score = score; best_rt = RT; best_c = C; }
            break;                                                 // largest C for this RT
        }
    }
    if (best_c > 1) {
        pl->RT = best_rt; pl->C = best_c;
        pl->tiles = (int)((Ng + best_rt - 1) / best_rt);
        pl->waves = 8;
    }
}

int choose_plan(int64_t Ng, int64_t m_pad, int groups, int requested, bool have_scratch, Plan* out)
{
    Plan pl;
    pl.C = 1; pl.tiles = 0;
    if (m_pad / gpfq::kSeg > 1024) return fail(GPFQ_ERR_UNSUPPORTED, "m > 1048576 calibration rows is not supported");
    pl.S = (int)(m_pad / gpfq::kSeg);
    if (requested < GPFQ_PLAN_AUTO || requested > GPFQ_PLAN_COOP) return fail(GPFQ_ERR_ARG, "unknown plan id");
    if (requested == GPFQ_PLAN_RESIDENT && pl.S > kMaxResidentSegments)
        return fail(GPFQ_ERR_UNSUPPORTED, "resident plan needs m_pad <= 16384");
    const int cus = device_cu_count();
    if (requested == GPFQ_PLAN_RESIDENT || (requested == GPFQ_PLAN_AUTO && pl.S <= kMaxResidentSegments)) {
        pl.kind = GPFQ_PLAN_RESIDENT;
        pl.waves = pl.S;
        // rows per workgroup: share the activation registers between rows once there are more rows than the
        // chip has room for one-row workgroups (the roomier variants exist for <= 8 waves)
        pl.RT = 1;
        const int force_rt = env_int("GPFQ_RESIDENT_RT", 0);
        if (pl.S <= 8) {
            // measured (tools/layer_bench.py): one row per workgroup is never slower than two or four on the
            // ResNet-50 shapes; the larger variants exist for experiments (GPFQ_RESIDENT_RT)
            if (force_rt == 1 || force_rt == 2 || force_rt == 4) pl.RT = force_rt;
        }
        *out = pl;
        return GPFQ_OK;
    }
    if (requested == GPFQ_PLAN_COOP || (requested == GPFQ_PLAN_AUTO && !env_int("GPFQ_COOP_DISABLE", 0))) {
        if (groups == 1 && have_scratch && choose_coop(Ng, pl.S, cus, &pl)) {
            *out = pl;
            return GPFQ_OK;
        }
        if (requested == GPFQ_PLAN_COOP)
            return fail(GPFQ_ERR_UNSUPPORTED, "cooperative plan needs groups == 1, a scratch buffer and a shape that fits");
    }
    choose_stream(Ng, pl.S, groups, have_scratch, &pl);
    *out = pl;
    return GPFQ_OK;
}

template <int RT>
int launch_stream(const Plan& pl, const gpfq::LoopParams& p, int groups, bool vec, void* scratch, hipStream_t st)
{
    const int S = p.S;
    const int C = pl.C > 1 ? pl.C : 1;
    const int n_max = (S + C - 1) / C;
    dim3 block((unsigned)(64 * pl.waves), 1, 1);
    const size_t shm = sizeof(float) * (2 * RT * (size_t)n_max + 2 * (RT + 1));
    gpfq::StreamCoop sc;
    sc.C = C; sc.tiles = pl.tiles; sc.xbuf = nullptr; sc.status = nullptr;
    sc.spin_limit = (unsigned)env_int("GPFQ_COOP_SPIN_LIMIT", 1 << 21);
    hipError_t e;
    if (C > 1) {
        const int nblocks = pl.tiles * C;
        int nb = 0;
        e = vec ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gpfq::gpfq_stream_kernel<RT, true, true>, 64 * pl.waves, shm)
                : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gpfq::gpfq_stream_kernel<RT, false, true>, 64 * pl.waves, shm);
        if (e != hipSuccess) return hip_fail(e, "occupancy query");
        const int cus = device_cu_count();
        if (nb < 1 || (nblocks + cus - 1) / cus > nb) return fail(GPFQ_ERR_UNSUPPORTED, "cooperative grid does not fit on the device");
        size_t xbytes = (size_t)pl.tiles * 2 * C * RT * sizeof(unsigned long long);
        xbytes = (xbytes + 15) & ~(size_t)15;
        if (xbytes > kScratchStatusOffset) return fail(GPFQ_ERR_UNSUPPORTED, "exchange buffer larger than the scratch area");
        e = hipMemsetAsync(scratch, 0, xbytes, st);
        if (e != hipSuccess) return hip_fail(e, "exchange buffer memset");
        sc.xbuf = reinterpret_cast<unsigned long long*>(scratch);
        sc.status = reinterpret_cast<int*>(static_cast<char*>(scratch) + kScratchStatusOffset);
        dim3 grid((unsigned)nblocks, 1, 1);
        if (vec) hipLaunchKernelGGL((gpfq::gpfq_stream_kernel<RT, true, true>), grid, block, shm, st, p, sc);
        else hipLaunchKernelGGL((gpfq::gpfq_stream_kernel<RT, false, true>), grid, block, shm, st, p, sc);
    } else {
        dim3 grid((unsigned)((p.Ng + RT - 1) / RT), (unsigned)groups, 1);
        if (vec) hipLaunchKernelGGL((gpfq::gpfq_stream_kernel<RT, true, false>), grid, block, shm, st, p, sc);
        else hipLaunchKernelGGL((gpfq::gpfq_stream_kernel<RT, false, false>), grid, block, shm, st, p, sc);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "GPFQ streaming kernel launch");
    return GPFQ_OK;
}

gpfq::SlabParams make_slab_params(const Plan& pl, const gpfq::LoopParams& p, bool vec, void* scratch)
{
    gpfq::SlabParams sp;
    sp.W = p.W; sp.Q = p.Q; sp.U = p.U; sp.idx = p.idx; sp.AT = p.AT; sp.XT = p.XT; sp.nrm2 = p.nrm2;
    sp.xbuf = reinterpret_cast<unsigned long long*>(scratch);
    sp.status = scratch ? reinterpret_cast<int*>(static_cast<char*>(scratch) + kScratchStatusOffset) : nullptr;
    sp.ldw = p.ldw; sp.ldq = p.ldq; sp.ldu = p.ldu; sp.ldi = p.ldi; sp.m = p.m; sp.m_pad = p.m_pad;
    sp.Ng = (int)p.Ng; sp.d = (int)p.d; sp.S = pl.S; sp.C = pl.C; sp.tiles = pl.tiles; sp.idx_bytes = p.idx_bytes;
    sp.vec = vec ? 1 : 0;
    sp.step = p.qc.step; sp.Kf = p.qc.Kf; sp.lamb = p.qc.lamb;
    sp.spin_limit = (unsigned)env_int("GPFQ_COOP_SPIN_LIMIT", 1 << 21);
    sp.seed = p.qc.seed; sp.row_id0 = p.row_id0;
    return sp;
}

template <int RT, int MODE, bool COOP, int MAXW>
int launch_slab_t(const Plan& pl, const gpfq::SlabParams& sp, int groups, void* scratch, hipStream_t st)
{
    if (pl.waves > MAXW) return fail(GPFQ_ERR_UNSUPPORTED, "internal: waves exceed the kernel variant's bound");
    // one more wave for the reducer role when the variant's register budget allows it
    const int nwaves = pl.waves + ((pl.waves + 1 <= MAXW && !env_int("GPFQ_NO_REDUCER_WAVE", 0)) ? 1 : 0);
    const int threads = 64 * nwaves;
    const size_t shm = sizeof(float) * (2 * RT * (size_t)nwaves + 2 * (RT + 1) + 2 * RT * 64);
    hipError_t e;
    dim3 grid;
    if (COOP) {
        const int nblocks = pl.tiles * pl.C;
        // every workgroup must be resident at once: check the grid against the occupancy query (the query is
        // known to over-report by one only near the SGPR limit of >= 6 waves per SIMD; these kernels run at
        // <= 4, so the answer is taken as is -- and every spin is bounded anyway)
        int nb = 0;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gpfq::gpfq_slab_kernel<RT, MODE, COOP, MAXW>, threads, shm);
        if (e != hipSuccess) return hip_fail(e, "occupancy query");
        const int cus = device_cu_count();
        const int need = (nblocks + cus - 1) / cus;
        if (nb < 1 || need > nb || (need * nwaves + 3) / 4 > 4)
            return fail(GPFQ_ERR_UNSUPPORTED, "cooperative grid does not fit on the device");
        size_t xbytes = (size_t)pl.tiles * 2 * pl.C * RT * sizeof(unsigned long long);
        xbytes = (xbytes + 15) & ~(size_t)15;
        if (xbytes > kScratchStatusOffset) return fail(GPFQ_ERR_UNSUPPORTED, "exchange buffer larger than the scratch area");
        e = hipMemsetAsync(scratch, 0, xbytes, st);
        if (e != hipSuccess) return hip_fail(e, "exchange buffer memset");
        grid = dim3((unsigned)nblocks, 1, 1);
    } else {
        grid = dim3((unsigned)((sp.Ng + RT - 1) / RT), (unsigned)groups, 1);
    }
    hipLaunchKernelGGL((gpfq::gpfq_slab_kernel<RT, MODE, COOP, MAXW>), grid, dim3((unsigned)threads), shm, st, sp);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "GPFQ slab kernel launch");
    return GPFQ_OK;
}

template <int RT, bool COOP, int MAXW>
int launch_slab_m(const Plan& pl, const gpfq::SlabParams& sp, int mode, int groups, void* scratch, hipStream_t st)
{
    switch (mode) {
    case gpfq::MODE_SOFT: return launch_slab_t<RT, gpfq::MODE_SOFT, COOP, MAXW>(pl, sp, groups, scratch, st);
    case gpfq::MODE_HARD: return launch_slab_t<RT, gpfq::MODE_HARD, COOP, MAXW>(pl, sp, groups, scratch, st);
    case gpfq::MODE_STOCHASTIC: return launch_slab_t<RT, gpfq::MODE_STOCHASTIC, COOP, MAXW>(pl, sp, groups, scratch, st);
    default: return launch_slab_t<RT, gpfq::MODE_MSQ, COOP, MAXW>(pl, sp, groups, scratch, st);
    }
}

// The instantiated (rows per workgroup, wave bound) pairs -- keep slab_variant_ok() in step with this switch.
int launch_slab(const Plan& pl, const gpfq::LoopParams& p, int groups, bool vec, void* scratch, hipStream_t st)
{
    const gpfq::SlabParams sp = make_slab_params(pl, p, vec, scratch);
    const int m = p.qc.mode;
    if (pl.kind == GPFQ_PLAN_RESIDENT && pl.S == 1 && !env_int("GPFQ_NO_WAVE_KERNEL", 0)) {
        // one wave per row tile; four rows per wave once there are more rows than the chip has wave slots for
        // (the quantizer then runs once per four rows: the step is VALU-issue-bound at 16 waves per CU)
        const int wrt = env_int("GPFQ_WAVE_RT", sp.Ng >= 4096 ? 4 : 1) == 4 ? 4 : 1;     // measured: 4096 rows 1.5x faster, 2048 rows 1.3x slower
        dim3 grid((unsigned)((sp.Ng + 4 * wrt - 1) / (4 * wrt)), (unsigned)groups, 1);
#define GPFQ_LAUNCH_WAVE(RTV, MODEV) hipLaunchKernelGGL((gpfq::gpfq_wave_kernel<RTV, MODEV>), grid, dim3(256), 0, st, sp)
        if (wrt == 4) {
            switch (m) {
            case gpfq::MODE_SOFT: GPFQ_LAUNCH_WAVE(4, gpfq::MODE_SOFT); break;
            case gpfq::MODE_HARD: GPFQ_LAUNCH_WAVE(4, gpfq::MODE_HARD); break;
            case gpfq::MODE_STOCHASTIC: GPFQ_LAUNCH_WAVE(4, gpfq::MODE_STOCHASTIC); break;
            default: GPFQ_LAUNCH_WAVE(4, gpfq::MODE_MSQ); break;
            }
        } else {
            switch (m) {
            case gpfq::MODE_SOFT: GPFQ_LAUNCH_WAVE(1, gpfq::MODE_SOFT); break;
            case gpfq::MODE_HARD: GPFQ_LAUNCH_WAVE(1, gpfq::MODE_HARD); break;
            case gpfq::MODE_STOCHASTIC: GPFQ_LAUNCH_WAVE(1, gpfq::MODE_STOCHASTIC); break;
            default: GPFQ_LAUNCH_WAVE(1, gpfq::MODE_MSQ); break;
            }
        }
#undef GPFQ_LAUNCH_WAVE
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "GPFQ wave kernel launch");
        return GPFQ_OK;
    }
    if (pl.kind == GPFQ_PLAN_RESIDENT) {
        if (pl.RT == 1) return launch_slab_m<1, false, 16>(pl, sp, m, groups, scratch, st);
        if (pl.RT == 2) return launch_slab_m<2, false, 8>(pl, sp, m, groups, scratch, st);
        return launch_slab_m<4, false, 8>(pl, sp, m, groups, scratch, st);
    }
    if (pl.RT == 1) return launch_slab_m<1, true, 12>(pl, sp, m, groups, scratch, st);
    if (pl.RT == 2) {
        if (pl.waves <= 8) return launch_slab_m<2, true, 8>(pl, sp, m, groups, scratch, st);
        return launch_slab_m<2, true, 12>(pl, sp, m, groups, scratch, st);
    }
    if (pl.waves <= 8) return launch_slab_m<4, true, 8>(pl, sp, m, groups, scratch, st);
    return launch_slab_m<4, true, 12>(pl, sp, m, groups, scratch, st);   // 168-VGPR budget: spills a little
}

// most waves per workgroup an instantiation exists for
int slab_max_waves(bool coop, int RT)
{
    if (!coop) return RT == 1 ? 16 : 8;
    return 12;
}

int run_loop(gpfq::LoopParams p, int groups, int plan, void* scratch, size_t scratch_bytes, hipStream_t st)
{
    if (p.Ng <= 0 || p.d <= 0 || groups <= 0) return GPFQ_OK;   // nothing to do
    if (p.Ng > 0x7fffffff || p.d > 0x7fffffff) return fail(GPFQ_ERR_UNSUPPORTED, "N or d beyond 2^31");
    const bool have_scratch = scratch && scratch_bytes >= kScratchBytes && !(reinterpret_cast<uintptr_t>(scratch) & 255);
    Plan pl;
    // the register-resident plans start from U = 0; a caller-provided initial residual (the in-place
    // _quantization surface) streams through memory
    if (p.u_has_init && plan == GPFQ_PLAN_AUTO) plan = GPFQ_PLAN_STREAM;
    if (p.u_has_init && plan != GPFQ_PLAN_STREAM)
        return fail(GPFQ_ERR_UNSUPPORTED, "an initial residual needs the streaming plan");
    int rc = choose_plan(p.Ng, p.m_pad, groups, plan, have_scratch, &pl);
    if (rc) return rc;
    if (groups > 65535) return fail(GPFQ_ERR_UNSUPPORTED, "groups > 65535");
    p.S = pl.S;
    const bool vec = ((p.ldu & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.U) & 15) == 0);
    if (pl.kind == GPFQ_PLAN_COOP) {
        rc = launch_slab(pl, p, groups, vec, scratch, st);
        if (rc != GPFQ_ERR_UNSUPPORTED || plan == GPFQ_PLAN_COOP) return rc;
        rc = choose_plan(p.Ng, p.m_pad, groups, GPFQ_PLAN_STREAM, have_scratch, &pl);   // does not fit: stream instead
        if (rc) return rc;
    }
    if (pl.kind == GPFQ_PLAN_RESIDENT) return launch_slab(pl, p, groups, vec, scratch, st);
    for (int attempt = 0; attempt < 2; ++attempt) {
        switch (pl.RT) {
        case 4: rc = launch_stream<4>(pl, p, groups, vec, scratch, st); break;
        case 2: rc = launch_stream<2>(pl, p, groups, vec, scratch, st); break;
        default: rc = launch_stream<1>(pl, p, groups, vec, scratch, st); break;
        }
        if (rc != GPFQ_ERR_UNSUPPORTED || pl.C <= 1) return rc;
        choose_stream(p.Ng, pl.S, groups, false, &pl);          // cooperative grid did not fit: whole rows
    }
    return rc;
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

size_t gpfq_scratch_bytes(void) { return kScratchBytes; }

static size_t ws_cols_bytes(int64_t d_g, int64_t m, int groups)
{
    const size_t D = (size_t)d_g * (size_t)groups;
    return D * (size_t)gpfq_padded_m(m) * sizeof(float);
}
static size_t ws_nrm_bytes(int64_t d_g, int groups)
{
    return (((size_t)d_g * (size_t)groups * sizeof(float) + 255) / 256) * 256;
}

size_t gpfq_workspace_bytes(int64_t N, int64_t d_g, int64_t m, int groups)
{
    (void)N;
    if (d_g < 0 || m < 0 || groups < 1) return 0;
    return kScratchBytes + 2 * ws_cols_bytes(d_g, m, groups) + ws_nrm_bytes(d_g, groups) + 256;
}

int gpfq_read_status(void* scratch, int* status_host4, void* stream)
{
    if (!scratch || !status_host4) return fail(GPFQ_ERR_ARG, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    char* sp = static_cast<char*>(scratch) + kScratchStatusOffset;
    hipError_t e = hipMemcpyAsync(status_host4, sp, 4 * sizeof(int), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return hip_fail(e, "status read");
    if (status_host4[0] != 0) {
        e = hipMemsetAsync(sp, 0, 4 * sizeof(int), st);
        if (e != hipSuccess) return hip_fail(e, "status reset");
        return fail(GPFQ_ERR_TIMEOUT, "cooperative kernel timed out waiting for a peer workgroup");
    }
    return GPFQ_OK;
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
                          void* idx, int64_t ldi, int idx_bytes, int plan, void* scratch, size_t scratch_bytes,
                          void* stream)
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
    return run_loop(p, 1, plan, scratch, scratch_bytes, (hipStream_t)stream);
}

int gpfq_quantize_groups_prepared_f32(const float* W, float* Q, float* U, const float* AT, const float* XT,
                                      const float* nrm2, int64_t N, int64_t d_g, int64_t m, int64_t m_pad,
                                      int groups, float step, int K, int mode, float lamb, uint64_t seed,
                                      uint64_t row_id0, void* idx, int idx_bytes, int plan, void* scratch,
                                      size_t scratch_bytes, void* stream)
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
    return run_loop(p, groups, plan, scratch, scratch_bytes, (hipStream_t)stream);
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
    char* ws = static_cast<char*>(workspace);          // [scratch][AT][XT][nrm2]
    const size_t cb = ws_cols_bytes(d_g, m, groups);
    float* AT = reinterpret_cast<float*>(ws + kScratchBytes);
    float* XT = reinterpret_cast<float*>(ws + kScratchBytes + cb);
    float* nrm2 = reinterpret_cast<float*>(ws + kScratchBytes + 2 * cb);
    int rc = gpfq_prepare_columns_f32(A, lda, X, ldx, m, D, AT, XT, nrm2, mp, stream);
    if (rc) return rc;
    return gpfq_quantize_groups_prepared_f32(W, Q, U, AT, XT, nrm2, N, d_g, m, mp, groups, step, K, mode, lamb, seed,
                                             row_id0, idx, idx_bytes, plan, ws, kScratchBytes, stream);
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

int gpfq_column_norms_f32(const float* XT, int64_t D, int64_t m, int64_t m_pad, float* nrm2, void* stream)
{
    if (!XT || !nrm2) return fail(GPFQ_ERR_ARG, "null pointer");
    if (D < 0 || m_pad != gpfq_padded_m(m)) return fail(GPFQ_ERR_ARG, "bad shape (m_pad must equal gpfq_padded_m(m))");
    if (D == 0) return GPFQ_OK;
    const int S = (int)(m_pad / gpfq::kSeg);
    hipLaunchKernelGGL(gpfq::gpfq_colnorm_kernel, dim3((unsigned)D), dim3(256), sizeof(float) * (size_t)S,
                       (hipStream_t)stream, XT, m_pad, S, nrm2);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "colnorm launch");
    return GPFQ_OK;
}

int gpfq_gather_patches_f32(const float* x, int64_t B, int64_t C, int64_t H, int64_t W, int kh, int kw, int pad_h,
                            int pad_w, int dil_h, int dil_w, const int64_t* patch_index, int64_t m, float* outT,
                            int64_t m_pad, void* stream)
{
    if (!x || !outT || (!patch_index && m > 0)) return fail(GPFQ_ERR_ARG, "null pointer");
    if (B < 1 || C < 1 || H < 1 || W < 1 || kh < 1 || kw < 1 || dil_h < 1 || dil_w < 1 || pad_h < 0 || pad_w < 0 || m < 0)
        return fail(GPFQ_ERR_ARG, "bad shape");
    if (m_pad != gpfq_padded_m(m)) return fail(GPFQ_ERR_ARG, "m_pad must equal gpfq_padded_m(m)");
    // nn.Unfold(kernel, dilation, padding, stride = kernel): blocks per axis
    const int64_t Lh = (H + 2 * pad_h - (int64_t)dil_h * (kh - 1) - 1) / kh + 1;
    const int64_t Lw = (W + 2 * pad_w - (int64_t)dil_w * (kw - 1) - 1) / kw + 1;
    if (Lh < 1 || Lw < 1) return fail(GPFQ_ERR_ARG, "kernel larger than the padded input");
    const int64_t D = C * kh * kw;
    if (D > 0x7fffffff || C * H * W > 0x7fffffffffffLL) return fail(GPFQ_ERR_UNSUPPORTED, "feature map too large");
    dim3 grid((unsigned)(m_pad / 64), (unsigned)((D + 63) / 64), 1);
    if (grid.y > 65535) return fail(GPFQ_ERR_UNSUPPORTED, "too many features");
    hipLaunchKernelGGL(gpfq::gpfq_gather_patches_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, (int)C, (int)H, (int)W,
                       kh, kw, pad_h, pad_w, dil_h, dil_w, (int)Lw, Lh * Lw, patch_index, m, outT, m_pad, (int)D);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "gather_patches launch");
    (void)B;
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
    int rc = choose_plan(N / groups, gpfq_padded_m(m), groups, plan, true, &pl);
    if (rc) return rc;
    if (buf && buf_bytes) {
        if (pl.kind == GPFQ_PLAN_COOP)
            snprintf(buf, buf_bytes, "coop RT=%d C=%d waves=%d S=%d grid=%d d=%lld", pl.RT, pl.C, pl.waves, pl.S,
                     pl.tiles * pl.C, (long long)d_g);
        else if (pl.kind == GPFQ_PLAN_STREAM && pl.C > 1)
            snprintf(buf, buf_bytes, "stream RT=%d C=%d waves=%d S=%d grid=%d d=%lld", pl.RT, pl.C, pl.waves, pl.S,
                     pl.tiles * pl.C, (long long)d_g);
        else
            snprintf(buf, buf_bytes, "%s RT=%d waves=%d S=%d grid=(%lld,%d) d=%lld",
                     pl.kind == GPFQ_PLAN_RESIDENT ? "resident" : "stream", pl.RT, pl.waves, pl.S,
                     (long long)((N / groups + pl.RT - 1) / pl.RT), groups, (long long)d_g);
    }
    return pl.kind;
}

}  // extern "C"
