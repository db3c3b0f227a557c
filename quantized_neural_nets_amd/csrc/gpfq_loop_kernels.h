// gpfq_loop_kernels.h -- the GPFQ loop kernels (gfx950): gpfq_resident_* (residual resident in registers, whole rows per
// workgroup; _w1 = one-segment rows, a workgroup of one wave), gpfq_coop_* (the same with rows split over co-operating
// workgroups) and gpfq_stream_kernel (residual streamed through HBM / L2).  Reference: StepAlgorithm._quantization,
// step_algorithm.py:107-148.  One launch runs the WHOLE column loop of a layer (all groups): rows of the residual U
// are independent, and per step a workgroup makes ONE pass over its rows, fusing
//   u -= q_{t-1} x_{t-1};  u += w_t a_t;  <u, x_t>.
#pragma once
#include <type_traits>
#include "gpfq_device.h"

namespace gpfq {

struct LoopParams {
    const float* W; int64_t ldw;
    float* Q; int64_t ldq;
    float* U; int64_t ldu; int u_has_init;
    const float* AT; const float* XT;
    const float* nrm2; // [columns][2]: ||x_t||^2 and its reciprocal (0 for a zero column)
    int64_t Ng;        // rows per group
    int64_t d;         // columns per group
    int64_t m; int64_t m_pad; int S;
    QuantCfg qc;
    uint64_t row_id0;
    void* idx; int64_t ldi; int idx_bytes;
    float* usq;        // optional [rows][S]: per-segment sums of squares of the final residual (error-metric epilogue)
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
// Register-resident plans (resident / cooperative): the residual lives in registers for the whole column loop.
//
// A workgroup owns an RT x (n_own segments) slab of U: RT rows sharing the activation registers, one wave
// per canonical segment.  Per step: every wave sweeps its segment (fused update + fma chain, sweep16) and
// reduces the 64 lane chains (wave_tree64_lane63); then the canonical slot tree is finished, the sum divided by the
// column norm and quantized (the RT rows in RT different lanes).
//
//   resident: one workgroup holds whole rows (S <= 16 segments); every wave finishes the tree for itself, one
//     barrier per step -- gpfq_resident_kernel below.
//   cooperative (gpfq_coop_kernel): a row is split by columns over C workgroups ("members"), needed when rows are
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
    float* usq;
    unsigned long long* xbuf; int* status;
    int64_t ldw, ldq, ldu, ldi, m, m_pad;
    int Ng, d, S, C, tiles, idx_bytes, vec;
    int pace;          // cooperative kernels: pauses (s_sleep 1 each) between the column requests issued in the exchange window; 0 = off
    int xcd_tiles;     // cooperative kernels: keep the members of a row tile on one XCD (needs tiles % 8 == 0)
    float step, Kf, lamb;
    float inv_step;    // fl(1 / step) for quant_msq_from_dot; NaN switches the division-free path off
    float msq_thr;     // 0.5 - (K + 4) * 2^-18: how far from the middle of [n, n + 1) that path trusts its floor
    unsigned spin_limit;    // cooperative kernels: 256 * (polls before an exchange gives up, host-clamped to < 2^24) + (low five bits: pause before the first poll, in units of 256 clocks, 0 .. 31)
    uint64_t seed, row_id0;
    int prefetch_lines; // resident kernels: lines of every segment one workgroup's agent touches (32 / workgroups per XCD, at least 1)
    int prefetch_ahead; // resident kernels: columns the prefetch agent (one extra wave, active in one workgroup per XCD) runs ahead of the sweeps; 0 = no agent wave
    unsigned salt;     // pipelined kernels: 8-bit launch number carried in every granule's epoch word (a line left behind by an earlier launch never matches)
    int allow_local;   // pipelined kernels: members of a tile that find themselves on ONE XCD may publish with plain stores (the XCD's L2 is their coherence point)
    int reducer_prio;  // twelve-row pipelined kernels: the reducer wave's issue priority (s_setprio 0 .. 3)
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
// for all RT rows at once, exchange with the other members, divide by the column norm, quantize the RT rows in RT
// lanes, hand q back through LDS and write Q / idx.
// Lane layout: every row owns an aligned block of lanes padded with +0.0f -- 16 lanes (one DPP row) for the member's
// own slots (nl <= 16 of them: a member has at most 12 segments), `stride` = max(16, C) lanes for the gathered
// members (max(16, C/2) when RT * C > 64) -- so that both trees run all their levels without testing how many are needed: a taken branch costs
// ~20 cycles on this chain, and adding +0.0f is exact (a partial sum is never -0.0f, see wave_tree16_zero_padded).
// Returns whether the exchange timed out (wave-uniform).  Nothing else learns of it inside the launch: the status word is
// raised, the outputs of the launch are void, and the sweep waves simply finish the loop on whatever q comes out (a
// per-step abort test in every wave cost more than it could ever save).
// ABORTWORD: the old protocol, kept for the four-row 12-wave variant only (no register left for the `gave_up` state):
// the reducer leaves the fact in a word of qs and every wave tests it behind barrier 2 and leaves the loop.
// QUAD: the one-row 16-wave variant gathers FOUR members per lane (4j .. 4j+3: two levels of the tree over the members in
// the lane), so that up to 256 members -- a whole chip for one row of up to 4096 segments -- fit the 64 lanes.  With two
// rows (256 members x 2 rows = 512 granules) a lane gathers EIGHT members, 8j .. 8j+7, three levels in the lane.
// GOVR (QUAD only) overrides the members per lane: 16 for FOUR rows on 256 members (1024 granules, four batches of four).
template <int RT, int MODE, bool FAST, bool ABORTWORD, bool QUAD = false, int GOVR = 0>
__device__ __forceinline__ bool reducer_section(const SlabParams& p, const float* seg, float* qs, const SlotMap smap,
                                        int NW, int nl, int lane, int tile, int c, int C, int par,
                                        int t, float n2cur, float in2cur, int row0, int64_t grow0, int seg_lo, bool gave_up)
{
    // this workgroup's block of the slot tree for all RT rows at once: lane = 16 * row + slot
    float v;
    {
        const int r16 = lane >> 4;
        const int rr = r16 < RT ? r16 : 0;
        const float val = seg[rr * NW + (smap.s0 - seg_lo)];
        v = wave_tree16_zero_padded(((smap.mask & 1u) && r16 < RT) ? val : 0.0f);
    }
    bool timed_out = false;
    // Gather layout: lane = stride * row + member.  More than 64 granules (RT * C <= 128: long rows that run in rounds)
    // are gathered two members per lane, 2j and 2j+1 -- adjacent member blocks, the pair the first level of the tree
    // over the members adds anyway.
    static_assert(!QUAD || RT == 1 || RT == 2 || RT == 4, "four (two rows: eight) members per lane");
    constexpr int G = GOVR ? GOVR : (RT == 2 ? 8 : 4);       // QUAD: members per lane
    const bool wide = !QUAD && RT * C > 64;
    const int per_row = QUAD ? C / G : (wide ? C >> 1 : C);  // lanes per row
    const int sh = per_row <= 16 ? 4 : (per_row <= 32 ? 5 : 6);   // log2 of the lane stride of a row in the gather
    const int gr_ = lane >> sh;              // row of this lane
    const int member = lane & ((1 << sh) - 1);
    {
        const unsigned epoch = (unsigned)t + 1u;
        // (32-bit offset arithmetic: the exchange area is 96 KiB; as size_t this was four 64-bit multiplies per step)
        unsigned long long* xb_ = p.xbuf + (unsigned)(tile * 2 + par) * (unsigned)(C * RT);
        if ((lane & 15) == 0 && (lane >> 4) < RT)
            __hip_atomic_store(xb_ + (size_t)c * RT + (lane >> 4),
                               ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(v),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool want = member < per_row && gr_ < RT;
        const unsigned long long* src = xb_ + (want ? (size_t)(QUAD ? G * member : (wide ? 2 * member : member)) * RT + gr_ : 0);
        unsigned long long gv = 0, gw = 0;
        // once an exchange of this launch has timed out (status raised, the host redoes the layer) the later ones give up
        // at their first unanswered poll: the launch runs to its end on whatever q comes out, nobody needs an exit path
        unsigned spins = gave_up ? p.spin_limit : 0u;
        // lanes that gather nothing count as arrived; the compare mask goes straight into the scalar unit
        const unsigned long long idle = ~__builtin_amdgcn_ballot_w64(want);
        // Pacing.  A poll that comes back without every granule costs a whole round trip AND sits in the way of the granules
        // still travelling, so the first poll waits -- the host says how long (low five bits of the spin limit, units of 256
        // clocks; gpfq_capi.hip first_poll_pause: it grows with the members of the tile, 0 for small tiles, where a pause
        // only costs: 1 x 16 granules 1.52 -> 1.79 us per column, 2 x 8 1.31 -> 1.47, 4 x 4 + 2-6 % over a layer).  The gap
        // between two polls makes no difference that survives a whole-workload run.
        // (The pause rides in the spin limit, which this loop keeps in a scalar register anyway: a value of its own, or a
        // test on C, is one more live scalar, and the four-row kernels then restore spilt SGPRs on this very path: + 2.6 %
        // on layer3.0.conv2.)
        // (ONE opaque statement: a C++ loop here is two more basic blocks in the middle of the step, and the register
        // allocator answers them with SGPR spills that are restored on this very path)
        {
            unsigned w = p.spin_limit & 31u;
            asm volatile("s_cmp_eq_u32 %0, 0\n\t"
                         "s_cbranch_scc1 2f\n"
                         "1:\n\t"
                         "s_sleep 4\n\t"
                         "s_sub_u32 %0, %0, 1\n\t"
                         "s_cmp_lg_u32 %0, 0\n\t"
                         "s_cbranch_scc1 1b\n"
                         "2:" : "+s"(w) :: "scc", "memory");
        }
        if constexpr (QUAD) {
            // (eight members per lane: two batches of four loads, each summed as soon as it has been checked -- sixteen
            // registers of granules in flight do not fit beside two residual rows and the column window)
            unsigned long long g[4];
            float v4 = 0.0f;
            for (;;) {
                unsigned long long ok = ~0ull;
                float vq[G / 4];                     // the lane's members in fours: the tree over them stays balanced
#pragma unroll
                for (int b = 0; b < G / 4; ++b) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        g[i] = __hip_atomic_load(src + (4 * b + i) * RT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (idle lanes: granules 0 .. G-1, in bounds: C >= G)
#pragma unroll
                    for (int i = 0; i < 4; ++i) ok &= __builtin_amdgcn_ballot_w64((unsigned)(g[i] >> 32) == epoch);
                    vq[b] = (__uint_as_float((unsigned)g[0]) + __uint_as_float((unsigned)g[1])) +
                            (__uint_as_float((unsigned)g[2]) + __uint_as_float((unsigned)g[3]));
                    if constexpr (G >= 8) __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (G == 4) v4 = vq[0];
                else if constexpr (G == 8) v4 = vq[0] + vq[1];
                else v4 = (vq[0] + vq[1]) + (vq[2] + vq[3]);
                if ((ok | idle) == __builtin_amdgcn_read_exec()) break;
                if ((spins += 256) > p.spin_limit) { timed_out = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            v = want ? v4 : 0.0f;
        } else if (!wide) {
            for (;;) {
                gv = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((__builtin_amdgcn_ballot_w64((unsigned)(gv >> 32) == epoch) | idle) == __builtin_amdgcn_read_exec()) break;
                if ((spins += 256) > p.spin_limit) { timed_out = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            v = want ? __uint_as_float((unsigned)gv) : 0.0f;
        } else {
            const unsigned long long* src1 = src + (want ? RT : 0);
            for (;;) {
                gv = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                gw = __hip_atomic_load(src1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (((__builtin_amdgcn_ballot_w64((unsigned)(gv >> 32) == epoch) &
                      __builtin_amdgcn_ballot_w64((unsigned)(gw >> 32) == epoch)) | idle) == __builtin_amdgcn_read_exec()) break;
                if ((spins += 256) > p.spin_limit) { timed_out = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            v = want ? __uint_as_float((unsigned)gv) + __uint_as_float((unsigned)gw) : 0.0f;
        }
        // upper levels of the slot tree over the members of each row
        v = wave_tree16_zero_padded(v);
        if (sh > 4) v = xor16_add(v);
        if (sh > 5) v = xor32_add(v);
    }
    const bool lead = member == 0 && gr_ < RT;
    const bool rvalid = lead && (row0 + gr_ < p.Ng);
    const int64_t growl = grow0 + (rvalid ? gr_ : 0);
    int id;
    float q;
    // MSQ: index and value straight from the dot product (gpfq_device.h quant_msq_from_dot), the divisions only on the
    // rare step whose quotient lies within (K + 4) * 2^-18 of a rounding boundary in some lane; used first, checked second (see
    // resident_body).  FAST is off in the four-row 12-wave variant: both paths do not fit the 56 registers it leaves
    // the compiler.
    bool redo = false;
    auto divide_and_quantize = [&]() {
        const float sarg = (n2cur > 0.0f) ? v / n2cur : 0.0f;
        q = quantize_mode<MODE>(p, sarg, p.row_id0 + (uint64_t)growl, (uint64_t)t, id);
    };
    if (FAST) redo = !quant_msq_from_dot(v, in2cur, p.inv_step, p.step, p.Kf, p.msq_thr, ~__builtin_amdgcn_ballot_w64(lead), q, id);
    else divide_and_quantize();
    // Q / idx leave through a 64-step history in LDS and one coalesced store per row every 64 steps: a
    // store per step would queue behind the sweep waves' column loads in the vector-memory pipe and hold up
    // the reducer's arrival at the second barrier.
    float* hist = qs + 2 * (RT + 1);                 // [RT][64] values, then [RT][64] indices (as int bits)
    auto commit = [&]() {
        if (lead) {
            qs[par * (RT + 1) + gr_] = q;
            hist[gr_ * 64 + (t & 63)] = q;
            hist[(RT + gr_) * 64 + (t & 63)] = __int_as_float(id);
        }
    };
    commit();
    if (FAST && __builtin_expect(redo, 0)) {
        divide_and_quantize();
        commit();
    }
    if (__builtin_expect((t & 63) == 63 || t + 1 == p.d, 0)) {
        const int t0 = t & ~63;
        const int n = t - t0 + 1;                    // steps in this history block
        // the history was written by the lead lanes, it is read by lanes 0..n-1: lanes are threads to the compiler, and
        // nothing but a fence orders one lane's store before another lane's load (the LDS itself executes a wave's
        // operations in order)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // The lane number is recomputed here, opaque to the compiler: everything derived from it below -- the per-lane
        // 64-bit store addresses of every row, for both index widths -- is then defined INSIDE this rare block.  Left
        // visible, the loop-invariant part of those addresses was hoisted out of the column loop and kept live across
        // it: 24 VGPRs, which the 64-register variants (coop_lds_body) had to spill to scratch.
        const int ln = fresh_lane_id();
        if (c == 0 && ln < n) {
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                if (row0 + r < p.Ng) {
                    const int64_t gw = grow0 + r;
                    p.Q[gw * p.ldq + t0 + ln] = hist[r * 64 + ln];
                    if (p.idx) {
                        const int iv = __float_as_int(hist[(RT + r) * 64 + ln]);
                        if (p.idx_bytes == 1) reinterpret_cast<int8_t*>(p.idx)[gw * p.ldi + t0 + ln] = (int8_t)iv;
                        else reinterpret_cast<int16_t*>(p.idx)[gw * p.ldi + t0 + ln] = (int16_t)iv;
                    }
                }
            }
        }
    }
    if constexpr (ABORTWORD) {
        if (lane == 0) qs[par * (RT + 1) + RT] = timed_out ? 1.0f : 0.0f;
    }
    if (__builtin_expect(timed_out && !gave_up, 0) && lane == 0) {
        atomicExch(p.status, 1);
        p.status[1] = t; p.status[2] = tile; p.status[3] = c;
    }
    return timed_out;
}

// Sum of squares of one canonical segment of the final residual (lane chain from +0.0f, then the lane tree): the
// in-kernel part of the error metrics ||U||_F and ||U^T||_col (step_algorithm.py:216-219, :239-243), so that no pass
// over U is needed afterwards.  Wave-uniform call (EXEC full); lane 63 stores.
__device__ __forceinline__ void store_segment_sumsq(float* usq, int64_t slot, const float (&u)[16], int lane)
{
    float acc = 0.0f;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc = __builtin_fmaf(u[e], u[e], acc);
    const float sg = wave_tree64_lane63(acc);
    if (lane == 63) usq[slot] = sg;
}

// Tail of the window kernels for one row: the pending subtraction of the last step (in place, in the window), then the
// residual leaves the registers four elements at a time (step_algorithm.py:148), with the fused sum of squares.
// STRIDE 2: the row is one of an interleaved pair (win_sweep16_pair): its registers are U, U + 2, U + 4, ...
// SUB false: the caller has done the pending subtraction itself (its x_{d-1} is not in the window).
template <int U, int XL, int STRIDE = 1, bool SUB = true>
__device__ __forceinline__ void finish_row_w(const SlabParams& p, float qlast, bool valid, int64_t grow, int seg, int lane)
{
    if constexpr (SUB) {
        if constexpr (STRIDE == 1) win_final_sub16<U, XL>(qlast);
        else win_final_sub16_s2<U, XL>(qlast);
    }
    if (!valid) return;
    // computed here, from a lane number the compiler cannot see through: nothing 64-bit and per-lane is hoisted above
    // the column loop and kept live across it (see the Q / idx flush in reducer_section)
    lane = fresh_lane_id();
    const int64_t kbase = (int64_t)seg * kSeg + 4 * lane;
    float* Urow = p.U + grow * p.ldu;
    float acc = 0.0f;
    auto chunk = [&](auto c_) {
        constexpr int c = decltype(c_)::value;
        float v[4];
        if constexpr (STRIDE == 1) win_read4<U + 4 * c>(v);
        else win_read4_s2<U + 8 * c>(v);
        const int64_t k0 = kbase + 256 * c;
        if (p.vec && k0 + 3 < p.m) {
            *reinterpret_cast<float4*>(Urow + k0) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (k0 + j < p.m) Urow[k0 + j] = v[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_fmaf(v[j], v[j], acc);       // element order e = 4c + j, as store_segment_sumsq
    };
    chunk(std::integral_constant<int, 0>{});
    chunk(std::integral_constant<int, 1>{});
    chunk(std::integral_constant<int, 2>{});
    chunk(std::integral_constant<int, 3>{});
    if (p.usq) {
        const float sg = wave_tree64_lane63(acc);
        if (lane == 63) p.usq[grow * p.S + seg] = sg;
    }
}

// Cooperative plan (rows split by columns over C workgroups): the column buffers and the RT residual rows of every
// sweep wave live in the column window (gpfq_device.h), as in the resident kernel below.
// DEPTH = look-ahead of the column loads in steps.
//   DEPTH 2: three x buffers and two a buffers rotate by name in a six-fold unrolled loop (x_t in X[t % 3], a_t in
//     A[t % 2]); column t+2 is requested into the registers sweep t has finished with, three quarters behind the lane
//     tree and the last one behind barrier 2, and sweep t+1 waits for column t+1 only (vmcnt(8)).  With 12 waves a
//     workgroup pulls 96 KB of columns per step through the CU's vector-memory pipe, which takes longer than the rest
//     of the step unless it never pauses -- and the granule store and the polls of the exchange travel through the
//     same per-CU queue and wait for every request ahead of them.  Measured on the ResNet-50 shapes: all eight
//     requests behind barrier 2 (the queue is empty for the exchange, but every wave then stalls issuing into a full
//     queue before its next sweep) 2.45 / 1.84 / 2.33 us per column, this split 2.30 / 1.74 / 2.28, all of them behind
//     the lane tree 2.61 / 1.75 / 2.27.
//   DEPTH 1 (four rows at 12 waves: 168 registers hold the four rows and three column buffers, not five): two x
//     buffers alternate, one a buffer; column t+1 is requested into the registers sweep t has finished with -- it has
//     the whole exchange to land -- and sweep t+1 waits for all of it (vmcnt(0)).
// WB = first register of the window: x buffers, a buffers, then the RT residual rows.
// GROUPED: every row is a group of its own (depthwise convolutions: groups == out channels, one row per group) with its
// OWN d columns: row r reads columns [r * d, (r + 1) * d) and their norms.  The host presents such a layer as Ng = groups
// rows of one group (run_loop); everything else -- exchange, rounds, Philox keys by global row -- is the plain kernel.
template <int RT, int MODE, int DEPTH, int WB, bool QUAD = false, bool GROUPED = false>
__device__ __forceinline__ void coop_body(const SlabParams& p)
{
    constexpr int NX = DEPTH == 2 ? 3 : 2, NA = DEPTH == 2 ? 2 : 1;
    constexpr int X0 = WB, X1 = WB + 16, X2 = WB + 32, A0 = WB + 16 * NX, A1 = A0 + 16, U0 = WB + 16 * (NX + NA);
    extern __shared__ float smem[];                 // seg[2][RT][NW], then qs[2][RT + 1], then the Q / idx history
    const int NW = blockDim.x >> 6;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int S = p.S, C = p.C;
    const int P = pow2_ceil(S);
    // A layer that runs in rounds (one launch per block of rows) stops at the first launch whose exchange timed out: the
    // status word stays raised until the host has read it, and the layer is redone on another plan anyway.
    if (*static_cast<volatile const int*>(p.status) != 0) return;
    int tile, c;
    // keep the members of one row tile on one XCD when the tile count allows it (blocks b and b+8 share an
    // XCD under round-robin dispatch; speed only, never correctness)
    if (p.xcd_tiles && (p.tiles & 7) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        tile = (j / C) * 8 + xcd;
        c = j % C;
    } else {
        tile = blockIdx.x / C;
        c = blockIdx.x % C;
    }
    const int seg_lo = (c * S + C - 1) / C, seg_hi = ((c + 1) * S + C - 1) / C;
    const int n_own = seg_hi - seg_lo;              // <= max_own
    const int max_own = (S + C - 1) / C;            // sweep waves of the fullest member
    // The reducer (slot tree, exchange, quantizer) is a wave of its own when the launch has one to spare
    // (NW == max_own + 1); otherwise wave 0 doubles as the reducer.
    const int rwave = NW > max_own ? max_own : 0;
    const bool active = wave < n_own;
    const int myseg = seg_lo + (active ? wave : 0);
    const int nl = P / C;                           // slots of this workgroup's block, <= 16 (at most 12 segments per member)
    const SlotMap smap = make_slot_map(S, P, c * nl, 1, lane & 15, nl);   // reducer: lane = 16 * row + slot

    float* segs = smem;                             // [2][RT][NW]
    float* qs = smem + 2 * RT * NW;                 // [2][RT + 1] (last = abort flag), then the history [2*RT][64]

    const int row0 = tile * RT;
    const int64_t grow0 = row0;                     // groups == 1
    static_assert(!GROUPED || RT == 1, "one row per group");
    const int64_t gcol = GROUPED ? (int64_t)tile * p.d : 0;             // first column of this row's group
    const float* xload = uniform_ptr(p.XT + gcol * p.m_pad + (int64_t)myseg * kSeg);      // wave-uniform column pointers
    const float* aload = uniform_ptr(p.AT + gcol * p.m_pad + (int64_t)myseg * kSeg);
    const unsigned lane_off = 16u * (unsigned)lane;
    const kfloat* nrm = as_scalar(p.nrm2 + 2 * gcol);

    const kfloat* wrow[RT];
    float qprev[RT], wcur[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const int64_t gr = grow0 + ((row0 + r < p.Ng) ? r : (p.Ng - 1 - row0));
        wrow[r] = as_scalar(p.W + gr * p.ldw);
        qprev[r] = 0.0f;
        wcur[r] = wrow[r][0];
    }
    float n2cur = nrm[0];
    float in2cur = nrm[1];                          // fl(1 / ||x_t||^2) for quant_msq_from_dot

    // the residual starts at 0 (a non-zero initial residual is the streaming plan's job); every buffer starts defined
    win_zero16<U0>();
    if constexpr (RT >= 2) win_zero16<U0 + 16>();
    if constexpr (RT >= 4) { win_zero16<U0 + 32>(); win_zero16<U0 + 48>(); }
    win_zero16<X0>(); win_zero16<X1>(); win_zero16<A0>();
    if constexpr (DEPTH == 2) { win_zero16<X2>(); win_zero16<A1>(); }
    if constexpr (DEPTH == 2) {
        // x_0 -> X0, a_0 -> A0, x_1 -> X1, a_1 -> A1 (a one-column layer re-reads column 0); X2 = x_{-1} = 0
        if (active) {
            win_load16<X0>(xload, lane_off);
            win_load16<A0>(aload, lane_off);
        }
        const int64_t adv = (1 < p.d) ? p.m_pad : 0;
        xload += adv;
        aload += adv;
        if (active) {
            win_load16<X1>(xload, lane_off);
            win_load16<A1>(aload, lane_off);
        }
    } else {
        // x_0 -> X0, a_0 -> A0; X1 = x_{-1} = 0
        if (active) {
            win_load16<X0>(xload, lane_off);
            win_load16<A0>(aload, lane_off);
        }
    }

#ifdef GPFQ_STAMPS
    // diagnostic build only: cycles per phase, summed by the reducer wave and one sweep wave of block 0 into status[16..]
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
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
#else
#define GPFQ_STAMP(i)
#endif
    int t = 0;
    constexpr bool ABORTWORD = (RT == 4 && DEPTH == 1);
    bool gave_up = false;                           // reducer wave: an exchange of this launch has timed out
    bool dead = false;                              // ABORTWORD only
    // Column requests in the exchange window (below) only when the reducer is a wave of its own: where wave 0 doubles as
    // the reducer (12 sweep waves) the window requests of the other eleven waves sit in front of its polls and the
    // exchange gets slower than the sweep phase gets faster (measured per column: 2.17 -> 2.4-2.5 us; with a dedicated
    // reducer 1.65 -> 1.60 and 2.20 -> 2.12).
    const bool trickle = p.pace > 0 && NW > max_own;
    // one step; XP holds x_{t-1}, XC x_t, AC a_t.  Returns false after the last column or on a timeout.
    auto step = [&](auto xp_, auto xc_, auto ac_) -> bool {
        constexpr int XP = decltype(xp_)::value, XC = decltype(xc_)::value, AC = decltype(ac_)::value;
        GPFQ_STAMP(0)
        const int par = t & 1;
        const bool more = t + 1 < p.d;
        float* seg = segs + par * RT * NW;
        // the pointers advance in every wave (uniform values must not change under a per-wave condition, or they
        // stop being scalar); the last steps re-read the last column rather than branch
        {
            const int64_t adv = (t + DEPTH < p.d) ? p.m_pad : 0;
            xload += adv;
            aload += adv;
        }
        if (active) {
            win_wait<DEPTH == 2 ? 8 : 0>();         // column t has landed (DEPTH 2: the loads of column t+1 stay in flight)
            float acc[RT];
            if constexpr (RT == 1) {
                acc[0] = win_sweep16<U0, XP, AC, XC>(qprev[0], wcur[0]);
            } else {                                // rows in interleaved pairs: 80 instructions per pair instead of 96
                const v2f a01 = win_sweep16_pair<U0, XP, AC, XC>(qprev[0], qprev[1], wcur[0], wcur[1]);
                acc[0] = a01.x; acc[1] = a01.y;
                if constexpr (RT >= 4) {
                    const v2f a23 = win_sweep16_pair<U0 + 32, XP, AC, XC>(qprev[2], qprev[3], wcur[2], wcur[3]);
                    acc[2] = a23.x; acc[3] = a23.y;
                }
            }
            GPFQ_STAMP(1)
            if constexpr (RT == 1) {
                const float sg = wave_tree64_lane63(acc[0]);
                if (lane == 63) seg[wave] = sg;
            } else {                                // row r's total in lane row r: one LDS write for all rows
                const float tot = wave_tree64_rows<RT>(acc);
                if ((lane & 15) == 0 && (lane >> 4) < RT) seg[(lane >> 4) * NW + wave] = tot;
            }
            // The next column wanted goes into the registers the sweeps have just finished with (x_{t-1}'s and a_t's).
            // A wave that is also the reducer requests three quarters here and the last one behind barrier 2, as in
            // round 1 (its exchange comes next: the queue must be short by then).  Every other sweep wave waits for the
            // exchange window below.
            if (!trickle || wave == rwave) {
                win_load4<XP, 0>(xload, lane_off);
                win_load4<AC, 0>(aload, lane_off);
                win_load4<XP, 1>(xload, lane_off);
                win_load4<AC, 1>(aload, lane_off);
                win_load4<XP, 2>(xload, lane_off);
                win_load4<AC, 2>(aload, lane_off);
            }
            GPFQ_STAMP(2)
        }
        __syncthreads();
        GPFQ_STAMP(3)
        // next column's weights and norm through the scalar cache (the last step re-reads its own, unused, ones rather
        // than branch), requested here: at the top of the step their round trip sat in front of the sweep (scalar loads
        // and LDS share lgkmcnt, any wait on it waits for all of them); behind barrier 1 the sweep waves have nothing to do
        float wn[RT];
        const int tn = more ? t + 1 : t;
#pragma unroll
        for (int r = 0; r < RT; ++r) wn[r] = sload(wrow[r], 4u * (unsigned)tn);
        const float n2n = sload(nrm, 8u * (unsigned)tn), in2n = sload(nrm, 8u * (unsigned)tn + 4u);
        if (wave == rwave) {
            GPFQ_STAMP(4)
            if constexpr (ABORTWORD)
                reducer_section<RT, MODE, false, true>(p, seg, qs, smap, NW, nl, lane, tile, c, C, par, t, n2cur, in2cur, row0,
                                                       grow0, seg_lo, false);
            else
                gave_up |= reducer_section<RT, MODE, MODE == MODE_MSQ, false, QUAD>(p, seg, qs, smap, NW, nl, lane, tile, c, C, par, t,
                                                                                    n2cur, in2cur, row0, grow0, seg_lo, gave_up);
            GPFQ_STAMP(5)
        } else if (trickle && active) {
            // The exchange window: the sweep waves idle here for ~2 000 cycles while the reducer's granules travel.  Their
            // eight column requests go out NOW, one at a time with a pause between: issued together behind the lane tree
            // (round 1) they filled the CU's in-order vector-memory queue exactly when every wave wanted to reach barrier
            // 1 -- 96 requests of 16 clocks each at 12 waves, the waves stalling at issue -- and issued together here they
            // would sit in front of the reducer's granule store and polls.  Paced, the queue stays a few entries deep.
            __builtin_amdgcn_s_sleep(2);                                 // the reducer's granule store goes first
            auto pause = [&]() { for (int i = 0; i < p.pace; ++i) __builtin_amdgcn_s_sleep(1); };
            win_load4<XP, 0>(xload, lane_off); pause();
            win_load4<AC, 0>(aload, lane_off); pause();
            win_load4<XP, 1>(xload, lane_off); pause();
            win_load4<AC, 1>(aload, lane_off); pause();
            win_load4<XP, 2>(xload, lane_off); pause();
            win_load4<AC, 2>(aload, lane_off); pause();
            win_load4<XP, 3>(xload, lane_off); pause();
            win_load4<AC, 3>(aload, lane_off);
        }
        GPFQ_STAMP(6)
        __syncthreads();
        GPFQ_STAMP(7)
        if (active && (!trickle || wave == rwave)) {
            win_load4<XP, 3>(xload, lane_off);
            win_load4<AC, 3>(aload, lane_off);
        }
#pragma unroll
        for (int r = 0; r < RT; ++r) qprev[r] = qs[par * (RT + 1) + r];
        if constexpr (ABORTWORD) {
            if (qs[par * (RT + 1) + RT] != 0.0f) { dead = true; return false; }   // an exchange timed out: status word is set
        }
        if (!more) return false;
#pragma unroll
        for (int r = 0; r < RT; ++r) wcur[r] = wn[r];
        n2cur = n2n;
        in2cur = in2n;
        ++t;
        return true;
    };
    using I0 = std::integral_constant<int, X0>; using I1 = std::integral_constant<int, X1>; using I2 = std::integral_constant<int, X2>;
    using J0 = std::integral_constant<int, A0>; using J1 = std::integral_constant<int, A1>;
    int k = 0;                                       // buffer that holds x_t when the loop ends
    if constexpr (DEPTH == 2) {
        for (;;) {
            k = 0; if (!step(I2{}, I0{}, J0{})) break;
            k = 1; if (!step(I0{}, I1{}, J1{})) break;
            k = 2; if (!step(I1{}, I2{}, J0{})) break;
            k = 0; if (!step(I2{}, I0{}, J1{})) break;
            k = 1; if (!step(I0{}, I1{}, J0{})) break;
            k = 2; if (!step(I1{}, I2{}, J1{})) break;
        }
    } else {
        for (;;) {
            k = 0; if (!step(I1{}, I0{}, J0{})) break;
            k = 1; if (!step(I0{}, I1{}, J0{})) break;
        }
    }
    // every load issued above has landed before the last column is used again (and before the wave ends)
    win_wait<0>();
#ifdef GPFQ_STAMPS
    if (blockIdx.x == 0 && lane == 0 && (wave == rwave || wave == (rwave == 0 ? NW - 1 : 0)) && p.status) {
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(p.status + 16) + (wave == rwave ? 0 : 8);
        for (int i = 0; i < 8; ++i) dbg[i] = stamp_sum[i];
    }
#endif
    if (dead || !active) return;
    auto finish = [&](auto xl_) {
        constexpr int XL = decltype(xl_)::value;
        if constexpr (RT == 1) {
            finish_row_w<U0, XL>(p, qprev[0], row0 < p.Ng, grow0, myseg, lane);
        } else {                                    // interleaved pairs: row r sits at U0 + 32 (r / 2) + (r % 2), stride 2
            finish_row_w<U0, XL, 2>(p, qprev[0], row0 < p.Ng, grow0, myseg, lane);
            finish_row_w<U0 + 1, XL, 2>(p, qprev[1], row0 + 1 < p.Ng, grow0 + 1, myseg, lane);
            if constexpr (RT >= 4) {
                finish_row_w<U0 + 32, XL, 2>(p, qprev[2], row0 + 2 < p.Ng, grow0 + 2, myseg, lane);
                finish_row_w<U0 + 33, XL, 2>(p, qprev[3], row0 + 3 < p.Ng, grow0 + 3, myseg, lane);
            }
        }
    };
    if (k == 0) finish(I0{});
    else if (k == 1) finish(I1{});
    else if constexpr (DEPTH == 2) finish(I2{});
}

// FOUR rows at 13 sweep waves: the variant for rows whose members need a 13th wave (m = 1024 (L/4 + 1), L = 14^2 4^k --
// every 1x1 convolution of ResNet-50 at batch 1024 -- is 12.27..12.5 segments per member at any power-of-two member count).
// The 128-register budget of 13..16 waves holds the four residual rows (64) and nothing like a column window, so the
// columns live in LDS: gfx950's global_load_lds_dwordx4 writes a wave's 1 KB quarters straight into its three 4 KB
// buffers (x ring of two, a), one step ahead (column t+1 is requested when sweep t has read its buffers for the last
// time and has the whole exchange to land), and the sweep reads them back 16 bytes per lane per quarter (conflict-free).
// 13 x 12 KB = 156 of the CU's 160 KB.  Same arithmetic, same order: the interleaved pair sweep, a quarter at a time.
template <int MODE, bool QUAD, int GOVR = 0>
__device__ __forceinline__ void coop_lds_body(const SlabParams& p)
{
    constexpr int RT = 4, U0 = 64;                  // window = the four residual rows only (two interleaved pairs)
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [sweep waves][3][1024], seg[2][RT][NW], qs[2][RT+1], history
    const int NW = blockDim.x >> 6;
    // the wave number as a SCALAR: everything derived from it (segment, LDS buffers, "am I a sweep wave") then lives in
    // SGPRs -- the compiler has 64 VGPRs here, and as vector values the segment number and its 64-bit multiples were
    // kept live across the column loop for the epilogue, in scratch
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int S = p.S, C = p.C;
    const int P = pow2_ceil(S);
    if (*static_cast<volatile const int*>(p.status) != 0) return;       // (a layer in rounds stops at the first timed-out launch)
    int tile, c;
    if (p.xcd_tiles && (p.tiles & 7) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        tile = (j / C) * 8 + xcd;
        c = j % C;
    } else {
        tile = blockIdx.x / C;
        c = blockIdx.x % C;
    }
    const int seg_lo = (c * S + C - 1) / C, seg_hi = ((c + 1) * S + C - 1) / C;
    const int n_own = seg_hi - seg_lo;
    const int max_own = (S + C - 1) / C;
    const int rwave = NW > max_own ? max_own : 0;
    const bool active = wave < n_own;
    const int myseg = seg_lo + (active ? wave : 0);
    const int nl = P / C;
    const SlotMap smap = make_slot_map(S, P, c * nl, 1, lane & 15, nl);

    float* cols = smem + (size_t)(wave < max_own ? wave : 0) * 3072;    // this wave's x ring (2 x 1024) and a buffer (1024)
    float* segs = smem + (size_t)max_own * 3072;    // [2][RT][NW]
    float* qs = segs + 2 * RT * NW;                 // [2][RT + 1], then the history [2*RT][64]

    const int row0 = tile * RT;
    const int64_t grow0 = row0;
    const kfloat* nrm = as_scalar(p.nrm2);
    const kfloat* wrow[RT];
    float qprev[RT], wcur[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const int64_t gr = grow0 + ((row0 + r < p.Ng) ? r : (p.Ng - 1 - row0));
        wrow[r] = as_scalar(p.W + gr * p.ldw);
        qprev[r] = 0.0f;
        wcur[r] = wrow[r][0];
    }
    float n2cur = nrm[0];
    float in2cur = nrm[1];
    win_zero16<U0>(); win_zero16<U0 + 16>(); win_zero16<U0 + 32>(); win_zero16<U0 + 48>();

    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* glb_ptr_t;
    // one column segment (1024 floats) of this wave into one LDS buffer: lane l brings elements 256 q + 4 l .. + 3 of quarter q
    // (the global address is a wave-uniform base plus a 32-bit lane offset: one VGPR, where two per-lane 64-bit pointers
    // advanced every step were four)
    const unsigned lane_off = 16u * (unsigned)lane;
    auto dma = [&](float* buf, const float* gbase) {
        lds_ptr_t l = (lds_ptr_t)(uintptr_t)buf;
        glb_ptr_t g = (glb_ptr_t)(reinterpret_cast<const char*>(gbase) + lane_off);
        __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(g, l, 16, 1024, 0);
        __builtin_amdgcn_global_load_lds(g, l, 16, 2048, 0);
        __builtin_amdgcn_global_load_lds(g, l, 16, 3072, 0);
    };
    const float* xg = p.XT + (int64_t)myseg * kSeg;          // wave-uniform (the wave number is a scalar)
    const float* ag = p.AT + (int64_t)myseg * kSeg;
    if (active) {
        // x_0 -> ring[0], a_0 -> a; ring[1] = x_{-1} = 0 (q_{-1} = 0)
        const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) *reinterpret_cast<float4*>(cols + 1024 + 256 * q4 + 4 * lane) = z;
        dma(cols, xg);
        dma(cols + 2048, ag);
    }
    int t = 0;
    bool gave_up = false;
    for (;;) {
        const int par = t & 1;
        const bool more = t + 1 < p.d;
        float* seg = segs + par * RT * NW;
        if (active) {
            const float* xc = cols + par * 1024;         // x_t
            const float* xp = cols + (par ^ 1) * 1024;   // x_{t-1}
            const float* ab = cols + 2048;               // a_t
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // column t has landed in LDS
            v2f acc01 = {0.0f, 0.0f}, acc23 = {0.0f, 0.0f};
            const v2f qq01 = {qprev[0], qprev[1]}, ww01 = {wcur[0], wcur[1]};
            const v2f qq23 = {qprev[2], qprev[3]}, ww23 = {wcur[2], wcur[3]};
            // the LDS reads of quarter q + 1 are issued before the arithmetic of quarter q (two quarters' 24 registers of
            // columns at a time; the budget is 64)
            float4 x4 = *reinterpret_cast<const float4*>(xc + 4 * lane);
            float4 a4 = *reinterpret_cast<const float4*>(ab + 4 * lane);
            float4 p4 = *reinterpret_cast<const float4*>(xp + 4 * lane);
            auto quarter = [&](auto c_) {
                constexpr int cq = decltype(c_)::value;
                float4 nx = x4, na = a4, np = p4;
                if constexpr (cq < 3) {
                    nx = *reinterpret_cast<const float4*>(xc + 256 * (cq + 1) + 4 * lane);
                    na = *reinterpret_cast<const float4*>(ab + 256 * (cq + 1) + 4 * lane);
                    np = *reinterpret_cast<const float4*>(xp + 256 * (cq + 1) + 4 * lane);
                }
                __builtin_amdgcn_sched_barrier(0);
                const v2f x01 = {x4.x, x4.y}, x23 = {x4.z, x4.w}, a01 = {a4.x, a4.y}, a23 = {a4.z, a4.w};
                const v2f p01 = {p4.x, p4.y}, p23 = {p4.z, p4.w};
                win_sweep4_pair_lds<U0 + 8 * cq>(acc01, qq01, ww01, p01, p23, a01, a23, x01, x23);
                win_sweep4_pair_lds<U0 + 32 + 8 * cq>(acc23, qq23, ww23, p01, p23, a01, a23, x01, x23);
                __builtin_amdgcn_sched_barrier(0);
                x4 = nx; a4 = na; p4 = np;
            };
            quarter(std::integral_constant<int, 0>{});
            quarter(std::integral_constant<int, 1>{});
            quarter(std::integral_constant<int, 2>{});
            quarter(std::integral_constant<int, 3>{});
            const float acc[RT] = {acc01.x, acc01.y, acc23.x, acc23.y};
            const float tot = wave_tree64_rows<RT>(acc);
            if ((lane & 15) == 0 && (lane >> 4) < RT) seg[(lane >> 4) * NW + wave] = tot;
            // column t+1 into the buffers this sweep has read for the last time (the last step re-reads its own)
            const int64_t adv = more ? p.m_pad : 0;
            xg += adv;
            ag += adv;
            dma(cols + (par ^ 1) * 1024, xg);
            dma(cols + 2048, ag);
        }
        // Not __syncthreads(): its release fence waits for the LDS-DMA just issued (s_waitcnt vmcnt(0): the DMA writes
        // LDS), and the columns would land BEFORE the exchange instead of under it -- 785-segment rows 80.2 -> 71.8 us per
        // column, 197 segments 52.9 -> 50.2 (in-kernel stamps: the last sweep wave reaches the barrier 4 700 cycles into
        // the step and then waited 1 900 more for its own columns).  The buffers they land in are this wave's own, and
        // the wave waits for them itself at the top of its next sweep; what the barrier has to order is the LDS word
        // written above (and the reducer's q below): lgkmcnt.  (Issuing the DMA behind the barrier instead, so that the
        // reducer's granule goes first: 91.7 and 59.8.)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        float wn[RT];
        const int tn = more ? t + 1 : t;
#pragma unroll
        for (int r = 0; r < RT; ++r) wn[r] = sload(wrow[r], 4u * (unsigned)tn);
        const float n2n = sload(nrm, 8u * (unsigned)tn), in2n = sload(nrm, 8u * (unsigned)tn + 4u);
        if (wave == rwave)
            gave_up |= reducer_section<RT, MODE, MODE == MODE_MSQ, false, QUAD, GOVR>(p, seg, qs, smap, NW, nl, lane, tile, c, C, par, t,
                                                                                      n2cur, in2cur, row0, grow0, seg_lo, gave_up);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int r = 0; r < RT; ++r) qprev[r] = qs[par * (RT + 1) + r];
        if (!more) break;
#pragma unroll
        for (int r = 0; r < RT; ++r) wcur[r] = wn[r];
        n2cur = n2n;
        in2cur = in2n;
        ++t;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!active) return;
    {   // the pending subtraction of the last step: u -= q_{d-1} x_{d-1}, x_{d-1} from its ring buffer
        const float* xl = cols + (t & 1) * 1024;
        const v2f qq01 = {qprev[0], qprev[1]}, qq23 = {qprev[2], qprev[3]};
        auto quarter = [&](auto c_) {
            constexpr int cq = decltype(c_)::value;
            const float4 x4 = *reinterpret_cast<const float4*>(xl + 256 * cq + 4 * lane);
            const v2f x01 = {x4.x, x4.y}, x23 = {x4.z, x4.w};
            win_final_sub4_pair_lds<U0 + 8 * cq>(qq01, x01, x23);
            win_final_sub4_pair_lds<U0 + 32 + 8 * cq>(qq23, x01, x23);
        };
        quarter(std::integral_constant<int, 0>{});
        quarter(std::integral_constant<int, 1>{});
        quarter(std::integral_constant<int, 2>{});
        quarter(std::integral_constant<int, 3>{});
    }
    finish_row_w<U0, 0, 2, false>(p, qprev[0], row0 < p.Ng, grow0, myseg, lane);
    finish_row_w<U0 + 1, 0, 2, false>(p, qprev[1], row0 + 1 < p.Ng, grow0 + 1, myseg, lane);
    finish_row_w<U0 + 32, 0, 2, false>(p, qprev[2], row0 + 2 < p.Ng, grow0 + 2, myseg, lane);
    finish_row_w<U0 + 33, 0, 2, false>(p, qprev[3], row0 + 3 < p.Ng, grow0 + 3, myseg, lane);
}

#define GPFQ_DEFINE_COOP_LDS(MODE, QUADV, GOVRV, SUFFIX)                                                          \
    __global__ void __launch_bounds__(64 * 16) __attribute__((amdgpu_num_vgpr(64 / 2)))                            \
    gpfq_coop_rt4_m##MODE##_w16##SUFFIX(const SlabParams p)                                                       \
    {                                                                                                             \
        asm volatile("" ::: "v127");                                                                              \
        coop_lds_body<MODE, QUADV, GOVRV>(p);                                                                     \
    }
GPFQ_DEFINE_COOP_LDS(0, false, 0, l) GPFQ_DEFINE_COOP_LDS(1, false, 0, l) GPFQ_DEFINE_COOP_LDS(2, false, 0, l) GPFQ_DEFINE_COOP_LDS(3, false, 0, l)
GPFQ_DEFINE_COOP_LDS(0, true, 0, lq) GPFQ_DEFINE_COOP_LDS(1, true, 0, lq) GPFQ_DEFINE_COOP_LDS(2, true, 0, lq) GPFQ_DEFINE_COOP_LDS(3, true, 0, lq)
// four rows on 256 members (1024 granules, sixteen gathered per lane): rows so long that one row tile takes the whole chip
// (EfficientNet-B1's 112 x 112 maps at batch 1024: 3137 segments) pull every column pair from HBM once per ROW TILE --
// four rows per pass halve that traffic against the two-row 256-member kernel (gpfq_coop_rt2_*_w16o)
GPFQ_DEFINE_COOP_LDS(0, true, 16, lh) GPFQ_DEFINE_COOP_LDS(1, true, 16, lh) GPFQ_DEFINE_COOP_LDS(2, true, 16, lh) GPFQ_DEFINE_COOP_LDS(3, true, 16, lh)

// One __global__ per (rows per workgroup, quantizer, wave bound): see GPFQ_DEFINE_RESIDENT below for the attribute.
#define GPFQ_DEFINE_COOP(RT, MODE, MAXW, DEPTH, WB, LAST)                                                         \
    __global__ void __launch_bounds__(64 * MAXW) __attribute__((amdgpu_num_vgpr(WB / 2)))                          \
    gpfq_coop_rt##RT##_m##MODE##_w##MAXW(const SlabParams p)                                                     \
    {                                                                                                             \
        asm volatile("" ::: LAST);      /* makes the kernel descriptor allocate the whole budget */               \
        coop_body<RT, MODE, DEPTH, WB>(p);                                                                        \
    }
#define GPFQ_DEFINE_COOP_MODES(RT, MAXW, DEPTH, WB, LAST)                                                         \
    GPFQ_DEFINE_COOP(RT, 0, MAXW, DEPTH, WB, LAST) GPFQ_DEFINE_COOP(RT, 1, MAXW, DEPTH, WB, LAST)                 \
    GPFQ_DEFINE_COOP(RT, 2, MAXW, DEPTH, WB, LAST) GPFQ_DEFINE_COOP(RT, 3, MAXW, DEPTH, WB, LAST)
GPFQ_DEFINE_COOP_MODES(1, 12, 2, 72, "v167")       // 168 - 80 - 16
GPFQ_DEFINE_COOP_MODES(2, 8, 2, 144, "v255")       // 256 - 80 - 32
GPFQ_DEFINE_COOP_MODES(2, 12, 2, 56, "v167")       // 168 - 80 - 32
GPFQ_DEFINE_COOP_MODES(4, 8, 2, 112, "v255")       // 256 - 80 - 64
// 168 - 48 - 64: three column buffers, one step of look-ahead, 56 registers for the compiler (the stochastic variant fits
// them since the Q / idx flush and the epilogue recompute their lane number -- fresh_lane_id -- instead of keeping
// per-lane store addresses live across the loop).
GPFQ_DEFINE_COOP(4, 0, 12, 1, 56, "v167") GPFQ_DEFINE_COOP(4, 1, 12, 1, 56, "v167") GPFQ_DEFINE_COOP(4, 2, 12, 1, 56, "v167")
GPFQ_DEFINE_COOP(4, 3, 12, 1, 56, "v167")
// 128 - 48 - 32: two rows at 13..16 waves, for rows whose members would otherwise need a 13th sweep wave (P / C = 16 slots
// per member and S / C just above 12: m = 803 840 is 785 segments, 12.3 per member at C = 64), one step of look-ahead.
GPFQ_DEFINE_COOP_MODES(2, 16, 1, 48, "v127")
// 128 - 48 - 16: ONE row at 13..16 waves, four gathered members per lane (up to 256 members: a whole chip for one row of
// up to 4096 segments -- EfficientNet-B1's 112 x 112 maps at batch 1024 are 3137), one step of look-ahead.
#define GPFQ_DEFINE_COOP_QUAD(MODE)                                                                               \
    __global__ void __launch_bounds__(64 * 16) __attribute__((amdgpu_num_vgpr(64 / 2)))                            \
    gpfq_coop_rt1_m##MODE##_w16(const SlabParams p)                                                               \
    {                                                                                                             \
        asm volatile("" ::: "v127");                                                                              \
        coop_body<1, MODE, 1, 64, true>(p);                                                                       \
    }
GPFQ_DEFINE_COOP_QUAD(0) GPFQ_DEFINE_COOP_QUAD(1) GPFQ_DEFINE_COOP_QUAD(2) GPFQ_DEFINE_COOP_QUAD(3)
// two rows on up to 256 members (512 granules, eight gathered per lane): rows so long that one row alone takes the whole
// chip (EfficientNet-B1's 112 x 112 maps at batch 1024, 3137 segments) pull every column from HBM once per ROW TILE, and
// that traffic, not the exchange, is what their step waits for
#define GPFQ_DEFINE_COOP_OCT(MODE)                                                                                \
    __global__ void __launch_bounds__(64 * 16) __attribute__((amdgpu_num_vgpr(48 / 2)))                            \
    gpfq_coop_rt2_m##MODE##_w16o(const SlabParams p)                                                              \
    {                                                                                                             \
        asm volatile("" ::: "v127");                                                                              \
        coop_body<2, MODE, 1, 48, true>(p);                                                                       \
    }
GPFQ_DEFINE_COOP_OCT(0) GPFQ_DEFINE_COOP_OCT(1) GPFQ_DEFINE_COOP_OCT(2) GPFQ_DEFINE_COOP_OCT(3)
// one row per group (depthwise convolutions with long rows), 12 waves: the one-row variant with per-row columns
#define GPFQ_DEFINE_COOP_GROUPED(MODE)                                                                            \
    __global__ void __launch_bounds__(64 * 12) __attribute__((amdgpu_num_vgpr(72 / 2)))                            \
    gpfq_coop_rt1g_m##MODE##_w12(const SlabParams p)                                                              \
    {                                                                                                             \
        asm volatile("" ::: "v167");                                                                              \
        coop_body<1, MODE, 2, 72, false, true>(p);                                                                \
    }
GPFQ_DEFINE_COOP_GROUPED(0) GPFQ_DEFINE_COOP_GROUPED(1) GPFQ_DEFINE_COOP_GROUPED(2) GPFQ_DEFINE_COOP_GROUPED(3)

// ------------------------------------------------------------------------------------------------
// Resident plan (whole rows in one workgroup, S <= 16 segments, one wave per segment): NO reducer role and ONE
// barrier per step.  Every wave leaves its segment sums in LDS, and behind the barrier every wave finishes the slot
// tree, divides and quantizes for itself (the same bits in every wave), so q never travels through LDS and there
// is no second barrier.
//
// RT rows per workgroup share the column registers.  What bounds this kernel is the CU's vector-memory pipe: a wave's
// 16-byte-per-lane load moves 1 KB in ~16 clocks, so the 8 KB a segment needs per step (x_t and a_t) cost ~53 ns of
// the pipe whatever else happens (measured: 0.42 / 0.65 / 1.10 us per step at 3 / 7 / 12 segments, and exactly twice
// that with two one-row workgroups on a CU; 22 TB/s of L2 -> CU column traffic chip-wide on every shape).  Two rows
// in one workgroup pull each column ONCE: half the pipe time per row.
//
// The five column buffers live in the column window (gpfq_device.h: physical registers reserved from the compiler),
// x_t in X[t % 3], a_t in A[t % 2], a six-fold unrolled loop naming them; the loads run two steps ahead, two requests
// at a time at four points of the step (eight at once from every wave fill the CU's vector-memory queue, and a wave
// whose request is not accepted stalls right there, on the critical path).  The step is a dependent chain of ~1000
// cycles, so it is also kept free of taken branches (each ~20 cycles): no level tests in the slot tree, no "is there
// a next column" test, the rare Q / idx flush out of line.
// ------------------------------------------------------------------------------------------------
// WB = first register of the window = the kernel's register budget minus 80 + 16 RT (see GPFQ_DEFINE_RESIDENT below):
// three x buffers, two a buffers, then the RT residual rows.
// ONE = rows of a single segment (m <= 1024: every fully connected layer, 1x1 convs on 1x1 maps): the workgroup is one
// wave, the lane tree's total IS the dot product -- no LDS word, no barrier, no slot tree.
// NQ (ONE only) = quarters of the segment that hold samples: m <= 256 -> 1, m <= 512 -> 2 (VGG-16's fully connected layers
// at batch 512, AlexNet's at 32): the zero padding is neither loaded nor swept.
// The PREFETCH AGENT of a resident workgroup (round 4): one extra wave that never sweeps.  A layer whose prepared columns
// exceed the 256-MB Infinity Cache (ResNet-50's layer4.0.conv2 at batch 1024: 264 MB) takes every column from HBM, and a
// column that misses holds the CU's outstanding-request slots about twice as long as one that hits: the
// sweeps' own two steps of look-ahead do not cover it, and a third did not help (round 3).  The agents touch ONE dword of
// every 128-byte line of columns t + K (x and a) so that the lines sit in the XCD's L2 when the sweep waves ask for them.
// Every workgroup reads the same columns, so the work is SHARED by the workgroups of an XCD (b and b + 8 share one under
// round-robin dispatch -- speed only: a wrong guess prefetches into another L2): workgroup j of its XCD (j = (b >> 3) & 31)
// touches line j of every segment, x in lanes 0 .. S-1 and a in lanes S .. 2S-1: ONE load instruction per step with 2 S
// active lanes (fewer workgroups per XCD: LP = 2 or 4 lines of every segment each).  (First version: the agents of eight workgroups touched everything, 64 different lines per instruction, 14
// instructions per step -- an uncoalesced load occupies the CU's address unit for as long as 64 separate ones, those eight
// workgroups fell behind and the layer went from 0.77 to 0.98 us per column.)  The loads land in a window register nobody
// reads; nothing waits for them.
// The ONE barrier of a resident step.  The sweep waves (resident_body::step) and the prefetch agent, which returns early and
// never sweeps, meet at it: both MUST pass exactly one per column and none outside the column loop -- a second or a
// conditional barrier in the step would not hang (ended waves drop out of barriers) but let the sweep waves run past
// segment sums that have not been written.  So both name it through these two helpers and nothing else in the resident
// kernels may synchronise the workgroup (tests/test_gpu_prefetch_agent.py runs every (rows, wave bound) variant with the
// agent against the oracle).
__device__ __forceinline__ void resident_step_barrier() { __syncthreads(); }
// (the agent has nothing in LDS and a load in flight that must NOT be waited for: lgkmcnt only)
__device__ __forceinline__ void resident_agent_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int WB>
__device__ __forceinline__ void resident_prefetch_agent(const SlabParams& p, int g, int lane)
{
    const int K = p.prefetch_ahead;
    const int S = p.S;
    // lines of a segment this workgroup touches (1 with 32+ workgroups per XCD); a launch with an agent wave but no lines
    // (the host never makes one: launch_resident) leaves the agent to pass the barriers and touch nothing
    const bool any = p.prefetch_lines > 0;
    const int LP = any ? p.prefetch_lines : 1;
    const int W = 32 / LP;                                       // workgroups of an XCD that share a segment's 32 lines
    const int64_t col_bytes = p.m_pad * (int64_t)sizeof(float);
    const unsigned j = (blockIdx.x >> 3) % (unsigned)W;          // this workgroup's share
    const int per_matrix = S * LP;
    const bool mine = any && lane < 2 * per_matrix;
    const int rem = lane < per_matrix ? lane : lane - per_matrix;
    const unsigned line = j + (unsigned)(rem % LP) * (unsigned)W;
    const char* base = reinterpret_cast<const char*>((lane < per_matrix ? p.XT : p.AT) + (int64_t)g * p.d * p.m_pad) +
                       (mine ? (unsigned)(rem / LP) * 4096u + 128u * line : 0u);
    for (int t = 0; t < p.d; ++t) {
        const int tc = t + K;
        if (mine && tc < p.d)
            asm volatile("global_load_dword v[%c1], %0, off" :: "v"(base + tc * col_bytes), "n"(WB) : "memory");
        resident_agent_barrier();                                       // the step's one barrier
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int RT, int MODE, int WB, bool ONE = false, int NQ = 4>
__device__ __forceinline__ void resident_body(const SlabParams& p)
{
    static_assert(NQ == 4 || ONE, "partial segments: one-segment variant only");
    static_assert(RT == 1 || RT == 2 || RT == 4, "one DPP row of 16 lanes per residual row");
    constexpr int X0 = WB, X1 = WB + 16, X2 = WB + 32, A0 = WB + 48, A1 = WB + 64, U0 = WB + 80;
    extern __shared__ float smem[];                 // seg[2][RT][S]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int S = p.S;                              // == sweep waves of the workgroup (+ the prefetch agent's wave, if launched with one)
    const int P = pow2_ceil(S);                     // slots of the canonical tree, <= 16: one DPP row per residual row
    const int row0 = blockIdx.x * RT, g = blockIdx.y;
    if constexpr (!ONE) {
        if (__builtin_amdgcn_readfirstlane(wave) == S) {     // launched with S + 1 waves: this one is the prefetch agent
            resident_prefetch_agent<WB>(p, g, lane);
            return;
        }
    }
    const SlotMap smap = make_slot_map(S, P, 0, 1, (lane & 15) % P, P);
    const int r16 = lane >> 4;                      // the residual row this lane serves in the slot tree
    const bool occupied = (smap.mask & 1u) && (lane & 15) < P && r16 < RT;   // other lanes contribute +0.0f
    const int seg_off = (r16 < RT ? r16 : 0) * S + smap.s0;                  // a valid LDS word in every lane
    const int64_t grow0 = (int64_t)g * p.Ng + row0;
    const float* xload = uniform_ptr(p.XT + (int64_t)g * p.d * p.m_pad + (int64_t)wave * kSeg);
    const float* aload = uniform_ptr(p.AT + (int64_t)g * p.d * p.m_pad + (int64_t)wave * kSeg);
    const unsigned lane_off = 16u * (unsigned)lane;
    const kfloat* nrm = as_scalar(p.nrm2 + 2 * (int64_t)g * p.d);
    // "this is wave 0" as a scalar, so that the flush test of the Q / idx history is a scalar branch
    const bool wave0 = __builtin_amdgcn_readfirstlane(wave) == 0;

    const kfloat* wrow[RT];
    float qprev[RT], wcur[RT], qhist[RT];
    int ihist[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        // rows past the end of the group repeat its last row (computed, never stored)
        const int64_t gr = grow0 + ((row0 + r < p.Ng) ? r : (p.Ng - 1 - row0));
        wrow[r] = as_scalar(p.W + gr * p.ldw);
        qprev[r] = 0.0f;
        wcur[r] = wrow[r][0];
        qhist[r] = 0.0f;
        ihist[r] = 0;
    }
    float n2cur = nrm[0];
    float in2cur = nrm[1];                          // fl(1 / ||x_t||^2) for quant_msq_from_dot

    // the residual starts at 0 (a non-zero initial residual is the streaming plan's job); x_t lives in X[t % 3],
    // a_t in A[t % 2]; X2 starts as x_{-1} = 0 (q_{-1} = 0)
    win_zero16<U0>();
    if constexpr (RT >= 2) win_zero16<U0 + 16>();
    if constexpr (RT >= 4) { win_zero16<U0 + 32>(); win_zero16<U0 + 48>(); }
    win_zero16<X2>();
    if constexpr (NQ < 4) { win_zero16<X0>(); win_zero16<X1>(); }     // (the quarters no load ever writes)
    win_load16<X0, NQ>(xload, lane_off);
    win_load16<A0, NQ>(aload, lane_off);
    {
        const int64_t adv = (1 < p.d) ? p.m_pad : 0;   // a one-column layer re-reads column 0
        xload += adv;
        aload += adv;
    }
    win_load16<X1, NQ>(xload, lane_off);
    win_load16<A1, NQ>(aload, lane_off);

#ifdef GPFQ_STAMPS
    // diagnostic build only: cycles per phase of a step, summed by wave 0 and the last wave of block 0 into status[16..]
    unsigned long long rst_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long rst_prev = 0;
#define GPFQ_RSTAMP(i)                                                                                  \
    {                                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        unsigned long long now_;                                                                        \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");                      \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        rst_sum[i] += now_ - rst_prev;                                                                  \
        rst_prev = now_;                                                                                \
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rst_prev)::"memory");
#else
#define GPFQ_RSTAMP(i)
#endif
    int t = 0;
    const int dlast = p.d - 1;
    // the lanes q is read from (v_readlane 16 * r): the only ones whose quantizer result is used
    constexpr unsigned long long kRowLanes = RT == 1 ? 0x1ull : (RT == 2 ? 0x10001ull : 0x1000100010001ull);
    // one step; XP holds x_{t-1}, XC x_t, AC a_t.  Returns false after the last column.
    auto step = [&](auto xp_, auto xc_, auto ac_) -> bool {
        constexpr int XP = decltype(xp_)::value, XC = decltype(xc_)::value, AC = decltype(ac_)::value;
        GPFQ_RSTAMP(0)
        const bool more = t + 1 < p.d;
        float* seg = smem + (t & 1) * RT * S;
        win_wait<2 * NQ>();                              // column t has landed; the eight loads of column t+1 stay in flight
        GPFQ_RSTAMP(1)
        float acc[RT];
        if constexpr (RT == 1) {
            acc[0] = win_sweep16<U0, XP, AC, XC, NQ>(qprev[0], wcur[0]);
        } else {                                    // rows in interleaved pairs: 80 instructions per pair instead of 96
            const v2f a01 = win_sweep16_pair<U0, XP, AC, XC, NQ>(qprev[0], qprev[1], wcur[0], wcur[1]);
            acc[0] = a01.x; acc[1] = a01.y;
            if constexpr (RT >= 4) {
                const v2f a23 = win_sweep16_pair<U0 + 32, XP, AC, XC, NQ>(qprev[2], qprev[3], wcur[2], wcur[3]);
                acc[2] = a23.x; acc[3] = a23.y;
            }
        }
        GPFQ_RSTAMP(2)
        // column t+2 goes into the registers the sweeps have just finished with (x_{t-1}'s and a_t's); the last two
        // steps re-read the last column rather than branch
        {
            const int64_t adv = (t + 2 < p.d) ? p.m_pad : 0;
            xload += adv;
            aload += adv;
        }
        win_load4<XP, 0, NQ>(xload, lane_off);
        win_load4<AC, 0, NQ>(aload, lane_off);
        float v1 = 0.0f;                            // ONE: the dot products, row r in lane row r
        if constexpr (RT == 1) {
            const float sg = wave_tree64_lane63(acc[0]);
            if constexpr (ONE) v1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sg), 63));
            else if (lane == 63) seg[wave] = sg;
        } else {                                    // row r's total in lane row r: one LDS write for all rows
            const float tot = wave_tree64_rows<RT>(acc);
            if constexpr (ONE) v1 = tot;
            else if ((lane & 15) == 0 && r16 < RT) seg[r16 * S + wave] = tot;
        }
        float uni = 0.0f;
        if (MODE == MODE_STOCHASTIC)
            uni = philox_uniform(p.seed, p.row_id0 + (uint64_t)(grow0 + ((r16 < RT && row0 + r16 < p.Ng) ? r16 : 0)), (uint64_t)t);
        win_load4<XP, 1, NQ>(xload, lane_off);
        win_load4<AC, 1, NQ>(aload, lane_off);
        GPFQ_RSTAMP(3)
        if constexpr (!ONE) resident_step_barrier();
        GPFQ_RSTAMP(4)
        win_load4<XP, 2, NQ>(xload, lane_off);
        win_load4<AC, 2, NQ>(aload, lane_off);
        // the slot tree in every wave, all RT rows at once: lane = 16 * row + slot; the other lanes hold +0.0f, so the
        // four levels need no tests
        float v;
        if constexpr (ONE) {
            v = v1;
        } else {
            const float val = seg[seg_off];
            v = wave_tree16_zero_padded(occupied ? val : 0.0f);
        }
        // Next column's weights and norm through the scalar cache (the last step re-reads its own, unused, ones rather
        // than branch) -- requested HERE, behind the LDS read: scalar loads and LDS share one counter (lgkmcnt) and
        // return out of order, so every wait on that counter is a wait for all of them; requested at the top of the
        // step they made the `s_waitcnt lgkmcnt(0)` in front of the sweep wait a whole scalar-cache round trip.  From
        // here the next wait on the counter is a full quantizer away.  The scheduling barrier keeps the tree above, the
        // opaque asm (its result is the address) keeps the requests below.
        // byte offset of the next column's weight (the last step re-reads its own); min, not a select on `more`: a
        // select on a lane-mask boolean is vector code
        unsigned tn4 = 4u * (unsigned)(t + 1 < dlast ? t + 1 : dlast);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" : "+s"(tn4)::"memory");
        float wn[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) wn[r] = sload(wrow[r], tn4);
        const float n2n = sload(nrm, 2u * tn4), in2n = sload(nrm, 2u * tn4 + 4u);
        __builtin_amdgcn_sched_barrier(0);          // ... and keeps them from sinking below the quantizer
        int id;
        float q;
        // MSQ: index and value straight from the dot product (gpfq_device.h quant_msq_from_dot), and the reference's two
        // divisions only on the rare step whose quotient lies within (K + 4) * 2^-18 of a rounding boundary.  The answer is used
        // first and checked second: the check is a chain of its own, as long as the one to q, and a branch in front of
        // the readlanes would put it back on the critical path; behind them its condition has long been computed.  (The
        // same bits in every wave, so every wave takes the same side.)
        bool redo = false;
        auto divide_and_quantize = [&]() {
            const float sarg = (n2cur > 0.0f) ? v / n2cur : 0.0f;
            if (MODE == MODE_SOFT) q = quant_soft(p.step, sarg, p.Kf, p.lamb, id);
            else if (MODE == MODE_HARD) q = quant_hard(p.step, sarg, p.Kf, p.lamb, id);
            else if (MODE == MODE_STOCHASTIC) q = quant_stochastic(p.step, sarg, p.Kf, uni, id);
            else q = quant_msq(p.step, sarg, p.Kf, id);
        };
        if (MODE == MODE_MSQ) redo = !quant_msq_from_dot(v, in2cur, p.inv_step, p.step, p.Kf, p.msq_thr, ~kRowLanes, q, id);
        else divide_and_quantize();
        // Q / idx: 64 steps of history in registers (lane l holds step t0 + l), one coalesced store every 64 steps
        auto commit = [&]() {
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                qprev[r] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, q), 16 * r));
                const int idr = __builtin_amdgcn_readlane(id, 16 * r);
                if (lane == (t & 63)) { qhist[r] = qprev[r]; ihist[r] = idr; }
            }
        };
        commit();
        GPFQ_RSTAMP(5)
        if (MODE == MODE_MSQ && __builtin_expect(redo, 0)) {
            divide_and_quantize();
            commit();
        }
        if (__builtin_expect(((t & 63) == 63 || !more) && wave0, 0)) {
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
        win_load4<XP, 3, NQ>(xload, lane_off);
        win_load4<AC, 3, NQ>(aload, lane_off);
        if (!more) return false;
#pragma unroll
        for (int r = 0; r < RT; ++r) wcur[r] = wn[r];
        n2cur = n2n;
        in2cur = in2n;
        ++t;
        return true;
    };
    using I0 = std::integral_constant<int, X0>; using I1 = std::integral_constant<int, X1>; using I2 = std::integral_constant<int, X2>;
    using J0 = std::integral_constant<int, A0>; using J1 = std::integral_constant<int, A1>;
    int k = 0;                                       // buffer that holds x_t when the loop ends (t % 3)
    for (;;) {
        k = 0; if (!step(I2{}, I0{}, J0{})) break;
        k = 1; if (!step(I0{}, I1{}, J1{})) break;
        k = 2; if (!step(I1{}, I2{}, J0{})) break;
        k = 0; if (!step(I2{}, I0{}, J1{})) break;
        k = 1; if (!step(I0{}, I1{}, J0{})) break;
        k = 2; if (!step(I1{}, I2{}, J1{})) break;
    }
#ifdef GPFQ_STAMPS
    if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && (wave == 0 || wave == S - 1) && p.status) {
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(p.status + 16) + (wave == 0 ? 0 : 8);
        for (int i = 0; i < 8; ++i) dbg[i] = rst_sum[i];
    }
#endif
    // every load issued above has landed before the last column is used again
    win_wait<0>();
    auto finish = [&](auto xl_) {
        constexpr int XL = decltype(xl_)::value;
        if constexpr (RT == 1) {
            finish_row_w<U0, XL>(p, qprev[0], row0 < p.Ng, grow0, wave, lane);
        } else {                                    // interleaved pairs: row r sits at U0 + 32 (r / 2) + (r % 2), stride 2
            finish_row_w<U0, XL, 2>(p, qprev[0], row0 < p.Ng, grow0, wave, lane);
            finish_row_w<U0 + 1, XL, 2>(p, qprev[1], row0 + 1 < p.Ng, grow0 + 1, wave, lane);
            if constexpr (RT >= 4) {
                finish_row_w<U0 + 32, XL, 2>(p, qprev[2], row0 + 2 < p.Ng, grow0 + 2, wave, lane);
                finish_row_w<U0 + 33, XL, 2>(p, qprev[3], row0 + 3 < p.Ng, grow0 + 3, wave, lane);
            }
        }
    };
    if (k == 0) finish(I0{});
    else if (k == 1) finish(I1{});
    else finish(I2{});
}

// One __global__ per (rows per workgroup, quantizer, wave bound): amdgpu_num_vgpr takes a literal, and it is what
// reserves the window -- WB = budget(MAXW) - 80 - 16 RT registers for the compiler, the rest for the window.  On
// gfx950 (one register file for VGPRs and AGPRs) amdgpu_num_vgpr(n) leaves the compiler 2n registers, v0 .. v(2n-1)
// (probe: a kernel that needs 200 registers uses v0..v143 and spills the rest under amdgpu_num_vgpr(72)), hence
// WB / 2; the build's ISA check (tools/check_async_loads.py: no compiler instruction may name a window register) is
// what guarantees it for every kernel of every build.
// MAXW = most waves the kernel may be launched with: 8 -> 256 VGPRs, 12 -> 168, 16 -> 128.
#define GPFQ_DEFINE_RESIDENT(RT, MODE, MAXW, WB, LAST)                                                            \
    __global__ void __launch_bounds__(64 * MAXW) __attribute__((amdgpu_num_vgpr(WB / 2)))                          \
    gpfq_resident_rt##RT##_m##MODE##_w##MAXW(const SlabParams p)                                                 \
    {                                                                                                             \
        asm volatile("" ::: LAST);      /* makes the kernel descriptor allocate the whole budget */               \
        resident_body<RT, MODE, WB>(p);                                                                           \
    }
#define GPFQ_DEFINE_RESIDENT_MODES(RT, MAXW, WB, LAST)                                                            \
    GPFQ_DEFINE_RESIDENT(RT, 0, MAXW, WB, LAST) GPFQ_DEFINE_RESIDENT(RT, 1, MAXW, WB, LAST)                       \
    GPFQ_DEFINE_RESIDENT(RT, 2, MAXW, WB, LAST) GPFQ_DEFINE_RESIDENT(RT, 3, MAXW, WB, LAST)
GPFQ_DEFINE_RESIDENT_MODES(1, 8, 160, "v255")
GPFQ_DEFINE_RESIDENT_MODES(2, 8, 144, "v255")
GPFQ_DEFINE_RESIDENT_MODES(4, 8, 112, "v255")
GPFQ_DEFINE_RESIDENT_MODES(1, 12, 72, "v167")
GPFQ_DEFINE_RESIDENT_MODES(2, 12, 56, "v167")
GPFQ_DEFINE_RESIDENT_MODES(1, 16, 32, "v127")
// one-segment rows: a workgroup is one wave (same budgets as the 8-wave variants)
#define GPFQ_DEFINE_RESIDENT_ONE(RT, MODE, WB)                                                                    \
    __global__ void __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(WB / 2)))                                \
    gpfq_resident_rt##RT##_m##MODE##_w1(const SlabParams p)                                                      \
    {                                                                                                             \
        asm volatile("" ::: "v255");                                                                              \
        resident_body<RT, MODE, WB, true>(p);                                                                     \
    }
#define GPFQ_DEFINE_RESIDENT_ONE_MODES(RT, WB)                                                                    \
    GPFQ_DEFINE_RESIDENT_ONE(RT, 0, WB) GPFQ_DEFINE_RESIDENT_ONE(RT, 1, WB) GPFQ_DEFINE_RESIDENT_ONE(RT, 2, WB)    \
    GPFQ_DEFINE_RESIDENT_ONE(RT, 3, WB)
GPFQ_DEFINE_RESIDENT_ONE_MODES(1, 160)
GPFQ_DEFINE_RESIDENT_ONE_MODES(2, 144)
GPFQ_DEFINE_RESIDENT_ONE_MODES(4, 112)
// ... and of m <= 256 / 512 samples (one / two quarters of the segment): half or a quarter of the loads and of the sweep
#define GPFQ_DEFINE_RESIDENT_ONE_Q(RT, MODE, WB, NQ)                                                              \
    __global__ void __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(WB / 2)))                                \
    gpfq_resident_rt##RT##_m##MODE##_w1q##NQ(const SlabParams p)                                                 \
    {                                                                                                             \
        asm volatile("" ::: "v255");                                                                              \
        resident_body<RT, MODE, WB, true, NQ>(p);                                                                 \
    }
#define GPFQ_DEFINE_RESIDENT_ONE_Q_MODES(RT, WB, NQ)                                                              \
    GPFQ_DEFINE_RESIDENT_ONE_Q(RT, 0, WB, NQ) GPFQ_DEFINE_RESIDENT_ONE_Q(RT, 1, WB, NQ)                           \
    GPFQ_DEFINE_RESIDENT_ONE_Q(RT, 2, WB, NQ) GPFQ_DEFINE_RESIDENT_ONE_Q(RT, 3, WB, NQ)
GPFQ_DEFINE_RESIDENT_ONE_Q_MODES(1, 160, 1) GPFQ_DEFINE_RESIDENT_ONE_Q_MODES(2, 144, 1) GPFQ_DEFINE_RESIDENT_ONE_Q_MODES(4, 112, 1)
GPFQ_DEFINE_RESIDENT_ONE_Q_MODES(1, 160, 2) GPFQ_DEFINE_RESIDENT_ONE_Q_MODES(2, 144, 2) GPFQ_DEFINE_RESIDENT_ONE_Q_MODES(4, 112, 2)

// ------------------------------------------------------------------------------------------------
// Streaming plan: any (N, m).  The residual rows stay in the caller's U (HBM / L2 / Infinity Cache) and
// are read and written once per step; wave w owns segments w, w+NW, ... of the workgroup's RT rows.
// ------------------------------------------------------------------------------------------------
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
    const kfloat* nrm = as_scalar(p.nrm2 + 2 * (int64_t)g * p.d);
    const kfloat* Wk = as_scalar(p.W);

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
        for (int r = 0; r < RT; ++r) w[r] = last ? 0.0f : Wk[grow[r] * p.ldw + t];
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
                if (last && p.usq && valid[r]) store_segment_sumsq(p.usq, grow[r] * S + s, u[r], lane);
            }
        }
        if (last) break;
        __syncthreads();
        if (wave == 0) {
            // this member's block of the slot tree, row r's value parked in lane r
            float mine = 0.0f;
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const float pr = combine_slots<(RT <= 2)>(seg + r * n_max - seg_lo, smap, per, nl, seg_hi - 1);
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
            const float n2 = nrm[2 * t];
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

}  // namespace gpfq
