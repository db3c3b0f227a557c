// gpfq_pipe_kernels.h -- the PIPELINED cooperative kernels (round 4): gpfq_pipe_rg{1,2}_*.
//
// Reference: StepAlgorithm._quantization, step_algorithm.py:107-148 -- the same recurrence, the same canonical arithmetic
// and the same granule exchange as gpfq_coop_* (gpfq_loop_kernels.h).  What changes is WHEN things happen.
//
// The lock-step cooperative kernel does, per column t and for all RT rows of the workgroup at once:
//     sweep -> barrier -> [slot tree -> publish -> gather (a fabric round trip) -> quantize] -> barrier
// and every sweep wave idles through the bracket (62-71 % of their life: profiles/r03_v5_pmc_counters.json).  But rows are
// independent (step_algorithm.py:141-148 is row-wise), so here the workgroup's rows are cut into G = 4 GROUPS of RG rows
// and a step becomes four PHASES, one barrier each.  In phase p = 4 t + g
//     the sweep waves   sweep group g with column t (needs q_g(t-1), which the reducer left in LDS at least one phase ago);
//     the publisher wave  (a) finishes the slot tree of the group swept in phase p-1 and PUBLISHES its granules;
//     the gatherer wave   (b) GATHERS the granules of the group swept in phase p-3 (published by every member in its
//                             phase p-2; the load itself was requested in phase p-1), finishes the tree over the members,
//                             quantizes and leaves q in LDS -- one phase before that group's next sweep (phase p+1).
// Three exchanges are in flight while the fourth group is swept: a step costs 4 x max(sweep of one group, the longer of the
// two reducer roles' phases, what a granule needs to travel) instead of (sweep of all rows + one exposed exchange).  The
// two reducer roles are ALWAYS waves of their own (one wave doing both was measured: 1 200 cycles per phase, the sweep
// waves 600-700 at the barrier -- profiles/NOTES.md, round 4).
//
// Bit-exactness is untouched: per row the same sweep (win_sweep16 / win_sweep16_pair), the same lane tree, the same slot
// tree split at the same member boundaries, the same quantizer -- only interleaved differently in time.
//
// Columns: x_t in X[t % 3], a_t in A[t % 2] (the column window of gpfq_device.h).  All four phases of step t read
// x_{t-1}, x_t, a_t; column t+1 is requested in phases 0 and 1 of step t into the two buffers step t does not use
// (x_{t-2}'s and a_{t-1}'s) and waited for in front of phase 0 of step t+1: two to three phases of look-ahead.
// Granules: xbuf[tile][t & 1][member][row]; epoch t + 1.  A member can run at most two phases ahead of another (its
// gather of phase p needs the other's publish of phase p-2), so the slot written for step t+1 (phase p+5 of the group)
// is never the one still being gathered for step t (phase p+3), and the slot of step t is rewritten at step t+2 only.
#pragma once
#include "gpfq_loop_kernels.h"

namespace gpfq {

// s_waitcnt lgkmcnt(0) + s_barrier, NOT __syncthreads(): the sweep waves keep window loads in flight across barriers and
// the reducer a granule store; what a barrier has to order here are LDS words only (segment sums, q).
__device__ __forceinline__ void pipe_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

#ifdef GPFQ_STAMPS
// diagnostic build only (make stamps): cycles per part of a phase, summed over the launch by the reducer wave and by sweep
// wave 0 of workgroup 0 into status[16..] (tools/stamps.py)
#define GPFQ_PSTAMP(i)                                                                                  \
    {                                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        unsigned long long now_;                                                                        \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");                    \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        pst_sum[i] += now_ - pst_prev;                                                                  \
        pst_prev = now_;                                                                                \
    }
#define GPFQ_PSTAMP_DECL                                                                                \
    unsigned long long pst_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};                                           \
    unsigned long long pst_prev = 0;                                                                    \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pst_prev)::"memory");
#define GPFQ_PSTAMP_DUMP(base)                                                                          \
    if (blockIdx.x == 0 && lane == 0 && p.status) {                                                     \
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(p.status + 16) + (base);        \
        for (int i = 0; i < 8; ++i) dbg[i] = pst_sum[i];                                                \
    }
#else
#define GPFQ_PSTAMP(i)
#define GPFQ_PSTAMP_DECL
#define GPFQ_PSTAMP_DUMP(base)
#endif

// The reducer wave of a pipelined workgroup: 4 d + 3 phases, in phase ph part (a) for the group swept in phase ph-1 and
// part (b) for the group swept in phase ph-3.  Lane layouts as in reducer_section (gpfq_loop_kernels.h): (a) lane =
// 16 * row + slot; (b) lane = stride * row + member, stride = max(16, C) -- RG * C <= 64 granules per group.
// The epoch word of a granule: bits 0..19 column + 1, bits 20..27 the launch number (a line an earlier launch left in an L2
// can never match), bits 28..31 the publisher's XCD + 1.
__device__ __forceinline__ unsigned pipe_epoch_tag(unsigned salt) { return (salt & 255u) << 20; }
// HW_REG_XCC_ID (hardware register 20, bits 3:0): the XCD this wave runs on, + 1 so that a zeroed granule names none
__device__ __forceinline__ unsigned pipe_xcc_id() { return (__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15u) + 1u; }

// The reducer roles of a pipelined workgroup, 4 d + 3 phases each.  ROLE 0: the PUBLISHER wave, 1: the GATHERER wave, 2: both
// roles in ONE wave (the variant with seven sweep waves).
//   publisher, phase ph: part (a) for the group swept in phase ph-1 -- this member's block of the slot tree (lane = 16 * row +
//     slot, as in reducer_section, gpfq_loop_kernels.h) and one granule per row;
//   gatherer, phase ph: part (b) for the group swept in phase ph-3 -- lane = stride * row + member, stride = max(16, C),
//     RG * C <= 64 granules per group.
// XCD-LOCAL PUBLISHING.  A device-scope store is a write-through to the fabric, and the line leaves the XCD's L2: a member on
// the same XCD then reads it back from memory (0.67 us per exchange on an idle chip; 0.36 us with a plain store, which stays in
// that L2, where device-scope loads -- they bypass only the vector L1 -- find it: profiles/r03_xchg_probe.txt).  Members on
// OTHER XCDs never see a plain store, so locality is ESTABLISHED, not assumed: every granule's epoch word names its
// publisher's XCD (HW_REG_XCC_ID); steps 0 of all groups are published device-scope; the gatherer's first gather compares
// the members' XCDs with its own and raises a flag in LDS if they all match; from then on the publisher stores plainly.
// Every member of a tile sees the same set of XCDs, so a tile switches as a whole; one member elsewhere and it never does.
// GV = a register PAIR of the column window, free in the gatherer's wave (it sweeps nothing): the gather requested one phase
// ahead lands there.  It cannot be a C++ value: the compiler waits for a load of its own with `s_waitcnt vmcnt(0)` wherever
// it sees fit (at the head of the re-poll loop, in front of the next request, behind the publisher's store) and would treat
// the register as free meanwhile.  As asm statements the request, the wait and the two moves out of the window are ours;
// operations the compiler issues itself (the rare Q / idx flush) only make the waits longer, never shorter -- the counter
// retires in order.  In the one-wave variant exactly one younger operation, part (a)'s store, is outstanding at the wait:
// vmcnt(1), or the wave would sit out that store's round trip every phase.
// GPL = granules per lane of a gather (1, or 4: members 4 j .. 4 j + 3 of a row in lane j -- adjacent blocks of the slot tree,
// added (g0 + g1) + (g2 + g3) in the lane: the first two levels of the tree over the members -- so that two rows x 128 members,
// 256 granules, fit the 64 lanes: rows of 769 .. 896 segments, ResNet-50's 56 x 56 maps at batch 1024).
template <int RG, int MODE, int GV, int ROLE, int GPL = 1>
__device__ __forceinline__ void pipe_reducer(const SlabParams& p, const float* segs, float* qs, float* local_flag, int NS, int lane,
                                             int tile, int c, int C, int nl, int seg_lo, int row0)
{
    constexpr int G = 4, RT = G * RG;
    constexpr bool PUB = ROLE != 1, GAT = ROLE != 0;
    constexpr bool FAST = MODE == MODE_MSQ;
    const int nph = 4 * p.d;
    unsigned long long* const xb = p.xbuf + (unsigned)(tile * 2) * (unsigned)(C * RT);
    // ---- publisher state
    const int P = pow2_ceil(p.S);
    const SlotMap smap = make_slot_map(p.S, P, c * nl, 1, lane & 15, nl);
    const int r16 = lane >> 4;
    const bool mine = (smap.mask & 1u) && r16 < RG;      // this lane holds an occupied slot of row r16 of the group
    const int seg_word = (r16 < RG ? r16 : 0) * NS + (smap.s0 - seg_lo);
    const unsigned pub_off = (unsigned)c * RT + (unsigned)(r16 < RG ? r16 : 0);         // + ga * RG
    const unsigned my_xcc = pipe_xcc_id();
    const unsigned tag = pipe_epoch_tag(p.salt);         // the launch number, in every epoch word
    bool local = false;                                  // the gatherer has found every member of the tile on this XCD
    // ---- gatherer state
    const kfloat* nrm = as_scalar(p.nrm2);
    float* hist = qs + RT;                               // [RT][64] values, then [RT][64] indices (as int bits)
    static_assert(GPL == 1 || GPL == 4, "one granule per lane, or four");
    const int lpr = C / GPL;                             // lanes per row of a gather
    const int sh = lpr <= 16 ? 4 : (lpr <= 32 ? 5 : 6);
    const int grow = lane >> sh, member = lane & ((1 << sh) - 1);
    const bool want = member < lpr && grow < RG;
    const bool lead = member == 0 && grow < RG;
    const unsigned long long idle = ~__builtin_amdgcn_ballot_w64(want);
    const unsigned long long unused = ~__builtin_amdgcn_ballot_w64(lead);
    const unsigned src_off = want ? (unsigned)(GPL * member) * RT + (unsigned)grow : 0u; // + gb * RG (idle lanes: member 0's granules)
    // the window registers of a gather: GPL pairs from GV up; the request, the wait + read-out and the epoch test of all of them
    // (scalar base + per-lane 32-bit byte offset: the base -- tile, parity of the step, group -- is computed by the scalar unit,
    // the lane part never changes; as per-lane 64-bit pointers every request cost four 64-bit vector additions)
    const unsigned lane_bytes = 8u * src_off;
    auto request = [&](const unsigned long long* base) {
        asm volatile("global_load_dwordx2 v[%c2:%c2+1], %0, %1 sc1" :: "v"(lane_bytes), "s"(base), "n"(GV) : "memory");
        if constexpr (GPL == 4) {
            asm volatile("global_load_dwordx2 v[%c2:%c2+1], %0, %1 offset:%c3 sc1" :: "v"(lane_bytes), "s"(base), "n"(GV + 2), "n"(8 * RT) : "memory");
            asm volatile("global_load_dwordx2 v[%c2:%c2+1], %0, %1 offset:%c3 sc1" :: "v"(lane_bytes), "s"(base), "n"(GV + 4), "n"(16 * RT) : "memory");
            asm volatile("global_load_dwordx2 v[%c2:%c2+1], %0, %1 offset:%c3 sc1" :: "v"(lane_bytes), "s"(base), "n"(GV + 6), "n"(24 * RT) : "memory");
        }
    };
    // ... and the re-poll's: its base may come straight out of a spill slot (v_readlane, a VALU write of the SGPR pair), which an
    // asm VMEM instruction may read only five wait states later -- the compiler pads its own loads, not ours
    // (ONE statement with the first load: operands are reloaded in front of a statement, never inside it)
    auto request_again = [&](const unsigned long long* base) {
        asm volatile("s_nop 4\n\tglobal_load_dwordx2 v[%c2:%c2+1], %0, %1 sc1" :: "v"(lane_bytes), "s"(base), "n"(GV) : "memory");
        if constexpr (GPL == 4) {
            asm volatile("s_nop 4\n\tglobal_load_dwordx2 v[%c2:%c2+1], %0, %1 offset:%c3 sc1" :: "v"(lane_bytes), "s"(base), "n"(GV + 2), "n"(8 * RT) : "memory");
            asm volatile("s_nop 4\n\tglobal_load_dwordx2 v[%c2:%c2+1], %0, %1 offset:%c3 sc1" :: "v"(lane_bytes), "s"(base), "n"(GV + 4), "n"(16 * RT) : "memory");
            asm volatile("s_nop 4\n\tglobal_load_dwordx2 v[%c2:%c2+1], %0, %1 offset:%c3 sc1" :: "v"(lane_bytes), "s"(base), "n"(GV + 6), "n"(24 * RT) : "memory");
        }
    };
    unsigned glo[GPL], ghi[GPL];
    auto read_out = [&]() {
#define GPFQ_RD(i) asm volatile("v_mov_b32 %0, v[%c2]\n\tv_mov_b32 %1, v[%c2+1]\n\ts_nop 0" : "=v"(glo[i]), "=v"(ghi[i]) : "n"(GV + 2 * i) : "memory");
        GPFQ_RD(0)
        if constexpr (GPL == 4) { GPFQ_RD(1) GPFQ_RD(2) GPFQ_RD(3) }
#undef GPFQ_RD
    };
    auto all_arrived = [&](unsigned epoch) {
        unsigned long long ok = __builtin_amdgcn_ballot_w64((ghi[0] & 0x0fffffffu) == epoch);
        if constexpr (GPL == 4) {
#pragma unroll
            for (int i = 1; i < 4; ++i) ok &= __builtin_amdgcn_ballot_w64((ghi[i] & 0x0fffffffu) == epoch);
        }
        return (ok | idle) == __builtin_amdgcn_read_exec();
    };
    bool gave_up = false;
    __builtin_amdgcn_s_setprio(3);                       // a short dependent chain among long sweeps: issue it first
    GPFQ_PSTAMP_DECL
    // one phase.  STEADY: 4 <= ph < 4 d -- every part of the phase runs (a publish, a gather, a request) and the locality flag
    // has been decided: no range tests, no flag read; the seven phases of fill and drain take the general form
    auto phase = [&](auto steady_, const int ph) {
        constexpr bool STEADY = decltype(steady_)::value;
        GPFQ_PSTAMP(0)                                   // the barrier
        bool published = false;
        if constexpr (PUB) {
            // ---- (a) the group swept in the phase before: this member's block of the slot tree, published
            published = STEADY || (ph >= 1 && ph <= nph);
            if (published) {
                const int pa = ph - 1, ga = pa & 3, ta = pa >> 2;
                const float val = segs[ga * RG * NS + seg_word];
                if constexpr (!STEADY) {
                    if (!local && ph >= 4) local = __builtin_amdgcn_readfirstlane(__float_as_int(*local_flag)) != 0;
                }
                const float v = wave_tree16_zero_padded(mine ? val : 0.0f);
                unsigned long long* dst = xb + (unsigned)(ta & 1) * (unsigned)(C * RT) + pub_off + (unsigned)(ga * RG);
                const unsigned long long granule = ((unsigned long long)(tag | (my_xcc << 28) | ((unsigned)ta + 1u)) << 32) |
                                                   (unsigned long long)__float_as_uint(v);
                if ((lane & 15) == 0 && r16 < RG) {
                    if (local) __hip_atomic_store(dst, granule, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else __hip_atomic_store(dst, granule, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if constexpr (ROLE == 0) { GPFQ_PSTAMP(1) }
        }
        if constexpr (GAT) {
            // ---- (b) the group swept three phases ago: gather, finish the tree over the members, quantize, q into LDS
            const bool gathers = STEADY || ph >= 3;
            const int pb = ph - 3, gb = pb & 3, tb = pb >> 2;
            bool timed_out = false;
            float n2cur = 0.0f, in2cur = 0.0f;
            if (gathers) {
                const unsigned epoch = tag | ((unsigned)tb + 1u);
                n2cur = sload(nrm, 8u * (unsigned)tb);
                in2cur = sload(nrm, 8u * (unsigned)tb + 4u);
                const unsigned long long* src = xb + (unsigned)(tb & 1) * (unsigned)(C * RT) + (unsigned)(gb * RG);    // (uniform: + the lane's offset in the load)
                // the first look at these granules was REQUESTED in the phase before (below): a device-scope load is a round
                // trip of its own even when the data has long arrived
                if (published) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                read_out();
                GPFQ_PSTAMP(1)                           // the gather requested a phase ago lands
                // (the answered first look falls straight through: as a `while` the compiler's layout put three taken branches on it)
                if (__builtin_expect(!all_arrived(epoch), 0)) {
                    unsigned spins = gave_up ? p.spin_limit : 0u;
                    do {
                        if ((spins += 256) > p.spin_limit) { timed_out = true; break; }
                        __builtin_amdgcn_s_sleep(1);
                        request_again(src);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        read_out();
                    } while (!all_arrived(epoch));
                }
                // a limit of zero polls: no exchange may be waited for at all, and none counts as answered -- the first gather
                // reports a timeout whether its granules had arrived or not (with XCD-local publishing they always have: the
                // tests that force the timeout path need it to be deterministic)
                if (!STEADY && pb == 0 && (p.spin_limit >> 8) == 0u) timed_out = true;
                if (!STEADY && pb == 0 && p.allow_local && !timed_out) {
                    // the first gather names every member's XCD: all on this one -> the publisher may store plainly
                    bool elsewhere = want && ((ghi[0] >> 28) != my_xcc);
                    if constexpr (GPL == 4) {
#pragma unroll
                        for (int i = 1; i < 4; ++i) elsewhere |= want && ((ghi[i] >> 28) != my_xcc);
                    }
                    if (__builtin_amdgcn_ballot_w64(elsewhere) == 0 && lane == 0) *local_flag = __int_as_float(1);
                }
            }
            GPFQ_PSTAMP(2)                               // re-polls (granules that had not arrived)
            // ---- the gather of the NEXT phase, requested as soon as the window pair is free again: the group swept two
            // phases ago, published by every member at the top of its phase ph - 1; the load travels under this phase's
            // quantizer and the barrier
            if (STEADY || (ph >= 2 && ph - 2 < nph)) {
                const int pn = ph - 2, gn = pn & 3, tn = pn >> 2;
                request(xb + (unsigned)(tn & 1) * (unsigned)(C * RT) + (unsigned)(gn * RG));
            }
            GPFQ_PSTAMP(3)                               // the request for the next phase
            if (gathers) {
                float v = __uint_as_float(glo[0]);
                if constexpr (GPL == 4) v = (v + __uint_as_float(glo[1])) + (__uint_as_float(glo[2]) + __uint_as_float(glo[3]));
                v = want ? v : 0.0f;
                v = wave_tree16_zero_padded(v);
                if (sh > 4) v = xor16_add(v);
                if (sh > 5) v = xor32_add(v);
                const int rr = gb * RG + (lead ? grow : 0);      // row of the tile this lane quantizes
                const bool rvalid = lead && (row0 + rr < p.Ng);
                const int64_t growl = (int64_t)row0 + (rvalid ? rr : 0);
                int id;
                float q;
                bool redo = false;
                auto divide_and_quantize = [&]() {
                    const float sarg = (n2cur > 0.0f) ? v / n2cur : 0.0f;
                    q = quantize_mode<MODE>(p, sarg, p.row_id0 + (uint64_t)growl, (uint64_t)tb, id);
                };
                if (FAST) redo = !quant_msq_from_dot(v, in2cur, p.inv_step, p.step, p.Kf, p.msq_thr, unused, q, id);
                else divide_and_quantize();
                auto commit = [&]() {
                    if (lead) {
                        qs[rr] = q;
                        hist[rr * 64 + (tb & 63)] = q;
                        hist[(RT + rr) * 64 + (tb & 63)] = __int_as_float(id);
                    }
                };
                commit();
                if (FAST && __builtin_expect(redo, 0)) {
                    divide_and_quantize();
                    commit();
                }
                // Q / idx of this group leave through the 64-step history, one coalesced store per row (see reducer_section)
                if (__builtin_expect((tb & 63) == 63 || tb + 1 == p.d, 0)) {
                    const int t0 = tb & ~63;
                    const int n = tb - t0 + 1;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    const int ln = fresh_lane_id();
                    if (c == 0 && ln < n) {
#pragma unroll
                        for (int r = 0; r < RG; ++r) {
                            const int row = gb * RG + r;
                            if (row0 + row < p.Ng) {
                                const int64_t gw = (int64_t)row0 + row;
                                p.Q[gw * p.ldq + t0 + ln] = hist[row * 64 + ln];
                                if (p.idx) {
                                    const int iv = __float_as_int(hist[(RT + row) * 64 + ln]);
                                    if (p.idx_bytes == 1) reinterpret_cast<int8_t*>(p.idx)[gw * p.ldi + t0 + ln] = (int8_t)iv;
                                    else reinterpret_cast<int16_t*>(p.idx)[gw * p.ldi + t0 + ln] = (int16_t)iv;
                                }
                            }
                        }
                    }
                }
                if (__builtin_expect(timed_out && !gave_up, 0) && lane == 0) {
                    atomicExch(p.status, 1);
                    p.status[1] = tb; p.status[2] = tile; p.status[3] = c;
                }
                gave_up |= timed_out;
            }
            GPFQ_PSTAMP(4)                               // tree over the members, quantizer, q into LDS, rare flush
        }
        pipe_barrier();
    };
    {
        int ph = 0;
        const int fill = nph < 5 ? nph + 3 : 5;          // phases 0 .. 4: the flag is read in phase 4 (written in phase 3)
        for (; ph < fill; ++ph) phase(std::false_type{}, ph);
        if (PUB && !local && ph >= 5) local = __builtin_amdgcn_readfirstlane(__float_as_int(*local_flag)) != 0;
        for (; ph < nph; ++ph) phase(std::true_type{}, ph);
        for (; ph < nph + 3; ++ph) phase(std::false_type{}, ph);
    }
#ifdef GPFQ_STAMPS
    if (blockIdx.x == 0 && lane == 0 && p.status) {
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(p.status + 16);
        if constexpr (ROLE == 0) { dbg[6] = pst_sum[0]; dbg[7] = pst_sum[1]; }      // (the publisher: barrier, part (a))
        else for (int i = 0; i < 6; ++i) dbg[i] = pst_sum[i];
    }
#endif
}

// WB = first register of the window: three x buffers, two a buffers, then the RT = 4 RG residual rows (RG = 2: every group
// an interleaved pair, win_sweep16_pair).
// SINGLE: one reducer wave plays both roles, leaving SEVEN sweep waves (rows whose members need a seventh segment: every
// 1 x 1 convolution of ResNet-50 at batch 1024 has 12.3-12.5 segments per member at a power-of-two member count, i.e. 6.1-6.25
// at twice as many); otherwise publisher and gatherer are waves of their own beside up to six sweep waves.
template <int RG, int MODE, int WB, bool SINGLE, int GPL = 1>
__device__ __forceinline__ void coop_pipe_body(const SlabParams& p)
{
    static_assert(RG == 1 || RG == 2, "groups of one row or of one interleaved pair");
    constexpr int G = 4, RT = G * RG;
    constexpr int X0 = WB, X1 = WB + 16, X2 = WB + 32, A0 = WB + 48, A1 = WB + 64, U0 = WB + 80;
    extern __shared__ float smem[];                 // seg[RT][NS], qs[RT], history [2 RT][64]
    const int NW = blockDim.x >> 6;                 // sweep waves + the publisher wave + the gatherer wave (SINGLE: one wave for both)
    const int NS = NW - (SINGLE ? 1 : 2);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int S = p.S, C = p.C;
    const int P = pow2_ceil(S);
    if (*static_cast<volatile const int*>(p.status) != 0) return;      // (a layer in rounds stops at the first timed-out launch)
    int tile, c;
    if (p.xcd_tiles) {
        // The members of a row tile on one XCD (workgroups b and b + 8 share one under round-robin dispatch; the kernel
        // verifies it before it relies on it, pipe_reducer).  The host pads the grid to a multiple of eight tiles where the
        // chip has the room: the workgroups of the tiles that do not exist leave at once -- nobody waits for them.
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        tile = (j / C) * 8 + xcd;
        c = j % C;
        if (tile >= p.tiles) return;
    } else {
        tile = blockIdx.x / C;
        c = blockIdx.x % C;
    }
    const int seg_lo = (c * S + C - 1) / C, seg_hi = ((c + 1) * S + C - 1) / C;
    const int n_own = seg_hi - seg_lo;              // <= NS
    const bool active = wave < n_own;
    const int myseg = seg_lo + (active ? wave : 0);
    const int nl = P / C;

    float* segs = smem;                             // [RT][NS]
    float* qs = smem + RT * NS;                     // [RT], then the Q / idx history [2 RT][64], then the locality flag
    float* local_flag = qs + RT + 2 * RT * 64;
    const int row0 = tile * RT;
    if (threadIdx.x < RT) qs[threadIdx.x] = 0.0f;   // q_{-1} = 0
    if (threadIdx.x == 0) *local_flag = 0.0f;
    pipe_barrier();
    if (wave >= NS) {
        if constexpr (SINGLE) pipe_reducer<RG, MODE, X0, 2, GPL>(p, segs, qs, local_flag, NS, lane, tile, c, C, nl, seg_lo, row0);
        else if (wave == NS) pipe_reducer<RG, MODE, X0, 0>(p, segs, qs, local_flag, NS, lane, tile, c, C, nl, seg_lo, row0);
        else pipe_reducer<RG, MODE, X0, 1>(p, segs, qs, local_flag, NS, lane, tile, c, C, nl, seg_lo, row0);
        return;
    }

    const float* xnext = uniform_ptr(p.XT + (int64_t)myseg * kSeg);    // column t+1 while step t runs
    const float* anext = uniform_ptr(p.AT + (int64_t)myseg * kSeg);
    const unsigned lane_off = 16u * (unsigned)lane;
    const kfloat* wrow[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const int64_t gr = (int64_t)row0 + ((row0 + r < p.Ng) ? r : (p.Ng - 1 - row0));   // rows past the end repeat the last one
        wrow[r] = as_scalar(p.W + gr * p.ldw);
    }
    // the residual starts at 0 (a non-zero initial residual is the streaming plan's job); every buffer starts defined
    win_zero16<U0>(); win_zero16<U0 + 16>(); win_zero16<U0 + 32>(); win_zero16<U0 + 48>();
    if constexpr (RG == 2) { win_zero16<U0 + 64>(); win_zero16<U0 + 80>(); win_zero16<U0 + 96>(); win_zero16<U0 + 112>(); }
    win_zero16<X0>(); win_zero16<X1>(); win_zero16<X2>(); win_zero16<A0>(); win_zero16<A1>();
    if (active) {                                   // x_0 -> X0, a_0 -> A0; X2 = x_{-1} = 0 (q_{-1} = 0)
        win_load16<X0>(xnext, lane_off);
        win_load16<A0>(anext, lane_off);
    }
    float wn[RG];                                   // weights of the NEXT phase's group, fetched a phase ahead
#pragma unroll
    for (int r = 0; r < RG; ++r) wn[r] = wrow[r][0];
    int t = 0;
    const int dlast = p.d - 1;
    GPFQ_PSTAMP_DECL

    // one phase: group g of step t; XP holds x_{t-1}, XC x_t, AC a_t; XN / AN take column t+1
    auto phase = [&](auto xp_, auto xc_, auto ac_, auto xn_, auto an_, auto g_) {
        constexpr int XP = decltype(xp_)::value, XC = decltype(xc_)::value, AC = decltype(ac_)::value;
        constexpr int XN = decltype(xn_)::value, AN = decltype(an_)::value, g = decltype(g_)::value;
        constexpr int UG = U0 + 16 * RG * g;
        GPFQ_PSTAMP(0)                                   // the barrier
        float q[RG], w[RG];
#pragma unroll
        for (int r = 0; r < RG; ++r) { q[r] = qs[g * RG + r]; w[r] = wn[r]; }
        if constexpr (g == 0) {
            // column t+1 (the last step re-reads its own rather than branch): the pointers advance in every wave
            const int64_t adv = (t + 1 < p.d) ? p.m_pad : 0;
            xnext += adv;
            anext += adv;
        }
        if (active) {
            if constexpr (g == 0) win_wait<0>();    // column t has landed (requested in phases 0 and 1 of step t-1)
            GPFQ_PSTAMP(1)                               // q from LDS, the wait for column t
            if constexpr (RG == 1) {
                const float acc = win_sweep16<UG, XP, AC, XC>(q[0], w[0]);
                GPFQ_PSTAMP(2)                           // the sweep
                const float sg = wave_tree64_lane63(acc);
                if (lane == 63) segs[g * NS + wave] = sg;
            } else {
                const v2f a01 = win_sweep16_pair<UG, XP, AC, XC>(q[0], q[1], w[0], w[1]);
                GPFQ_PSTAMP(2)
                const float acc[2] = {a01.x, a01.y};
                const float tot = wave_tree64_rows<2>(acc);
                if ((lane & 15) == 0 && (lane >> 4) < 2) segs[(g * 2 + (lane >> 4)) * NS + wave] = tot;
            }
            GPFQ_PSTAMP(3)                               // lane tree, LDS word
            if constexpr (g < 2) {                  // half of column t+1 per phase: four requests, not eight, at a time
                win_load4<XN, 2 * g>(xnext, lane_off);
                win_load4<AN, 2 * g>(anext, lane_off);
                win_load4<XN, 2 * g + 1>(xnext, lane_off);
                win_load4<AN, 2 * g + 1>(anext, lane_off);
            }
        }
        // the next phase's weights through the scalar cache (group g+1 of this step, or group 0 of the next; the last step
        // re-reads its own), requested behind the sweep: a scalar load shares lgkmcnt with the LDS read of q above
        {
            constexpr int gn = (g + 1) & 3;
            unsigned tn4 = 4u * (unsigned)(g == 3 ? (t < dlast ? t + 1 : dlast) : t);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" : "+s"(tn4)::"memory");
#pragma unroll
            for (int r = 0; r < RG; ++r) wn[r] = sload(wrow[gn * RG + r], tn4);
        }
        GPFQ_PSTAMP(4)                                   // column requests, weight requests
        pipe_barrier();
    };
    auto step = [&](auto xp_, auto xc_, auto ac_, auto xn_, auto an_) -> bool {
        phase(xp_, xc_, ac_, xn_, an_, std::integral_constant<int, 0>{});
        phase(xp_, xc_, ac_, xn_, an_, std::integral_constant<int, 1>{});
        phase(xp_, xc_, ac_, xn_, an_, std::integral_constant<int, 2>{});
        phase(xp_, xc_, ac_, xn_, an_, std::integral_constant<int, 3>{});
        return ++t < p.d;
    };
    using I0 = std::integral_constant<int, X0>; using I1 = std::integral_constant<int, X1>; using I2 = std::integral_constant<int, X2>;
    using J0 = std::integral_constant<int, A0>; using J1 = std::integral_constant<int, A1>;
    int k = 0;                                       // buffer that holds x_{d-1} when the loop ends
    for (;;) {
        k = 0; if (!step(I2{}, I0{}, J0{}, I1{}, J1{})) break;
        k = 1; if (!step(I0{}, I1{}, J1{}, I2{}, J0{})) break;
        k = 2; if (!step(I1{}, I2{}, J0{}, I0{}, J1{})) break;
        k = 0; if (!step(I2{}, I0{}, J1{}, I1{}, J0{})) break;
        k = 1; if (!step(I0{}, I1{}, J0{}, I2{}, J1{})) break;
        k = 2; if (!step(I1{}, I2{}, J1{}, I0{}, J0{})) break;
    }
    // the reducer is three phases behind the last sweep: its last publishes, gathers and quantizers
    pipe_barrier(); pipe_barrier(); pipe_barrier();
    win_wait<0>();                                   // every load issued above has landed before the wave ends
#ifdef GPFQ_STAMPS
    if (wave == 0) { GPFQ_PSTAMP_DUMP(8) }
#endif
    if (!active) return;
    float qlast[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) qlast[r] = qs[r];
    auto finish = [&](auto xl_) {
        constexpr int XL = decltype(xl_)::value;
        if constexpr (RG == 1) {
            finish_row_w<U0, XL>(p, qlast[0], row0 < p.Ng, (int64_t)row0, myseg, lane);
            finish_row_w<U0 + 16, XL>(p, qlast[1], row0 + 1 < p.Ng, (int64_t)row0 + 1, myseg, lane);
            finish_row_w<U0 + 32, XL>(p, qlast[2], row0 + 2 < p.Ng, (int64_t)row0 + 2, myseg, lane);
            finish_row_w<U0 + 48, XL>(p, qlast[3], row0 + 3 < p.Ng, (int64_t)row0 + 3, myseg, lane);
        } else {                                     // interleaved pairs: row 2 g + h sits at U0 + 32 g + h, stride 2
            finish_row_w<U0, XL, 2>(p, qlast[0], row0 < p.Ng, (int64_t)row0, myseg, lane);
            finish_row_w<U0 + 1, XL, 2>(p, qlast[1], row0 + 1 < p.Ng, (int64_t)row0 + 1, myseg, lane);
            finish_row_w<U0 + 32, XL, 2>(p, qlast[2], row0 + 2 < p.Ng, (int64_t)row0 + 2, myseg, lane);
            finish_row_w<U0 + 33, XL, 2>(p, qlast[3], row0 + 3 < p.Ng, (int64_t)row0 + 3, myseg, lane);
            finish_row_w<U0 + 64, XL, 2>(p, qlast[4], row0 + 4 < p.Ng, (int64_t)row0 + 4, myseg, lane);
            finish_row_w<U0 + 65, XL, 2>(p, qlast[5], row0 + 5 < p.Ng, (int64_t)row0 + 5, myseg, lane);
            finish_row_w<U0 + 96, XL, 2>(p, qlast[6], row0 + 6 < p.Ng, (int64_t)row0 + 6, myseg, lane);
            finish_row_w<U0 + 97, XL, 2>(p, qlast[7], row0 + 7 < p.Ng, (int64_t)row0 + 7, myseg, lane);
        }
    };
    if (k == 0) finish(I0{});
    else if (k == 1) finish(I1{});
    else finish(I2{});
}

// <= 8 waves (6 sweep waves + the publisher + the gatherer; `s`: 7 sweep waves + one wave for both roles): 256 registers =
// window (80 columns + 64 / 128 rows) + 112 / 48 for the compiler
#define GPFQ_DEFINE_PIPE(RG, MODE, WB, SINGLE, SUFFIX)                                                            \
    __global__ void __launch_bounds__(64 * 8) __attribute__((amdgpu_num_vgpr(WB / 2)))                            \
    gpfq_pipe_rg##RG##_m##MODE##_w8##SUFFIX(const SlabParams p)                                                  \
    {                                                                                                             \
        asm volatile("" ::: "v255");                                                                              \
        coop_pipe_body<RG, MODE, WB, SINGLE>(p);                                                                  \
    }
GPFQ_DEFINE_PIPE(1, 0, 112, false, ) GPFQ_DEFINE_PIPE(1, 1, 112, false, ) GPFQ_DEFINE_PIPE(1, 2, 112, false, ) GPFQ_DEFINE_PIPE(1, 3, 112, false, )
GPFQ_DEFINE_PIPE(2, 0, 48, false, ) GPFQ_DEFINE_PIPE(2, 1, 48, false, ) GPFQ_DEFINE_PIPE(2, 2, 48, false, ) GPFQ_DEFINE_PIPE(2, 3, 48, false, )
GPFQ_DEFINE_PIPE(1, 0, 112, true, s) GPFQ_DEFINE_PIPE(1, 1, 112, true, s) GPFQ_DEFINE_PIPE(1, 2, 112, true, s) GPFQ_DEFINE_PIPE(1, 3, 112, true, s)
GPFQ_DEFINE_PIPE(2, 0, 48, true, s) GPFQ_DEFINE_PIPE(2, 1, 48, true, s) GPFQ_DEFINE_PIPE(2, 2, 48, true, s) GPFQ_DEFINE_PIPE(2, 3, 48, true, s)

// two rows x 128 members per gather: four granules per lane (one reducer wave, seven sweep waves)
#define GPFQ_DEFINE_PIPE_Q(MODE)                                                                                  \
    __global__ void __launch_bounds__(64 * 8) __attribute__((amdgpu_num_vgpr(48 / 2)))                            \
    gpfq_pipe_rg2_m##MODE##_w8sq(const SlabParams p)                                                             \
    {                                                                                                             \
        asm volatile("" ::: "v255");                                                                              \
        coop_pipe_body<2, MODE, 48, true, 4>(p);                                                                  \
    }
GPFQ_DEFINE_PIPE_Q(0) GPFQ_DEFINE_PIPE_Q(1) GPFQ_DEFINE_PIPE_Q(2) GPFQ_DEFINE_PIPE_Q(3)

}  // namespace gpfq
