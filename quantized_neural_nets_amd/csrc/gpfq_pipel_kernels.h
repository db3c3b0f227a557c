// gpfq_pipel_kernels.h -- the pipelined cooperative kernels with LDS-staged columns and TWELVE rows per workgroup (round 5):
// gpfq_pipel_m{0..3}_w8.
//
// Reference: StepAlgorithm._quantization, step_algorithm.py:107-148 -- the same recurrence, the same canonical arithmetic,
// the same granule exchange as gpfq_coop_* / gpfq_pipe_* (gpfq_loop_kernels.h, gpfq_pipe_kernels.h).  This family is for the
// layers that run in ROUNDS because their residual does not fit the chip's registers at once (every 1 x 1 convolution of
// ResNet-50 at calibration batch 1024: rows of 785 / 197 / 50 segments): what a round costs is one cooperative step per
// column, so what counts is (segment-rows a CU holds) / (time of a step).
//
//   * ROWS PER WAVE.  A CU runs 8 waves of 256 registers or 16 of 128 -- the same 512 KB.  The LDS-staged lock-step kernel
//     (gpfq_coop_rt4_*_w16l*) has 13 sweep waves x 4 rows = 52 segment-rows per CU (64 of a wave's 128 registers hold residual),
//     the pipelined register-window kernels (gpfq_pipe_rg2_*) 6-7 sweep waves x 8 rows = 48-56 (128 of 256; the column window
//     takes 80).  Here a sweep wave keeps TWELVE rows -- 192 of its 256 registers, the other 64 are the compiler's -- and the
//     columns live in LDS: 7 sweep waves x 12 rows = 84 segment-rows per CU, 1.5-1.6 x fewer rounds.
//   * FIVE column buffers per sweep wave in LDS (x ring of three, a ring of two: 20 KB per wave, 140 KB of the CU's 160), which
//     is what two or more row groups that alternate need (a group's pending update still reads x_{t-1} while the next column
//     is on its way) and what thirteen waves could not have (DESIGN.md 10.3 of round 4: 260 KB).  Column t+1 is requested by
//     global_load_lds_dwordx4 at the top of step t and waited for at the top of step t+1: a whole step of look-ahead.
//   * THREE row groups of FOUR rows (two interleaved pairs), three phases per step, one barrier each.  In phase p = 3 t + g the
//     sweep waves sweep group g with column t; the reducer wave (one wave, both roles) publishes the group swept in phase
//     p-1, consumes the gather of the group swept in phase p-2 (quantizer, q into LDS: one phase before that group's next
//     sweep) and, last, requests the gather of the group it has just published.  A phase is a sweep of two pairs by two waves per
//     SIMD -- longer than what an exchange needs (publish -> visible -> requested -> landed >= 1 250 cycles) -- so
//     the exchange hides under the arithmetic and the step is bound by the vector ALU's issue rate.
//   * GRANULES for up to 128 members (rows of 785 segments: 6.13 per member): xbuf[tile][parity][group][position][row], a
//     member's four rows of a group in 32 contiguous bytes, and positions ordered so that the gather is COALESCED and still adds
//     adjacent members first: lane (pair, j) takes 16 bytes -- two rows -- of members GPL j .. GPL j + GPL - 1 (GPL = members
//     per lane, C / 32), member GPL j + i at position 32 i + j, so that load i of the gather reads one contiguous KB and the lane
//     adds (g0 + g1) + (g2 + g3): the first levels of the canonical tree over the members; five DPP / permlane levels finish it.
//     (The 256-granule gather of gpfq_pipe_rg2_*_w8sq reads lines that hold all four groups, a quarter of each used, from 128
//     members at once: 4.65 us per column and round, profiles/NOTES.md round 4.)
//
// Built, bit-exact and measured SLOWER (round 5, not kept): two groups of six rows (a group's exchange then has one phase, not two)
// and LDS-word dataflow synchronisation instead of the barrier per phase -- profiles/r05_probe_pipel_two_groups.txt,
// profiles/r05_probe_pipel_dataflow.txt.
//
// Bit-exactness is untouched: per row the same sweep (win_sweep4_pair_lds), the same lane tree, the same slot tree split at the
// same member boundaries, the same quantizer -- only interleaved differently in time (tests/test_gpu_pipel.py).
#pragma once
#include "gpfq_pipe_kernels.h"

namespace gpfq {

constexpr int kPipelGroups = 3, kPipelGroupRows = 4, kPipelRows = 12;
constexpr int kPipelColFloats = 5 * 1024;      // a sweep wave's column buffers in LDS: x ring [3][1024], a ring [2][1024]

// LDS bytes of a workgroup with NS sweep waves (the launch passes exactly this)
__host__ __device__ constexpr size_t pipel_lds_bytes(int NS)
{
    return sizeof(float) * ((size_t)NS * kPipelColFloats + (size_t)kPipelRows * NS + kPipelRows + 2 * kPipelRows * 64 + 4);
}

// One column segment (1024 floats) of this wave into one of its LDS buffers: lane l brings elements 256 q + 4 l .. + 3 of
// quarter q.  Issued by asm, M0 and all: the compiler must not know that an LDS DMA is in flight, or it waits for it
// (s_waitcnt vmcnt(0)) in front of the next LDS read it cannot prove disjoint -- and the sweep that follows reads the OTHER
// buffers of the same array.  The wave waits for its own DMA itself, a step later (pipel_wait_columns).
__device__ __forceinline__ void pipel_dma(const float* gbase /* wave-uniform */, unsigned lane_off, unsigned lds_byte_addr)
{
    asm volatile("s_mov_b32 m0, %2\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %0, %1\n\t"
                 "global_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                 "global_load_lds_dwordx4 %0, %1 offset:2048\n\t"
                 "global_load_lds_dwordx4 %0, %1 offset:3072"
                 :: "v"(lane_off), "s"(gbase), "s"(lds_byte_addr) : "memory", "m0");     // (m0 is reserved: the compiler sets it in front of
}                                                                                         //  every use of its own and keeps nothing in it)
__device__ __forceinline__ void pipel_wait_columns() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// The reducer wave: 3 d + 2 phases.  Phase ph: (a) publish the group swept in phase ph-1; (b) consume the gather of the
// group swept in phase ph-2 (requested at the end of phase ph-1); (c) request the gather of the group just published.
// Lane layouts: (a) lane = 16 * row + slot (four rows of a group: all 64 lanes), as in reducer_section; (b) lane =
// 32 * pair + j: pair 0 = rows 0, 1 of the group in the two halves of a 16-byte load, pair 1 = rows 2, 3.
// GV = first register of the gather's landing zone in the window (the reducer sweeps nothing: the residual's registers
// are free in its wave): four loads of four registers.
template <int MODE>
__device__ __forceinline__ void pipel_reducer(const SlabParams& p, const float* segs, float* qs, int NS, int lane, int tile, int c,
                                              int C, int nl, int seg_lo, int row0)
{
    constexpr int G = kPipelGroups, RG = kPipelGroupRows, RT = kPipelRows, GV = 64;
    constexpr bool FAST = MODE == MODE_MSQ;
    const int nph = G * p.d;
    unsigned long long* const xb = p.xbuf + (unsigned)tile * (unsigned)(2 * G * RG) * (unsigned)C;
    const unsigned block = (unsigned)(C * RG);                 // granules of one (parity, group)
    // ---- publisher
    const int P = pow2_ceil(p.S);
    const SlotMap smap = make_slot_map(p.S, P, c * nl, 1, lane & 15, nl);
    const int r16 = lane >> 4;                                 // the row of the group this lane serves
    const bool mine = (smap.mask & 1u) != 0;
    const int seg_word = r16 * NS + (smap.s0 - seg_lo);
    const int GPL = C > 32 ? C >> 5 : 1;                       // members per lane of a gather: 1, 2 or 4 (C <= 128)
    const unsigned pos = (unsigned)(c % GPL) * 32u + (unsigned)(c / GPL);
    const unsigned pub_off = pos * RG + (unsigned)r16;
    const unsigned my_xcc = pipe_xcc_id();
    const unsigned tag = pipe_epoch_tag(p.salt);
    // ---- gatherer
    const kfloat* nrm = as_scalar(p.nrm2);
    float* hist = qs + RT;                                     // [RT][64] values, then [RT][64] indices (as int bits)
    const int pair = lane >> 5, j = lane & 31;
    const int lpr = C / GPL;                                   // lanes per pair of rows, <= 32
    const bool want = j < lpr;
    const bool lead = (lane & 15) == 0;                        // lane 16 r quantizes row r of the group
    const unsigned long long idle = ~__builtin_amdgcn_ballot_w64(want);
    const unsigned long long unused = ~__builtin_amdgcn_ballot_w64(lead);
    const unsigned lane_bytes = want ? 32u * (unsigned)j + 16u * (unsigned)pair : 0u;
    auto request = [&](const unsigned long long* base) {
        // (s_nop 4: the base may come straight out of a spill slot -- v_readlane, a VALU write of the SGPR pair, which an asm
        // VMEM instruction may read only five wait states later; the compiler pads its own loads, not ours)
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 v[%c2:%c2+3], %0, %1 sc1" :: "v"(lane_bytes), "s"(base), "n"(GV) : "memory");
        if (GPL >= 2)
            asm volatile("s_nop 4\n\tglobal_load_dwordx4 v[%c2:%c2+3], %0, %1 offset:1024 sc1" :: "v"(lane_bytes), "s"(base), "n"(GV + 4) : "memory");
        if (GPL >= 4) {
            asm volatile("s_nop 4\n\tglobal_load_dwordx4 v[%c2:%c2+3], %0, %1 offset:2048 sc1" :: "v"(lane_bytes), "s"(base), "n"(GV + 8) : "memory");
            asm volatile("s_nop 4\n\tglobal_load_dwordx4 v[%c2:%c2+3], %0, %1 offset:3072 sc1" :: "v"(lane_bytes), "s"(base), "n"(GV + 12) : "memory");
        }
    };
    unsigned glo0[4], ghi0[4], glo1[4], ghi1[4];               // [load]: first / second row of the pair, value and epoch word
    auto read_out = [&]() {
#define GPFQ_RDL(i)                                                                                                          \
        asm volatile("v_mov_b32 %0, v[%c4]\n\tv_mov_b32 %1, v[%c4+1]\n\tv_mov_b32 %2, v[%c4+2]\n\tv_mov_b32 %3, v[%c4+3]\n\ts_nop 0" \
                     : "=v"(glo0[i]), "=v"(ghi0[i]), "=v"(glo1[i]), "=v"(ghi1[i]) : "n"(GV + 4 * i) : "memory");
        GPFQ_RDL(0) GPFQ_RDL(1) GPFQ_RDL(2) GPFQ_RDL(3)
#undef GPFQ_RDL
    };
    auto all_arrived = [&](unsigned epoch) {
        unsigned long long ok = __builtin_amdgcn_ballot_w64((ghi0[0] & 0x0fffffffu) == epoch) &
                                __builtin_amdgcn_ballot_w64((ghi1[0] & 0x0fffffffu) == epoch);
        if (GPL >= 2)
            ok &= __builtin_amdgcn_ballot_w64((ghi0[1] & 0x0fffffffu) == epoch) & __builtin_amdgcn_ballot_w64((ghi1[1] & 0x0fffffffu) == epoch);
        if (GPL >= 4) {
            ok &= __builtin_amdgcn_ballot_w64((ghi0[2] & 0x0fffffffu) == epoch) & __builtin_amdgcn_ballot_w64((ghi1[2] & 0x0fffffffu) == epoch);
            ok &= __builtin_amdgcn_ballot_w64((ghi0[3] & 0x0fffffffu) == epoch) & __builtin_amdgcn_ballot_w64((ghi1[3] & 0x0fffffffu) == epoch);
        }
        return (ok | idle) == __builtin_amdgcn_read_exec();
    };
    bool gave_up = false;
    // the reducer's issue priority (the host's choice, launch_pipel): a short dependent chain among long sweeps -- but it shares
    // its SIMD with a sweep wave, whose instructions it then delays
    if (p.reducer_prio == 3) __builtin_amdgcn_s_setprio(3);
    else if (p.reducer_prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (p.reducer_prio == 1) __builtin_amdgcn_s_setprio(1);
    // the running (group, column) of the three parts: (a) / (c) the group swept in phase ph-1, (b) the one swept in phase ph-2
    int ga = 0, ta = 0, gb = 0, tb = 0;
    GPFQ_PSTAMP_DECL
    for (int ph = 0; ph < nph + 2; ++ph) {
        const bool publishes = ph >= 1 && ph <= nph;
        const bool consumes = ph >= 2;
        GPFQ_PSTAMP(0)                                       // the barrier
        if (publishes) {
            // ---- (a) this member's block of the slot tree for the four rows of the group, published
            const float val = segs[ga * RG * NS + seg_word];
            const float v = wave_tree16_zero_padded(mine ? val : 0.0f);
            unsigned long long* dst = xb + (unsigned)((ta & 1) * G + ga) * block;
            const unsigned long long granule = ((unsigned long long)(tag | (my_xcc << 28) | ((unsigned)ta + 1u)) << 32) |
                                               (unsigned long long)__float_as_uint(v);
            if ((lane & 15) == 0) __hip_atomic_store(dst + pub_off, granule, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        GPFQ_PSTAMP(1)                                       // slot tree + publish
        if (consumes) {
            // ---- (b) the group swept two phases ago: its gather (requested at the end of the phase before) lands
            const unsigned epoch = tag | ((unsigned)tb + 1u);
            const float n2cur = sload(nrm, 8u * (unsigned)tb);
            const float in2cur = sload(nrm, 8u * (unsigned)tb + 4u);
            const unsigned long long* src = xb + (unsigned)((tb & 1) * G + gb) * block;
            bool timed_out = false;
            if (publishes) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");      // (exactly one younger operation: (a)'s store)
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            read_out();
            GPFQ_PSTAMP(2)                                   // the gather requested a phase ago lands
            if (__builtin_expect(!all_arrived(epoch), 0)) {
                unsigned spins = gave_up ? p.spin_limit : 0u;
                do {
                    if ((spins += 256) > p.spin_limit) { timed_out = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                    request(src);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    read_out();
                } while (!all_arrived(epoch));
            }
            // (a limit of zero polls: the first gather reports a timeout whether its granules had arrived or not -- the tests
            // that force the timeout path need it to be deterministic)
            if (ph == 2 && (p.spin_limit >> 8) == 0u) timed_out = true;
            GPFQ_PSTAMP(3)                                   // re-polls
            // the lane's members, adjacent blocks of the slot tree: (g0 + g1) + (g2 + g3); loads beyond GPL hold nothing
            float x0 = __uint_as_float(glo0[0]), y0 = __uint_as_float(glo1[0]);
            float x1 = GPL >= 2 ? __uint_as_float(glo0[1]) : 0.0f, y1 = GPL >= 2 ? __uint_as_float(glo1[1]) : 0.0f;
            float x2 = GPL >= 4 ? __uint_as_float(glo0[2]) : 0.0f, y2 = GPL >= 4 ? __uint_as_float(glo1[2]) : 0.0f;
            float x3 = GPL >= 4 ? __uint_as_float(glo0[3]) : 0.0f, y3 = GPL >= 4 ? __uint_as_float(glo1[3]) : 0.0f;
            float x = (x0 + x1) + (x2 + x3), y = (y0 + y1) + (y2 + y3);
            x = want ? x : 0.0f;
            y = want ? y : 0.0f;
            // the tree over the 32 lanes of a pair, both rows: every lane of the half ends with the row's sum
            x = xor16_add(wave_tree16_zero_padded(x));
            y = xor16_add(wave_tree16_zero_padded(y));
            const float v = (lane & 16) ? y : x;                // lane row r (lane >> 4) holds row r of the group
            const int rr = gb * RG + (lane >> 4);               // row of the tile
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
            if (++gb == G) { gb = 0; ++tb; }
        }
        GPFQ_PSTAMP(4)                                       // tree over the members, quantizer, q into LDS, rare flush
        if (publishes) {
            // ---- (c) the gather of the group published above, requested at the END of this phase: the other members publish
            // it in this same phase of theirs, and what is asked for before it is visible is asked for twice
            // (the pause: low five bits of the spin limit, units of 64 clocks -- the host's choice, launch_pipel)
            {
                unsigned w = p.spin_limit & 31u;
                asm volatile("s_cmp_eq_u32 %0, 0\n\t"
                             "s_cbranch_scc1 2f\n"
                             "1:\n\t"
                             "s_sleep 1\n\t"
                             "s_sub_u32 %0, %0, 1\n\t"
                             "s_cmp_lg_u32 %0, 0\n\t"
                             "s_cbranch_scc1 1b\n"
                             "2:" : "+s"(w) :: "scc", "memory");
            }
            request(xb + (unsigned)((ta & 1) * G + ga) * block);
            if (++ga == G) { ga = 0; ++ta; }
        }
        GPFQ_PSTAMP(5)                                       // the pause and the request
        pipe_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GPFQ_PSTAMP_DUMP(0)
}

template <int MODE>
__device__ __forceinline__ void coop_pipel_body(const SlabParams& p)
{
    constexpr int G = kPipelGroups, RG = kPipelGroupRows, RT = kPipelRows, U0 = 64;   // window = the twelve residual rows
    extern __shared__ __attribute__((aligned(16))) float smem[];   // cols [NS][5][1024], segs [RT][NS], qs [RT], history [2 RT][64]
    const int NW = blockDim.x >> 6;                 // sweep waves + the reducer wave
    const int NS = NW - 1;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int S = p.S, C = p.C;
    const int P = pow2_ceil(S);
    if (*static_cast<volatile const int*>(p.status) != 0) return;      // (a layer in rounds stops at the first timed-out launch)
    const int tile = blockIdx.x / C, c = blockIdx.x % C;
    const int seg_lo = (c * S + C - 1) / C, seg_hi = ((c + 1) * S + C - 1) / C;
    const int n_own = seg_hi - seg_lo;              // <= NS
    const bool active = wave < n_own;
    const int myseg = seg_lo + (active ? wave : 0);
    const int nl = P / C;

    float* cols = smem + (size_t)(wave < NS ? wave : 0) * kPipelColFloats;   // this wave's x ring [3][1024], then its a ring [2][1024]
    float* segs = smem + (size_t)NS * kPipelColFloats;                        // [RT][NS]
    float* qs = segs + RT * NS;                                               // [RT], then the Q / idx history [2 RT][64]
    const int row0 = tile * RT;
    if (threadIdx.x < RT) qs[threadIdx.x] = 0.0f;   // q_{-1} = 0
    pipe_barrier();
    if (wave == NS) {
        pipel_reducer<MODE>(p, segs, qs, NS, lane, tile, c, C, nl, seg_lo, row0);
        return;
    }

    const kfloat* wrow[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const int64_t gr = (int64_t)row0 + ((row0 + r < p.Ng) ? r : (p.Ng - 1 - row0));   // rows past the end repeat the last one
        wrow[r] = as_scalar(p.W + gr * p.ldw);
    }
    // the residual starts at 0 (a non-zero initial residual is the streaming plan's job)
    win_zero16<U0>(); win_zero16<U0 + 16>(); win_zero16<U0 + 32>(); win_zero16<U0 + 48>();
    win_zero16<U0 + 64>(); win_zero16<U0 + 80>(); win_zero16<U0 + 96>(); win_zero16<U0 + 112>();
    win_zero16<U0 + 128>(); win_zero16<U0 + 144>(); win_zero16<U0 + 160>(); win_zero16<U0 + 176>();

    const unsigned lane_off = 16u * (unsigned)lane;
    const unsigned cols_lds = (unsigned)(uintptr_t)cols;       // LDS byte address of this wave's buffers (an LDS pointer's low word)
    const float* xg = p.XT + (int64_t)myseg * kSeg;            // wave-uniform (the wave number is a scalar)
    const float* ag = p.AT + (int64_t)myseg * kSeg;
    if (active) {
        // x_0 -> X[0], a_0 -> A[0]; X[2] = x_{-1} = 0 (q_{-1} = 0)
        const float4 z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) *reinterpret_cast<float4*>(cols + 2048 + 256 * q4 + 4 * lane) = z;
        pipel_dma(xg, lane_off, cols_lds);
        pipel_dma(ag, lane_off, cols_lds + 3u * 4096u);
    }
    float wn[RG];                                   // weights of the NEXT phase's group, fetched a phase ahead
#pragma unroll
    for (int r = 0; r < RG; ++r) wn[r] = wrow[r][0];
    int t = 0;
    int xi = 0;                                     // t % 3: x_t lives in X[xi], x_{t-1} in X[(xi + 2) % 3], column t+1 goes to X[(xi + 1) % 3]
    const int dlast = p.d - 1;
    GPFQ_PSTAMP_DECL

    // one phase: group g of step t
    auto phase = [&](auto g_) {
        constexpr int g = decltype(g_)::value;
        constexpr int UG = U0 + 16 * RG * g;
        GPFQ_PSTAMP(0)                                   // the barrier
        const float4 qv = *reinterpret_cast<const float4*>(qs + RG * g);
        const float w0 = wn[0], w1 = wn[1], w2 = wn[2], w3 = wn[3];
        const int xn = xi == 2 ? 0 : xi + 1, xpi = xi == 0 ? 2 : xi - 1;
        if constexpr (g == 0) {
            // column t+1 (the last step re-reads its own rather than branch): the pointers advance in every wave
            const int64_t adv = (t + 1 < p.d) ? p.m_pad : 0;
            xg += adv;
            ag += adv;
            if (active) {
                pipel_wait_columns();               // column t has landed (requested at the top of step t-1)
                // ... and column t+1 goes into the buffers step t-1 has read for the last time (x_{t-2}'s and a_{t-1}'s)
                pipel_dma(xg, lane_off, cols_lds + (unsigned)xn * 4096u);
                pipel_dma(ag, lane_off, cols_lds + (3u + (unsigned)((t + 1) & 1)) * 4096u);
            }
        }
        GPFQ_PSTAMP(1)                                   // q from LDS, the wait for column t, the requests for column t+1
        if (active) {
            const float* xc = cols + xi * 1024;          // x_t
            const float* xp = cols + xpi * 1024;         // x_{t-1}
            const float* ab = cols + (3 + (t & 1)) * 1024;   // a_t
            v2f acc01 = {0.0f, 0.0f}, acc23 = {0.0f, 0.0f};
            const v2f qq01 = {qv.x, qv.y}, ww01 = {w0, w1};
            const v2f qq23 = {qv.z, qv.w}, ww23 = {w2, w3};
            // the LDS reads of quarter q + 1 are issued before the arithmetic of quarter q
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
                win_sweep4_pair_lds<UG + 8 * cq>(acc01, qq01, ww01, p01, p23, a01, a23, x01, x23);
                win_sweep4_pair_lds<UG + 32 + 8 * cq>(acc23, qq23, ww23, p01, p23, a01, a23, x01, x23);
                __builtin_amdgcn_sched_barrier(0);
                x4 = nx; a4 = na; p4 = np;
            };
            quarter(std::integral_constant<int, 0>{});
            quarter(std::integral_constant<int, 1>{});
            quarter(std::integral_constant<int, 2>{});
            quarter(std::integral_constant<int, 3>{});
            GPFQ_PSTAMP(2)                               // the sweep of two pairs (LDS reads included)
            const float acc[4] = {acc01.x, acc01.y, acc23.x, acc23.y};
            const float tot = wave_tree64_rows<4>(acc);  // row r's total in lane row r
            if ((lane & 15) == 0) segs[(g * RG + (lane >> 4)) * NS + wave] = tot;
            GPFQ_PSTAMP(3)                               // lane tree, LDS word
        }
        // the next phase's weights through the scalar cache (group g+1 of this step, or group 0 of the next; the last step
        // re-reads its own), requested behind the sweep: a scalar load shares lgkmcnt with the LDS reads above
        {
            constexpr int gn = (g + 1) % G;
            unsigned tn4 = 4u * (unsigned)(g == G - 1 ? (t < dlast ? t + 1 : dlast) : t);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" : "+s"(tn4)::"memory");
#pragma unroll
            for (int r = 0; r < RG; ++r) wn[r] = sload(wrow[gn * RG + r], tn4);
        }
        GPFQ_PSTAMP(4)                                   // weight requests
        pipe_barrier();
    };
    for (;;) {
        phase(std::integral_constant<int, 0>{});
        phase(std::integral_constant<int, 1>{});
        phase(std::integral_constant<int, 2>{});
        if (++t >= p.d) break;
        xi = xi == 2 ? 0 : xi + 1;
    }
    // the reducer is two phases behind the last sweep: its last publish, gathers and quantizers
    pipe_barrier(); pipe_barrier();
    pipel_wait_columns();                            // every DMA issued above has landed before the wave ends
#ifdef GPFQ_STAMPS
    if (wave == (p.pace >= 100 ? p.pace - 100 : 0)) { GPFQ_PSTAMP_DUMP(8) }      // (GPFQ_COOP_PACE=100+w: sweep wave w reports)
#endif
    if (!active) return;
    {   // the pending subtraction of the last step: u -= q_{d-1} x_{d-1}, x_{d-1} from its ring buffer (xi is still (d-1) % 3)
        const float* xl = cols + xi * 1024;
        const float4 qa = *reinterpret_cast<const float4*>(qs), qb = *reinterpret_cast<const float4*>(qs + 4),
                     qc = *reinterpret_cast<const float4*>(qs + 8);
        const v2f q01 = {qa.x, qa.y}, q23 = {qa.z, qa.w}, q45 = {qb.x, qb.y}, q67 = {qb.z, qb.w}, q89 = {qc.x, qc.y}, qab = {qc.z, qc.w};
        auto quarter = [&](auto c_) {
            constexpr int cq = decltype(c_)::value;
            const float4 x4 = *reinterpret_cast<const float4*>(xl + 256 * cq + 4 * lane);
            const v2f x01 = {x4.x, x4.y}, x23 = {x4.z, x4.w};
            win_final_sub4_pair_lds<U0 + 8 * cq>(q01, x01, x23);
            win_final_sub4_pair_lds<U0 + 32 + 8 * cq>(q23, x01, x23);
            win_final_sub4_pair_lds<U0 + 64 + 8 * cq>(q45, x01, x23);
            win_final_sub4_pair_lds<U0 + 96 + 8 * cq>(q67, x01, x23);
            win_final_sub4_pair_lds<U0 + 128 + 8 * cq>(q89, x01, x23);
            win_final_sub4_pair_lds<U0 + 160 + 8 * cq>(qab, x01, x23);
        };
        quarter(std::integral_constant<int, 0>{});
        quarter(std::integral_constant<int, 1>{});
        quarter(std::integral_constant<int, 2>{});
        quarter(std::integral_constant<int, 3>{});
    }
    // interleaved pairs: row 2 k + h sits at U0 + 32 k + h, stride 2
#define GPFQ_FIN(k)                                                                                                          \
    finish_row_w<U0 + 32 * k, 0, 2, false>(p, 0.0f, row0 + 2 * k < p.Ng, (int64_t)row0 + 2 * k, myseg, lane);               \
    finish_row_w<U0 + 32 * k + 1, 0, 2, false>(p, 0.0f, row0 + 2 * k + 1 < p.Ng, (int64_t)row0 + 2 * k + 1, myseg, lane);
    GPFQ_FIN(0) GPFQ_FIN(1) GPFQ_FIN(2) GPFQ_FIN(3) GPFQ_FIN(4) GPFQ_FIN(5)
#undef GPFQ_FIN
}

// <= 8 waves (7 sweep waves + the reducer): 256 registers = the window's 192 (twelve residual rows) + 64 for the compiler
#define GPFQ_DEFINE_PIPEL(MODE)                                                                                   \
    __global__ void __launch_bounds__(64 * 8) __attribute__((amdgpu_num_vgpr(64 / 2)))                            \
    gpfq_pipel_m##MODE##_w8(const SlabParams p)                                                                   \
    {                                                                                                             \
        asm volatile("" ::: "v255");                                                                              \
        coop_pipel_body<MODE>(p);                                                                                 \
    }
GPFQ_DEFINE_PIPEL(0) GPFQ_DEFINE_PIPEL(1) GPFQ_DEFINE_PIPEL(2) GPFQ_DEFINE_PIPEL(3)

}  // namespace gpfq
