// gpfq_capi.hip -- host side of the MI355X GPFQ hot path: plan selection, launches and the C ABI of include/gpfq.h.
//
// Path (reference = YixuanSeanZhou/Quantized_Neural_Nets, src/):
//   StepAlgorithm._quantization   step_algorithm.py:107-148   -> gpfq_loop_kernels.h (resident / coop / wave / stream kernels),
//                                                                 gpfq_pipe_kernels.h (pipelined cooperative kernels)
//   quantizers                    step_algorithm.py:7-104     -> gpfq_device.h quant_*
//   column reads [:, t], norm     step_algorithm.py:141-144   -> gpfq_prep_kernels.h
//   conv activation capture       quantize_neural_net.py:334-347 -> gpfq_prep_kernels.h gpfq_gather_patches_kernel
// Compile with -ffp-contract=off (see gpfq_device.h).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <string>

#include "../../include/gpfq.h"
#include "gpfq_device.h"
#include "gpfq_loop_kernels.h"
#include "gpfq_pipe_kernels.h"
#include "gpfq_pipel_kernels.h"
#include "gpfq_prep_kernels.h"

// ================================================================================================
// C ABI
// ================================================================================================
namespace {

thread_local std::string g_err;
thread_local int g_used_exchange = 0;    // did this thread's last loop launch wait on other workgroups (status word matters)?

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
// Polls before an exchange gives up.  A poll is one sc1 load round trip (0.65-0.85 us, profiles/r02_xchg_probe.txt) plus
// a 64-clock pause, and an exchange of a co-resident grid completes within a handful of them; 2^14 polls are ~12 ms, so a
// grid whose co-residency broke (another process on the card) costs milliseconds before the layer is redone on the
// whole-row streaming plan -- not the ~1.5 s of the former 2^21.
constexpr int kDefaultSpinLimit = 1 << 14;

struct Plan {
    int kind;      // GPFQ_PLAN_STREAM / GPFQ_PLAN_RESIDENT / GPFQ_PLAN_COOP
    int RT;        // rows per workgroup
    int waves;     // waves per workgroup
    int S;         // segments per row
    int C;         // coop: members per row tile
    int tiles;     // coop: row tiles
    int rounds;    // coop: launches the rows are spread over (every launch must be co-resident); 1 = all rows at once
    int tiles_round;   // coop: row tiles per launch
    int grouped;   // coop: one row per group (depthwise convolutions), every row with its own columns
    int pipe;      // coop: 1 = the PIPELINED kernels (gpfq_pipe_kernels.h): RT = 4 or 8 rows in four groups, reducer wave(s) of their own;
                   //       2 = the pipelined kernels with LDS-staged columns (gpfq_pipel_kernels.h): RT = 12 rows in three groups
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

int slab_max_waves(bool coop, int RT);
int resident_max_rt(int waves);
// the epoch word of a pipelined granule: column + 1 in 20 bits, the launch number in 8, the publisher's XCD in 4
bool p_d_fits_epoch(int d) { return d < (1 << 20) - 1; }

// GPFQ_COOP_SPIN_LIMIT (polls before an exchange gives up), clamped to what the kernels' 32-bit word can carry scaled by 256.
// The kernels count `(spins += 256) > limit` with limit = 256 * polls + (pause, < 32): the counter has to be able to EXCEED the
// limit before it wraps, so the largest poll count is 2^24 - 2 (limit <= 0xFFFFFE1F, spins reaches 0xFFFFFF00 and gives up);
// at 2^24 - 1 the counter would reach 0xFFFFFF00 <= limit and wrap to 0 on the next poll: an unbounded spin.
constexpr int kMaxSpinPolls = (1 << 24) - 2;
int clamped_spin_limit()
{
    const int v = env_int("GPFQ_COOP_SPIN_LIMIT", kDefaultSpinLimit);
    return v < 0 ? 0 : (v > kMaxSpinPolls ? kMaxSpinPolls : v);
}

// Is there an instantiation of the cooperative kernel for (rows per workgroup, sweep waves, members) with this
// quantizer?  Answered by the same table the launch uses (coop_kernel_for, below), so that a pair without one is never
// CHOSEN: the plan that is described is the plan that launches.  (Today every pair exists for all four quantizers.)
bool coop_variant_exists(int RT, int NW, int C, int mode);

// wave bound of the cooperative kernel variant that takes NW sweep waves of RT rows split over C members (keep
// coop_kernel() in step)
int coop_wave_bound(int RT, int NW, int C)
{
    if (RT == 1) return (NW <= 12 && C <= 128) ? 12 : 16;    // (16: the variant that gathers four members per lane)
    if (RT == 4 && C >= 64) return 16;                  // (256 / 1024 granules: the LDS-staged variant only, up to 13 sweep waves)
    if (RT == 2 && C > 64) return 16;                   // (the variant that gathers eight members per lane)
    if (NW <= 8) return 8;
    if (RT == 4) return NW <= 12 ? 12 : 16;             // (16: the LDS-staged variant, 13 sweep waves)
    return NW <= 12 ? 12 : 16;
}

int pow2_ceil_host(int v)
{
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

// Cost model of one cooperative column step (microseconds): least squares over 50 measured (N, S, RT, C) points on an
// MI355X (tools/layer_bench.py sweep over the ResNet-50 cooperative shapes and their 2-, 4- and 8-way row shards; rms
// error 0.18 us, and it picks the fastest measured configuration for every one of those shapes).  A fixed part
// (barriers, exchange hop, quantizer), RT sweeps, the column requests of the waves on the CU, and the gather: more
// than 16 granules (RT * C) cost extra, the more so the more workgroups are exchanging at the same time; 32 members
// cost extra again; so does the one-step look-ahead of the variant without room for five column buffers.
double slab_step_cost(int RT, int waves, int C, int wgs)
{
    const int n = RT * C;
    (void)wgs;
    // (the gather terms were 0.28 + 1.02 wgs/256, 0.65 and 1.5 before the first poll was paced -- launch_coop -- and the
    // reducer went on its instruction diet; re-measured on the shapes that run in rounds: 50 segments 4 rows x 8 members
    // 2.35 us, 197 segments 4 x 32 2.26, 785 segments 2 x 64 3.45; 91 segments 2 x 8 2.20 against 4 x 16 2.50, and a
    // 32-row shard of them 1 x 8 2.06 against 2 x 16 2.26)
    return 0.74 + 0.119 * RT + 0.102 * waves + (n > 16 ? 0.45 + 0.004 * (n - 16) : 0.0) + (C >= 32 ? 0.1 : 0.0) + (C >= 64 ? 0.9 : 0.0) + (((RT == 4 && waves > 8) || (RT == 2 && waves > 12)) ? 0.43 : 0.0) + ((RT == 4 && waves > 12) ? 0.6 : 0.0);
}

// Rows per workgroup of the resident plan: the rows of a workgroup share every column load, and the CU's vector-memory
// pipe is what bounds that kernel -- so as many as still leave one workgroup for every CU (fewer workgroups than CUs
// would idle whole CUs: a step costs the same however few rows the chip holds), within the register budget.
int resident_rows_per_wg(int64_t Ng, int S, int cus)
{
    const int force = env_int("GPFQ_RESIDENT_RT", 0);
    if (S <= 1) {
        // one-segment rows (a workgroup is one wave, the step is that wave's instruction stream): one row per wave while
        // every SIMD holds at most one wave, then rows share the wave -- the quantizer runs once per step for all of them
        // (measured per column, N = 1152 / 2048 / 4096 / 8192 rows: one row 0.42 / 0.54 / 1.03 / 2.05 us, two rows
        // 0.43 / 0.46 / 0.66 / 1.25, four rows 0.64 / 0.66 / 0.67 / 1.07)
        if (force == 1 || force == 2 || force == 4) return force;
        return Ng <= 5 * (int64_t)cus ? 1 : (Ng <= 16 * (int64_t)cus ? 2 : 4);
    }
    const int cap = resident_max_rt(S);
    for (int rt = 4; rt >= 2; rt >>= 1)
        if (rt <= cap && (force ? rt == force : (Ng + rt - 1) / rt >= cus)) return rt;
    return 1;
}

// The resident plan's column step in microseconds (MI355X, measured: 0.42 / 0.63 / 1.10 at 3 / 7 / 12 segments with one
// row per workgroup and one workgroup per CU; 0.57 / 0.84 at 3 / 7 segments with two rows): a fixed chain (reduction,
// divisions, quantizer), ~57 ns of the CU's vector-memory pipe per segment (8 KB of columns) of every workgroup on the
// CU, and a sweep per further row.  The kernels take their whole register budget (256 / 168 / 128 VGPRs at <= 8 / 12 /
// 16 waves), so a CU holds 8 / 12 / 16 waves of them; more workgroups than fit run in rounds.
double resident_step_cost(int64_t Ng, int S, int cus)
{
    const int RT = resident_rows_per_wg(Ng, S, cus);
    const int slots = S <= 8 ? 8 : (S <= 12 ? 12 : 16);
    const int fit = slots / S > 0 ? slots / S : 1;
    const int64_t tiles = (Ng + RT - 1) / RT;
    const int64_t per_cu = (tiles + cus - 1) / cus;
    const int conc = (int)(per_cu < fit ? per_cu : fit);
    const int64_t rounds = (tiles + (int64_t)cus * conc - 1) / ((int64_t)cus * conc);
    const double segs = conc * S;                          // segments whose columns the CU pulls per step
    const double pipe = segs <= 8 ? 0.057 * segs : 0.057 * 8 + 0.085 * (segs - 8);   // (steeper once every SIMD holds 3+ waves)
    return (double)rounds * (0.25 + pipe + 0.18 * (RT - 1));
}

// What one column costs on the streaming plan (microseconds; measured on the ResNet-50 1x1 convolutions at batch
// 1024, profiles/r02_v1_bench_r50_all_layers.txt: 4.7-5.5 TB/s of algorithmic bytes, 6.3 when the residual fits the
// Infinity Cache and whole rows cover the chip; EfficientNet-B1's 3137-segment rows 4.1-4.3).
double stream_col_cost(int64_t Ng, int S, int cus)
{
    const double bytes = 8.0 * (double)Ng * S * 4096.0 / 4.0;
    const bool cached = (double)Ng * S * 4096.0 <= 220e6 && Ng >= 4 * (int64_t)cus;
    // (rows beyond 1024 segments: 4.1-4.3 TB/s measured -- few rows, the column split leaves a quarter of the CUs idle)
    return bytes / (cached ? 6.3e6 : (S > 1024 ? 4.3e6 : 5.0e6));
}

// Cooperative configuration: cheapest modelled layer among the (RT, C) pairs.  A grid that is not co-resident as a
// whole runs in ROUNDS: rows are independent, so the layer is cut into blocks of rows whose workgroups all fit on the
// chip, one launch per block, every launch walking all d columns with its block of U in registers (the columns come
// from L2 / the Infinity Cache again; U, the 8*N*m bytes per column of the streaming plan, never moves).
// Depends on (Ng, S, CU count) only -- never on the data.  cost_out: microseconds per column for all rows.
bool choose_pipe(int64_t Ng, int S, int cus, int mode, Plan* pl, double* cost_out, bool allow_rounds);
bool choose_pipel(int64_t Ng, int S, int cus, Plan* pl, double* cost_out, bool allow_rounds);

// allow_pipe false: the lock-step kernels only -- for layers the pipelined kernels cannot take (2^20 columns or more: their
// epoch word) and for the retry after a pipelined launch was refused (run_loop)
bool choose_coop(int64_t Ng, int S, int cus, int mode, Plan* pl, double* cost_out = nullptr, bool allow_rounds = true,
                 bool allow_pipe = true)
{
    const int force_rt = env_int("GPFQ_COOP_RT", 0), force_c = env_int("GPFQ_COOP_C", 0);
    const int wgs_per_cu = env_int("GPFQ_COOP_WGS_PER_CU", 1) > 1 ? 2 : 1;
    const int capacity = cus * wgs_per_cu;
    if (env_int("GPFQ_COOP_NO_ROUNDS", 0)) allow_rounds = false;
    double best = 1e30;
    bool found = false;
    for (int RT = 4; RT >= 1; RT >>= 1) {
        if (force_rt && RT != force_rt) continue;
        const int64_t tiles = (Ng + RT - 1) / RT;
        // the reducer gathers RT * C <= 128 granules (two per lane) -- 256 in the one-row 16-wave variant (four per lane),
        // 512 in the two-row one (eight per lane)
        for (int C = 256; C >= 2; C >>= 1) {
            if (RT == 4 && C == 128) continue;                                     // (512 granules of four rows: no variant gathers them)
            if (force_c && C != force_c) continue;
            if (C > S || C > capacity) continue;
            const int NW = (S + C - 1) / C;
            if (RT == 1 && (C > 128 || NW > 12) && (C < 4 || NW > 15)) continue;   // (that variant gathers in fours)
            if (RT == 4 && C >= 64 && NW > 13) continue;                           // (256 / 1024 granules: the LDS-staged variant only)
            if (RT == 2 && C > 64 && NW > 15) continue;                            // (256 / 512 granules: the 16-wave variant that gathers eight members per lane, reducer wave of its own)
            if (RT == 4 && NW == 13 && env_int("GPFQ_COOP_NO_LDS", 0)) continue;
            if (NW > slab_max_waves(true, RT) || pow2_ceil_host(S) / C > 16) continue;
            if (!coop_variant_exists(RT, NW, C, mode)) continue;
            const int64_t tiles_round = tiles * C <= capacity ? tiles : capacity / C;
            const int64_t rounds = (tiles + tiles_round - 1) / tiles_round;
            if (rounds > 1 && !allow_rounds) continue;
            const int wgs = (int)tiles_round * C;
            const int per_cu = (wgs + cus - 1) / cus;
            if ((per_cu * NW + 3) / 4 > 4) continue;
            // fewer workgroups than CUs simply leave CUs idle, which costs nothing per step
            const int vmax = coop_wave_bound(RT, NW, C);             // wave bound of the variant launch_slab picks
            const int launched = NW + (NW + 1 <= vmax ? 1 : 0);      // + the reducer wave when it fits
            const double cost = (double)rounds * slab_step_cost(RT, per_cu * launched, C, wgs);
            if (!found || cost < best - 1e-9) {
                found = true;
                best = cost;
                pl->kind = GPFQ_PLAN_COOP; pl->RT = RT; pl->C = C; pl->tiles = (int)tiles; pl->waves = NW; pl->S = S;
                pl->rounds = (int)rounds; pl->tiles_round = (int)tiles_round; pl->grouped = 0; pl->pipe = 0;
            }
        }
    }
    // the pipelined kernels where they are modelled cheaper (GPFQ_COOP_PIPE: 0 never, 1 whenever a configuration exists)
    const int pipe_mode = env_int("GPFQ_COOP_PIPE", -1);
    if (pipe_mode != 0 && allow_pipe) {
        Plan pp = *pl;
        double pcost = 0.0;
        // (two sweep waves per member and every row on the chip at once: the lock-step step is already little more than its
        // exchange -- measured 1.47 us per column against 1.55 pipelined on ResNet-50's layer2.{1,2,3}.conv2 -- whatever the
        // lock-step model, which overestimates its 8-wave kernels, says)
        if (choose_pipe(Ng, S, cus, mode, &pp, &pcost, allow_rounds) &&
            (!found || pipe_mode == 1 || (pcost < best && !(pp.waves <= 2 && pl->rounds == 1)))) {
            *pl = pp;
            best = pcost;
            found = true;
        }
    }
    // ... and the twelve-row pipelined kernels with LDS-staged columns (GPFQ_COOP_PIPEL: 0 never, 1 whenever a configuration
    // exists; default: where modelled cheaper -- layers in rounds, whose rounds they cut by a third)
    // (GPFQ_COOP_PIPE=0, "no pipelined kernels", switches this family off too unless it is asked for by name)
    const int pipel_mode = env_int("GPFQ_COOP_PIPEL", pipe_mode == 0 ? 0 : -1);
    if (pipel_mode != 0 && allow_pipe && (!force_rt || force_rt == gpfq::kPipelRows)) {
        Plan pp = *pl;
        double pcost = 0.0;
        // (by more than 3 %: at a modelled tie the measured one goes to the four-group kernels -- 256 rows of 197 segments
        // 9.99 against 10.36 us per column)
        if (choose_pipel(Ng, S, cus, &pp, &pcost, allow_rounds) && (!found || pipel_mode == 1 || pcost < 0.97 * best)) {
            *pl = pp;
            best = pcost;
            found = true;
        }
    }
    if (found && cost_out) *cost_out = best;
    return found;
}

// One column step of the PIPELINED cooperative kernels (gpfq_pipe_kernels.h), microseconds, measured on one MI355X
// (tools/layer_bench.py, round 4; the members of a tile on one XCD, publishing with plain stores): four phases, each the longer
// of one group's sweep on the fullest SIMD -- RG = 2: 0.20 us per sweep wave of that SIMD + 0.13 (lane tree, LDS, barrier),
// RG = 1: 0.13 + 0.13 -- and the gatherer wave's own phase (~0.38 us: landing of the requested granules, epoch check, tree
// over the members, quantizer, q and history into LDS), a little longer with more sweep waves queueing column requests in
// front of its loads and with 32+ members.  Measured / modelled per column: 8 rows x 16 members x 6 waves 2.12 / 2.12,
// 8 x 8 x 4 1.70 / 1.68, 8 x 16 x 2 1.52 / 1.52, 8 x 32 x 3 1.73 / 1.76, 4 x 16 x 6 1.73 / 1.68.
// `local`: the tile's members sit on one XCD (launch_pipe places them so when the tile count is a multiple of the XCDs and
// a tile has no more members than an XCD has CUs) and publish with plain stores; otherwise every granule is a write-through
// and a phase cannot be shorter than what its round trip needs (measured 0.53-0.60 us per phase before XCD-local
// publishing: 2.1-2.4 us per column).
double pipe_step_cost(int RG, int waves, int C, bool local)
{
    const int per_simd = (waves + 3) / 4;                                  // sweep waves on the fullest SIMD
    const double sweep = (RG == 1 ? 0.13 : 0.20) * per_simd + 0.13;
    // (64 granules per gather, seven sweep waves: measured in rounds 2.35-2.53 us per column on 197-segment rows, round 5)
    double gather = 0.38 + (waves >= 3 ? 0.04 : 0.0) + (RG * C >= 64 ? (waves == 7 ? 0.08 : 0.02) : 0.0) + (RG * C > 64 ? 0.06 : 0.0);
    // (single rows with at most one sweep wave per SIMD: the sweep waves are at the barrier early and the gatherer's phase is the
    // shortest measured -- N = 128, 26 segments, four rows x 8 members: 1.36 us per column against 1.43 as eight rows x 16 members
    // and 1.46 on the lock-step two-row kernel)
    // (up to 16 members, one DPP row per gather: with 32 the same layout loses to the lock-step one-row kernel, 1.64 against
    // 1.52-1.62 on 8 and 16 rows of 91 segments)
    if (RG == 1 && waves <= 4 && C <= 16) gather = 0.34;
    if (waves == 7 || RG * C > 64) gather += 0.12;                                        // (one wave for both reducer roles: + the slot tree and the store)
    if (!local && gather < 0.60) gather = 0.60;
    // (two rows x 128 members, four granules per lane, the members on four XCDs: measured 4.65 us per column and round
    // against 3.8 for the LDS-staged four-row kernel on 64 members -- the variant exists, AUTO does not take it)
    if (RG * C > 64 && gather < 1.15) gather = 1.15;
    return 4.0 * (sweep > gather ? sweep : gather);
}

// The pipelined configurations for (Ng rows, S segments): RT = 8 (four interleaved pairs) or 4 (four single rows), C members
// with at most 6 segments each (6 sweep waves + the publisher + the gatherer: the 256-register budget), RG * C <= 64 granules per gather.
bool choose_pipe(int64_t Ng, int S, int cus, int mode, Plan* pl, double* cost_out, bool allow_rounds)
{
    (void)mode;                                                            // (every quantizer has both variants)
    const int force_rt = env_int("GPFQ_COOP_RT", 0), force_c = env_int("GPFQ_COOP_C", 0);
    double best = 1e30;
    bool found = false;
    for (int RT = 8; RT >= 4; RT >>= 1) {
        if (force_rt && RT != force_rt) continue;
        const int RG = RT / 4;
        const int64_t tiles = (Ng + RT - 1) / RT;
        for (int C = RG == 2 ? 128 : 64; C >= 2; C >>= 1) {                 // RG * C <= 64 granules per gather, or two rows x 128 members (four per lane)
            if (RG * C > 64 && !(RG == 2 && C == 128 && force_c == 128)) continue;   // (256 granules: only when asked for, see pipe_step_cost)
            if (force_c && C != force_c) continue;
            if (C > S || C > cus) continue;
            const int NW = (S + C - 1) / C;
            if (NW > 7 || pow2_ceil_host(S) / C > 16) continue;
            const int64_t tiles_round = tiles * C <= cus ? tiles : cus / C;
            if (tiles_round < 1) continue;
            const int64_t rounds = (tiles + tiles_round - 1) / tiles_round;
            if (rounds > 1 && !allow_rounds) continue;
            if ((size_t)tiles_round * 2 * C * RT * sizeof(unsigned long long) > kScratchStatusOffset) continue;
            // (a tile's members on one XCD: launch_pipe pads the grid to a multiple of eight tiles where the chip has the room)
            const bool local = C <= cus / 8 && ((tiles_round + 7) & ~(int64_t)7) * C <= cus && env_int("GPFQ_PIPE_LOCAL", 1) &&
                               env_int("GPFQ_COOP_XCD_TILES", 1);
            const double cost = (double)rounds * pipe_step_cost(RG, NW, C, local);
            if (!found || cost < best - 1e-9) {
                found = true;
                best = cost;
                pl->kind = GPFQ_PLAN_COOP; pl->RT = RT; pl->C = C; pl->tiles = (int)tiles; pl->waves = NW; pl->S = S;
                pl->rounds = (int)rounds; pl->tiles_round = (int)tiles_round; pl->grouped = 0; pl->pipe = 1;
            }
        }
    }
    if (found && cost_out) *cost_out = best;
    return found;
}

// One column step of the pipelined kernels with LDS-staged columns (gpfq_pipel_kernels.h), microseconds: three phases, each the
// sweep of one group -- four rows, two interleaved pairs: 160 packed instructions per sweep wave -- on the fullest SIMD plus
// the lane tree, the LDS word and the barrier; the exchange hides under it.  (First measurements, round 5: see profiles/NOTES.md.)
// Measured per column and round (tools/layer_bench.py, round 5): 128 members x 7 waves 3.67, 32 x 7 3.2-3.3, 8 x 7 3.3-3.4,
// 2 x 7 3.34, 64 x 5 3.0.  What bounds a phase is the SIMD's issue rate: 250 instructions per sweep wave (160 of them packed, 4.45
// SIMD cycles each with two waves on the SIMD: profiles/r05_probe_valu.txt), two sweep waves per SIMD: ~2 400 cycles.
double pipel_step_cost(int waves, int C)
{
    const int per_simd = (waves + 3) / 4;
    const double sweep = 0.45 * per_simd + (waves > 4 ? 0.19 : 0.29);
    const double gather = C > 64 ? 1.22 : (C > 32 ? 1.0 : 0.60);      // (128 members: the reducer's re-polls; device scope throughout)
    return 3.0 * (sweep > gather ? sweep : gather);
}

// The configurations of that family for (Ng rows, S segments): RT = 12, C members with at most 7 segments each (7 sweep waves +
// the reducer: eight waves of 256 registers), up to 128 members (four granules of a row pair per lane of the gather).
bool choose_pipel(int64_t Ng, int S, int cus, Plan* pl, double* cost_out, bool allow_rounds)
{
    const int force_c = env_int("GPFQ_COOP_C", 0);
    const int RT = gpfq::kPipelRows;
    const int64_t tiles = (Ng + RT - 1) / RT;
    double best = 1e30;
    bool found = false;
    for (int C = 128; C >= 2; C >>= 1) {
        if (force_c && C != force_c) continue;
        if (C > S || C > cus) continue;
        const int NW = (S + C - 1) / C;
        if (NW > 7 || pow2_ceil_host(S) / C > 16) continue;
        const int64_t tiles_round = tiles * C <= cus ? tiles : cus / C;
        if (tiles_round < 1) continue;
        const int64_t rounds = (tiles + tiles_round - 1) / tiles_round;
        if (rounds > 1 && !allow_rounds) continue;
        if ((size_t)tiles_round * 2 * C * RT * sizeof(unsigned long long) > kScratchStatusOffset) continue;
        const double cost = (double)rounds * pipel_step_cost(NW, C);
        if (!found || cost < best - 1e-9) {
            found = true;
            best = cost;
            pl->kind = GPFQ_PLAN_COOP; pl->RT = RT; pl->C = C; pl->tiles = (int)tiles; pl->waves = NW; pl->S = S;
            pl->rounds = (int)rounds; pl->tiles_round = (int)tiles_round; pl->grouped = 0; pl->pipe = 2;
        }
    }
    if (found && cost_out) *cost_out = best;
    return found;
}

// Depthwise convolutions with long rows (groups == out channels, ONE row per group, 9 or 25 columns each, m up to 370 688
// in EfficientNet-B1 at batch 1024): every group is a row of its own with its own columns, so the cooperative kernel's
// one-row variant takes them as Ng = groups rows (GROUPED: per-row column base), in rounds of as many groups as fit the
// chip.  The streaming plan gives each group one workgroup (17 GB/s of algorithmic bytes each: 96 groups of 362
// segments 174 us per column); a round costs a cooperative step plus its launch spread over the few columns.
bool choose_coop_grouped(int groups, int S, int cus, Plan* pl, double* cost_out)
{
    const int P = pow2_ceil_host(S);
    double best = 1e30;
    bool found = false;
    for (int C = 128; C >= 2; C >>= 1) {
        if (C > S || C > cus || P / C > 16) continue;
        const int NW = (S + C - 1) / C;
        if (NW > 12) continue;                                        // (the grouped variant is the 12-wave one-row kernel)
        const int tiles_round = groups * C <= cus ? groups : cus / C;
        const int rounds = (groups + tiles_round - 1) / tiles_round;
        const int launched = NW + (NW + 1 <= 12 ? 1 : 0);
        const double cost = rounds * (slab_step_cost(1, launched, C, tiles_round * C) + 1.3);
        if (cost < best) {
            found = true;
            best = cost;
            pl->kind = GPFQ_PLAN_COOP; pl->RT = 1; pl->C = C; pl->tiles = groups; pl->waves = NW; pl->S = S;
            pl->rounds = rounds; pl->tiles_round = tiles_round; pl->grouped = 1; pl->pipe = 0;
        }
    }
    if (found && cost_out) *cost_out = best;
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
    pl->rounds = 1; pl->tiles_round = 0; pl->grouped = 0; pl->pipe = 0;
    pl->RT = Ng >= 1024 ? 4 : (Ng >= 512 ? 2 : 1);
    while (pl->RT > 1 && (size_t)2 * pl->RT * S * sizeof(float) > 48 * 1024) pl->RT >>= 1;   // the segment sums of RT rows live in LDS
    if (S > 1024 && pl->RT > 2) pl->RT = 2;               // (the four-row kernel has no 32 / 64-slots-per-lane tree)
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
            if (RT > 2 && S > 1024 * C) continue;                 // (the four-row kernel has no 32 / 64-slots-per-lane tree)
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

int choose_plan(int64_t Ng, int64_t m_pad, int groups, int requested, bool have_scratch, int mode, Plan* out, bool allow_pipe = true)
{
    Plan pl;
    pl.C = 1; pl.tiles = 0; pl.rounds = 1; pl.tiles_round = 0; pl.grouped = 0; pl.pipe = 0;
    if (m_pad / gpfq::kSeg > 4096) return fail(GPFQ_ERR_UNSUPPORTED, "m > 4194304 calibration rows is not supported");
    pl.S = (int)(m_pad / gpfq::kSeg);
    if (requested < GPFQ_PLAN_AUTO || requested > GPFQ_PLAN_STREAM_ROWS) return fail(GPFQ_ERR_ARG, "unknown plan id");
    if (requested == GPFQ_PLAN_STREAM_ROWS) {            // whole rows per workgroup: never waits for another workgroup
        choose_stream(Ng, pl.S, groups, false, &pl);
        *out = pl;
        return GPFQ_OK;
    }
    if (requested == GPFQ_PLAN_RESIDENT && pl.S > kMaxResidentSegments)
        return fail(GPFQ_ERR_UNSUPPORTED, "resident plan needs m_pad <= 16384");
    const int cus = device_cu_count();
    if (requested == GPFQ_PLAN_RESIDENT || (requested == GPFQ_PLAN_AUTO && pl.S <= kMaxResidentSegments)) {
        pl.kind = GPFQ_PLAN_RESIDENT;
        pl.waves = pl.S;
        pl.RT = resident_rows_per_wg(Ng, pl.S, cus);
        // Long rows of which a CU holds only one at a time run in rounds; four rows per cooperative workgroup can then
        // be cheaper (N = 512, m = 13 312, VGG-16's 512-channel convs at batch 512: 2.39 -> 2.17 us per column).
        if (requested == GPFQ_PLAN_AUTO && groups == 1 && have_scratch && Ng > cus && !env_int("GPFQ_COOP_DISABLE", 0)) {
            Plan cp = pl;
            double ccost = 0.0;
            // (the cooperative alternative may itself run in rounds where the resident plan does: 2048 rows of 13 segments
            // are eight rounds of one-row workgroups or four of four-row tiles split in two, 9.5 vs 8.1 us per column)
            const double rcost = resident_step_cost(Ng, pl.S, cus);
            const bool resident_in_rounds = (Ng + pl.RT - 1) / pl.RT > (int64_t)cus * ((pl.S <= 8 ? 8 : (pl.S <= 12 ? 12 : 16)) / pl.S);
            if (choose_coop(Ng, pl.S, cus, mode, &cp, &ccost, resident_in_rounds, allow_pipe) && ccost < 0.97 * rcost) {
                *out = cp;
                return GPFQ_OK;
            }
        }
        *out = pl;
        return GPFQ_OK;
    }
    if (groups > 1 && Ng == 1 && pl.S > kMaxResidentSegments && have_scratch && requested != GPFQ_PLAN_STREAM &&
        !env_int("GPFQ_COOP_DISABLE", 0)) {
        Plan gp = pl;
        double gcost = 0.0;
        const double stream_cost = 8.0 * pl.S * 1024.0 / 0.017e6 * ((groups + cus - 1) / cus);
        if (choose_coop_grouped(groups, pl.S, cus, &gp, &gcost) && (requested == GPFQ_PLAN_COOP || gcost < 0.9 * stream_cost)) {
            *out = gp;
            return GPFQ_OK;
        }
    }
    if (requested == GPFQ_PLAN_COOP || (requested == GPFQ_PLAN_AUTO && !env_int("GPFQ_COOP_DISABLE", 0))) {
        double ccost = 0.0;
        if (groups == 1 && have_scratch && choose_coop(Ng, pl.S, cus, mode, &pl, &ccost, true, allow_pipe)) {
            // in rounds only where that beats moving U through memory every column
            if (pl.rounds == 1 || requested == GPFQ_PLAN_COOP || ccost < 0.9 * stream_col_cost(Ng, pl.S, cus)) {
                *out = pl;
                return GPFQ_OK;
            }
            pl.C = 1; pl.tiles = 0; pl.rounds = 1; pl.tiles_round = 0; pl.pipe = 0;
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
    sc.spin_limit = (unsigned)clamped_spin_limit();
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

// Pause before the first poll of an exchange, in units of 256 clocks.  A poll that comes back incomplete costs a round trip
// AND stands in the way of the granules still travelling: every member polls every line of the tile's block, members^2 / 16
// line requests per round on the one memory channel that block lives on -- so the first poll waits about as long as the
// slowest member's store needs, and that grows with the MEMBERS.  Measured optimum (tools/scratch/poll_delay_sweep.py; per
// column against the round-2 rule, 2 wherever 32 granules were awaited): 64 members 8 (4.72 -> 4.18 us), 128: 16
// (9.29 -> 8.16), 256: 16-24 (9.21 -> 8.65), 32: 4, 16 members with two or four rows 3-4 (in rounds 51.1 -> 47.8); beyond
// the optimum every unit costs its 0.1 us.  Small tiles reach the exchange close together and a pause only costs (one row
// x 16 members: 0 -> 4 is 1.46 -> 1.78 us; 2 x 8, 2 x 4, 4 x 4: 0) -- except four rows x 8 on 12 waves and the
// LDS-staged 13-wave kernel, whose members arrive further apart (four rows x 4 members in rounds: 3, 25.0 -> 22.7).
int first_poll_pause(int RT, int C, int waves, bool lds)
{
    if (C >= 128) return 16;
    if (C >= 64) return 8;
    if (C >= 32) return 4;
    if (C >= 16) return RT >= 2 ? 4 : 0;
    if (C == 4 && lds) return 3;
    // (four rows x 8 members: 2 with 12 sweep waves, 2.86 -> 2.76 us per column; none with 7 -- EfficientNet-B1's 50-segment
    // rows 1.14 -> 1.02 ms per layer: the fewer waves, the closer together the members arrive)
    return RT * C >= 32 && waves > 8 ? 2 : 0;
}

gpfq::SlabParams make_slab_params(const Plan& pl, const gpfq::LoopParams& p, bool vec, void* scratch)
{
    gpfq::SlabParams sp;
    sp.W = p.W; sp.Q = p.Q; sp.U = p.U; sp.idx = p.idx; sp.AT = p.AT; sp.XT = p.XT; sp.nrm2 = p.nrm2; sp.usq = p.usq;
    sp.xbuf = reinterpret_cast<unsigned long long*>(scratch);
    sp.status = scratch ? reinterpret_cast<int*>(static_cast<char*>(scratch) + kScratchStatusOffset) : nullptr;
    sp.ldw = p.ldw; sp.ldq = p.ldq; sp.ldu = p.ldu; sp.ldi = p.ldi; sp.m = p.m; sp.m_pad = p.m_pad;
    sp.Ng = (int)p.Ng; sp.d = (int)p.d; sp.S = pl.S; sp.C = pl.C; sp.tiles = pl.tiles; sp.idx_bytes = p.idx_bytes;
    sp.vec = vec ? 1 : 0;
    sp.step = p.qc.step; sp.Kf = p.qc.Kf; sp.lamb = p.qc.lamb;
    // the division-free MSQ path (gpfq_device.h quant_msq_from_dot) assumes a quotient that cannot underflow before the
    // sign is taken and a tolerance that grows with K: steps outside 2^-40 .. 2^40 and K > 1024 (no real alphabet) keep
    // the divisions (NaN fails every comparison)
    const bool fast_ok = p.qc.step >= 0x1p-40f && p.qc.step <= 0x1p40f && p.qc.Kf <= 1024.0f && !env_int("GPFQ_EXACT_DIVISIONS", 0);
    sp.inv_step = fast_ok ? 1.0f / p.qc.step : __builtin_nanf("");
    sp.msq_thr = 0.5f - (p.qc.Kf + 4.0f) * 0x1p-18f;
    // (low FIVE bits: pause before the first poll of an exchange, in units of 256 clocks, 0 .. 31 -- launch_coop, which knows
    // the kernel; the kernel reads `& 31u`.)  The poll count is clamped to [0, 2^24 - 2] before it is scaled: 256 * 2^24
    // wraps a 32-bit word to 0 -- every exchange would give up at once --, 2^24 - 1 could never be exceeded by a counter that
    // advances in steps of 256, and a negative value casts to a huge one (clamped_spin_limit).
    sp.spin_limit = 256u * (unsigned)clamped_spin_limit();
    sp.pace = env_int("GPFQ_COOP_PACE", 2);
    sp.xcd_tiles = 0;                               // launch_coop decides
    sp.seed = p.qc.seed; sp.row_id0 = p.row_id0;
    sp.salt = 0; sp.allow_local = 0;                // launch_pipe sets them
    sp.reducer_prio = 0;                            // launch_pipel sets it
    sp.prefetch_ahead = 0; sp.prefetch_lines = 1;   // launch_resident decides
    return sp;
}

typedef void (*SlabKernel)(const gpfq::SlabParams);

// the instantiated cooperative (rows per workgroup, wave bound) pairs -- keep slab_max_waves() in step
SlabKernel coop_kernel(int RT, int mode, int maxw)
{
#define GPFQ_PICK(RTV, MAXWV)                                                                                         \
    if (RT == RTV && maxw == MAXWV) {                                                                                 \
        switch (mode) {                                                                                               \
        case gpfq::MODE_SOFT: return gpfq::gpfq_coop_rt##RTV##_m1_w##MAXWV;                                           \
        case gpfq::MODE_HARD: return gpfq::gpfq_coop_rt##RTV##_m2_w##MAXWV;                                           \
        case gpfq::MODE_STOCHASTIC: return gpfq::gpfq_coop_rt##RTV##_m3_w##MAXWV;                                     \
        default: return gpfq::gpfq_coop_rt##RTV##_m0_w##MAXWV;                                                        \
        }                                                                                                             \
    }
    GPFQ_PICK(1, 12) GPFQ_PICK(1, 16) GPFQ_PICK(2, 8) GPFQ_PICK(2, 12) GPFQ_PICK(2, 16) GPFQ_PICK(4, 8) GPFQ_PICK(4, 12)
#undef GPFQ_PICK
    return nullptr;                                 // (four rows at 16: coop_kernel_lds)
}

// two rows on 256 members, eight granules gathered per lane
SlabKernel coop_kernel_oct(int mode)
{
    switch (mode) {
    case gpfq::MODE_SOFT: return gpfq::gpfq_coop_rt2_m1_w16o;
    case gpfq::MODE_HARD: return gpfq::gpfq_coop_rt2_m2_w16o;
    case gpfq::MODE_STOCHASTIC: return gpfq::gpfq_coop_rt2_m3_w16o;
    default: return gpfq::gpfq_coop_rt2_m0_w16o;
    }
}

// four rows at 13 sweep waves, columns staged through LDS (gpfq_loop_kernels.h coop_lds_body)
SlabKernel coop_kernel_lds(int mode, bool quad, bool hex)
{
    switch (mode) {
    case gpfq::MODE_SOFT: return hex ? gpfq::gpfq_coop_rt4_m1_w16lh : quad ? gpfq::gpfq_coop_rt4_m1_w16lq : gpfq::gpfq_coop_rt4_m1_w16l;
    case gpfq::MODE_HARD: return hex ? gpfq::gpfq_coop_rt4_m2_w16lh : quad ? gpfq::gpfq_coop_rt4_m2_w16lq : gpfq::gpfq_coop_rt4_m2_w16l;
    case gpfq::MODE_STOCHASTIC: return hex ? gpfq::gpfq_coop_rt4_m3_w16lh : quad ? gpfq::gpfq_coop_rt4_m3_w16lq : gpfq::gpfq_coop_rt4_m3_w16l;
    default: return hex ? gpfq::gpfq_coop_rt4_m0_w16lh : quad ? gpfq::gpfq_coop_rt4_m0_w16lq : gpfq::gpfq_coop_rt4_m0_w16l;
    }
}

// one row per group, 12 waves (gpfq_loop_kernels.h GPFQ_DEFINE_COOP_GROUPED)
SlabKernel coop_kernel_grouped(int mode)
{
    switch (mode) {
    case gpfq::MODE_SOFT: return gpfq::gpfq_coop_rt1g_m1_w12;
    case gpfq::MODE_HARD: return gpfq::gpfq_coop_rt1g_m2_w12;
    case gpfq::MODE_STOCHASTIC: return gpfq::gpfq_coop_rt1g_m3_w12;
    default: return gpfq::gpfq_coop_rt1g_m0_w12;
    }
}

// The kernel a cooperative configuration launches (nullptr: no instantiation), its wave bound and whether it stages its
// columns through LDS -- the ONE table behind launch_coop and coop_variant_exists.
SlabKernel coop_kernel_for(int RT, int NW, int C, int mode, bool grouped, int* maxw_out, bool* lds_out)
{
    const int maxw = grouped ? 12 : coop_wave_bound(RT, NW, C);
    const bool lds = RT == 4 && maxw == 16 && !grouped;
    const bool oct = RT == 2 && C > 64 && maxw == 16 && !grouped;
    if (maxw_out) *maxw_out = maxw;
    if (lds_out) *lds_out = lds;
    if (NW > maxw || (lds && NW > 13)) return nullptr;
    return grouped ? coop_kernel_grouped(mode) : lds ? coop_kernel_lds(mode, RT * C > 128, RT * C > 256) : oct ? coop_kernel_oct(mode) : coop_kernel(RT, mode, maxw);
}

bool coop_variant_exists(int RT, int NW, int C, int mode)
{
    return coop_kernel_for(RT, NW, C, mode, false, nullptr, nullptr) != nullptr;
}

// the pipelined cooperative kernels (gpfq_pipe_kernels.h): RG = 1 (four single rows) or 2 (four interleaved pairs)
SlabKernel pipe_kernel(int RG, int mode, bool single)
{
#define GPFQ_PICKP(RGV, SINGLEV, SUFFIX)                                                                              \
    if (RG == RGV && single == SINGLEV) {                                                                             \
        switch (mode) {                                                                                               \
        case gpfq::MODE_SOFT: return gpfq::gpfq_pipe_rg##RGV##_m1_w8##SUFFIX;                                         \
        case gpfq::MODE_HARD: return gpfq::gpfq_pipe_rg##RGV##_m2_w8##SUFFIX;                                         \
        case gpfq::MODE_STOCHASTIC: return gpfq::gpfq_pipe_rg##RGV##_m3_w8##SUFFIX;                                   \
        default: return gpfq::gpfq_pipe_rg##RGV##_m0_w8##SUFFIX;                                                      \
        }                                                                                                             \
    }
    GPFQ_PICKP(1, false, ) GPFQ_PICKP(2, false, ) GPFQ_PICKP(1, true, s) GPFQ_PICKP(2, true, s)
#undef GPFQ_PICKP
    return nullptr;
}

// two rows x 128 members per gather (four granules per lane), seven sweep waves + one reducer wave
SlabKernel pipe_kernel_quad(int mode)
{
    switch (mode) {
    case gpfq::MODE_SOFT: return gpfq::gpfq_pipe_rg2_m1_w8sq;
    case gpfq::MODE_HARD: return gpfq::gpfq_pipe_rg2_m2_w8sq;
    case gpfq::MODE_STOCHASTIC: return gpfq::gpfq_pipe_rg2_m3_w8sq;
    default: return gpfq::gpfq_pipe_rg2_m0_w8sq;
    }
}

// Has a cooperative launch of THIS process timed out on the device (gpfq_read_status saw its status word raised)?  Then the
// chip is shared with somebody, and every later cooperative grid on that device goes through hipLaunchCooperativeKernel.
std::atomic<int> g_contended[64];

int current_device_slot()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
    return dev;
}

bool device_contended()
{
    const int dev = current_device_slot();
    return dev >= 0 && g_contended[dev].load(std::memory_order_relaxed) != 0;
}

void note_contention()
{
    const int dev = current_device_slot();
    if (dev < 0) return;
    if (g_contended[dev].exchange(1) == 0 && env_int("GPFQ_COOP_LAUNCH_API", 0) >= 0)
        fprintf(stderr, "gpfq: a cooperative launch timed out on device %d (the card is shared?): the layer is redone on the "
                        "streaming plan, and cooperative grids on this device are launched through "
                        "hipLaunchCooperativeKernel from now on\n", dev);
}

// will the next cooperative grid on the current device go through the cooperative launch API?
bool use_coop_launch_api()
{
    const int mode = env_int("GPFQ_COOP_LAUNCH_API", 0);        // 1 always, -1 never, 0 (default) once contention was seen
    return mode > 0 || (mode == 0 && device_contended());
}

// Launch of a grid whose workgroups wait for each other.  Default: a plain launch behind the occupancy check of the caller
// (the grid is sized to be co-resident on an otherwise idle chip; every spin is bounded and a timeout is reported).
// hipLaunchCooperativeKernel, with which the RUNTIME refuses a grid that cannot be co-resident
// (hipErrorCooperativeLaunchTooLarge) instead of letting it spin to its bound, costs +5 us per launch on the pipelined
// family (round 4, profiles/NOTES.md) -- so it is not what an undisturbed process pays: it is used from the first timeout
// this process has seen on the device on (note_contention: a second tenant costs a 2^14-poll spin, ~12 ms, and a streaming
// redo per cooperative launch, silently, for as long as it stays), or always with GPFQ_COOP_LAUNCH_API=1 (-1: never).
hipError_t launch_waiting_grid(SlabKernel kern, dim3 grid, dim3 block, size_t shm, hipStream_t st, gpfq::SlabParams& sp)
{
    if (use_coop_launch_api()) {
        void* args[] = {&sp};
        return hipLaunchCooperativeKernel(reinterpret_cast<const void*>(kern), grid, block, args, (unsigned)shm, st);
    }
    hipLaunchKernelGGL(kern, grid, block, shm, st, sp);
    return hipGetLastError();
}


int launch_pipe(const Plan& pl, const gpfq::SlabParams& sp, int mode, void* scratch, hipStream_t st)
{
    if (!p_d_fits_epoch(sp.d)) return fail(GPFQ_ERR_UNSUPPORTED, "pipelined cooperative kernels take fewer than 2^20 columns");
    const int RT = pl.RT, RG = RT / 4;
    const bool quad = RG * pl.C > 64;                                  // 256 granules per gather: four per lane (two rows x 128 members only)
    const bool single = pl.waves == 7 || quad;                         // seven sweep waves: one wave plays both reducer roles
    SlabKernel kern = quad ? ((RT == 8 && pl.C == 128) ? pipe_kernel_quad(mode) : nullptr)
                           : ((RT == 4 || RT == 8) ? pipe_kernel(RG, mode, single) : nullptr);
    if (!kern || pl.waves < 1 || pl.waves > 7)
        return fail(GPFQ_ERR_UNSUPPORTED, "internal: no pipelined cooperative kernel for this (rows, waves, members) triple");
    const int nwaves = pl.waves + (single ? 1 : 2);                    // + the publisher wave and the gatherer wave (or one for both)
    const int threads = 64 * nwaves;
    const size_t shm = sizeof(float) * ((size_t)RT * pl.waves + RT + 2 * (size_t)RT * 64 + 4);   // (+ the locality flag)
    const int cus = device_cu_count();
    // XCD placement (speed; the kernel verifies it before relying on it): the grid is padded to a multiple of eight tiles
    // where the chip holds that many workgroups, so that tile k's members can sit on XCD k mod 8 whatever the tile count
    const int tiles_padded = (pl.tiles + 7) & ~7;
    const bool place = env_int("GPFQ_COOP_XCD_TILES", 1) && tiles_padded * pl.C <= cus;
    const int nblocks = (place ? tiles_padded : pl.tiles) * pl.C;
    int nb = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, threads, shm);
    if (e != hipSuccess) return hip_fail(e, "occupancy query");
    if (nb < 1 || nblocks > cus) return fail(GPFQ_ERR_UNSUPPORTED, "cooperative grid does not fit on the device");
    size_t xbytes = (size_t)pl.tiles * 2 * pl.C * RT * sizeof(unsigned long long);
    xbytes = (xbytes + 15) & ~(size_t)15;
    if (xbytes > kScratchStatusOffset) return fail(GPFQ_ERR_UNSUPPORTED, "exchange buffer larger than the scratch area");
    e = hipMemsetAsync(scratch, 0, xbytes, st);
    if (e != hipSuccess) return hip_fail(e, "exchange buffer memset");
    gpfq::SlabParams spx = sp;
    spx.spin_limit = sp.spin_limit & ~255u;                            // (no pause before a gather: the granules are two phases old)
    spx.xcd_tiles = place ? 1 : 0;
    static std::atomic<unsigned> launch_number{0};
    spx.salt = launch_number.fetch_add(1) & 255u;
    spx.allow_local = env_int("GPFQ_PIPE_LOCAL", 1) && p_d_fits_epoch(sp.d);
    e = launch_waiting_grid(kern, dim3((unsigned)nblocks, 1, 1), dim3((unsigned)threads), shm, st, spx);
    if (e != hipSuccess) return hip_fail(e, "GPFQ pipelined cooperative kernel launch");
    return GPFQ_OK;
}

SlabKernel pipel_kernel(int mode)
{
    switch (mode) {
    case gpfq::MODE_SOFT: return gpfq::gpfq_pipel_m1_w8;
    case gpfq::MODE_HARD: return gpfq::gpfq_pipel_m2_w8;
    case gpfq::MODE_STOCHASTIC: return gpfq::gpfq_pipel_m3_w8;
    default: return gpfq::gpfq_pipel_m0_w8;
    }
}

// twelve rows in three groups, columns staged through LDS (gpfq_pipel_kernels.h)
int launch_pipel(const Plan& pl, const gpfq::SlabParams& sp, int mode, void* scratch, hipStream_t st)
{
    if (!p_d_fits_epoch(sp.d)) return fail(GPFQ_ERR_UNSUPPORTED, "pipelined cooperative kernels take fewer than 2^20 columns");
    if (pl.RT != gpfq::kPipelRows || pl.waves < 1 || pl.waves > 7 || pl.C > 128 || pl.C < 1)
        return fail(GPFQ_ERR_UNSUPPORTED, "internal: no LDS-staged pipelined kernel for this (rows, waves, members) triple");
    SlabKernel kern = pipel_kernel(mode);
    const int threads = 64 * (pl.waves + 1);                           // + the reducer wave
    const size_t shm = gpfq::pipel_lds_bytes(pl.waves);
    hipError_t e = hipSuccess;
    {   // more than 64 KB of dynamic LDS: allowed once per (device, kernel), for the largest workgroup (seven sweep waves)
        static std::atomic<int> attr_set[64][4];
        const int dev = current_device_slot();
        if (dev < 0 || !attr_set[dev][mode & 3].load(std::memory_order_acquire)) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)gpfq::pipel_lds_bytes(7));
            if (e != hipSuccess) return hip_fail(e, "dynamic LDS size");
            if (dev >= 0) attr_set[dev][mode & 3].store(1, std::memory_order_release);
        }
    }
    const int cus = device_cu_count();
    const int nblocks = pl.tiles * pl.C;
    int nb = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, threads, shm);
    if (e != hipSuccess) return hip_fail(e, "occupancy query");
    if (nb < 1 || nblocks > cus) return fail(GPFQ_ERR_UNSUPPORTED, "cooperative grid does not fit on the device");
    size_t xbytes = (size_t)pl.tiles * 2 * pl.C * pl.RT * sizeof(unsigned long long);
    xbytes = (xbytes + 15) & ~(size_t)15;
    if (xbytes > kScratchStatusOffset) return fail(GPFQ_ERR_UNSUPPORTED, "exchange buffer larger than the scratch area");
    e = hipMemsetAsync(scratch, 0, xbytes, st);
    if (e != hipSuccess) return hip_fail(e, "exchange buffer memset");
    gpfq::SlabParams spx = sp;
    // low five bits: the reducer's pause in front of its gather request, in units of 64 clocks (gpfq_pipel_kernels.h (c))
    // (measured: 8 x 64 clocks and priority 2 are the best or within the noise of it at 2 / 8 / 32 / 128 members)
    spx.spin_limit = (sp.spin_limit & ~255u) | ((unsigned)env_int("GPFQ_PIPEL_REQUEST_PAUSE", 8) & 31u);
    spx.xcd_tiles = 0;
    spx.reducer_prio = env_int("GPFQ_PIPEL_REDUCER_PRIO", 2) & 3;
    static std::atomic<unsigned> launch_number{0};
    spx.salt = launch_number.fetch_add(1) & 255u;
    spx.allow_local = 0;
    e = launch_waiting_grid(kern, dim3((unsigned)nblocks, 1, 1), dim3((unsigned)threads), shm, st, spx);
    if (e != hipSuccess) return hip_fail(e, "GPFQ LDS-staged pipelined cooperative kernel launch");
    return GPFQ_OK;
}

int launch_coop(const Plan& pl, const gpfq::SlabParams& sp, int mode, void* scratch, hipStream_t st)
{
    if (pl.pipe == 2) return launch_pipel(pl, sp, mode, scratch, st);
    if (pl.pipe) return launch_pipe(pl, sp, mode, scratch, st);
    const int RT = pl.RT;
    int maxw = 0;
    bool lds = false;
    SlabKernel kern = coop_kernel_for(RT, pl.waves, pl.C, mode, pl.grouped != 0, &maxw, &lds);
    if (!kern)
        return fail(GPFQ_ERR_UNSUPPORTED, "internal: no cooperative kernel for this (rows, waves) pair");
    // one more wave for the reducer role when the variant's wave bound allows it
    const int nwaves = pl.waves + ((pl.waves + 1 <= maxw && !env_int("GPFQ_NO_REDUCER_WAVE", 0)) ? 1 : 0);
    const int threads = 64 * nwaves;
    const size_t shm = sizeof(float) * (2 * RT * (size_t)nwaves + 2 * (RT + 1) + 2 * RT * 64) +
                       (lds ? (size_t)pl.waves * 3 * 4096 : 0);          // + three 4-KB column buffers per sweep wave
    if (lds) {
        hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (ea != hipSuccess) return hip_fail(ea, "dynamic LDS size");
    }
    const int nblocks = pl.tiles * pl.C;
    // every workgroup must be resident at once: check the grid against the occupancy query (the query is
    // known to over-report by one only near the SGPR limit of >= 6 waves per SIMD; these kernels run at
    // <= 4, so the answer is taken as is -- and every spin is bounded anyway)
    int nb = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, threads, shm);
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
    // Which workgroups share an XCD (blocks b and b+8 do under round-robin dispatch; speed only).  Members of one row
    // tile together: every XCD's L2 pulls every column, 4-8 workgroups share a line.  Member c of every tile together
    // (blockIdx = tile * C + c, C a multiple of 8): an L2 pulls 1/8 of each column for all tiles, but 32+ workgroups hit
    // the same line at the same time.  Measured per column: layer1 (91 segments, two steps of look-ahead) 2.21 vs 2.36 us,
    // 26 segments 1.90 vs 2.00 / 2.46 vs 2.56 -- tiles together; the variants with ONE step of look-ahead (their loads
    // must land within a step) 3.47 vs 3.29 and 21.9 vs 20.6, and 197 segments in rounds (1.6 MB of columns per step
    // and XCD) 139 vs 102 -- members together.
    gpfq::SlabParams spx = sp;
    spx.spin_limit = (sp.spin_limit & ~255u) | ((unsigned)env_int("GPFQ_COOP_POLL_DELAY", first_poll_pause(RT, pl.C, pl.waves, lds)) & 31u);
    const bool depth1 = (RT == 4 && maxw == 12) || (RT == 2 && maxw == 16);
    spx.xcd_tiles = env_int("GPFQ_COOP_XCD_TILES", (!depth1 && pl.S <= 128) ? 1 : 0);
    e = launch_waiting_grid(kern, dim3((unsigned)nblocks, 1, 1), dim3((unsigned)threads), shm, st, spx);
    if (e != hipSuccess) return hip_fail(e, "GPFQ cooperative kernel launch");
    return GPFQ_OK;
}


// the instantiated (rows per workgroup, wave bound) pairs: keep resident_max_rt() in step
SlabKernel resident_kernel(int RT, int mode, int maxw)
{
#define GPFQ_PICK(RTV, MAXWV)                                                                                         \
    if (RT == RTV && maxw == MAXWV) {                                                                                 \
        switch (mode) {                                                                                               \
        case gpfq::MODE_SOFT: return gpfq::gpfq_resident_rt##RTV##_m1_w##MAXWV;                                       \
        case gpfq::MODE_HARD: return gpfq::gpfq_resident_rt##RTV##_m2_w##MAXWV;                                       \
        case gpfq::MODE_STOCHASTIC: return gpfq::gpfq_resident_rt##RTV##_m3_w##MAXWV;                                 \
        default: return gpfq::gpfq_resident_rt##RTV##_m0_w##MAXWV;                                                    \
        }                                                                                                             \
    }
    GPFQ_PICK(1, 8) GPFQ_PICK(2, 8) GPFQ_PICK(4, 8) GPFQ_PICK(1, 12) GPFQ_PICK(2, 12) GPFQ_PICK(1, 16)
    GPFQ_PICK(1, 1) GPFQ_PICK(2, 1) GPFQ_PICK(4, 1)
#undef GPFQ_PICK
    return nullptr;
}

// one-segment rows whose samples fill one or two quarters of the segment
SlabKernel resident_kernel_partial(int RT, int mode, int nq)
{
#define GPFQ_PICKQ(RTV, NQV)                                                                                          \
    if (RT == RTV && nq == NQV) {                                                                                     \
        switch (mode) {                                                                                               \
        case gpfq::MODE_SOFT: return gpfq::gpfq_resident_rt##RTV##_m1_w1q##NQV;                                       \
        case gpfq::MODE_HARD: return gpfq::gpfq_resident_rt##RTV##_m2_w1q##NQV;                                       \
        case gpfq::MODE_STOCHASTIC: return gpfq::gpfq_resident_rt##RTV##_m3_w1q##NQV;                                 \
        default: return gpfq::gpfq_resident_rt##RTV##_m0_w1q##NQV;                                                    \
        }                                                                                                             \
    }
    GPFQ_PICKQ(1, 1) GPFQ_PICKQ(2, 1) GPFQ_PICKQ(4, 1) GPFQ_PICKQ(1, 2) GPFQ_PICKQ(2, 2) GPFQ_PICKQ(4, 2)
#undef GPFQ_PICKQ
    return nullptr;
}

// most rows per workgroup the register budget of a wave bound leaves room for (the window takes 80 of it)
int resident_max_rt(int waves) { return waves <= 8 ? 4 : (waves <= 12 ? 2 : 1); }

int launch_resident(const Plan& pl, const gpfq::SlabParams& sp, int mode, int groups, hipStream_t st)
{
    if (pl.waves != pl.S || pl.S > 16) return fail(GPFQ_ERR_UNSUPPORTED, "internal: resident plan needs one wave per segment");
    const int maxw = pl.S == 1 ? 1 : (pl.waves <= 8 ? 8 : (pl.waves <= 12 ? 12 : 16));     // 1: the one-segment variant
    // one-segment rows of m <= 256 / 512 samples: the variants that load and sweep one / two quarters of the segment
    const int nq = (pl.S == 1 && !env_int("GPFQ_FULL_SEGMENT", 0)) ? (sp.m <= 256 ? 1 : (sp.m <= 512 ? 2 : 4)) : 4;
    SlabKernel k = nq < 4 ? resident_kernel_partial(pl.RT, mode, nq) : resident_kernel(pl.RT, mode, maxw);
    if (!k) return fail(GPFQ_ERR_UNSUPPORTED, "internal: no resident kernel for this (rows, waves) pair");
    const size_t shm = sizeof(float) * 2 * (size_t)pl.RT * (size_t)pl.S;
    dim3 grid((unsigned)((sp.Ng + pl.RT - 1) / pl.RT), (unsigned)groups, 1);
    // The prefetch agent (gpfq_loop_kernels.h resident_prefetch_agent): one more wave per workgroup, where the variant's wave
    // bound has room for it, touching the lines of column t + K so that they sit in the XCD's L2 when the sweeps ask -- the
    // work shared by the workgroups of an XCD.  Measured per column (tools/layer_bench.py, K = 16 against none):
    // N = 512, m = 7168 with its 264 MB of columns beyond the Infinity Cache 0.79 -> 0.66 us; N = 256, m = 7168 0.69 -> 0.54;
    // N = 512, m = 3072 0.51 -> 0.47; N = 1024, m = 5120 (four rows) 1.07 -> 1.02; N = 2048, m = 2048 1.09 -> 1.03: never a
    // loss, so every multi-segment resident layer of one group with the room gets one.  GPFQ_RESIDENT_PREFETCH: 0 = never,
    // a positive value = the distance K in columns (default 16).
    gpfq::SlabParams spx = sp;
    int nwaves = pl.waves;
    {
        const int want = env_int("GPFQ_RESIDENT_PREFETCH", 16);
        const int per_xcd = (int)((grid.x + 7) / 8);                       // workgroups that share an XCD's L2 (round-robin dispatch)
        const int lp = per_xcd >= 32 ? 1 : (per_xcd >= 16 ? 2 : (per_xcd >= 8 ? 4 : 0));
        const bool room = pl.S > 1 && pl.waves + 1 <= maxw && groups == 1 && lp > 0 && 2 * pl.S * lp <= 64;
        if (room && want > 0) {
            spx.prefetch_ahead = want;
            spx.prefetch_lines = lp;
            nwaves = pl.waves + 1;
        }
    }
    // (the agent wave exists only together with its share of lines: the kernel takes wave S for the agent)
    if (nwaves != pl.waves && spx.prefetch_lines < 1) return fail(GPFQ_ERR_UNSUPPORTED, "internal: prefetch agent without lines");
    hipLaunchKernelGGL(k, grid, dim3((unsigned)(64 * nwaves)), shm, st, spx);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "GPFQ resident kernel launch");
    return GPFQ_OK;
}

// Launch of the register-resident plans ("slab" = the RT x n block of U a workgroup keeps in registers): the
// instantiated (rows per workgroup, wave bound) pairs -- keep slab_max_waves() in step with these switches.
int launch_slab(const Plan& pl, const gpfq::LoopParams& p, int groups, bool vec, void* scratch, hipStream_t st)
{
    const gpfq::SlabParams sp = make_slab_params(pl, p, vec, scratch);
    const int m = p.qc.mode;
    if (pl.kind == GPFQ_PLAN_RESIDENT) return launch_resident(pl, sp, m, groups, st);
    return launch_coop(pl, sp, m, scratch, st);
}

// most waves per workgroup an instantiation exists for
int slab_max_waves(bool coop, int RT)
{
    if (!coop) return 16;
    return RT <= 2 ? 16 : 13;                           // (four rows: 13 sweep waves x 12 KB of LDS is what a CU holds)
}

int run_loop(gpfq::LoopParams p, int groups, int plan, void* scratch, size_t scratch_bytes, hipStream_t st)
{
    if (p.Ng <= 0 || p.d <= 0 || groups <= 0) return GPFQ_OK;   // nothing to do
    if (p.Ng > 0x7fffffff || p.d > 0x7fffffff) return fail(GPFQ_ERR_UNSUPPORTED, "N or d beyond 2^31");
    const bool have_scratch = scratch && scratch_bytes >= kScratchBytes && !(reinterpret_cast<uintptr_t>(scratch) & 255);
    Plan pl;
    // the register-resident plans start from U = 0; a caller-provided initial residual (the in-place
    // _quantization surface) streams through memory
    g_used_exchange = 0;
    if (p.u_has_init && plan == GPFQ_PLAN_AUTO) plan = GPFQ_PLAN_STREAM;
    if (p.u_has_init && plan != GPFQ_PLAN_STREAM && plan != GPFQ_PLAN_STREAM_ROWS)
        return fail(GPFQ_ERR_UNSUPPORTED, "an initial residual needs a streaming plan");
    int rc = choose_plan(p.Ng, p.m_pad, groups, plan, have_scratch, p.qc.mode, &pl, p_d_fits_epoch((int)p.d));
    if (rc) return rc;
    if (groups > 65535) return fail(GPFQ_ERR_UNSUPPORTED, "groups > 65535");
    p.S = pl.S;
    const bool vec = ((p.ldu & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.U) & 15) == 0);
    if (pl.kind == GPFQ_PLAN_COOP && pl.grouped) {
        // one row per group: the kernel sees Ng = (groups of this round) rows of ONE group, each row with its own
        // columns (GROUPED); a round is a block of groups
        rc = GPFQ_OK;
        for (int64_t g0 = 0; g0 < groups && rc == GPFQ_OK; g0 += pl.tiles_round) {
            gpfq::LoopParams q = p;
            Plan pr = pl;
            q.Ng = (groups - g0 < pl.tiles_round) ? groups - g0 : pl.tiles_round;
            pr.tiles = (int)q.Ng;
            q.W = p.W + g0 * p.ldw; q.Q = p.Q + g0 * p.ldq; q.U = p.U + g0 * p.ldu;
            if (p.idx) q.idx = static_cast<char*>(p.idx) + g0 * p.ldi * p.idx_bytes;
            if (p.usq) q.usq = p.usq + g0 * pl.S;
            q.AT = p.AT + g0 * p.d * p.m_pad; q.XT = p.XT + g0 * p.d * p.m_pad; q.nrm2 = p.nrm2 + 2 * g0 * p.d;
            q.row_id0 = p.row_id0 + (uint64_t)g0;
            rc = launch_slab(pr, q, 1, vec, scratch, st);
            if (rc == GPFQ_ERR_UNSUPPORTED && g0 > 0) return fail(GPFQ_ERR_HIP, "internal: a later round of a cooperative layer did not fit");
        }
        if (rc == GPFQ_OK) g_used_exchange = 1;
        if (rc != GPFQ_ERR_UNSUPPORTED || plan == GPFQ_PLAN_COOP) return rc;
        rc = choose_plan(p.Ng, p.m_pad, groups, GPFQ_PLAN_STREAM, have_scratch, p.qc.mode, &pl);   // does not fit: stream instead
        if (rc) return rc;
    }
    // (a pipelined configuration the launch refuses -- the occupancy query is the launch's -- gets ONE more try on the lock-step
    // family, the configuration choose_coop had found for the shape before the pipelined one was modelled cheaper)
    for (int attempt = 0; attempt < 2 && pl.kind == GPFQ_PLAN_COOP; ++attempt) {
        if (pl.rounds <= 1) {
            rc = launch_slab(pl, p, groups, vec, scratch, st);
        } else {
            // one launch per block of rows; a launch that finds the status word raised by an earlier one returns at once
            const int64_t block = (int64_t)pl.tiles_round * pl.RT;
            rc = GPFQ_OK;
            for (int64_t r0 = 0; r0 < p.Ng && rc == GPFQ_OK; r0 += block) {
                gpfq::LoopParams q = p;
                Plan pr = pl;
                q.Ng = (p.Ng - r0 < block) ? p.Ng - r0 : block;
                pr.tiles = (int)((q.Ng + pl.RT - 1) / pl.RT);
                q.W = p.W + r0 * p.ldw; q.Q = p.Q + r0 * p.ldq; q.U = p.U + r0 * p.ldu;
                if (p.idx) q.idx = static_cast<char*>(p.idx) + r0 * p.ldi * p.idx_bytes;
                if (p.usq) q.usq = p.usq + r0 * pl.S;
                q.row_id0 = p.row_id0 + (uint64_t)r0;
                rc = launch_slab(pr, q, groups, vec, scratch, st);
                if (rc == GPFQ_ERR_UNSUPPORTED && r0 > 0) return fail(GPFQ_ERR_HIP, "internal: a later round of a cooperative layer did not fit");
            }
        }
        if (rc == GPFQ_OK) g_used_exchange = 1;
        if (rc != GPFQ_ERR_UNSUPPORTED) return rc;
        const bool was_pipe = pl.pipe != 0;
        if (was_pipe && attempt == 0) {
            Plan lp;
            if (choose_plan(p.Ng, p.m_pad, groups, plan, have_scratch, p.qc.mode, &lp, false) == GPFQ_OK && lp.kind == GPFQ_PLAN_COOP && !lp.pipe) {
                pl = lp;
                continue;
            }
        }
        if (plan == GPFQ_PLAN_COOP) return rc;
        rc = choose_plan(p.Ng, p.m_pad, groups, GPFQ_PLAN_STREAM, have_scratch, p.qc.mode, &pl);   // does not fit: stream instead
        if (rc) return rc;
    }
    if (pl.kind == GPFQ_PLAN_RESIDENT) return launch_slab(pl, p, groups, vec, scratch, st);
    for (int attempt = 0; attempt < 2; ++attempt) {
        switch (pl.RT) {
        case 4: rc = launch_stream<4>(pl, p, groups, vec, scratch, st); break;
        case 2: rc = launch_stream<2>(pl, p, groups, vec, scratch, st); break;
        default: rc = launch_stream<1>(pl, p, groups, vec, scratch, st); break;
        }
        if (rc == GPFQ_OK && pl.C > 1) g_used_exchange = 1;
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

int gpfq_last_launch_used_exchange(void) { return g_used_exchange; }

unsigned gpfq_spin_limit_word(void) { return 256u * (unsigned)clamped_spin_limit(); }

int gpfq_coop_launch_api_active(void) { return use_coop_launch_api() ? 1 : 0; }

void gpfq_clear_contention(void)
{
    const int dev = current_device_slot();
    if (dev >= 0) g_contended[dev].store(0);
}

const char* gpfq_last_error(void) { return g_err.c_str(); }

int64_t gpfq_padded_m(int64_t m)
{
    if (m < 1) m = 1;
    return ((m + gpfq::kSeg - 1) / gpfq::kSeg) * gpfq::kSeg;
}

size_t gpfq_scratch_bytes(void) { return kScratchBytes; }

// waves per column of the norm kernel: 4, or 16 when the columns are few and long (every wave takes whole segments)
static unsigned colnorm_threads(int64_t D, int S)
{
    return (D < 512 && S >= 64) ? 1024u : 256u;
}

static size_t ws_cols_bytes(int64_t d_g, int64_t m, int groups)
{
    const size_t D = (size_t)d_g * (size_t)groups;
    return D * (size_t)gpfq_padded_m(m) * sizeof(float);
}
static size_t ws_nrm_bytes(int64_t d_g, int groups)
{
    return ((2 * (size_t)d_g * (size_t)groups * sizeof(float) + 255) / 256) * 256;       // {norm, reciprocal} per column
}

size_t gpfq_workspace_bytes(int64_t N, int64_t d_g, int64_t m, int groups)
{
    (void)N;
    if (d_g < 0 || m < 0 || groups < 1) return 0;
    return kScratchBytes + 2 * ws_cols_bytes(d_g, m, groups) + ws_nrm_bytes(d_g, groups) + 256 +
           gpfq_prepare_ws_bytes(d_g * (int64_t)groups, m);                 // [scratch][AT][XT][nrm2][segment sums]
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
        note_contention();
        e = hipMemsetAsync(sp, 0, 4 * sizeof(int), st);
        if (e != hipSuccess) return hip_fail(e, "status reset");
        return fail(GPFQ_ERR_TIMEOUT, "cooperative kernel timed out waiting for a peer workgroup");
    }
    return GPFQ_OK;
}

int gpfq_prepare_columns_f32(const float* A, int64_t lda, const float* X, int64_t ldx, int64_t m, int64_t D,
                             float* AT, float* XT, float* nrm2, int64_t m_pad, void* stream)
{
    // one matrix only: A == NULL (AT untouched) or X == NULL (XT and nrm2 untouched) -- a caller that prepares the analog
    // columns ahead of time, on another stream, and the quantized ones when they exist
    if ((!A && !X) || (A && !AT) || (X && (!XT || !nrm2))) return fail(GPFQ_ERR_ARG, "null pointer");
    if (m < 0 || D < 0 || (A && lda < D) || (X && ldx < D)) return fail(GPFQ_ERR_ARG, "bad shape (need lda, ldx >= D)");
    if (m_pad != gpfq_padded_m(m)) return fail(GPFQ_ERR_ARG, "m_pad must equal gpfq_padded_m(m)");
    if ((A && (reinterpret_cast<uintptr_t>(AT) & 15)) || (X && (reinterpret_cast<uintptr_t>(XT) & 15)))
        return fail(GPFQ_ERR_ARG, "AT / XT must be 16-byte aligned");
    if (D == 0) return GPFQ_OK;
    hipStream_t st = (hipStream_t)stream;
    if (D <= 60 && (!A || lda == D) && (!X || ldx == D) && !env_int("GPFQ_NO_SMALL_TRANSPOSE", 0)) {      // (256 x 61 floats of LDS)
        // few columns of contiguous matrices (first convs, EfficientNet's narrow 1x1 convs at 112 x 112): flat reads
        const size_t shm = sizeof(float) * 256 * (size_t)((int)D | 1);
        hipLaunchKernelGGL(gpfq::gpfq_transpose_pad_small_kernel, dim3((unsigned)(m_pad / 256), 2), dim3(256), shm, st, A, X, m,
                           (int)D, AT, XT, m_pad);
    } else {
        dim3 grid((unsigned)(m_pad / 64), (unsigned)((D + 63) / 64), 2);
        if (grid.y > 65535) return fail(GPFQ_ERR_UNSUPPORTED, "too many columns");
        hipLaunchKernelGGL(gpfq::gpfq_transpose_pad_kernel, grid, dim3(256), 0, st, A, lda, X, ldx, m, D, AT, XT, m_pad);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "transpose launch");
    if (!X) return GPFQ_OK;
    const int S = (int)(m_pad / gpfq::kSeg);
    hipLaunchKernelGGL(gpfq::gpfq_colnorm_kernel, dim3((unsigned)D), dim3(colnorm_threads(D, S)), sizeof(float) * (size_t)S, st,
                       XT, m_pad, S, nrm2);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "colnorm launch");
    return GPFQ_OK;
}

size_t gpfq_prepare_ws_bytes(int64_t D, int64_t m)
{
    if (D < 0 || m < 0) return 0;
    const size_t S = (size_t)(gpfq_padded_m(m) / gpfq::kSeg);
    return (((size_t)D * S * sizeof(float) + 255) / 256) * 256;          // one segment sum per (column, segment)
}

int gpfq_prepare_columns_ws_f32(const float* A, int64_t lda, const float* X, int64_t ldx, int64_t m, int64_t D,
                                float* AT, float* XT, float* nrm2, int64_t m_pad, void* ws, size_t ws_bytes, void* stream)
{
    // without a workspace for the segment sums (or with the fused pass switched off): transpose, then norms from XT
    // (the segment sums belong to X: the analog matrix alone needs no workspace)
    if ((X && (!ws || ws_bytes < gpfq_prepare_ws_bytes(D, m))) || env_int("GPFQ_NO_FUSED_PREP", 0))
        return gpfq_prepare_columns_f32(A, lda, X, ldx, m, D, AT, XT, nrm2, m_pad, stream);
    if ((!A && !X) || (A && !AT) || (X && (!XT || !nrm2))) return fail(GPFQ_ERR_ARG, "null pointer");
    if (m < 0 || D < 0 || (A && lda < D) || (X && ldx < D)) return fail(GPFQ_ERR_ARG, "bad shape (need lda, ldx >= D)");
    if (m_pad != gpfq_padded_m(m)) return fail(GPFQ_ERR_ARG, "m_pad must equal gpfq_padded_m(m)");
    if ((A && (reinterpret_cast<uintptr_t>(AT) & 15)) || (X && (reinterpret_cast<uintptr_t>(XT) & 15)) || (reinterpret_cast<uintptr_t>(ws) & 3))
        return fail(GPFQ_ERR_ARG, "AT / XT must be 16-byte aligned");
    if (D == 0) return GPFQ_OK;
    hipStream_t st = (hipStream_t)stream;
    const int S = (int)(m_pad / gpfq::kSeg);
    if (!A) lda = ldx;                              // (the skipped matrix takes no part in the choice of the load mode)
    if (!X) ldx = lda;
    const bool aligned = !((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(X)) & 15);
    const bool flat = D <= 64 && lda == D && ldx == D && aligned;
    const bool vec = !flat && aligned && (lda & 3) == 0 && (ldx & 3) == 0 && (D & 3) == 0;
    // columns per workgroup: 64 (two workgroups per CU) where there are workgroups enough to balance over the chip,
    // else 32 (four per CU, twice as many of them)
    const int cus = device_cu_count();
    const int64_t wg64 = (int64_t)S * ((D + 63) / 64) * ((A ? 1 : 0) + (X ? 1 : 0));
    int TC = env_int("GPFQ_PREP_TC", 0);
    if (TC != 32 && TC != 64) TC = (flat || wg64 >= 16 * (int64_t)cus) ? 64 : 32;
    if (flat) TC = 64;
    const int ntile = (int)((D + TC - 1) / TC);
    // Column tiles side by side (workgroups that run together read neighbouring pieces of the same input rows): up to 2 KB of
    // every row, in groups of EQUAL size.  With groups of 512 / TC tiles and a ragged last one -- D = 576 is 9 tiles of 64:
    // 8 + 1 -- the last group is a tail of S workgroups per matrix with no neighbours and an emptying chip: 3.84 TB/s on
    // 93 184 x 576, 4.78 with 5 + 4 (tools/prep_bench.py; 4.29 -> 4.77 at D = 1 152, 4.34 -> 4.77 at 26 624 x 2 304).
    const int gmax = 512 / TC;
    const int ngr = (ntile + gmax - 1) / gmax;
    const int G = (ntile + ngr - 1) / ngr;
    const int64_t ngroups = (ntile + G - 1) / G;
    const int64_t nblocks = ngroups * S * 2 * G;
    if (nblocks > 0x7fffffffLL) return fail(GPFQ_ERR_UNSUPPORTED, "too many column tiles");
    const size_t shm = 256 * (size_t)(TC + 1) * sizeof(float);
    float* part = static_cast<float*>(ws);
    hipError_t e;
#define GPFQ_LAUNCH_TN(FLATV, VECV, TCV)                                                                              \
    {                                                                                                                 \
        static bool attr_set = false;                                                                                 \
        if (!attr_set) {                                                                                              \
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(gpfq::gpfq_transpose_norm_kernel<FLATV, VECV, TCV>), \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);                            \
            if (e != hipSuccess) return hip_fail(e, "dynamic LDS size");                                              \
            attr_set = true;                                                                                          \
        }                                                                                                             \
        hipLaunchKernelGGL((gpfq::gpfq_transpose_norm_kernel<FLATV, VECV, TCV>), dim3((unsigned)nblocks), dim3(256), shm, st, \
                           A, lda, X, ldx, m, D, AT, XT, m_pad, part, S, ntile, G);                                    \
    }
    if (flat) GPFQ_LAUNCH_TN(true, false, 64)
    else if (vec && TC == 64) GPFQ_LAUNCH_TN(false, true, 64)
    else if (vec) GPFQ_LAUNCH_TN(false, true, 32)
    else if (TC == 64) GPFQ_LAUNCH_TN(false, false, 64)
    else GPFQ_LAUNCH_TN(false, false, 32)
#undef GPFQ_LAUNCH_TN
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "transpose + norm launch");
    if (!X) return GPFQ_OK;
    hipLaunchKernelGGL(gpfq::gpfq_colnorm_finish_kernel, dim3((unsigned)D), dim3(64), 0, st, part, S, nrm2);
    e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "colnorm finish launch");
    return GPFQ_OK;
}

int gpfq_quantization_f32(const float* W, int64_t ldw, float* Q, int64_t ldq, float* U, int64_t ldu,
                          int u_has_init, const float* AT, const float* XT, const float* nrm2,
                          int64_t N, int64_t d, int64_t m, int64_t m_pad,
                          float step, int K, int mode, float lamb, uint64_t seed, uint64_t row_id0,
                          void* idx, int64_t ldi, int idx_bytes, float* usq_seg, int plan, void* scratch,
                          size_t scratch_bytes, void* stream)
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
    p.row_id0 = row_id0; p.idx = idx; p.ldi = ldi; p.idx_bytes = idx_bytes; p.usq = usq_seg;
    return run_loop(p, 1, plan, scratch, scratch_bytes, (hipStream_t)stream);
}

int gpfq_quantize_groups_prepared_f32(const float* W, float* Q, float* U, const float* AT, const float* XT,
                                      const float* nrm2, int64_t N, int64_t d_g, int64_t m, int64_t m_pad,
                                      int groups, float step, int K, int mode, float lamb, uint64_t seed,
                                      uint64_t row_id0, void* idx, int idx_bytes, float* usq_seg, int plan,
                                      void* scratch, size_t scratch_bytes, void* stream)
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
    p.row_id0 = row_id0; p.idx = idx; p.ldi = d_g; p.idx_bytes = idx_bytes; p.usq = usq_seg;
    return run_loop(p, groups, plan, scratch, scratch_bytes, (hipStream_t)stream);
}

int gpfq_quantize_layer_f32(const float* W, const float* A, int64_t lda, const float* X, int64_t ldx,
                            int64_t N, int64_t d_g, int64_t m, int groups,
                            float step, int K, int mode, float lamb, uint64_t seed, uint64_t row_id0,
                            float* Q, void* idx, int idx_bytes, float* U, float* usq_seg,
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
    char* part = ws + kScratchBytes + 2 * cb + ws_nrm_bytes(d_g, groups) + 256;
    int rc = gpfq_prepare_columns_ws_f32(A, lda, X, ldx, m, D, AT, XT, nrm2, mp, part, gpfq_prepare_ws_bytes(D, m), stream);
    if (rc) return rc;
    return gpfq_quantize_groups_prepared_f32(W, Q, U, AT, XT, nrm2, N, d_g, m, mp, groups, step, K, mode, lamb, seed,
                                             row_id0, idx, idx_bytes, usq_seg, plan, ws, kScratchBytes, stream);
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

int gpfq_philox_uniform_f32(uint64_t seed, uint64_t row_id0, uint64_t column, int64_t n, float* out, void* stream)
{
    if (!out || n < 0) return fail(GPFQ_ERR_ARG, "bad argument");
    if (n == 0) return GPFQ_OK;
    hipLaunchKernelGGL(gpfq::gpfq_philox_uniform_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       seed, row_id0, column, n, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "philox launch");
    return GPFQ_OK;
}

int gpfq_column_norms_f32(const float* XT, int64_t D, int64_t m, int64_t m_pad, float* nrm2, void* stream)
{
    if (!XT || !nrm2) return fail(GPFQ_ERR_ARG, "null pointer");
    if (D < 0 || m_pad != gpfq_padded_m(m)) return fail(GPFQ_ERR_ARG, "bad shape (m_pad must equal gpfq_padded_m(m))");
    if (D == 0) return GPFQ_OK;
    const int S = (int)(m_pad / gpfq::kSeg);
    hipLaunchKernelGGL(gpfq::gpfq_colnorm_kernel, dim3((unsigned)D), dim3(colnorm_threads(D, S)), sizeof(float) * (size_t)S,
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
    const int64_t nrows = C * kh;                                       // feature rows (c, i): kw contiguous values each
    dim3 grid((unsigned)(m_pad / 64), (unsigned)((nrows + 15) / 16), 1);
    if (grid.y > 65535) return fail(GPFQ_ERR_UNSUPPORTED, "too many features");
#define GPFQ_LAUNCH_GATHER(KWV)                                                                                       \
    hipLaunchKernelGGL(gpfq::gpfq_gather_patches_kernel<KWV>, grid, dim3(256), 0, (hipStream_t)stream, x, (int)C, (int)H, (int)W, \
                       kh, kw, pad_h, pad_w, dil_h, dil_w, (int)Lw, Lh * Lw, patch_index, m, outT, m_pad, (int)D)
    if (kw == 3) GPFQ_LAUNCH_GATHER(3);
    else if (kw == 5) GPFQ_LAUNCH_GATHER(5);
    else if (kw == 7) GPFQ_LAUNCH_GATHER(7);
    else GPFQ_LAUNCH_GATHER(0);
#undef GPFQ_LAUNCH_GATHER
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
    return gpfq_describe_plan_mode(N, d_g, m, groups, plan, GPFQ_MODE_MSQ, buf, buf_bytes);
}

int gpfq_describe_plan_mode(int64_t N, int64_t d_g, int64_t m, int groups, int plan, int mode, char* buf, size_t buf_bytes)
{
    if (groups < 1 || N % groups != 0) return fail(GPFQ_ERR_ARG, "bad groups");
    if (mode < 0 || mode > 3) return fail(GPFQ_ERR_ARG, "mode must be 0..3");
    Plan pl;
    int rc = choose_plan(N / groups, gpfq_padded_m(m), groups, plan, true, mode, &pl, d_g < 0 || p_d_fits_epoch((int)(d_g > 0x7fffffff ? 0x7fffffff : d_g)));
    if (rc) return rc;
    if (buf && buf_bytes) {
        if (pl.kind == GPFQ_PLAN_COOP && pl.grouped)
            snprintf(buf, buf_bytes, "coop RT=1 C=%d waves=%d S=%d grid=%d rounds=%d groups=%d d=%lld", pl.C, pl.waves, pl.S,
                     pl.tiles_round * pl.C, pl.rounds, groups, (long long)d_g);
        else if (pl.kind == GPFQ_PLAN_COOP && pl.rounds > 1)
            snprintf(buf, buf_bytes, "coop RT=%d C=%d waves=%d S=%d grid=%d rounds=%d%s d=%lld", pl.RT, pl.C, pl.waves, pl.S,
                     pl.tiles_round * pl.C, pl.rounds, pl.pipe == 2 ? " pipel=1" : pl.pipe ? " pipe=1" : "", (long long)d_g);
        else if (pl.kind == GPFQ_PLAN_COOP)
            snprintf(buf, buf_bytes, "coop RT=%d C=%d waves=%d S=%d grid=%d%s d=%lld", pl.RT, pl.C, pl.waves, pl.S,
                     pl.tiles * pl.C, pl.pipe == 2 ? " pipel=1" : pl.pipe ? " pipe=1" : "", (long long)d_g);
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
