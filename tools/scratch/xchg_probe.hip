// xchg_probe.hip -- what one granule exchange of the cooperative kernels costs on its own (gfx950, test tool, not product).
// 256 workgroups of one wave, one per CU; the C members of a tile publish an 8-byte {value, epoch} granule each and
// gather all C of them, step after step, exactly as reducer_section (gpfq_loop_kernels.h) does -- without any sweep.
// Prints microseconds per step by members, placement (members of a tile spread over the XCDs / on one XCD), variant of
// the store / load pair and length of the pause that stands in for the sweep.
//   hipcc --offload-arch=gfx950 -O3 -o xchg_probe xchg_probe.hip && ./xchg_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define XB_BYTES (64u << 20)
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

// VARIANT 0: relaxed agent-scope atomic store / load (global_store_dwordx2 sc1, global_load_dwordx2 sc1) -- the product
//         1: system scope (sc0 sc1)
//         2: publish with an atomic exchange, poll with an atomic fetch-add of 0 (both executed at the coherence point)
//         3: store as 0, poll with fetch-add 0
//         4: store as 0, poll with global_load_dwordx2 sc0 -- STALE for ever: workgroup scope, a hit in the CU's own vector
//            cache satisfies it
//         6: store as 0, poll = buffer_inv sc1 + plain load
//         7: store as 0, poll = buffer_inv sc0 + plain load
//         8: plain store, poll = buffer_inv sc0 + plain load
//         9: PLAIN store (the line stays in the XCD's L2), poll with the sc1 load (bypasses the vector L1, L2-served): what
//            members that share an XCD could use (round 3)
//        10: sc0 store, sc1 load
//        11: PLAIN vector store, poll through the SCALAR path (s_load_dwordx16 glc: the scalar cache's port to L2, not the
//            vector-memory queue the column requests stand in) -- members on one XCD only
//        12: scalar store too (s_store_dwordx2 glc + s_dcache_wb)
typedef unsigned v16u __attribute__((ext_vector_type(16)));
template <int OFF>
__device__ __forceinline__ v16u sload16(const void* p)
{
    v16u r;
    asm volatile("s_load_dwordx16 %0, %1, %2 glc" : "=s"(r) : "s"(p), "n"(OFF) : "memory");
    return r;
}
// all C (<= 32) granules of a tile: epochs complete?  sum of the values in v
template <int C>
__device__ __forceinline__ bool scalar_poll(const unsigned long long* xb_, unsigned epoch, float& v)
{
    v16u r0 = sload16<0>(xb_), r1 = r0, r2 = r0, r3 = r0;
    if (C > 8) r1 = sload16<64>(xb_);
    if (C > 16) { r2 = sload16<128>(xb_); r3 = sload16<192>(xb_); }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r0), "+s"(r1), "+s"(r2), "+s"(r3) :: "memory");
    bool ok = true;
    v = 0.0f;
#pragma unroll
    for (int g = 0; g < C; ++g) {
        const v16u& r = g < 8 ? r0 : g < 16 ? r1 : g < 24 ? r2 : r3;
        ok = ok && r[2 * (g & 7) + 1] == epoch;
        v += __uint_as_float(r[2 * (g & 7)]);
    }
    return ok;
}
template <int VARIANT>
__device__ __forceinline__ void publish(unsigned long long* p, unsigned long long v)
{
    if (VARIANT == 12) {
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        const unsigned long long sv = ((unsigned long long)hi << 32) | lo;
        asm volatile("s_store_dwordx2 %0, %1, 0x0 glc\n\ts_dcache_wb\n\ts_waitcnt lgkmcnt(0)" :: "s"(sv), "s"(p) : "memory");
        return;
    }
    if (VARIANT == 8 || VARIANT == 9 || VARIANT == 11) { asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(v) : "memory"); return; }
    if (VARIANT == 10) { asm volatile("global_store_dwordx2 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory"); return; }
    if (VARIANT == 1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else if (VARIANT == 2) (void)__hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int VARIANT>
__device__ __forceinline__ unsigned long long peek(unsigned long long* p)
{
    if (VARIANT == 1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (VARIANT >= 9) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (VARIANT >= 4) {
        unsigned long long r;
        if (VARIANT == 4) asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
        else if (VARIANT == 6) asm volatile("buffer_inv sc1\n\tglobal_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
        else asm volatile("buffer_inv sc0\n\tglobal_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
        return r;
    }
    if (VARIANT >= 2) return __hip_atomic_fetch_add(p, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ int g_lstride = 16;       // granules between the starts of two consecutive 128-byte lines of a block (16 = contiguous)
template <int VARIANT>
__global__ void __launch_bounds__(64) xchg(unsigned long long* xb, int C, int steps, int one_xcd, int work, int first_pause,
                                           unsigned* bad, float* sink)
{
    const int lstride = g_lstride;
#define GOFF(g) (((g) >> 4) * lstride + ((g) & 15))
    const int wg = blockIdx.x, lane = threadIdx.x;
    const int tiles = gridDim.x / C;
    int tile, c;
    if (!one_xcd) { tile = wg / C; c = wg % C; }            // consecutive workgroup ids = round robin over the 8 XCDs
    else { const int xcd = wg & 7, slot = wg >> 3; tile = xcd * (tiles >> 3) + slot / C; c = slot % C; }
    const int blk = ((C + 15) >> 4) * lstride;                 // granules a (tile, parity) block spans
    unsigned long long* base = xb + (size_t)tile * 2 * blk;
    const int per = (C + 63) >> 6;                           // granules per lane
    float acc = 0.0f;
    bool dead = false;
    for (int t = 0; t < steps; ++t) {
        const unsigned epoch = (unsigned)t + 1u;
        unsigned long long* xb_ = base + (size_t)(t & 1) * blk;
        for (int i = 0; i < work; ++i) __builtin_amdgcn_s_sleep(16);
        if (VARIANT == 12) publish<VARIANT>(xb_ + GOFF(c), ((unsigned long long)epoch << 32) | (unsigned)__float_as_uint(1.0f + c));
        else if (lane == 0) publish<VARIANT>(xb_ + GOFF(c), ((unsigned long long)epoch << 32) | (unsigned)__float_as_uint(1.0f + c));
        for (int i = 0; i < first_pause; ++i) __builtin_amdgcn_s_sleep(4);
        unsigned spins = dead ? (1u << 13) : 0u;
        float v = 0.0f;
        if (VARIANT >= 11) {
            for (;;) {
                bool ok;
                switch (C) {
                case 2: ok = scalar_poll<2>(xb_, epoch, v); break;
                case 4: ok = scalar_poll<4>(xb_, epoch, v); break;
                case 8: ok = scalar_poll<8>(xb_, epoch, v); break;
                case 16: ok = scalar_poll<16>(xb_, epoch, v); break;
                default: ok = scalar_poll<32>(xb_, epoch, v); break;
                }
                if (ok) break;
                if (++spins > (1u << 13)) { dead = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            acc += v;
            continue;
        }
        for (;;) {
            unsigned long long ok = ~0ull;
            v = 0.0f;
            for (int i = 0; i < per; ++i) {
                const int g = i * 64 + lane;
                const bool want = g < C;
                const unsigned long long gv = peek<VARIANT>(xb_ + (want ? GOFF(g) : 0));
                ok &= __builtin_amdgcn_ballot_w64(!want || (unsigned)(gv >> 32) == epoch);
                v += want ? __uint_as_float((unsigned)gv) : 0.0f;
            }
            if (ok == __builtin_amdgcn_read_exec()) break;
            if (++spins > (1u << 13)) { dead = true; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        acc += v;
    }
    if (dead && lane == 0) atomicExch(bad, 1u);
    if (lane == 0) sink[wg] = acc;
}

template <int VARIANT>
static float run(unsigned long long* xb, unsigned* bad, float* sink, int C, int steps, int one_xcd, int work, int pause)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(xb, 0, XB_BYTES));
        CK(hipEventRecord(a));
        xchg<VARIANT><<<256, 64>>>(xb, C, steps, one_xcd, work, pause, bad, sink);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    unsigned h = 0; CK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
    if (h) { CK(hipMemset(bad, 0, 4)); return -1.0f; }       // stale for ever: printed as -1
    return best * 1000.0f / steps;
}

int main()
{
    unsigned long long* xb; unsigned* bad; float* sink;
    CK(hipMalloc(&xb, XB_BYTES)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&sink, 256 * 4));
    CK(hipMemset(bad, 0, 4));
    const int steps = 4000;
    if (getenv("XCHG_STRIDE")) {       // round 3: the lines of a tile's granule block spread over memory (bytes between line starts)
        const int strides[] = {128, 256, 512, 1024, 4096, 65536};
        printf("sc1 store / sc1 load, members spread over the XCDs, sweep stand-in 2 x s_sleep 16, first poll after 3 x s_sleep 4\n%-8s", "members");
        for (int sb : strides) printf(" %8d", sb);
        printf("\n");
        for (int C = 16; C <= 256; C <<= 1) {
            printf("%-8d", C);
            for (int sb : strides) {
                int ls = sb / 8;
                CK(hipMemcpyToSymbol(HIP_SYMBOL(g_lstride), &ls, sizeof(int)));
                printf(" %8.3f", run<0>(xb, bad, sink, C, steps, 0, 2, 3));
                fflush(stdout);
            }
            printf("\n");
        }
        return 0;
    }
    if (getenv("XCHG_SCALAR")) {        // round 3: polls (and stores) through the scalar path, members on one XCD
        for (int work = 0; work <= 2; work += 2) {
            printf("sweep stand-in %d x s_sleep 16\n%-8s %9s %12s %14s %14s\n", work, "members", "sc1/sc1", "plain/sc1", "plain/s_load", "s_store/s_load");
            for (int C = 2; C <= 32; C <<= 1) {
                printf("%-8d %9.3f %12.3f %14.3f", C, run<0>(xb, bad, sink, C, steps, 1, work, 0), run<9>(xb, bad, sink, C, steps, 1, work, 0),
                       run<11>(xb, bad, sink, C, steps, 1, work, 0));
                fflush(stdout);
                printf(" %14.3f\n", getenv("XCHG_SSTORE") ? run<12>(xb, bad, sink, C, steps, 1, work, 0) : 0.0f);
                fflush(stdout);
            }
        }
        return 0;
    }
    if (getenv("XCHG_ROUND3")) {        // round 3: plain / sc0 stores with sc1 loads, members on one XCD (and spread, for the record)
        for (int work = 0; work <= 2; work += 2) {
            printf("sweep stand-in %d x s_sleep 16\n%-8s %-10s %9s %12s %12s\n", work, "members", "placement", "sc1/sc1", "plain/sc1", "sc0/sc1");
            for (int C = 2; C <= 32; C <<= 1)
                for (int one = 1; one >= 0; --one) {
                    printf("%-8d %-10s %9.3f %12.3f %12.3f\n", C, one ? "one XCD" : "spread", run<0>(xb, bad, sink, C, steps, one, work, 0),
                           run<9>(xb, bad, sink, C, steps, one, work, 0), run<10>(xb, bad, sink, C, steps, one, work, 0));
                    fflush(stdout);
                }
        }
        return 0;
    }
    for (int work = 0; work <= 2; work += 2) {
        const float base0 = run<0>(xb, bad, sink, 1, steps, 0, work, 0);
        printf("pause standing in for the sweep: %d x s_sleep 16; a step without partners (C = 1): %.3f us   (-1: stale for ever)\n", work, base0);
        printf("%-8s %-10s %9s %9s %11s %11s %11s\n", "members", "placement", "sc1", "xchg/add0", "inv sc1+ld", "inv sc0+ld", "st,inv0+ld");
        for (int C = 2; C <= 256; C <<= 1) {
            for (int one = 0; one <= 1; ++one) {
                if (one && C > 32) continue;
                printf("%-8d %-10s %9.3f %9.3f %11.3f %11.3f %11.3f\n", C, one ? "one XCD" : "spread", run<0>(xb, bad, sink, C, steps, one, work, 0),
                       run<2>(xb, bad, sink, C, steps, one, work, 0), run<6>(xb, bad, sink, C, steps, one, work, 0),
                       run<7>(xb, bad, sink, C, steps, one, work, 0), run<8>(xb, bad, sink, C, steps, one, work, 0));
                fflush(stdout);
            }
        }
    }
    // the first poll of an exchange after n x s_sleep 4 (256 clocks each): a poll that comes back incomplete costs a round trip
    printf("first poll after n x s_sleep 4 (sc1 store / load, sweep stand-in 2 x s_sleep 16)\n%-8s %-10s", "members", "placement");
    const int ns[] = {0, 1, 2, 3, 4, 6, 8, 12, 16};
    for (int n : ns) printf(" %7d", n);
    printf("\n");
    for (int C = 8; C <= 256; C <<= 1) {
        for (int one = 0; one <= 1; ++one) {
            if (one && C > 32) continue;
            printf("%-8d %-10s", C, one ? "one XCD" : "spread");
            for (int n : ns) printf(" %7.3f", run<0>(xb, bad, sink, C, steps, one, 2, n));
            printf("\n");
            fflush(stdout);
        }
    }
    return 0;
}
