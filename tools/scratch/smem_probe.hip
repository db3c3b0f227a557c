// Probe (gfx950): can two workgroups exchange an 8-byte {value, epoch} granule through the SCALAR memory path
// (s_store_dwordx2 + s_dcache_wb / s_load_dwordx2 glc), how long does a round trip take compared with the
// vector agent-scope path (global_store/global_load sc1), and which XCC do blocks land on?
//   hipcc --offload-arch=gfx950 -O2 -o smem_probe smem_probe.hip && ./smem_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ unsigned xcc_id()
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v;
}

__device__ __forceinline__ void sstore2(unsigned long long* p, unsigned lo, unsigned hi)
{
    unsigned long long v = ((unsigned long long)hi << 32) | lo;
    asm volatile("s_store_dwordx2 %0, %1, 0x0 glc\n\ts_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::"s"(v), "s"(p) : "memory");
}
__device__ __forceinline__ unsigned long long sload2(const unsigned long long* p)
{
    unsigned long long v;
    asm volatile("s_load_dwordx2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}

__device__ __forceinline__ unsigned long long vload2_sc0(const unsigned long long* p)
{
    unsigned long long v;
    asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ unsigned long long vload2_plain(const unsigned long long* p)
{
    unsigned long long v;
    asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void vstore2_plain(unsigned long long* p, unsigned long long v)
{
    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void vstore2_sc0(unsigned long long* p, unsigned long long v)
{
    asm volatile("global_store_dwordx2 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ unsigned long long vload2_inv(const unsigned long long* p, int which)
{
    unsigned long long v;
    if (which == 0) asm volatile("buffer_inv sc0\n\tglobal_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (which == 1) asm volatile("buffer_inv sc1\n\tglobal_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx2 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ unsigned long long xload(const unsigned long long* p, int mode)
{
    if (mode >= 5) return vload2_inv(p, mode - 5);
    if (mode == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (mode == 1) return sload2(p);
    if (mode == 4) return vload2_plain(p);
    return vload2_sc0(p);                         // modes 2, 3
}
__device__ __forceinline__ void xstore(unsigned long long* p, unsigned lo, unsigned hi, int mode)
{
    const unsigned long long v = ((unsigned long long)hi << 32) | lo;
    if (mode == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (mode == 1) sstore2(p, lo, hi);
    else if (mode == 3) vstore2_sc0(p, v);
    else vstore2_plain(p, v);                     // modes 2, 4
}

// mode 0: vector agent-scope atomics (sc1); 1: scalar path; 2: plain store + sc0 load (L1 bypass, L2 hit);
// 3: sc0 store + sc0 load; 4: plain store + plain load (expected to fail: the poll may hit a stale L1 line).  Blocks a and b ping-pong `iters` times through
// slots[0] (a -> b) and slots[1] (b -> a); every other block optionally streams `bg` bytes per iteration through
// its vector memory pipe to load the CU queues (background = 1 also in the two ping-pong blocks' other waves).
__global__ void pingpong(unsigned long long* slots, int a, int b, int iters, int mode, const float4* bgbuf, int bg_iters,
                         unsigned long long* out, unsigned* xcc, unsigned limit)
{
    const int blk = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x == 0) xcc[blk] = xcc_id();
    if (wave > 0) {
        // background stream in the same CU: bg_iters x 1 KB per wave
        float4 acc = make_float4(0, 0, 0, 0);
        const float4* p = bgbuf + (size_t)(blk * 16 + wave) * 65536 + lane;
        for (int i = 0; i < bg_iters; ++i) {
            float4 v = p[(size_t)(i & 1023) * 64];
            acc.x += v.x; acc.y += v.y;
        }
        if (acc.x == 12345.678f) out[100] = 1;
        return;
    }
    if (blk != a && blk != b) return;
    unsigned long long* mine = slots + (blk == a ? 0 : 16);
    unsigned long long* theirs = slots + (blk == a ? 16 : 0);
    unsigned long long t0 = 0, t1 = 0;
    bool fail = false;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 1; it <= iters && !fail; ++it) {
        if (blk == a) {
            // send it, then wait for the echo
            xstore(mine, (unsigned)(it * 3), (unsigned)it, mode);
            unsigned spins = 0;
            for (;;) {
                unsigned long long v = xload(theirs, mode);
                if ((unsigned)(v >> 32) == (unsigned)it) { if ((unsigned)v != (unsigned)(it * 5)) fail = true; break; }
                if (++spins > limit) { fail = true; break; }
            }
        } else {
            unsigned spins = 0;
            for (;;) {
                unsigned long long v = xload(theirs, mode);
                if ((unsigned)(v >> 32) == (unsigned)it) { if ((unsigned)v != (unsigned)(it * 3)) fail = true; break; }
                if (++spins > limit) { fail = true; break; }
            }
            xstore(mine, (unsigned)(it * 5), (unsigned)it, mode);
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (lane == 0) { out[blk == a ? 0 : 1] = t1 - t0; out[blk == a ? 2 : 3] = fail ? 1 : 0; }
}

int main()
{
    unsigned long long *slots, *out;
    unsigned* xcc;
    float4* bg;
    const int nblk = 256;
    CK(hipMalloc(&slots, 4096));
    CK(hipMalloc(&out, 4096));
    CK(hipMalloc(&xcc, nblk * 4));
    CK(hipMalloc(&bg, (size_t)nblk * 16 * 65536 * sizeof(float4) + (1 << 20)));
    CK(hipMemset(bg, 0, (size_t)nblk * 16 * 65536 * sizeof(float4)));
    std::vector<unsigned> hx(nblk);
    const int iters = 2000;
    int pairs[3][2] = {{0, 8}, {0, 1}, {0, 16}};
    for (int bgw = 0; bgw <= 1; ++bgw) {
        for (int pi = 0; pi < 3; ++pi) {
            for (int mode = 0; mode <= 7; ++mode) {
                if (mode >= 2 && mode <= 4) continue;
                CK(hipMemset(slots, 0, 4096));
                CK(hipMemset(out, 0, 4096));
                const int threads = bgw ? 64 * 12 : 64;
                hipLaunchKernelGGL(pingpong, dim3(nblk), dim3(threads), 0, 0, slots, pairs[pi][0], pairs[pi][1], iters, mode, bg,
                                   bgw ? 200000 : 0, out, xcc, 20000u);
                CK(hipDeviceSynchronize());
                unsigned long long ho[4];
                CK(hipMemcpy(ho, out, sizeof(ho), hipMemcpyDeviceToHost));
                CK(hipMemcpy(hx.data(), xcc, nblk * 4, hipMemcpyDeviceToHost));
                printf("background=%d blocks (%d,%d) xcc (%u,%u) mode=%s: %.0f ticks per round trip%s\n", bgw, pairs[pi][0], pairs[pi][1],
                       hx[pairs[pi][0]], hx[pairs[pi][1]],
                       mode == 0 ? "vector-sc1" : mode == 1 ? "scalar" : mode == 2 ? "st-plain/ld-sc0" : mode == 3 ? "st-sc0/ld-sc0" : mode == 4 ? "plain/plain" : mode == 5 ? "inv-sc0+plain" : mode == 6 ? "inv-sc1+plain" : "plain/ld-nt",
                       (double)ho[0] / iters,
                       (ho[2] || ho[3]) ? "  FAILED/TIMED OUT" : "");
                fflush(stdout);
            }
        }
    }
    printf("xcc of blocks 0..15:");
    for (int i = 0; i < 16; ++i) printf(" %u", hx[i]);
    printf("\n");
    return 0;
}
