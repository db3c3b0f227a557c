// Probe (GPU box): what the shape of the tile does to a (m, D) -> (D, m) fp32 transpose that is otherwise built like
// gpfq_transpose_norm_kernel (registers -> LDS -> 16-byte stores, next block in flight while this one leaves).
//   BR x TC tile: a row of the input contributes TC * 4 contiguous bytes per read, a column of the output gets BR * 4
//   contiguous bytes per tile.  256 x 64 is the product's; 512 x 32 doubles the write run, 128 x 128 doubles the read run.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/transpose_probe tools/scratch/transpose_probe.hip && /tmp/transpose_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int BR, int TC>
__global__ void __launch_bounds__(256) tr_kernel(const float* __restrict__ in, int64_t ld, int64_t m, int64_t D, float* __restrict__ out,
                                                 int64_t m_pad, int S, int ntile, int G)
{
    extern __shared__ float tile[];                 // [BR][TC + 1]
    constexpr int TS = TC + 1;
    constexpr int NV = BR * TC / 4 / 256;           // 16-byte loads per thread and block
    constexpr int LPR = TC / 4;                     // lanes per row
    constexpr int RPP = 256 / LPR;                  // rows per pass of the 256 threads
    constexpr int NB = 1024 / BR;                   // blocks per 1024-sample segment
    unsigned idx = blockIdx.x;
    const int tig = (int)(idx % (unsigned)G); idx /= (unsigned)G;
    const int s = (int)(idx % (unsigned)S);
    const int grp = (int)(idx / (unsigned)S);
    const int tile_y = grp * G + tig;
    if (tile_y >= ntile) return;
    const int t0 = tile_y * TC;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c4 = (tid % LPR) * 4, r0 = tid / LPR;
    float4 v[NV];
    auto load_block = [&](int c) {
        const int64_t kb = (int64_t)s * 1024 + BR * c;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int64_t k = kb + r0 + RPP * i;
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < m && t0 + c4 < D) r = *reinterpret_cast<const float4*>(in + k * ld + t0 + c4);
            v[i] = r;
        }
    };
    load_block(0);
#pragma unroll 1
    for (int c = 0; c < NB; ++c) {
        if (c > 0) __syncthreads();
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            float* row = tile + (r0 + RPP * i) * TS + c4;
            row[0] = v[i].x; row[1] = v[i].y; row[2] = v[i].z; row[3] = v[i].w;
        }
        __syncthreads();
        if (c + 1 < NB) load_block(c + 1);
        const int64_t kb = (int64_t)s * 1024 + BR * c;
#pragma unroll
        for (int j = 0; j < TC / 4; ++j) {
            const int tt = wave + 4 * j;
            if (t0 + tt < D) {
#pragma unroll
                for (int h = 0; h < BR / 256; ++h) {
                    const float* col = tile + (256 * h + 4 * lane) * TS;
                    *reinterpret_cast<float4*>(out + (int64_t)(t0 + tt) * m_pad + kb + 256 * h + 4 * lane) =
                        make_float4(col[tt], col[TS + tt], col[2 * TS + tt], col[3 * TS + tt]);
                }
            }
        }
    }
}

// 128-row blocks: a wave's 64 lanes cover 128 samples of TWO columns (lanes 0..31 column tt, 32..63 column tt + 4 * ...)
template <int TC>
__global__ void __launch_bounds__(256) tr128_kernel(const float* __restrict__ in, int64_t ld, int64_t m, int64_t D, float* __restrict__ out,
                                                    int64_t m_pad, int S, int ntile, int G)
{
    extern __shared__ float tile[];                 // [128][TC + 1]
    constexpr int BR = 128, TS = TC + 1, NV = BR * TC / 4 / 256, LPR = TC / 4, RPP = 256 / LPR, NB = 1024 / BR;
    unsigned idx = blockIdx.x;
    const int tig = (int)(idx % (unsigned)G); idx /= (unsigned)G;
    const int s = (int)(idx % (unsigned)S);
    const int grp = (int)(idx / (unsigned)S);
    const int tile_y = grp * G + tig;
    if (tile_y >= ntile) return;
    const int t0 = tile_y * TC;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c4 = (tid % LPR) * 4, r0 = tid / LPR;
    float4 v[NV];
    auto load_block = [&](int c) {
        const int64_t kb = (int64_t)s * 1024 + BR * c;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int64_t k = kb + r0 + RPP * i;
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < m && t0 + c4 < D) r = *reinterpret_cast<const float4*>(in + k * ld + t0 + c4);
            v[i] = r;
        }
    };
    load_block(0);
    const int half = lane >> 5, l32 = lane & 31;
#pragma unroll 1
    for (int c = 0; c < NB; ++c) {
        if (c > 0) __syncthreads();
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            float* row = tile + (r0 + RPP * i) * TS + c4;
            row[0] = v[i].x; row[1] = v[i].y; row[2] = v[i].z; row[3] = v[i].w;
        }
        __syncthreads();
        if (c + 1 < NB) load_block(c + 1);
        const int64_t kb = (int64_t)s * 1024 + BR * c;
#pragma unroll
        for (int j = 0; j < TC / 8; ++j) {
            const int tt = 2 * (wave + 4 * j) + half;
            if (t0 + tt < D) {
                const float* col = tile + (4 * l32) * TS;
                *reinterpret_cast<float4*>(out + (int64_t)(t0 + tt) * m_pad + kb + 4 * l32) =
                    make_float4(col[tt], col[TS + tt], col[2 * TS + tt], col[3 * TS + tt]);
            }
        }
    }
}

template <typename F>
static double time_ms(F f, int n = 10)
{
    f();
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < n; ++i) f();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / n;
}

template <int BR, int TC, bool K128 = false>
static void run(const char* tag, const float* in, float* out, float* ref, int64_t m, int64_t D, int64_t m_pad, int Gbytes)
{
    const int S = (int)(m_pad / 1024);
    const int ntile = (int)((D + TC - 1) / TC);
    int G = Gbytes / (TC * 4);
    if (G < 1) G = 1;
    if (G > ntile) G = ntile;
    const int64_t ngroups = (ntile + G - 1) / G;
    const int64_t nblocks = ngroups * S * G;
    const size_t shm = (size_t)BR * (TC + 1) * sizeof(float);
    auto launch = [&]() {
        if constexpr (K128) {
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(tr128_kernel<TC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
            hipLaunchKernelGGL((tr128_kernel<TC>), dim3((unsigned)nblocks), dim3(256), shm, 0, in, D, m, D, out, m_pad, S, ntile, G);
        } else {
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(tr_kernel<BR, TC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
            hipLaunchKernelGGL((tr_kernel<BR, TC>), dim3((unsigned)nblocks), dim3(256), shm, 0, in, D, m, D, out, m_pad, S, ntile, G);
        }
    };
    CK(hipMemset(out, 0, (size_t)D * m_pad * 4));
    const double ms = time_ms(launch);
    // check against the reference transpose (first variant run)
    bool same = true;
    if (ref != out) {
        std::vector<float> a((size_t)1 << 16), b((size_t)1 << 16);
        for (int64_t off : {(int64_t)0, (int64_t)D * m_pad / 2, (int64_t)D * m_pad - (1 << 16)}) {
            CK(hipMemcpy(a.data(), out + off, a.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(b.data(), ref + off, b.size() * 4, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < a.size(); ++i) same = same && (a[i] == b[i]);
        }
    }
    const double by = 4.0 * ((double)m * D + (double)m_pad * D);
    printf("  %-22s %7.3f ms %5.2f TB/s%s\n", tag, ms, by / ms / 1e9, same ? "" : "  DIFFERENT");
}

int main()
{
    const int64_t shapes[][2] = {{93184, 576}, {93184, 1152}, {26624, 2304}, {7168, 4608}};
    for (auto& sh : shapes) {
        const int64_t m = sh[0], D = sh[1], m_pad = (m + 1023) / 1024 * 1024;
        float *in, *out, *ref;
        CK(hipMalloc(&in, (size_t)m * D * 4));
        CK(hipMalloc(&out, (size_t)D * m_pad * 4));
        CK(hipMalloc(&ref, (size_t)D * m_pad * 4));
        std::vector<float> h((size_t)m * D);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) * 1e-3f;
        CK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        printf("m=%lld D=%lld (%.0f MB moved; ONE matrix, no norms)\n", (long long)m, (long long)D, 4e-6 * ((double)m * D + (double)m_pad * D));
        run<256, 64>("256x64 (product) G=2K", in, ref, ref, m, D, m_pad, 2048);
        run<256, 64>("256x64 G=1 tile", in, out, ref, m, D, m_pad, 256);
        run<256, 64>("256x64 G=4K", in, out, ref, m, D, m_pad, 4096);
        run<256, 32>("256x32 G=2K", in, out, ref, m, D, m_pad, 2048);
        run<512, 32>("512x32 G=2K", in, out, ref, m, D, m_pad, 2048);
        run<512, 64>("512x64 G=2K", in, out, ref, m, D, m_pad, 2048);
        run<1024, 32>("1024x32 G=2K", in, out, ref, m, D, m_pad, 2048);
        run<128, 128, true>("128x128 G=2K", in, out, ref, m, D, m_pad, 2048);
        run<128, 64, true>("128x64 G=2K", in, out, ref, m, D, m_pad, 2048);
        CK(hipFree(in)); CK(hipFree(out)); CK(hipFree(ref));
    }
    return 0;
}
