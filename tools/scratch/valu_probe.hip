// What does the sweep's instruction mix cost on a SIMD?  cycles per instruction for (a) the five-instruction group of
// win_sweep4_pair_lds (two v_pk_mul_f32, two v_pk_add_f32, one v_pk_fma_f32), (b) v_pk_mul only, (c) v_pk_fma only,
// (d) v_fmac_f32 only -- with one and with two waves per SIMD.   hipcc --offload-arch=gfx950 -O3 valu_probe.hip -o valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ void probe(unsigned long long* out, int iters, float seed)
{
    v2f u0 = {seed, seed}, u1 = {seed, 1.f}, u2 = {2.f, seed}, u3 = {seed, 3.f}, acc = {0.f, 0.f}, t0 = {seed, 2.f}, t1 = {3.f, seed};
    v2f q = {seed, seed}, w = {seed, seed}, xp = {seed, 1.f}, a = {1.f, seed}, x = {seed, seed};
    float f0 = seed, f1 = 2.f * seed, f2 = 0.5f * seed, f3 = seed;
    unsigned long long t_begin, t_end;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_begin));
    for (int i = 0; i < iters; ++i) {
#define GRP(U)                                                                                     \
        if (KIND == 0) asm volatile("v_pk_mul_f32 %1, %4, %6 op_sel:[0,0] op_sel_hi:[1,0]\n\t"        \
                                    "v_pk_mul_f32 %2, %5, %7 op_sel:[0,0] op_sel_hi:[1,0]\n\t"        \
                                    "v_pk_add_f32 %3, %3, %1 neg_lo:[0,1] neg_hi:[0,1]\n\t"           \
                                    "v_pk_add_f32 %3, %3, %2\n\t"                                     \
                                    "v_pk_fma_f32 %0, %3, %8, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]"    \
                                    : "+v"(acc), "=&v"(t0), "=&v"(t1), "+v"(U) : "v"(q), "v"(w), "v"(xp), "v"(a), "v"(x)); \
        if (KIND == 1) asm volatile("v_pk_mul_f32 %0, %2, %3\n\tv_pk_mul_f32 %1, %2, %4\n\tv_pk_mul_f32 %0, %2, %3\n\tv_pk_mul_f32 %1, %2, %4\n\tv_pk_mul_f32 %0, %2, %3" : "=&v"(t0), "=&v"(t1) : "v"(q), "v"(U), "v"(xp)); \
        if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %2, %3, %0\n\tv_pk_fma_f32 %1, %2, %3, %1\n\tv_pk_fma_f32 %0, %2, %3, %0\n\tv_pk_fma_f32 %1, %2, %3, %1\n\tv_pk_fma_f32 %0, %2, %3, %0" : "+v"(acc), "+v"(t0) : "v"(U), "v"(x)) ; \
        if (KIND == 3) asm volatile("v_fmac_f32 %0, %2, %3\n\tv_fmac_f32 %1, %2, %3\n\tv_fmac_f32 %0, %2, %3\n\tv_fmac_f32 %1, %2, %3\n\tv_fmac_f32 %0, %2, %3" : "+v"(f0), "+v"(f1) : "v"(f2), "v"(f3));
        GRP(u0) GRP(u1) GRP(u2) GRP(u3) GRP(u0) GRP(u1) GRP(u2) GRP(u3) GRP(u0) GRP(u1) GRP(u2) GRP(u3) GRP(u0) GRP(u1) GRP(u2) GRP(u3)
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_end));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t_end - t_begin;
    if (acc.x + f0 + f1 + u0.x + u1.x + u2.x + u3.x == 12345.f && t0.x == 1.f && t1.x == 2.f) out[0] = 0;
}
template <int KIND>
void run(const char* name, int threads)
{
    unsigned long long* d;
    const int blocks = 256, waves = blocks * threads / 64, iters = 2000;
    hipMalloc(&d, sizeof(unsigned long long) * waves);
    hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0f);
    hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0f);
    hipDeviceSynchronize();
    unsigned long long* h = new unsigned long long[waves];
    hipMemcpy(h, d, sizeof(unsigned long long) * waves, hipMemcpyDeviceToHost);
    double mx = 0, sum = 0;
    for (int i = 0; i < waves; ++i) { sum += h[i]; if (h[i] > mx) mx = h[i]; }
    const double insts = 80.0 * iters;
    printf("%-34s %d waves/SIMD: %.2f cycles per instruction per wave (slowest wave %.2f) -> %.2f cycles of the SIMD per instruction\n", name,
           threads / 256, sum / waves / insts, mx / insts, mx / insts / (threads / 256));
    hipFree(d);
    delete[] h;
}
int main()
{
    for (int th : {256, 512}) {
        run<0>("sweep group (2 mul, 2 add, 1 fma)", th);
        run<1>("v_pk_mul_f32", th);
        run<2>("v_pk_fma_f32 (two chains)", th);
        run<3>("v_fmac_f32 (two chains)", th);
    }
    return 0;
}
