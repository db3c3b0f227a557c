#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
__global__ void k(const float* in, float* out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* stage = smem + wave * 2048;
    const float* g = in + wave * 1024 + lane * 4;
    lds_ptr_t l = (lds_ptr_t)(uintptr_t)(stage);
    __builtin_amdgcn_global_load_lds((glb_ptr_t)g, l, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((glb_ptr_t)g, l, 16, 1024, 0);
    __builtin_amdgcn_global_load_lds((glb_ptr_t)g, l, 16, 2048, 0);
    __builtin_amdgcn_global_load_lds((glb_ptr_t)g, l, 16, 3072, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float4 v0 = *reinterpret_cast<float4*>(stage + lane * 4);
    float4 v1 = *reinterpret_cast<float4*>(stage + 256 + lane * 4);
    float4 v2 = *reinterpret_cast<float4*>(stage + 512 + lane * 4);
    float4 v3 = *reinterpret_cast<float4*>(stage + 768 + lane * 4);
    float* o = out + wave * 1024 + lane * 4;
    *reinterpret_cast<float4*>(o) = v0; *reinterpret_cast<float4*>(o + 256) = v1;
    *reinterpret_cast<float4*>(o + 512) = v2; *reinterpret_cast<float4*>(o + 768) = v3;
}
int main() {
    const int n = 4 * 1024; float *in, *out; (void)hipMalloc(&in, n * 4); (void)hipMalloc(&out, n * 4);
    float* h = new float[n]; for (int i = 0; i < n; ++i) h[i] = (float)i;
    (void)hipMemcpy(in, h, n * 4, hipMemcpyHostToDevice); (void)hipMemset(out, 0, n * 4);
    k<<<1, 256, 4 * 8192>>>(in, out);
    float* r = new float[n]; (void)hipMemcpy(r, out, n * 4, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < n; ++i) if (r[i] != h[i]) { if (bad < 5) printf("i=%d got %f\n", i, r[i]); ++bad; }
    printf("glds bad=%d\n", bad); return bad != 0;
}
