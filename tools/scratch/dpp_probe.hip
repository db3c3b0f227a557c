#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* in, float* out) {
    float v = in[threadIdx.x];
    float w = v + 100.0f;
    int l = threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, w), false, false);
    out[0 * 64 + l] = __builtin_bit_cast(float, r[0]);
    out[1 * 64 + l] = __builtin_bit_cast(float, r[1]);
    auto r2 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, w), false, false);
    out[2 * 64 + l] = __builtin_bit_cast(float, r2[0]);
    out[3 * 64 + l] = __builtin_bit_cast(float, r2[1]);
}
int main() {
    float *in, *o; (void)hipMalloc(&in, 256); (void)hipMalloc(&o, 4 * 256);
    float h[64]; for (int i = 0; i < 64; ++i) h[i] = (float)i;
    (void)hipMemcpy(in, h, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(in, o);
    float a[4 * 64]; (void)hipMemcpy(a, o, 4 * 256, hipMemcpyDeviceToHost);
    const char* nm[4] = {"p16[0]", "p16[1]", "p32[0]", "p32[1]"};
    for (int s = 0; s < 4; ++s) { printf("%-8s", nm[s]); for (int i = 0; i < 64; ++i) printf(" %3d", (int)a[s * 64 + i]); printf("\n"); }
    return 0;
}
