#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../quantized_neural_nets_amd/csrc/gpfq_device.h"
__global__ void k(const float* in, float* out, float* out2, float* out3) {
    float v = in[threadIdx.x];
    out[threadIdx.x] = gpfq::wave_tree64(v);
    {   // lane-63 form: publish lane 63's value to every lane for the comparison below
        float h = gpfq::wave_tree64_lane63(v);
        float h63 = __shfl(h, 63, 64);
        if (h63 != out[threadIdx.x]) out[threadIdx.x] = -12345.0f;
    }
    float w = v;
    for (int off = 1; off < 64; off <<= 1) w = w + __shfl_xor(w, off, 64);
    out2[threadIdx.x] = w;
    for (int nl = 1, j = 0; nl <= 64; nl <<= 1, ++j) {
        float z = threadIdx.x < nl ? v : 0.0f;
        float a = gpfq::wave_tree_n(z, nl);
        float b = z;
        for (int off = 1; off < 64; off <<= 1) b = b + __shfl_xor(b, off, 64);
        out3[j * 64 + threadIdx.x] = (threadIdx.x == 0) ? (a == b ? 1.0f : 0.0f) : 1.0f;
    }
}
int main() {
    float *in, *o1, *o2, *o3; (void)hipMalloc(&in, 256); (void)hipMalloc(&o1, 256); (void)hipMalloc(&o2, 256); (void)hipMalloc(&o3, 7 * 256);
    float h[64]; for (int i = 0; i < 64; ++i) h[i] = 1.0f / (i + 3) * ((i % 3) ? 1 : -1) * 1.234567f;
    (void)hipMemcpy(in, h, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(in, o1, o2, o3);
    float a[64], b[64], c[7 * 64]; (void)hipMemcpy(a, o1, 256, hipMemcpyDeviceToHost); (void)hipMemcpy(b, o2, 256, hipMemcpyDeviceToHost);
    (void)hipMemcpy(c, o3, 7 * 256, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 64; ++i) if (a[i] != b[i] || a[i] != a[0]) bad++;
    int bad2 = 0; for (int i = 0; i < 7 * 64; ++i) if (c[i] != 1.0f) bad2++;
    printf("bad=%d bad_partial=%d a0=%.9g b0=%.9g\n", bad, bad2, a[0], b[0]);
    return bad + bad2;
}
