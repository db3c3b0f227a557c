// gpfq_prep_kernels.h -- everything around the loop (gfx950): column preparation (transpose + pad, canonical column
// norms), the fused conv activation capture, the standalone quantizers and the row |w| maximum.
// Reference: the strided column reads and norm of step_algorithm.py:141-144, SaveInputConv2d.__call__
// (quantize_neural_net.py:334-347), the quantizers step_algorithm.py:7-104 and the radius statistic :191.
#pragma once
#include "gpfq_device.h"

namespace gpfq {

// ------------------------------------------------------------------------------------------------
// Column preparation
// ------------------------------------------------------------------------------------------------
// out[t][k] = in[k][t] for k < m, 0 for m <= k < m_pad.  64x64 tiles via LDS; grid = (m_pad / 64, ceil(D / 64), 2).
// The loop kernel that follows reads the columns ONCE, from column 0 up, and its step is sensitive to where a column
// comes from (N = 256, m = 7 168: 0.65 us per column out of the 256-MB Infinity Cache, 0.82 out of HBM -- the loads
// run only two steps ahead).  So the column tiles are written from the LAST one down, A and X of a tile back to back
// (workgroups are dispatched in increasing (z, y, x) order): what the cache still holds when the loop starts is the
// columns it needs first.  The norm kernel walks the columns downwards for the same reason.
__global__ void __launch_bounds__(256) gpfq_transpose_pad_kernel(const float* __restrict__ A, int64_t lda,
                                                                 const float* __restrict__ X, int64_t ldx,
                                                                 int64_t m, int64_t D, float* __restrict__ AT,
                                                                 float* __restrict__ XT, int64_t m_pad)
{
    __shared__ float tile[64][65];
    const unsigned virt = blockIdx.z * gridDim.y + blockIdx.y;        // position in dispatch order, 0 .. 2 gridDim.y - 1
    const bool second = virt & 1u;                                     // X behind A of the same column tile
    const int64_t tile_y = (int64_t)gridDim.y - 1 - (virt >> 1);       // last column tile first
    const float* __restrict__ in = second ? X : A;
    const int64_t ld = second ? ldx : lda;
    float* __restrict__ out = second ? XT : AT;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t k0 = (int64_t)blockIdx.x * 64, t0 = tile_y * 64;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int64_t k = k0 + ty + 4 * i, t = t0 + tx;
        tile[ty + 4 * i][tx] = (k < m && t < D) ? in[k * ld + t] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int64_t t = t0 + ty + 4 * i, k = k0 + tx;
        if (t < D) out[t * m_pad + k] = tile[tx][ty + 4 * i];
    }
}

// The same transpose for FEW columns (D <= 64) of contiguous matrices (lda == ldx == D): a 64-wide tile would leave most
// of its lanes idle and read D * 4 bytes per row.  Here 256 rows of the matrix are one contiguous run of 256 * D floats:
// read flat (fully coalesced), staged in LDS with an odd row stride, written as D runs of 256 floats.
// grid = (m_pad / 256, 2): y = 0 the analog matrix, 1 the quantized one.
__global__ void __launch_bounds__(256) gpfq_transpose_pad_small_kernel(const float* __restrict__ A, const float* __restrict__ X,
                                                                       int64_t m, int D, float* __restrict__ AT,
                                                                       float* __restrict__ XT, int64_t m_pad)
{
    extern __shared__ float tile[];                 // [256][D | 1]
    const float* __restrict__ in = blockIdx.y ? X : A;
    float* __restrict__ out = blockIdx.y ? XT : AT;
    const int ldt = D | 1;
    const int64_t k0 = (int64_t)blockIdx.x * 256;
    const int64_t rows = (m - k0 < 256) ? (m - k0 > 0 ? m - k0 : 0) : 256;      // valid rows of this tile
    const int64_t nflat = rows * D;
    const float* __restrict__ src = in + k0 * D;
    for (int f = threadIdx.x; f < 256 * D; f += 256) {
        const int k = f / D, t = f - k * D;
        tile[k * ldt + t] = (f < nflat) ? src[f] : 0.0f;
    }
    __syncthreads();
    for (int t = 0; t < D; ++t) out[(int64_t)t * m_pad + k0 + threadIdx.x] = tile[threadIdx.x * ldt + t];
}

// nrm2[2t] = (sqrt(cdot(x_t, x_t)))^2, canonical order, nrm2[2t + 1] = its reciprocal (0 for a zero column).
// One workgroup per column, 4 waves -- or 16 when there are few columns (a layer with 16 columns of 3.2 M samples would
// otherwise keep 16 x 4 waves busy on the whole chip); dynamic LDS S floats.
__global__ void __launch_bounds__(1024) gpfq_colnorm_kernel(const float* __restrict__ XT, int64_t m_pad, int S,
                                                           float* __restrict__ nrm2)
{
    extern __shared__ float seg[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t col = (int64_t)gridDim.x - 1 - blockIdx.x;           // last column first (see gpfq_transpose_pad_kernel)
    const float* __restrict__ x = XT + col * m_pad + 4 * lane;
    const int nwaves = blockDim.x >> 6;
    for (int s = wave; s < S; s += nwaves) {
        float xv[16];
        load16(xv, x + (int64_t)s * kSeg);
        float acc = 0.0f;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc = __builtin_fmaf(xv[e], xv[e], acc);
        float sg = wave_tree64_lane63(acc);
        if (lane == 63) seg[s] = sg;
    }
    __syncthreads();
    if (wave == 0) {
        const int P = pow2_ceil(S);
        const int per = P > 64 ? P / 64 : 1, nl = P > 64 ? 64 : P;
        const SlotMap smap = make_slot_map(S, P, 0, per, lane, nl);
        float tot = combine_slots<true>(seg, smap, per, nl, S - 1);
        float r = sqrtf(tot);
        if (lane == 0) {
            const float n2 = r * r;
            nrm2[2 * col] = n2;
            nrm2[2 * col + 1] = (n2 > 0.0f) ? 1.0f / n2 : 0.0f;      // for quant_msq_from_dot (gpfq_device.h)
        }
    }
}

// Activation capture for Conv2d layers, fused: the sampled kernel-sized patches of an NCHW feature map go
// straight into the transposed, zero-padded column layout the loop kernels read (row f = feature (c, i, j),
// channel-major; column k = sampled patch k).  Patches sit on a grid whose stride is the KERNEL SIZE, as the
// reference's nn.Unfold(kernel_size, dilation, padding, kernel_size) does (quantize_neural_net.py:320).
// Replaces unfold + transpose + reshape + index (quantize_neural_net.py:334-347) and gpfq_transpose_pad_kernel.
// blockDim = 256: 64 patches x 4 feature lanes; grid = (m_pad / 64, ceil(D / 64)).
__global__ void __launch_bounds__(256) gpfq_gather_patches_kernel(const float* __restrict__ x, int C, int H, int W,
                                                                  int kh, int kw, int ph, int pw, int dh, int dw,
                                                                  int Lw, int64_t L, const int64_t* __restrict__ patch,
                                                                  int64_t m, float* __restrict__ outT, int64_t m_pad, int D)
{
    const int kx = threadIdx.x & 63, fy = threadIdx.x >> 6;
    const int64_t k = (int64_t)blockIdx.x * 64 + kx;
    const bool live = k < m;
    int64_t b = 0;
    int y0 = 0, x0 = 0;
    if (live) {
        const int64_t pi = patch[k];
        b = pi / L;
        const int l = (int)(pi - b * L);
        y0 = (l / Lw) * kh - ph;
        x0 = (l % Lw) * kw - pw;
    }
    const float* __restrict__ img = x + b * (int64_t)C * H * W;
    const int tile_y = (int)gridDim.y - 1 - (int)blockIdx.y;           // last feature tile first (see gpfq_transpose_pad_kernel)
    const int f_end = min(D, (tile_y + 1) * 64);
    for (int f = tile_y * 64 + fy; f < f_end; f += 4) {
        const int c = f / (kh * kw), r = f - c * (kh * kw);
        const int yy = y0 + (r / kw) * dh, xx = x0 + (r % kw) * dw;
        float v = 0.0f;
        if (live && yy >= 0 && yy < H && xx >= 0 && xx < W) v = img[((int64_t)c * H + yy) * W + xx];
        outT[(int64_t)f * m_pad + k] = v;
    }
}

__global__ void gpfq_quantizer_kernel(int mode, float step, const float* __restrict__ x, int64_t n, float Kf,
                                      float lamb, const float* __restrict__ uniform, float* __restrict__ out,
                                      int32_t* __restrict__ idx)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int id = 0;
    float q;
    switch (mode) {
    case MODE_SOFT: q = quant_soft(step, x[i], Kf, lamb, id); break;
    case MODE_HARD: q = quant_hard(step, x[i], Kf, lamb, id); break;
    case MODE_STOCHASTIC: q = quant_stochastic(step, x[i], Kf, uniform ? uniform[i] : 0.5f, id); break;
    default: q = quant_msq(step, x[i], Kf, id); break;
    }
    out[i] = q;
    if (idx) idx[i] = id;
}

// rowmax[i] = max_j |W[i][j]|  (max is exact, any order)
__global__ void __launch_bounds__(256) gpfq_row_absmax_kernel(const float* __restrict__ W, int64_t ldw, int64_t d,
                                                              float* __restrict__ rowmax)
{
    __shared__ float part[4];
    const float* __restrict__ w = W + (int64_t)blockIdx.x * ldw;
    float v = 0.0f;
    for (int64_t j = threadIdx.x; j < d; j += blockDim.x) v = fmaxf(v, fabsf(w[j]));
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) rowmax[blockIdx.x] = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
}

}  // namespace gpfq
