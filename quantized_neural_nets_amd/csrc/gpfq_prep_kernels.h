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
    if (in == nullptr) return;                                         // (one matrix only)
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
    if (in == nullptr) return;                      // (one matrix only)
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

// ------------------------------------------------------------------------------------------------
// Column preparation in ONE pass: transpose + pad of A and X AND the canonical sums of squares of X's columns.
//
// A workgroup owns (one canonical segment of 1024 samples) x (64 columns) of one matrix and walks the segment's four
// 256-sample blocks c = 0..3 in order: block -> registers (16-byte loads) -> LDS -> out, where wave w writes columns
// w, w + 4, ... and lane l the four samples 4l .. 4l + 3 of the block as ONE 16-byte store (a column's 256 samples are
// 1 KB contiguous; the segment's four blocks 4 KB).  Those four samples are exactly elements e = 4c .. 4c + 3 of lane
// l's chain in the canonical dot product (gpfq_device.h sweep16: e = 4c + j <-> k = 1024 s + 256 c + 4 l + j), so the
// storing thread extends the column's chain acc = fma(x, x, acc) in the canonical order on its way, and after block 3
// the wave's 64 chains go through the canonical lane tree: one segment sum per (column, segment), bit for bit what
// gpfq_colnorm_kernel computes from XT -- which is then never read again.  gpfq_colnorm_finish_kernel runs the slot
// tree over a column's segment sums.  (Non-temporal loads / stores change nothing measurable: +-3 % on the HBM-bound shapes.)  (Round 2: transpose at 4.4 TB/s with 4-byte stores of 256-byte runs, then a
// second kernel re-reading XT for the norms.)
//
// Dispatch order (1-D grid, x fastest): column tiles in groups of G from the LAST group down (what the Infinity Cache
// still holds when the loop starts is the columns it needs first, see gpfq_transpose_pad_kernel); inside a group the
// segments in order, and for each segment the group's column tiles next to each other, A and X of a tile back to
// back: workgroups that run together read the same input rows side by side (G x 256 contiguous bytes of every row).
// FLAT: the matrix is contiguous with few columns (lda == D <= 64: first convs, EfficientNet's narrow 1x1 convs): a
// 256-row block is one run of 256 * D floats, read flat with 16-byte loads whatever D is.
// VEC (tiled mode): 16-byte loads (needs ld % 4 == 0, D % 4 == 0, an aligned base); otherwise guarded 4-byte loads.
// TC = columns per workgroup: 64 (tile[256][65] floats of dynamic LDS, 66 560 bytes: two workgroups per CU) or 32
// (33 792 bytes: four per CU, half the work per workgroup -- twice as many workgroups to balance over the chip).
// The second half of a block of the one-pass kernels: a 256-sample x TC-column tile leaves LDS.  Wave w writes columns
// w, w + 4, ..., lane l the four samples 4l .. 4l + 3 as one 16-byte store, and extends lane l's canonical chain of the
// column by exactly those four elements (e = 4c .. 4c + 3), in order.
template <int TC>
__device__ __forceinline__ void prep_store_block(const float* tile, float* __restrict__ out, int t0, int ncols, int64_t m_pad,
                                                 int64_t kb, int wave, int lane, float (&acc)[TC / 4])
{
    constexpr int TS = TC + 1;
    const float* col = tile + 4 * lane * TS;
#pragma unroll
    for (int j = 0; j < TC / 4; ++j) {
        const int tt = wave + 4 * j;
        if (tt < ncols) {                           // (wave-uniform)
            const float x0 = col[tt], x1 = col[TS + tt], x2 = col[2 * TS + tt], x3 = col[3 * TS + tt];
            *reinterpret_cast<float4*>(out + (int64_t)(t0 + tt) * m_pad + kb + 4 * lane) = make_float4(x0, x1, x2, x3);
            float a = acc[j];
            a = __builtin_fmaf(x0, x0, a);
            a = __builtin_fmaf(x1, x1, a);
            a = __builtin_fmaf(x2, x2, a);
            a = __builtin_fmaf(x3, x3, a);
            acc[j] = a;
        }
    }
}
// ... and behind the segment's fourth block the wave's 64 chains of every column go through the canonical lane tree
template <int TC>
__device__ __forceinline__ void prep_store_sums(float* __restrict__ part, int t0, int ncols, int S, int s, int wave, int lane,
                                                const float (&acc)[TC / 4])
{
#pragma unroll
    for (int j = 0; j < TC / 4; ++j) {
        const int tt = wave + 4 * j;
        if (tt < ncols) {
            const float sg = wave_tree64_lane63(acc[j]);
            if (lane == 63) part[(int64_t)(t0 + tt) * S + s] = sg;
        }
    }
}

template <bool FLAT, bool VEC, int TC>
__global__ void __launch_bounds__(256) gpfq_transpose_norm_kernel(const float* __restrict__ A, int64_t lda,
                                                                  const float* __restrict__ X, int64_t ldx, int64_t m,
                                                                  int64_t D, float* __restrict__ AT, float* __restrict__ XT,
                                                                  int64_t m_pad, float* __restrict__ part, int S, int ntile,
                                                                  int G)
{
    static_assert(TC == 64 || TC == 32, "columns per workgroup");
    extern __shared__ float tile[];                 // [256][TC + 1]
    constexpr int TS = TC + 1;
    constexpr int NV = TC / 4;                      // 16-byte loads per thread and block: 256 rows x TC columns / 256 threads
    constexpr int LPR = TC / 4;                     // lanes per row of the tile (16 bytes each)
    constexpr int NJ = TC / 4;                      // columns per wave
    // block -> (group of column tiles, segment, matrix, tile in group)
    unsigned idx = blockIdx.x;
    const int tig = (int)(idx % (unsigned)G); idx /= (unsigned)G;
    const bool second = idx & 1u; idx >>= 1;
    const int s = (int)(idx % (unsigned)S);
    const int grp = (int)(idx / (unsigned)S);
    const int tile_rev = grp * G + tig;
    if (tile_rev >= ntile) return;                  // (ragged last group)
    const int tile_y = ntile - 1 - tile_rev;
    const float* __restrict__ in = second ? X : A;
    if (in == nullptr) return;                      // (one matrix only: gpfq_prepare_columns_ws_f32 with A or X NULL)
    const int64_t ld = second ? ldx : lda;
    float* __restrict__ out = second ? XT : AT;
    const int t0 = tile_y * TC;
    const int ncols = (int)((D - t0) < TC ? (D - t0) : TC);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    float acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = 0.0f;
    float4 v[NV];
    auto load_block = [&](int c) {
        const int64_t kb = (int64_t)s * kSeg + 256 * c;
        if constexpr (FLAT) {
            // 256 rows x D floats, contiguous: float4 number f covers elements 4 f .. 4 f + 3 of the run
            const int64_t valid = (m - kb < 256 ? (m - kb > 0 ? m - kb : 0) : 256) * D;    // elements of the run that exist
            const float* __restrict__ src = in + kb * D;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int64_t e0 = 4 * ((int64_t)tid + 256 * i);
                float4 r = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (e0 < 256 * D) {
                    if (e0 + 3 < valid) r = *reinterpret_cast<const float4*>(src + e0);
                    else {
                        if (e0 < valid) r.x = src[e0];
                        if (e0 + 1 < valid) r.y = src[e0 + 1];
                        if (e0 + 2 < valid) r.z = src[e0 + 2];
                    }
                }
                v[i] = r;
            }
        } else {
            const int c4 = (tid % LPR) * 4, r0 = tid / LPR;     // LPR lanes x 16 bytes = one row of the tile
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int64_t k = kb + r0 + (256 / LPR) * i;
                float4 r = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (k < m) {
                    const float* __restrict__ src = in + k * ld + t0 + c4;
                    if (VEC) {
                        if (c4 < ncols) r = *reinterpret_cast<const float4*>(src);
                    } else {
                        if (c4 < ncols) r.x = src[0];
                        if (c4 + 1 < ncols) r.y = src[1];
                        if (c4 + 2 < ncols) r.z = src[2];
                        if (c4 + 3 < ncols) r.w = src[3];
                    }
                }
                v[i] = r;
            }
        }
    };
    auto stage_block = [&]() {                      // registers -> tile[k][t]
        if constexpr (FLAT) {
            const int Di = (int)D;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int e0 = 4 * (tid + 256 * i);
                if (e0 < 256 * Di) {
                    int k = e0 / Di, t = e0 - k * Di;
                    const float el[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        tile[k * TS + t] = el[q];
                        if (++t == Di) { t = 0; ++k; }
                    }
                }
            }
        } else {
            const int c4 = (tid % LPR) * 4, r0 = tid / LPR;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                float* row = tile + (r0 + (256 / LPR) * i) * TS + c4;
                row[0] = v[i].x; row[1] = v[i].y; row[2] = v[i].z; row[3] = v[i].w;
            }
        }
    };
    load_block(0);
#pragma unroll 1
    for (int c = 0; c < 4; ++c) {
        if (c > 0) __syncthreads();                 // the previous block has been read out of the tile
        stage_block();
        __syncthreads();
        if (c < 3) load_block(c + 1);               // in flight while this block leaves
        prep_store_block<TC>(tile, out, t0, ncols, m_pad, (int64_t)s * kSeg + 256 * c, wave, lane, acc);
    }
    if (second) prep_store_sums<TC>(part, t0, ncols, S, s, wave, lane, acc);
}

// nrm2 pair of every column from its S segment sums (gpfq_transpose_norm_kernel's `part`): the canonical slot tree, the
// second half of gpfq_colnorm_kernel.  One wave per column.
__global__ void __launch_bounds__(64) gpfq_colnorm_finish_kernel(const float* __restrict__ part, int S, float* __restrict__ nrm2)
{
    const int lane = threadIdx.x;
    const int64_t col = blockIdx.x;
    const float* __restrict__ seg = part + col * S;
    const int P = pow2_ceil(S);
    const int per = P > 64 ? P / 64 : 1, nl = P > 64 ? 64 : P;
    const SlotMap smap = make_slot_map(S, P, 0, per, lane, nl);
    const float tot = combine_slots<true>(seg, smap, per, nl, S - 1);
    const float r = sqrtf(tot);
    if (lane == 0) {
        const float n2 = r * r;
        nrm2[2 * col] = n2;
        nrm2[2 * col + 1] = (n2 > 0.0f) ? 1.0f / n2 : 0.0f;
    }
}

// Activation capture for Conv2d layers, fused: the sampled kernel-sized patches of an NCHW feature map go
// straight into the transposed, zero-padded column layout the loop kernels read (row f = feature (c, i, j),
// channel-major; column k = sampled patch k).  Patches sit on a grid whose stride is the KERNEL SIZE, as the
// reference's nn.Unfold(kernel_size, dilation, padding, kernel_size) does (quantize_neural_net.py:320).
// Replaces unfold + transpose + reshape + index (quantize_neural_net.py:334-347) and the transposing pass.
// A thread owns one patch (lane = patch: the stores of one feature are 256 contiguous bytes per wave) and walks feature
// ROWS r = (c, i): the kw values (c, i, 0 .. kw-1) of a patch are contiguous in the image, so with KW == 3 / 5 / 7 and the
// row inside the image they are ONE 12-byte load (16 + 4, 16 + 12) instead of kw 4-byte ones -- the capture is bound by its scattered reads
// (a sampled patch uses kw * 4 bytes of every line it touches), not by its stores.  KW == 0: any kernel width, element by
// element.  blockDim = 256: 64 patches x 4 row lanes; grid = (m_pad / 64, ceil(C * kh / 16)): 16 feature rows per workgroup.
template <int KW>
__global__ void __launch_bounds__(256) gpfq_gather_patches_kernel(const float* __restrict__ x, int C, int H, int W,
                                                                  int kh, int kw, int ph, int pw, int dh, int dw,
                                                                  int Lw, int64_t L, const int64_t* __restrict__ patch,
                                                                  int64_t m, float* __restrict__ outT, int64_t m_pad, int D)
{
    typedef float f3 __attribute__((ext_vector_type(3), aligned(4)));
    typedef float f4 __attribute__((ext_vector_type(4), aligned(4)));
    const int kx = threadIdx.x & 63, ry = threadIdx.x >> 6;
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
    const int kwr = KW ? KW : kw;
    const int nrows = C * kh;
    const int tile_r = (int)gridDim.y - 1 - (int)blockIdx.y;           // last feature rows first (see gpfq_transpose_pad_kernel)
    const int r_end = min(nrows, (tile_r + 1) * 16);
    const bool row_inside = dw == 1 && x0 >= 0 && x0 + kwr <= W;       // the whole row of the patch lies in the image
    for (int r = tile_r * 16 + ry; r < r_end; r += 4) {
        const int c = r / kh, i = r - c * kh;
        const int yy = y0 + i * dh;
        const bool rowok = live && yy >= 0 && yy < H;
        const float* __restrict__ src = img + ((int64_t)c * H + yy) * W + x0;
        float* __restrict__ dst = outT + (int64_t)r * kwr * m_pad + k;
        if constexpr (KW == 3 || KW == 5 || KW == 7) {
            float v[KW];
#pragma unroll
            for (int j = 0; j < KW; ++j) v[j] = 0.0f;
            if (rowok) {
                if (row_inside) {                   // one or two loads for the whole row (4-byte aligned vector loads)
                    if constexpr (KW == 3) {
                        const f3 t = *reinterpret_cast<const f3*>(src);
                        v[0] = t.x; v[1] = t.y; v[2] = t.z;
                    } else {
                        const f4 t = *reinterpret_cast<const f4*>(src);
                        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
                        if constexpr (KW == 5) {
                            v[4] = src[4];
                        } else {
                            const f3 u = *reinterpret_cast<const f3*>(src + 4);
                            v[4] = u.x; v[5] = u.y; v[6] = u.z;
                        }
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < KW; ++j)
                        if (x0 + j * dw >= 0 && x0 + j * dw < W) v[j] = src[j * dw];
                }
            }
#pragma unroll
            for (int j = 0; j < KW; ++j) dst[(int64_t)j * m_pad] = v[j];
        } else {
            for (int j = 0; j < kwr; ++j) {
                const int xx = x0 + j * dw;
                float v = 0.0f;
                if (rowok && xx >= 0 && xx < W) v = src[j * dw];
                dst[(int64_t)j * m_pad] = v;
            }
        }
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

// The uniform draws of the stochastic quantizer as the LOOP kernels take them: out[i] = philox_uniform(seed, row_id0 + i,
// column) -- one draw per row of the layer at one column.  For the standalone StepAlgorithm._stochastic_msq, so that the
// name means one generator: quantizing the projections of column t with these draws is what the loop does at step t.
__global__ void gpfq_philox_uniform_kernel(uint64_t seed, uint64_t row_id0, uint64_t column, int64_t n, float* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = philox_uniform(seed, row_id0 + (uint64_t)i, column);
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
