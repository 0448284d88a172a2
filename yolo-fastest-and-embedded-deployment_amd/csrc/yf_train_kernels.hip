// yf_train_kernels.hip -- operators of the reference's TRAINING step (second slice): forward in train mode
// and backward of every layer type of YoloFastest (src/model_training/model/yolo_fastest.py:16-66), and the optimizer update of
// src/model_training/train.py:84 (Adam), on NCHW fp32 tensors like the reference's.
//
//   conv_norm_relu / conv_norm :16-38   Conv2d(bias=False) -> BatchNorm2d(train: batch statistics) -> [ReLU]
//   deconv_norm_relu           :42-48   ConvTranspose2d(k=2, s=2) -> BN -> ReLU
//   BasicResBlock              :52-66   three units + residual add
//   heads                      :136,146 Conv2d 1x1 with bias
//
// Pointwise / dense convs are GEMMs on the fp32 matrix pipe read straight from global memory; depthwise convs keep their weights
// wave-uniform; every reduction that spans workgroups (BatchNorm sums, weight gradients) stores partials into a scratch and a second
// pass adds them in a fixed order -- no float atomics, no in-kernel fences (both measured slow on this multi-XCD part; DESIGN.md
// section 4).  The generic one-thread-per-output kernels remain as the fallback for shapes the fast paths do not take (odd widths).
// They are NOT the tuned inference kernels (those fold BN, which training cannot: batch statistics).  Everything is stream-ordered;
// no allocation, no synchronisation.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <vector>

#include "yf_kernels.h"

namespace yf {

// ---- Conv2d forward: groups == 1 (dense / pointwise) or groups == C (depthwise); pad = (k - 1) / 2 ----
__global__ void __launch_bounds__(256) tconv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                        float* __restrict__ y, int N, int Cin, int H, int W, int Cout, int Ho, int Wo, int k,
                                                        int stride, int depthwise)
{
    const long idx = (long)blockIdx.x * 256 + threadIdx.x, total = (long)N * Cout * Ho * Wo;
    if (idx >= total) return;
    const int ox = (int)(idx % Wo), oy = (int)((idx / Wo) % Ho), co = (int)((idx / ((long)Wo * Ho)) % Cout), n = (int)(idx / ((long)Wo * Ho * Cout));
    const int pad = (k - 1) / 2;
    float s = bias ? bias[co] : 0.f;
    const int c0 = depthwise ? co : 0, c1 = depthwise ? co + 1 : Cin;
    for (int ci = c0; ci < c1; ++ci) {
        const float* xp = x + ((long)n * Cin + ci) * H * W;
        const float* wp = w + ((long)co * (depthwise ? 1 : Cin) + (depthwise ? 0 : ci)) * k * k;
        for (int ky = 0; ky < k; ++ky) {
            const int iy = oy * stride - pad + ky;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int ix = ox * stride - pad + kx;
                if (ix < 0 || ix >= W) continue;
                s = fmaf(xp[(long)iy * W + ix], wp[ky * k + kx], s);
            }
        }
    }
    y[idx] = s;
}

// ---- dense (groups == 1) convolution, CO_T output channels per thread: the input value is loaded once per CO_T outputs and the
// weights are wave-uniform (scalar loads).  Also the pointwise backward-data: out = ci, in = co, weight strides swapped. ----
template <int CO_T>
__global__ void __launch_bounds__(256) tconv_mc_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                       float* __restrict__ y, int N, int Cin, int H, int W, int Cout, int Ho, int Wo, int k,
                                                       int stride, long w_so, long w_si)
{
    const long q = (long)blockIdx.x * 256 + threadIdx.x, Q = (long)N * Ho * Wo;
    const int co0 = blockIdx.y * CO_T;
    if (q >= Q) return;
    const int ox = (int)(q % Wo), oy = (int)((q / Wo) % Ho), n = (int)(q / ((long)Wo * Ho));
    const int pad = (k - 1) / 2, kk = k * k;
    float acc[CO_T];
#pragma unroll
    for (int j = 0; j < CO_T; ++j) acc[j] = (bias && co0 + j < Cout) ? bias[co0 + j] : 0.f;
    const float* xn = x + (long)n * Cin * H * W;
    for (int ci = 0; ci < Cin; ++ci) {
        const float* xp = xn + (long)ci * H * W;
        for (int ky = 0; ky < k; ++ky) {
            const int iy = oy * stride - pad + ky;
            for (int kx = 0; kx < k; ++kx) {
                const int ix = ox * stride - pad + kx;
                const float xv = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? xp[(long)iy * W + ix] : 0.f;
                const float* wp = w + (long)co0 * w_so + (long)ci * w_si + ky * k + kx;
#pragma unroll
                for (int j = 0; j < CO_T; ++j)
                    if (co0 + j < Cout) acc[j] = fmaf(xv, wp[(long)j * w_so], acc[j]);
            }
        }
    }
    (void)kk;
    float* yp = y + ((long)n * Cout + co0) * Ho * Wo + (long)oy * Wo + ox;
#pragma unroll
    for (int j = 0; j < CO_T; ++j)
        if (co0 + j < Cout) yp[(long)j * Ho * Wo] = acc[j];
}

// ---- pointwise convolution as a GEMM on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32), operands straight from global memory:
//   Y[m][q] = sum_k A[m][k] X[k][q],  q = pixel over the batch (frame n = q / HW), X and Y NCHW.
// forward: A = weight [Cout][Cin] (sm = Cin, sk = 1); backward-data: A = weight^T (m = ci, k = co: sm = 1, sk = Cin), X = dY.
// One wave = 16 output channels x 64 pixels (four 16x16 tiles sharing the A fragment); a workgroup = 4 waves on 4 channel tiles.
// MFMA operand layout: A lane l = (row l % 16, k l / 16); B lane l = (k l / 16, col l % 16); D lane l = rows 4 (l / 16) + i, col l % 16.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__host__ __device__ inline int tpw_waves_m(int M) { return M > 32 ? 4 : M > 16 ? 2 : 1; }
template <int NT>   // NT 16-pixel tiles per wave: 4 for large maps, 1 when there are few pixels (more waves in flight)
__global__ void __launch_bounds__(256) tpw_mfma_kernel(const float* __restrict__ x, const float* __restrict__ a, const float* __restrict__ bias,
                                                       const float* __restrict__ addend, float* __restrict__ y, long Q, long HW, int M, int K,
                                                       long sm, long sk)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lk = lane >> 4;
    // 1-D grid in XCD-contiguous order with the channel tile fastest: the workgroups that read the same 16 NT pixels (all channel
    // tiles of that pixel tile) run on one XCD and share its L2
    // the 4 waves of a workgroup: wm of them along the channels, 4 / wm along the pixels (few channels: all four on pixels, no idle wave)
    const int wm = tpw_waves_m(M), wq = 4 / wm;
    const unsigned my = (unsigned)((M + 16 * wm - 1) / (16 * wm)), lb = (unsigned)xcd_tile(blockIdx.x, gridDim.x);
    const int m0 = (int)((lb % my) * wm + (wave % wm)) * 16;
    if (m0 >= M) return;
    const long q0 = ((long)(lb / my) * wq + wave / wm) * (16 * NT);
    // no predication inside the k loop: out-of-range pixels and rows read a valid (clamped) address and are not stored; the k tail
    // multiplies a clamped B element by an A element forced to zero
    const float* xp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        long q = q0 + t * 16 + lr;
        if (q > Q - 1) q = Q - 1;
        const long n = q / HW, i = q - n * HW;
        xp[t] = x + n * K * HW + i;                     // + k * HW
    }
    const int mr = m0 + lr < M ? m0 + lr : M - 1;
    const float* ap = a + (long)mr * sm;
    f32x4_t acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int K4 = K & ~3;
    for (int k0 = 0; k0 < K4; k0 += 4) {      // (four k-steps per trip with all loads up front was measured slower: 51 vs 44 us at batch 256)
        const int k = k0 + lk;
        const float av = ap[(long)k * sk];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xp[t][(long)k * HW], acc[t], 0, 0, 0);
    }
    if (K4 < K) {
        const int k = K4 + lk, kc = k < K ? k : K - 1;
        const float av = k < K ? ap[(long)kc * sk] : 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xp[t][(long)kc * HW], acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const long q = q0 + t * 16 + lr;
        if (q >= Q) continue;
        const long n = q / HW, i = q - n * HW;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + lk * 4 + r;
            if (m < M) {
                const long o = (n * M + m) * HW + i;
                const float v = acc[t][r] + (bias ? bias[m] : 0.f);
                y[o] = addend ? v + addend[o] : v;                  // addend: the skip gradient of a residual block (saves an add pass)
            }
        }
    }
}

// ---- BatchNorm statistics out of the conv's epilogue (the large maps: one pass over z less).  A workgroup leaves one (sum, sum of
// squares) pair per output channel and pixel block in `stat` ([channel][block], float2 of values summed in double over the block);
// tbn_stats_from_parts_kernel adds a channel's pairs in double, in block order.  Deterministic: fixed rotation / wave order.
__device__ __forceinline__ float tbn_affine(float x, float mean, float invstd, float gamma, float beta);   // below, with BatchNorm
// what a data-gradient kernel needs of the layer below to leave that layer's backward BatchNorm sums (yf_kernels.h: TBnRed)
struct TRedArgs { const float* z; const float* stats; const float* gamma; const float* beta; float2* part; int relu; };
template <int N_> __device__ __forceinline__ float row16_rotate(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N_, 0xf, 0xf, false));   // row_ror:N_
}
__device__ __forceinline__ float row16_sum(float v)         // the sum over the 16 lanes of a DPP row (= the lanes lr of one lk), in every lane
{
    v += row16_rotate<8>(v);
    v += row16_rotate<4>(v);
    v += row16_rotate<2>(v);
    v += row16_rotate<1>(v);
    return v;
}
// rows 16 t + 4 lk + r of a wave's tile: acc[t][e][r] = pixel e of row r; lanes with ok == false hold nothing.  A pair covers the 64
// pixels of the WAVE (no workgroup barrier in a kernel that lives on its waves not waiting for each other): summed in fp32 by a fixed
// tree -- 4 in the lane, the 16 lanes of the row by DPP rotations -- the pairs themselves are then added in double.
template <int MT>
__device__ __forceinline__ void tile_stats_store(const f32x4_t (&acc)[MT][4], bool ok, int m0, int M, float2* __restrict__ stat, long nwaves, long wave_index)
{
    const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s1 = 0.f, s2 = 0.f;
            if (ok) {
                const float a = acc[t][0][r], b = acc[t][1][r], c = acc[t][2][r], d = acc[t][3][r];
                s1 = (a + b) + (c + d);
                s2 = (a * a + b * b) + (c * c + d * d);
            }
            s1 = row16_sum(s1);
            s2 = row16_sum(s2);
            const int m = m0 + 16 * t + 4 * lk + r;
            if (lr == 0 && m < M) stat[(long)m * nwaves + wave_index] = make_float2(s1, s2);
        }
}

// The same GEMM for maps with many pixels, built for bandwidth: every lane loads float4 = 4 consecutive pixels of ONE k-row (a wave's
// load instruction covers 4 rows x 256 B), and MFMA e of a k-step takes element e -- column lr of accumulator e is pixel 4 lr + e, so
// the lane ends up with 4 consecutive pixels of each of its 4 output rows and stores float4 too.  One wave = MT 16-channel tiles x 64
// pixels (the B fragments are loaded once for all MT tiles); the 4 waves of a workgroup sit on 4 consecutive pixel tiles; m-groups of
// one pixel block are neighbours in the XCD-contiguous order.  Needs HW % 4 == 0 and K % 4 == 0 (every layer of this network).
// Two k-steps per trip with the loads up front.
// DECONV: the ConvTranspose2d(2, 2) forward is this GEMM with M = (co, a, b) rows (A = the weight [Cin][Cout 2 2] read by columns) and a
// scattering epilogue: the lane's 4 rows are the 2x2 output block of ONE channel, for each of its 4 input pixels (Wd = input width).
// (bid, nblocks: the workgroup's place in its grid -- blockIdx.x / gridDim.x, or a sub-range of a launch shared with another kernel body)
template <int MT, bool DECONV>
__device__ __forceinline__ void tpw4_body(const float* __restrict__ x, const float* __restrict__ a, const float* __restrict__ bias,
                                          const float* __restrict__ addend, float* __restrict__ y, long Q, long HW, int M, int K, long sm, long sk,
                                          int mgroups, int Wd, float2* __restrict__ stat, unsigned bid, unsigned nblocks)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lk = lane >> 4;
    const unsigned lb = (unsigned)xcd_tile(bid, nblocks);
    const int m0 = (int)(lb % (unsigned)mgroups) * (16 * MT);
    const long q0 = ((long)(lb / (unsigned)mgroups) * 4 + wave) * 64;
    if (q0 >= Q && !stat) return;                           // (with statistics a wave past the end still leaves its (zero) pairs)
    long q = q0 + 4 * lr;
    const bool qv = q < Q;
    if (!qv) q = Q - 4;
    const long n = q / HW, i = q - n * HW;
    const float* xp = x + (n * K + lk) * HW + i;          // + k0 * HW
    const float* ap[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = m0 + 16 * t + lr;
        ap[t] = a + (long)(m < M ? m : M - 1) * sm + (long)lk * sk;
    }
    f32x4_t acc[MT][4];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t][e] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    auto step = [&](const float4& b, const float (&av)[MT]) {
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], b.x, acc[t][0], 0, 0, 0);
            acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], b.y, acc[t][1], 0, 0, 0);
            acc[t][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], b.z, acc[t][2], 0, 0, 0);
            acc[t][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], b.w, acc[t][3], 0, 0, 0);
        }
    };
    // UK k-steps per trip, all of a trip's loads requested before its first MFMA: a wave's time is (trips) x (one memory latency), and
    // the small maps have too few waves per CU to hide it any other way (K = 224: 7 trips instead of 56 dependent ones)
    constexpr int UK = MT <= 2 ? 8 : 6;
    for (int k0 = 0; k0 < K; k0 += 4 * UK) {
        float4 b[UK];
        float av[UK][MT];
#pragma unroll
        for (int j = 0; j < UK; ++j) {
            if (k0 + 4 * j >= K) break;                                  // wave-uniform
            b[j] = *reinterpret_cast<const float4*>(xp + (long)(k0 + 4 * j) * HW);
#pragma unroll
            for (int t = 0; t < MT; ++t) av[j][t] = ap[t][(long)(k0 + 4 * j) * sk];
        }
#pragma unroll
        for (int j = 0; j < UK; ++j) {
            if (k0 + 4 * j >= K) break;
            step(b[j], av[j]);
        }
    }
    if constexpr (!DECONV && MT <= 2) {
        if (stat) tile_stats_store<MT>(acc, qv, m0, M, stat, (Q + 255) / 256 * 4, (long)(lb / (unsigned)mgroups) * 4 + wave);
    }
    if (!qv) return;
    if constexpr (DECONV) {
        const int Cout = M / 4;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int m = m0 + 16 * t + 4 * lk;                     // rows m .. m + 3 = channel m / 4, (a, b) = 0 .. 3
            if (m >= M) continue;
            float* yc = y + (n * Cout + m / 4) * 4 * HW;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int iy = (int)((i + e) / Wd), ix = (int)(i + e - (long)iy * Wd);
                float* o = yc + (long)(2 * iy) * (2 * Wd) + 2 * ix;
                *reinterpret_cast<float2*>(o) = make_float2(acc[t][e][0], acc[t][e][1]);
                *reinterpret_cast<float2*>(o + 2 * Wd) = make_float2(acc[t][e][2], acc[t][e][3]);
            }
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 16 * t + 4 * lk + r;
            if (m >= M) continue;
            const long o = (n * M + m) * HW + i;
            const float bv = bias ? bias[m] : 0.f;
            float4 v = make_float4(acc[t][0][r] + bv, acc[t][1][r] + bv, acc[t][2][r] + bv, acc[t][3][r] + bv);
            if (addend) {                                           // the skip gradient of a residual block (saves an add pass)
                const float4 ad = *reinterpret_cast<const float4*>(addend + o);
                v.x += ad.x; v.y += ad.y; v.z += ad.z; v.w += ad.w;
            }
            *reinterpret_cast<float4*>(y + o) = v;
        }
}

template <int MT, bool DECONV = false>
__global__ void __launch_bounds__(256) tpw4_mfma_kernel(const float* __restrict__ x, const float* __restrict__ a, const float* __restrict__ bias,
                                                        const float* __restrict__ addend, float* __restrict__ y, long Q, long HW, int M, int K,
                                                        long sm, long sk, int mgroups, int Wd = 0, float2* __restrict__ stat = nullptr)
{
    tpw4_body<MT, DECONV>(x, a, bias, addend, y, Q, HW, M, K, sm, sk, mgroups, Wd, stat, blockIdx.x, gridDim.x);
}

// The same GEMM with the weights stationary: where the A operand is big (conv4_1_1: 232 x 96 = 89 KB) every wave of tpw4_mfma_kernel pulls
// its MT x K slice of it through L2 -> L1 again -- 2560 waves x 44 KB at batch 256, more than the activations it multiplies, and the
// fill path is what the kernel then waits for (the inference engine's pointwise GEMMs hit the same wall in round 1).  Here a workgroup
// stages its 16 MT rows of A once, as a_lds[k][row] (row stride RS = 16, 48, 48, 80 floats: the four k-rows of a fragment read fall on
// disjoint banks), and walks pixel blocks with it (persistent grid); the A fragments of a trip are LDS reads.
template <int MT> __host__ __device__ constexpr int tpw4_rs() { return MT == 1 ? 16 : MT == 4 ? 80 : 48; }
template <int MT, bool DECONV = false>
__global__ void __launch_bounds__(256) tpw4_lds_kernel(const float* __restrict__ x, const float* __restrict__ a, const float* __restrict__ bias,
                                                       const float* __restrict__ addend, float* __restrict__ y, long Q, long HW, int M, int K,
                                                       long sm, long sk, int mgroups, int Wd = 0)
{
    extern __shared__ __attribute__((aligned(16))) float a_lds[];
    constexpr int RS = tpw4_rs<MT>(), ROWS = 16 * MT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lk = lane >> 4;
    const unsigned lb = (unsigned)xcd_tile(blockIdx.x, gridDim.x);
    const int m0 = (int)(lb % (unsigned)mgroups) * ROWS;
    {   // stage A: 8 loads in flight per thread and trip (a rolled copy loop waits for every load before the next: isa_serial_loads.py)
        const int total = ROWS * K;
        for (int base = 0; base < total; base += 256 * 8) {
            float v[8];
            int dst[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * 256 + (int)threadIdx.x;
                int r, k;
                if (sk == 1) { k = idx % K; r = idx / K; } else { r = idx % ROWS; k = idx / ROWS; }     // along the contiguous side of A
                const bool ok = idx < total && m0 + r < M;
                v[u] = ok ? a[(long)(m0 + r) * sm + (long)k * sk] : 0.f;
                dst[u] = idx < total ? k * RS + r : -1;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (dst[u] >= 0) a_lds[dst[u]] = v[u];
        }
    }
    __syncthreads();
    const float* al = a_lds + lk * RS + lr;                 // + k0 RS + 16 t
    const long npb = (Q + 255) / 256, pstep = gridDim.x / (unsigned)mgroups;
    constexpr int UK = MT <= 2 ? 8 : 6;
    for (long pb = lb / (unsigned)mgroups; pb < npb; pb += pstep) {
        const long q0 = (pb * 4 + wave) * 64;
        if (q0 >= Q) continue;
        long q = q0 + 4 * lr;
        const bool qv = q < Q;
        if (!qv) q = Q - 4;
        const long n = q / HW, i = q - n * HW;
        const float* xp = x + (n * K + lk) * HW + i;
        f32x4_t acc[MT][4];
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[t][e] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < K; k0 += 4 * UK) {
            float4 b[UK];
#pragma unroll
            for (int j = 0; j < UK; ++j) {
                if (k0 + 4 * j >= K) break;
                b[j] = *reinterpret_cast<const float4*>(xp + (long)(k0 + 4 * j) * HW);
            }
#pragma unroll
            for (int j = 0; j < UK; ++j) {
                if (k0 + 4 * j >= K) break;
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    const float av = al[(k0 + 4 * j) * RS + 16 * t];
                    acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[j].x, acc[t][0], 0, 0, 0);
                    acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[j].y, acc[t][1], 0, 0, 0);
                    acc[t][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[j].z, acc[t][2], 0, 0, 0);
                    acc[t][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[j].w, acc[t][3], 0, 0, 0);
                }
            }
        }
        if (!qv) continue;
        if constexpr (DECONV) {
            const int Cout = M / 4;
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const int m = m0 + 16 * t + 4 * lk;
                if (m >= M) continue;
                float* yc = y + (n * Cout + m / 4) * 4 * HW;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int iy = (int)((i + e) / Wd), ix = (int)(i + e - (long)iy * Wd);
                    float* o = yc + (long)(2 * iy) * (2 * Wd) + 2 * ix;
                    *reinterpret_cast<float2*>(o) = make_float2(acc[t][e][0], acc[t][e][1]);
                    *reinterpret_cast<float2*>(o + 2 * Wd) = make_float2(acc[t][e][2], acc[t][e][3]);
                }
            }
            continue;
        }
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 16 * t + 4 * lk + r;
                if (m >= M) continue;
                const long o = (n * M + m) * HW + i;
                const float bv = bias ? bias[m] : 0.f;
                float4 v = make_float4(acc[t][0][r] + bv, acc[t][1][r] + bv, acc[t][2][r] + bv, acc[t][3][r] + bv);
                if (addend) {
                    const float4 ad = *reinterpret_cast<const float4*>(addend + o);
                    v.x += ad.x; v.y += ad.y; v.z += ad.z; v.w += ad.w;
                }
                *reinterpret_cast<float4*>(y + o) = v;
            }
    }
}
template <int MT, bool DECONV>
static int launch_tpw4_lds(const float* x, const float* a, const float* bias, const float* addend, float* y, long Q, long HW, int M, int K, long sm,
                           long sk, int mgroups, int Wd, hipStream_t s)
{
    static bool attr_done[YF_MAX_DEVICES] = {};
    const int dev = current_device(), n_cu = device_cu_count(dev);
    if (dev < 0 || n_cu <= 0) return -1;
    const size_t lds = (size_t)tpw4_rs<MT>() * K * sizeof(float);
    if (lds > 96 * 1024) return -1;
    if (!attr_done[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(tpw4_lds_kernel<MT, DECONV>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess)
            return -1;
        attr_done[dev] = true;
    }
    const long npb = (Q + 255) / 256;
    long per = (2L * n_cu + mgroups - 1) / mgroups;                     // ~2 workgroups per CU in all
    if (per > npb) per = npb;
    hipLaunchKernelGGL((tpw4_lds_kernel<MT, DECONV>), dim3((unsigned)(per * mgroups)), dim3(256), lds, s, x, a, bias, addend, y, Q, HW, M, K, sm, sk,
                       mgroups, Wd);
    return 0;
}

// backward-data of the ConvTranspose2d(2, 2): dx[ci][p] = sum over k = (co, a, b) of w[ci][k] dY[co][2 iy + a][2 ix + b] -- the pointwise
// GEMM again, lane (lk, lr) = tap (a, b) = lk of pixel lr, so a k-step is one channel of dY and the lane's operand address only
// advances by a plane.  One wave = MT 16-channel tiles x 4 pixel tiles of 16 (the old gather kernel recomputed indices per element).
template <int MT>
__global__ void __launch_bounds__(256) tdeconv_bwd_mfma_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx,
                                                               long Q, int H, int W, int Cin, int Cout, int mgroups)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lk = lane >> 4;
    const unsigned lb = (unsigned)xcd_tile(blockIdx.x, gridDim.x);
    const int m0 = (int)(lb % (unsigned)mgroups) * (16 * MT);
    const long q0 = ((long)(lb / (unsigned)mgroups) * 4 + wave) * 64, HW = (long)H * W;
    if (q0 >= Q) return;
    const float* bp[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        long q = q0 + 16 * t + lr;
        if (q > Q - 1) q = Q - 1;
        const long n = q / HW, i = q - n * HW;
        const int iy = (int)(i / W), ix = (int)(i - (long)iy * W);
        bp[t] = dy + n * Cout * 4 * HW + (long)(2 * iy + (lk >> 1)) * (2 * W) + 2 * ix + (lk & 1);     // + co * 4 HW
    }
    const float* ap[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = m0 + 16 * t + lr;
        ap[t] = w + (long)(m < Cin ? m : Cin - 1) * 4 * Cout + lk;                                  // + co * 4
    }
    f32x4_t acc[MT][4];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[t][u] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int co0 = 0; co0 < Cout; co0 += 4) {                // 4 channels of dY per trip, all of the trip's loads requested before its first MFMA
        float b[4][4], av[4][MT];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (co0 + j >= Cout) break;                          // wave-uniform
#pragma unroll
            for (int u = 0; u < 4; ++u) b[j][u] = bp[u][(long)(co0 + j) * 4 * HW];
#pragma unroll
            for (int t = 0; t < MT; ++t) av[j][t] = ap[t][(co0 + j) * 4];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (co0 + j >= Cout) break;
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][t], b[j][u], acc[t][u], 0, 0, 0);
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long q = q0 + 16 * u + lr;
        if (q >= Q) continue;
        const long n = q / HW, i = q - n * HW;
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 16 * t + 4 * lk + r;
                if (m < Cin) dx[(n * Cin + m) * HW + i] = acc[t][u][r];
            }
    }
}

// dense conv forward for k > 1 (conv0, conv1_9) on the matrix pipe: the same GEMM with the B operand gathered (im2col on the fly):
// k-index r = (ci, ky, kx); A = weight [Cout][Cin k k] as stored.  One wave = 16 output channels x 64 output pixels.
template <int KS>
__global__ void __launch_bounds__(256) tconv_im2col_mfma_kernel(const float* __restrict__ x, const float* __restrict__ a, const float* __restrict__ bias,
                                                                float* __restrict__ y, int N, int Cin, int H, int W, int Ho, int Wo, int M, int stride)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lk = lane >> 4;
    const int m0 = (blockIdx.y * 4 + wave) * 16;
    if (m0 >= M) return;
    const long Q = (long)N * Ho * Wo, q0 = (long)blockIdx.x * 64;
    constexpr int KK = KS * KS, PAD = (KS - 1) / 2;
    const int K = Cin * KK;
    const float* xn[4];
    int iy0[4], ix0[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        long q = q0 + t * 16 + lr;
        if (q > Q - 1) q = Q - 1;
        const int ox = (int)(q % Wo), oy = (int)((q / Wo) % Ho), n = (int)(q / ((long)Wo * Ho));
        xn[t] = x + (long)n * Cin * H * W;
        iy0[t] = oy * stride - PAD;
        ix0[t] = ox * stride - PAD;
    }
    const int mr = m0 + lr < M ? m0 + lr : M - 1;
    const float* ap = a + (long)mr * K;
    f32x4_t acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 4) {
        const int r = k0 + lk;
        const bool kv = r < K;
        const int rc = kv ? r : K - 1;
        const int ci = rc / KK, tap = rc - ci * KK, ky = tap / KS, kx = tap - ky * KS;
        const float av = kv ? ap[rc] : 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int iy = iy0[t] + ky, ix = ix0[t] + kx;
            const float bv = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? xn[t][((long)ci * H + iy) * W + ix] : 0.f;
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[t], 0, 0, 0);
        }
    }
    const long HWo = (long)Ho * Wo;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const long q = q0 + t * 16 + lr;
        if (q >= Q) continue;
        const long n = q / HWo, i = q - n * HWo;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + lk * 4 + r;
            if (m < M) y[(n * M + m) * HWo + i] = acc[t][r] + (bias ? bias[m] : 0.f);
        }
    }
}

// dense 3x3 stride-2 pad-1 convolution (conv1_9: 24 -> 24 on the 128x160 map, the most expensive layer of the iteration) on the matrix
// pipe without a gather: k-index = (tap, 4 input channels), lane (lk, lr) = input channel ci0 + lk and a GROUP of 4 consecutive output
// pixels; per (ci0, ky) the lane loads the 9 input columns 8 ox4 - 1 .. 8 ox4 + 7 of its row as two aligned float4 and one scalar, and
// MFMA (kx, e) takes column 2 e + kx - 1 -- column lr of accumulator e is output pixel 4 lr + e, stored as float4 (cf. tpw4_mfma_kernel).
// Needs Cin % 4 == 0, H even, W % 8 == 0.  One wave = MT 16-channel tiles x 64 output pixels.
template <int MT>
__global__ void __launch_bounds__(256) tconv3s2_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                            float* __restrict__ y, int N, int Cin, int H, int W, int M)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lk = lane >> 4;
    const int Ho = H / 2, Wo = W / 2, per_row = Wo / 4;
    const long G = (long)N * Ho * per_row;                              // groups of 4 output pixels
    const long g0 = ((long)blockIdx.x * 4 + wave) * 16;
    if (g0 >= G) return;
    long g = g0 + lr;
    const bool gv = g < G;
    if (!gv) g = G - 1;
    const int ox4 = (int)(g % per_row), oy = (int)((g / per_row) % Ho), n = (int)(g / ((long)per_row * Ho));
    const float* xp = x + ((long)n * Cin + lk) * H * W + 8 * ox4;          // + ci0 H W + iy W
    const float* wp[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = 16 * t + lr;
        wp[t] = w + ((long)(m < M ? m : M - 1) * Cin + lk) * 9;          // + ci0 * 9 + ky * 3 + kx
    }
    f32x4_t acc[MT][4];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t][e] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int ci0 = 0; ci0 < Cin; ci0 += 4) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = 2 * oy + ky - 1;
            const bool ok = iy >= 0;
            const float* xr = xp + ((long)ci0 * H + (ok ? iy : 0)) * W;
            float4 lo = *reinterpret_cast<const float4*>(xr), hi = *reinterpret_cast<const float4*>(xr + 4);
            float m1 = xr[ox4 > 0 ? -1 : 0];
            if (!ok) { lo = make_float4(0.f, 0.f, 0.f, 0.f); hi = lo; }
            if (!ok || ox4 == 0) m1 = 0.f;
            const float v[9] = {m1, lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};     // v[c + 1] = column 8 ox4 + c
            float av[MT][3];
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) av[t][kx] = wp[t][ci0 * 9 + ky * 3 + kx];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int t = 0; t < MT; ++t) acc[t][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t][kx], v[2 * e + kx], acc[t][e], 0, 0, 0);
        }
    }
    // (requesting the next step's operands before this step's MFMAs was measured slower: 259 -> 327 us for conv1_9 at batch 256)
    if (!gv) return;
    const long HWo = (long)Ho * Wo;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = 16 * t + 4 * lk + r;
            if (m >= M) continue;
            const float bv = bias ? bias[m] : 0.f;
            *reinterpret_cast<float4*>(y + ((long)n * M + m) * HWo + (long)oy * Wo + 4 * ox4) =
                make_float4(acc[t][0][r] + bv, acc[t][1][r] + bv, acc[t][2][r] + bv, acc[t][3][r] + bv);
        }
}

// conv0 (1 -> 8, 3x3 stride 2): nothing to multiply, 190 MB to move.  A thread = 4 consecutive output pixels x all CO channels from the
// 3 x 9 input window (two aligned float4 + one scalar per row), CO float4 stores.
template <int CO>
__global__ void __launch_bounds__(256) tconv3s2_c1_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                          float* __restrict__ y, int N, int H, int W, int M)
{
    const int Ho = H / 2, Wo = W / 2, per_row = Wo / 4;
    const long G = (long)N * Ho * per_row, g = (long)blockIdx.x * 256 + threadIdx.x;
    if (g >= G) return;
    const int ox4 = (int)(g % per_row), oy = (int)((g / per_row) % Ho), n = (int)(g / ((long)per_row * Ho));
    float acc[CO][4];
#pragma unroll
    for (int c = 0; c < CO; ++c) { const float bv = (bias && c < M) ? bias[c] : 0.f; acc[c][0] = acc[c][1] = acc[c][2] = acc[c][3] = bv; }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = 2 * oy + ky - 1;
        if (iy < 0) continue;
        const float* xr = x + ((long)n * H + iy) * W + 8 * ox4;
        const float4 lo = *reinterpret_cast<const float4*>(xr), hi = *reinterpret_cast<const float4*>(xr + 4);
        const float m1 = ox4 > 0 ? xr[-1] : 0.f;
        const float v[9] = {m1, lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int c = 0; c < CO; ++c)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float wv = w[(c < M ? c : M - 1) * 9 + ky * 3 + kx];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[c][e] = fmaf(v[2 * e + kx], wv, acc[c][e]);
            }
    }
    const long HWo = (long)Ho * Wo;
#pragma unroll
    for (int c = 0; c < CO; ++c)
        if (c < M) *reinterpret_cast<float4*>(y + ((long)n * M + c) * HWo + (long)oy * Wo + 4 * ox4) = make_float4(acc[c][0], acc[c][1], acc[c][2], acc[c][3]);
}

// dense conv weight gradient on the matrix pipe: dW[co][r] = sum_q dY[co][q] Xcol[r][q], r = (ci, ky, kx), split over q.  One wave =
// 16 co x 64 r and a slice of q; per step (16 output pixels) every lane loads float4 (4 consecutive pixels) of one dY row and the
// four matching elements of four Xcol rows: MFMA e of the step takes element e, i.e. k-index l / 16 stands for pixel 4 (l / 16) + e
// in both operands.  Needs Ho Wo % 4 == 0 (the 4 pixels lie in one frame).  KS == 1: the Xcol elements are one float4 as well.
// KS == 2 (pad 0, stride 2) with the operands swapped is the ConvTranspose2d(2, 2) weight gradient: dW[ci][(co, a, b)] = sum_p
// X[ci][p] dY[co][2 iy + a][2 ix + b].
// Slice s writes its tile into dw + s * part_stride (a slab of the scratch; tsum_partials_kernel adds the slabs in order): device-scope
// float atomics on this multi-XCD part are executed memory-side and serialise per address -- 100 slices on one tile cost more than the GEMM.
// NW waves per workgroup share a slice (32-pixel trips dealt round-robin) and add their tiles through LDS in wave order: the layers with
// the most pixels have ONE tile, and 1024 single-wave workgroups (the slab limit) leave a CU with 4 waves = 8 KB of loads in flight.
template <int KS, int NW>
__device__ __forceinline__ void twgrad_body(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw, int N, int Cin, int H, int W,
                                            int Cout, int Ho, int Wo, int stride, long q_per, long part_stride, unsigned bid, unsigned nblocks)
{
    const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
    const int wv = NW > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
    constexpr int KK = KS * KS, PAD = (KS - 1) / 2;
    const int R = Cin * KK;
    // 1-D grid in XCD-contiguous order, tiles fastest: the waves of one pixel slice (they all read the same rows of dY and X) share an L2
    const unsigned nct = (unsigned)((Cout + 15) / 16), nrt = (unsigned)((R + 63) / 64), lb = (unsigned)xcd_tile(bid, nblocks);
    const unsigned slice = lb / (nct * nrt), tile = lb - slice * (nct * nrt);
    const int c0 = (int)(tile % nct) * 16, r0 = (int)(tile / nct) * 64;
    const long HWo = (long)Ho * Wo, Q = (long)N * HWo;
    const long qb = (long)slice * q_per, qe = qb + q_per < Q ? qb + q_per : Q;
    const int cr = c0 + lr < Cout ? c0 + lr : Cout - 1;
    const bool cv = c0 + lr < Cout;
    int rci[4], rky[4], rkx[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        int r = r0 + 16 * t + lr;
        if (r > R - 1) r = R - 1;                     // clamped rows produce columns that are never stored
        rci[t] = r / KK;
        const int tap = r - rci[t] * KK;
        rky[t] = tap / KS;
        rkx[t] = tap - rky[t] * KS;
    }
    f32x4_t acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int ntu = (R - r0 + 15) / 16;            // 16-row tiles of Xcol this wave really has (wave-uniform): the layers with the most pixels
                                                   // have the fewest channels, and loading / multiplying three tiles of clamped rows is what they cost
    auto load = [&](long q0, float4& av, float4* bv) {
        long q = q0 + 4 * lk;
        const bool qv = q < qe;                                       // q_per, Q multiples of 4
        if (!qv) q = qb;
        const long n = q / HWo, i = q - n * HWo;
        av = *reinterpret_cast<const float4*>(dy + (n * Cout + cr) * HWo + i);
        if (!(qv && cv)) av = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (KS == 1) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (t < ntu) bv[t] = *reinterpret_cast<const float4*>(x + (n * Cin + rci[t]) * HWo + i);
        } else {
            int oyj[4], oxj[4];                                     // the 4 pixels may straddle rows (Wo % 4 != 0)
#pragma unroll
            for (int j = 0; j < 4; ++j) { oyj[j] = (int)((i + j) / Wo); oxj[j] = (int)(i + j - (long)oyj[j] * Wo); }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t >= ntu) continue;
                const float* xc = x + (n * Cin + rci[t]) * H * W;
                float e[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int iy = oyj[j] * stride - PAD + rky[t], ix = oxj[j] * stride - PAD + rkx[t];
                    e[j] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? xc[(long)iy * W + ix] : 0.f;
                }
                bv[t] = make_float4(e[0], e[1], e[2], e[3]);
            }
        }
    };
    auto mac = [&](const float4& av, const float4* bv) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (t < ntu) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(((const float*)&av)[e], ((const float*)&bv[t])[e], acc[t], 0, 0, 0);
    };
    long q0 = qb + 32L * wv;
    for (; q0 + 16 < qe; q0 += 32L * NW) {                            // two steps per trip: ten float4 loads in flight per lane
        float4 a0, a1, b0[4], b1[4];
        load(q0, a0, b0);
        load(q0 + 16, a1, b1);
        mac(a0, b0);
        mac(a1, b1);
    }
    if (q0 < qe) {
        float4 a0, b0[4];
        load(q0, a0, b0);
        mac(a0, b0);
    }
    if constexpr (NW > 1) {
        __shared__ f32x4_t red[NW - 1][4][64];
        if (wv > 0) {
#pragma unroll
            for (int t = 0; t < 4; ++t) red[wv - 1][t][lane] = acc[t];
        }
        __syncthreads();
        if (wv > 0) return;
#pragma unroll
        for (int w = 0; w < NW - 1; ++w)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] += red[w][t][lane];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = c0 + lk * 4 + r, rr = r0 + 16 * t + lr;
            if (co < Cout && rr < R) dw[(long)slice * part_stride + (long)co * R + rr] = acc[t][r];
        }
}

template <int KS, int NW>
__global__ void __launch_bounds__(64 * NW) tconv_wgrad_mfma_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw,
                                                                   int N, int Cin, int H, int W, int Cout, int Ho, int Wo, int stride, long q_per,
                                                                   long part_stride)
{
    twgrad_body<KS, NW>(x, dy, dw, N, Cin, H, W, Cout, Ho, Wo, stride, q_per, part_stride, blockIdx.x, gridDim.x);
}

// Backward-data and weight gradient of a pointwise layer as ONE launch: workgroups [0, nA) run the data-gradient GEMM (tpw4_body), the
// rest the weight-gradient GEMM (twgrad_body, 4 waves per slice).  They only share their input dz; at the reference's batch 16 every
// kernel of the backward is a few microseconds of mostly waiting, and two that can run side by side cost one launch and one drain.
struct TPwBwdArgs {
    const float* dz; const float* w; const float* addend; float* dx;          // data gradient: dx[ci] = sum_co dz[co] w[co][ci] (+ addend)
    const float* x; float* dw;                                                // weight gradient: dW[co][ci] = sum_q dz[co][q] x[ci][q]
    long Q, HW; int N, Cin, Cout, H, W; int mgroups; unsigned nA; long q_per, part_stride;
};
template <int MT>
__global__ void __launch_bounds__(256) tpw_bwd_dual_kernel(TPwBwdArgs a)
{
    if (blockIdx.x < a.nA)
        tpw4_body<MT, false>(a.dz, a.w, nullptr, a.addend, a.dx, a.Q, a.HW, a.Cin, a.Cout, 1L, (long)a.Cin, a.mgroups, 0, nullptr, blockIdx.x, a.nA);
    else
        twgrad_body<1, 4>(a.x, a.dz, a.dw, a.N, a.Cin, a.H, a.W, a.Cout, a.H, a.W, 1, a.q_per, a.part_stride, blockIdx.x - a.nA, gridDim.x - a.nA);
}

// Weight gradient of the dense 3x3 stride-2 pad-1 convolution (conv1_9) as NINE GEMMs that share their operands, no gather:
//   dW[co][ci][ky][kx] = sum over pixels of dY[co][p] X[ci][2 oy + ky - 1][2 ox + kx - 1]:   M = co, N = ci, K = pixels, one accumulator per tap.
// Lane (lk, lr) of the A operand = channel co = lr, pixel GROUP lk (4 consecutive output pixels: one float4 of dY); of the B operand =
// channel ci = lr, the same group: per tap row ky the 9 input columns 8 ox4 - 1 .. 8 ox4 + 7 (two aligned float4 + one scalar), of which
// MFMA (ky, kx, e) takes column 2 e + kx - 1 against element e of dY (cf. tconv3s2_mfma_kernel).  A workgroup = 4 waves = the (co tile,
// ci tile) pairs of a slice of the pixel groups (they read the same operands: L1 serves the second reader); a step = 4 groups = 36
// MFMAs per wave for 10 loads, the next step's operands requested before the current step's MFMAs.  Cout, Cin <= 32; H even, W % 8 == 0.
// Slice s writes its tiles into dw + s * part_stride.
__global__ void __launch_bounds__(256) tconv3s2_wgrad_mfma_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw,
                                                                  int N, int Cin, int H, int W, int Cout, long g_per, long part_stride)
{
    const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), c0 = (wv & 1) * 16, i0 = (wv >> 1) * 16;
    if (c0 >= Cout || i0 >= Cin) return;
    const int Ho = H / 2, Wo = W / 2, per_row = Wo / 4;
    const long G = (long)N * Ho * per_row, gb = (long)blockIdx.x * g_per, ge = gb + g_per < G ? gb + g_per : G;
    const int co = c0 + lr, ci = i0 + lr;
    const bool cov = co < Cout, civ = ci < Cin;
    const float* dyc = dy + (long)(cov ? co : Cout - 1) * Ho * Wo;
    const float* xc = x + (long)(civ ? ci : Cin - 1) * H * W;
    struct Frag { float4 a, lo[3], hi[3]; float m1[3]; };
    auto load = [&](long g0, Frag& f) {
        long g = g0 + lk;
        const bool gv = g < ge;
        if (!gv) g = gb;
        const int ox4 = (int)(g % per_row), oy = (int)((g / per_row) % Ho);
        const long n = g / ((long)per_row * Ho);
        f.a = *reinterpret_cast<const float4*>(dyc + n * Cout * Ho * Wo + (long)oy * Wo + 4 * ox4);
        if (!(gv && cov)) f.a = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* xn = xc + n * Cin * H * W + 8 * ox4;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = 2 * oy + ky - 1;
            const bool ok = iy >= 0;
            const float* xr = xn + (long)(ok ? iy : 0) * W;
            f.lo[ky] = *reinterpret_cast<const float4*>(xr);
            f.hi[ky] = *reinterpret_cast<const float4*>(xr + 4);
            f.m1[ky] = xr[ox4 > 0 ? -1 : 0];
            if (!ok) { f.lo[ky] = make_float4(0.f, 0.f, 0.f, 0.f); f.hi[ky] = f.lo[ky]; }
            if (!ok || ox4 == 0) f.m1[ky] = 0.f;
        }
    };
    f32x4_t acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    auto mac = [&](const Frag& f) {
        const float av[4] = {f.a.x, f.a.y, f.a.z, f.a.w};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const float v[9] = {f.m1[ky], f.lo[ky].x, f.lo[ky].y, f.lo[ky].z, f.lo[ky].w, f.hi[ky].x, f.hi[ky].y, f.hi[ky].z, f.hi[ky].w};
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], v[2 * e + kx], acc[ky * 3 + kx], 0, 0, 0);
        }
    };
    Frag cur, nxt;                                           // (without the prefetch: 463 us instead of 349 for conv1_9 at batch 256)
    load(gb, cur);
    for (long g0 = gb; g0 < ge; g0 += 4) {
        load(g0 + 4 < ge ? g0 + 4 : g0, nxt);
        mac(cur);
        cur = nxt;
    }
    // acc[tap][r] = dW[c0 + 4 lk + r][i0 + lr][tap]
    if (!civ) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = c0 + 4 * lk + r;
        if (m >= Cout) continue;
        float* o = dw + (long)blockIdx.x * part_stride + ((long)m * Cin + ci) * 9;
#pragma unroll
        for (int t = 0; t < 9; ++t) o[t] = acc[t][r];
    }
}

// conv0's weight gradient (1 -> CO channels, 3x3 stride 2): 72 numbers out of 190 MB.  A thread walks groups of 4 output pixels: CO float4
// of dY and the 3 x 9 input window per group, CO x 9 sums in registers; wave shuffle + LDS reduction, one slab per workgroup.
template <int CO>
__global__ void __launch_bounds__(256) tconv3s2_c1_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw,
                                                                int N, int H, int W, int M, long part_stride)
{
    __shared__ float red[4][CO * 9];
    const int Ho = H / 2, Wo = W / 2, per_row = Wo / 4;
    const long G = (long)N * Ho * per_row, HWo = (long)Ho * Wo;
    float acc[CO][9];
#pragma unroll
    for (int c = 0; c < CO; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[c][t] = 0.f;
    for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < G; g += (long)gridDim.x * 256) {
        const int ox4 = (int)(g % per_row), oy = (int)((g / per_row) % Ho);
        const long n = g / ((long)per_row * Ho);
        float4 d[CO];
#pragma unroll
        for (int c = 0; c < CO; ++c) d[c] = c < M ? *reinterpret_cast<const float4*>(dy + (n * M + c) * HWo + (long)oy * Wo + 4 * ox4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = 2 * oy + ky - 1;
            if (iy < 0) continue;
            const float* xr = x + (n * H + iy) * W + 8 * ox4;
            const float4 lo = *reinterpret_cast<const float4*>(xr), hi = *reinterpret_cast<const float4*>(xr + 4);
            const float m1 = ox4 > 0 ? xr[-1] : 0.f;
            const float v[9] = {m1, lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
            for (int c = 0; c < CO; ++c)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[c][ky * 3 + kx] = fmaf(((const float*)&d[c])[e], v[2 * e + kx], acc[c][ky * 3 + kx]);
        }
    }
#pragma unroll
    for (int c = 0; c < CO; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            float v = acc[c][t];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][c * 9 + t] = v;
        }
    __syncthreads();
    if (threadIdx.x < M * 9)
        dw[(long)blockIdx.x * part_stride + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// depthwise weight gradient: dW[c][ky][kx] = sum_q dY[c][q] X[c][q shifted].  grid (chunks, C): every thread walks its output pixels,
// loads dY once and the KS x KS neighbourhood of X, keeps the KS*KS sums in registers; wave + workgroup reduction, one partial per tap
// and chunk (slab blockIdx.x of the scratch).
template <int KS>
__global__ void __launch_bounds__(256) tdw_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw, int N,
                                                        int C, int H, int W, int Ho, int Wo, int stride, long part_stride)
{
    constexpr int KK = KS * KS, PAD = (KS - 1) / 2;
    __shared__ float red[4][KK];
    const int c = blockIdx.y;
    const long HWo = (long)Ho * Wo, Q = (long)N * HWo;
    float acc[KK];
#pragma unroll
    for (int t = 0; t < KK; ++t) acc[t] = 0.f;
    if (stride == 1 && (Wo & 3) == 0) {
        // stride 1, widths multiple of 4: four output pixels per trip -- dY and each window row as aligned float4 loads (see tdw_conv_kernel)
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (long q4 = (long)blockIdx.x * 256 + threadIdx.x; q4 < Q / 4; q4 += (long)gridDim.x * 256) {
            const long q = q4 * 4, n = q / HWo, i = q - n * HWo;
            const int oy = (int)(i / Wo), ox0 = (int)(i - (long)oy * Wo);
            const float4 g4 = *reinterpret_cast<const float4*>(dy + (n * C + c) * HWo + i);
            const float* xp = x + (n * C + c) * H * W;
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) {
                const int iy = oy - PAD + ky;
                if (iy < 0 || iy >= H) continue;
                const float* xr = xp + (long)iy * W;
                const float4 c4 = *reinterpret_cast<const float4*>(xr + ox0);
                const float4 l4 = ox0 >= 4 ? *reinterpret_cast<const float4*>(xr + ox0 - 4) : z4;
                const float4 r4 = ox0 + 4 < W ? *reinterpret_cast<const float4*>(xr + ox0 + 4) : z4;
                float win[4 + 2 * PAD];
#pragma unroll
                for (int j = 0; j < PAD; ++j) { win[j] = ((const float*)&l4)[4 - PAD + j]; win[PAD + 4 + j] = ((const float*)&r4)[j]; }
#pragma unroll
                for (int j = 0; j < 4; ++j) win[PAD + j] = ((const float*)&c4)[j];
#pragma unroll
                for (int kx = 0; kx < KS; ++kx)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[ky * KS + kx] = fmaf(((const float*)&g4)[j], win[j + kx], acc[ky * KS + kx]);
            }
        }
    } else
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < Q; q += (long)gridDim.x * 256) {
        const long n = q / HWo, i = q - n * HWo;
        const int oy = (int)(i / Wo), ox = (int)(i - (long)oy * Wo);
        const float g = dy[(n * C + c) * HWo + i];
        const float* xp = x + (n * C + c) * H * W;
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
            const int iy = oy * stride - PAD + ky;
            const bool yv = iy >= 0 && iy < H;
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                const int ix = ox * stride - PAD + kx;
                const float xv = (yv && ix >= 0 && ix < W) ? xp[(long)iy * W + ix] : 0.f;
                acc[ky * KS + kx] = fmaf(g, xv, acc[ky * KS + kx]);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < KK; ++t) {
        float v = acc[t];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][t] = v;
    }
    __syncthreads();
    if (threadIdx.x < KK)
        dw[(long)blockIdx.x * part_stride + (long)c * KK + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// dw[i] = sum over the slabs, in slab order (deterministic).  One wave per 64 / SPL outputs: SPL lanes share an output when there are many slabs.
__global__ void __launch_bounds__(256) tsum_partials_kernel(const float* __restrict__ part, int nsplit, long nw, long part_stride,
                                                            float* __restrict__ dw, int spl)
{
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const long i = t / spl;
    const int j = (int)(t - i * spl);
    float v = 0.f;
    if (i < nw) {
        int sidx = j;
        for (; sidx + 3 * spl < nsplit; sidx += 4 * spl) {                // four slabs requested at once, added in slab order
            const float p0 = part[(long)sidx * part_stride + i], p1 = part[(long)(sidx + spl) * part_stride + i];
            const float p2 = part[(long)(sidx + 2 * spl) * part_stride + i], p3 = part[(long)(sidx + 3 * spl) * part_stride + i];
            v += p0; v += p1; v += p2; v += p3;
        }
        for (; sidx < nsplit; sidx += spl) v += part[(long)sidx * part_stride + i];
    }
    for (int o = spl >> 1; o > 0; o >>= 1) v += __shfl_down(v, o);       // spl is a power of two <= 64: the lanes of one output are adjacent
    if (i < nw && j == 0) dw[i] = v;
}

// ---- backward-data of the dense 3x3 stride-2 pad-1 convolution (conv1_9): one thread = the 2x2 input block (2a.., 2b..) -- all four
// parities, so every thread runs the same taps -- for CI_T input channels; the weights are wave-uniform.  H = 2 Ho, W = 2 Wo.
//   dx[2a][2b]     = dy[a][b] w11
//   dx[2a][2b+1]   = dy[a][b] w12 + dy[a][b+1] w10
//   dx[2a+1][2b]   = dy[a][b] w21 + dy[a+1][b] w01
//   dx[2a+1][2b+1] = dy[a][b] w22 + dy[a][b+1] w20 + dy[a+1][b] w02 + dy[a+1][b+1] w00
template <int CI_T>
__global__ void __launch_bounds__(256) tconv3s2_bwd_data_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx,
                                                                int N, int Cin, int Cout, int Ho, int Wo)
{
    const long q = (long)blockIdx.x * 256 + threadIdx.x, Q = (long)N * Ho * Wo;
    const int ci0 = blockIdx.y * CI_T;
    if (q >= Q) return;
    const int b = (int)(q % Wo), a = (int)((q / Wo) % Ho), n = (int)(q / ((long)Wo * Ho));
    const bool vb = b + 1 < Wo, va = a + 1 < Ho;
    float acc[4][CI_T];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int j = 0; j < CI_T; ++j) acc[e][j] = 0.f;
    const float* dp = dy + (long)n * Cout * Ho * Wo + (long)a * Wo + b;
    for (int co = 0; co < Cout; ++co) {
        const float* d = dp + (long)co * Ho * Wo;
        const float d00 = d[0], d01 = vb ? d[1] : 0.f, d10 = va ? d[Wo] : 0.f, d11 = (va && vb) ? d[Wo + 1] : 0.f;
        const float* wp = w + ((long)co * Cin + ci0) * 9;
#pragma unroll
        for (int j = 0; j < CI_T; ++j) {
            if (ci0 + j >= Cin) break;
            const float* k = wp + j * 9;
            acc[0][j] = fmaf(d00, k[4], acc[0][j]);
            acc[1][j] = fmaf(d00, k[5], fmaf(d01, k[3], acc[1][j]));
            acc[2][j] = fmaf(d00, k[7], fmaf(d10, k[1], acc[2][j]));
            acc[3][j] = fmaf(d00, k[8], fmaf(d01, k[6], fmaf(d10, k[2], fmaf(d11, k[0], acc[3][j]))));
        }
    }
    const int H = 2 * Ho, W = 2 * Wo;
#pragma unroll
    for (int j = 0; j < CI_T; ++j) {
        if (ci0 + j >= Cin) break;
        float* o = dx + (((long)n * Cin + ci0 + j) * H + 2 * a) * W + 2 * b;
        *reinterpret_cast<float2*>(o) = make_float2(acc[0][j], acc[1][j]);
        *reinterpret_cast<float2*>(o + W) = make_float2(acc[2][j], acc[3][j]);
    }
}

// Backward-data of the dense 3x3 stride-2 pad-1 convolution (conv1_9) on the matrix pipe: the four parity classes of dx are four small
// stride-1 convolutions of dY with 1, 2, 2 and 4 of the nine taps,
//   dx[2a][2b]     = dy[a][b] w11                              dx[2a][2b+1]   = dy[a][b] w12 + dy[a][b+1] w10
//   dx[2a+1][2b]   = dy[a][b] w21 + dy[a+1][b] w01             dx[2a+1][2b+1] = dy[a][b] w22 + dy[a][b+1] w20 + dy[a+1][b] w02 + dy[a+1][b+1] w00
// all on the same operands: lane (lk, lr) = output channel co0 + lk of dY and a GROUP of 4 consecutive columns b (one aligned float4 +
// the next column, for rows a and a + 1); MFMA (class, tap, e) takes column e or e + 1; M = ci (A = the weight, a scalar load per tap).
// The lane ends up with 8 consecutive columns of two rows of dx for each of its 4 MT input channels: float4 stores.
// Needs Cout % 4 == 0, Wo % 4 == 0, Cin <= 16 MT.  One wave = 64 positions of dY = 256 of dx.  (The VALU kernel above: 318 us for conv1_9 at batch 256.)
template <int MT>
__global__ void __launch_bounds__(256) tconv3s2_bwd_mfma_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx, int N,
                                                                int Cin, int Cout, int Ho, int Wo)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lk = lane >> 4;
    const int per_row = Wo / 4;
    const long G = (long)N * Ho * per_row, g0 = ((long)blockIdx.x * 4 + wave) * 16;
    if (g0 >= G) return;
    long g = g0 + lr;
    const bool gv = g < G;
    if (!gv) g = G - 1;
    const int b4 = (int)(g % per_row), a = (int)((g / per_row) % Ho), n = (int)(g / ((long)per_row * Ho));
    const bool row1 = a + 1 < Ho, col4 = 4 * b4 + 4 < Wo;
    const float* dp = dy + (((long)n * Cout + lk) * Ho + a) * Wo + 4 * b4;      // + co0 Ho Wo (+ Wo for row a + 1)
    const float* wp[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int ci = 16 * t + lr;
        wp[t] = w + ((long)lk * Cin + (ci < Cin ? ci : Cin - 1)) * 9;          // + co0 Cin 9 + tap
    }
    // class (py, px) -> accumulators [py][px][t][e]
    f32x4_t acc[2][2][MT][4];
#pragma unroll
    for (int py = 0; py < 2; ++py)
#pragma unroll
        for (int px = 0; px < 2; ++px)
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[py][px][t][e] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    struct Step { float4 r0, r1; float c0, c1; float wk[MT][9]; };
    auto load = [&](int co0, Step& f) {
        const float* d0 = dp + (long)co0 * Ho * Wo;
        f.r0 = *reinterpret_cast<const float4*>(d0);
        f.r1 = *reinterpret_cast<const float4*>(d0 + (row1 ? Wo : 0));
        f.c0 = d0[col4 ? 4 : 0];
        f.c1 = d0[(row1 ? Wo : 0) + (col4 ? 4 : 0)];
        if (!row1) { f.r1 = make_float4(0.f, 0.f, 0.f, 0.f); f.c1 = 0.f; }
        if (!col4) { f.c0 = 0.f; f.c1 = 0.f; }
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int k = 0; k < 9; ++k) f.wk[t][k] = wp[t][(long)co0 * Cin * 9 + k];
    };
    Step cur, nxt;
    load(0, cur);
    for (int co0 = 0; co0 < Cout; co0 += 4) {                // the next 4 channels of dY requested before this step's 18 MT MFMAs
        load(co0 + 4 < Cout ? co0 + 4 : co0, nxt);
        const float v0[5] = {cur.r0.x, cur.r0.y, cur.r0.z, cur.r0.w, cur.c0}, v1[5] = {cur.r1.x, cur.r1.y, cur.r1.z, cur.r1.w, cur.c1};
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f32x4_t& p00 = acc[0][0][t][e]; f32x4_t& p01 = acc[0][1][t][e]; f32x4_t& p10 = acc[1][0][t][e]; f32x4_t& p11 = acc[1][1][t][e];
                const float (&wk)[9] = cur.wk[t];
                p00 = __builtin_amdgcn_mfma_f32_16x16x4f32(wk[4], v0[e], p00, 0, 0, 0);         // w11 dy[a][b]
                p01 = __builtin_amdgcn_mfma_f32_16x16x4f32(wk[5], v0[e], p01, 0, 0, 0);         // w12 dy[a][b]
                p01 = __builtin_amdgcn_mfma_f32_16x16x4f32(wk[3], v0[e + 1], p01, 0, 0, 0);     // w10 dy[a][b+1]
                p10 = __builtin_amdgcn_mfma_f32_16x16x4f32(wk[7], v0[e], p10, 0, 0, 0);         // w21 dy[a][b]
                p10 = __builtin_amdgcn_mfma_f32_16x16x4f32(wk[1], v1[e], p10, 0, 0, 0);         // w01 dy[a+1][b]
                p11 = __builtin_amdgcn_mfma_f32_16x16x4f32(wk[8], v0[e], p11, 0, 0, 0);         // w22 dy[a][b]
                p11 = __builtin_amdgcn_mfma_f32_16x16x4f32(wk[6], v0[e + 1], p11, 0, 0, 0);     // w20 dy[a][b+1]
                p11 = __builtin_amdgcn_mfma_f32_16x16x4f32(wk[2], v1[e], p11, 0, 0, 0);         // w02 dy[a+1][b]
                p11 = __builtin_amdgcn_mfma_f32_16x16x4f32(wk[0], v1[e + 1], p11, 0, 0, 0);     // w00 dy[a+1][b+1]
            }
        cur = nxt;
    }
    if (!gv) return;
    const int H = 2 * Ho, W = 2 * Wo;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ci = 16 * t + 4 * lk + r;
            if (ci >= Cin) continue;
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                float* o = dx + (((long)n * Cin + ci) * H + 2 * a + py) * W + 8 * b4;
                *reinterpret_cast<float4*>(o) = make_float4(acc[py][0][t][0][r], acc[py][1][t][0][r], acc[py][0][t][1][r], acc[py][1][t][1][r]);
                *reinterpret_cast<float4*>(o + 4) = make_float4(acc[py][0][t][2][r], acc[py][1][t][2][r], acc[py][0][t][3][r], acc[py][1][t][3][r]);
            }
        }
}

// ---- depthwise convolution, one (frame, channel) plane per blockIdx.x so that the KS*KS weights are wave-uniform; a thread computes 4
// consecutive outputs of a row (Wo % 4 == 0) from the KS x (3 S + KS) input window.  FLIP: the weights reversed -- the backward-data
// of a stride-1 depthwise conv is the same conv of dY with the flipped kernel. ----
template <int KS, int S, bool FLIP>
__global__ void tdw_conv_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y, int C, int H, int W, int Ho, int Wo)
{
    constexpr int KK = KS * KS, PAD = (KS - 1) / 2, WIN = 3 * S + KS;
    const int plane = blockIdx.x, c = plane % C;                    // planes in x (N * C may exceed 65535), the plane's thread chunks in y
    const int t = blockIdx.y * blockDim.x + threadIdx.x, per_row = Wo / 4;
    if (t >= Ho * per_row) return;
    const int oy = t / per_row, ox0 = (t - oy * per_row) * 4;
    const float* xp = x + (long)plane * H * W;
    const float* wp = w + (long)c * KK;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < KS; ++ky) {
        const int iy = oy * S - PAD + ky;
        if (iy < 0 || iy >= H) continue;
        const float* xr = xp + (long)iy * W;
        float win[WIN];
        if constexpr (S == 1) {
            // stride 1: the window is [ox0 - PAD, ox0 + 3 + PAD]; ox0 and W are multiples of 4, so it is three ALIGNED float4 loads -- the
            // 4 centre inputs, the quad before (its last PAD elements) and the quad after (its first PAD) -- instead of 4 + 2 PAD scalar ones
            const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 c4 = *reinterpret_cast<const float4*>(xr + ox0);
            const float4 l4 = ox0 >= 4 ? *reinterpret_cast<const float4*>(xr + ox0 - 4) : z4;
            const float4 r4 = ox0 + 4 < W ? *reinterpret_cast<const float4*>(xr + ox0 + 4) : z4;
#pragma unroll
            for (int j = 0; j < PAD; ++j) { win[j] = ((const float*)&l4)[4 - PAD + j]; win[PAD + 4 + j] = ((const float*)&r4)[j]; }
#pragma unroll
            for (int j = 0; j < 4; ++j) win[PAD + j] = ((const float*)&c4)[j];
        } else {
#pragma unroll
            for (int j = 0; j < WIN; ++j) {
                const int ix = ox0 * S - PAD + j;
                win[j] = (ix >= 0 && ix < W) ? xr[ix] : 0.f;
            }
        }
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
            const float wv = FLIP ? wp[KK - 1 - (ky * KS + kx)] : wp[ky * KS + kx];
#pragma unroll
            for (int o = 0; o < 4; ++o) acc[o] = fmaf(win[o * S + kx], wv, acc[o]);
        }
    }
    *reinterpret_cast<float4*>(y + ((long)plane * Ho + oy) * Wo + ox0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

// ---- stride-1 depthwise convolution for the large maps, built for bandwidth: a thread owns 4 columns x R rows of the output and walks
// the R + KS - 1 input rows once, ONE aligned float4 per row; the PAD columns either side come from the neighbouring lanes (the quads
// ox0 -+ 4 of the same row are lanes -+ 1: a wave is a run of consecutive quads), by global loads only at the wave's two ends.
// (tdw_conv_kernel: 3 float4 loads per input row and output row, 9 per output quad -- the texture path, not HBM, was its limit.)
// No lane leaves before the last cross-lane exchange; threads past the plane compute on clamped addresses and store nothing. ----
template <int PAD>
__device__ __forceinline__ void tdw_row_window(const float* __restrict__ xr, bool row_ok, int ox0, int W, int lane, float (&win)[4 + 2 * PAD])
{
    float4 c4 = *reinterpret_cast<const float4*>(xr + ox0);
    if (!row_ok) c4 = make_float4(0.f, 0.f, 0.f, 0.f);
    win[PAD] = c4.x; win[PAD + 1] = c4.y; win[PAD + 2] = c4.z; win[PAD + 3] = c4.w;
#pragma unroll
    for (int j = 0; j < PAD; ++j) {
        // element ox0 - PAD + j = component 4 - PAD + j of the quad before; element ox0 + 4 + j = component j of the quad after
        float l = __shfl_up(win[PAD + 4 - PAD + j], 1), r = __shfl_down(win[PAD + j], 1);
        if (lane == 0) l = (row_ok && ox0 > 0) ? xr[ox0 - PAD + j] : 0.f;
        if (lane == 63) r = (row_ok && ox0 + 4 < W) ? xr[ox0 + 4 + j] : 0.f;
        win[j] = ox0 > 0 ? l : 0.f;
        win[PAD + 4 + j] = ox0 + 4 < W ? r : 0.f;
    }
}

// MANY: small planes (16x20: 20 threads' worth) -- the planes are numbered through the thread index as well, a workgroup covers a dozen
// of them and the weights are per-lane loads; otherwise one plane per blockIdx.x and wave-uniform weights.
template <int KS, bool FLIP, int R, bool MANY = false>
__global__ void __launch_bounds__(256) tdw_rows_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y, int C, int H, int W,
                                                       long nplanes = 0, float2* __restrict__ stat = nullptr,
                                                       TRedArgs red = TRedArgs{nullptr, nullptr, nullptr, nullptr, nullptr, 0})
{
    constexpr int KK = KS * KS, PAD = (KS - 1) / 2;
    const int lane = threadIdx.x & 63;
    const int per_row = W / 4, per_plane = (H / R) * per_row;
    long plane, t, count;
    if constexpr (MANY) {
        const long g = (long)blockIdx.x * 256 + threadIdx.x;
        count = nplanes * per_plane;
        const long gc = g < count ? g : count - 1;
        plane = gc / per_plane;
        t = g < count ? gc - plane * per_plane : per_plane;      // (>= per_plane: nothing to store)
        count = per_plane;
    } else {
        plane = blockIdx.x;
        t = blockIdx.y * blockDim.x + threadIdx.x;          // (the workgroup is sized to the plane: 64 .. 256 threads)
        count = per_plane;
    }
    const int c = (int)(plane % C);
    const int tc = (int)(t < count ? t : count - 1);
    const int rb = tc / per_row, ox0 = (tc - rb * per_row) * 4, oy0 = rb * R;
    const float* xp = x + plane * H * W;
    float wk[KK];
#pragma unroll
    for (int i = 0; i < KK; ++i) wk[i] = w[(long)c * KK + (FLIP ? KK - 1 - i : i)];
    float acc[R][4];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[r][o] = 0.f;
#pragma unroll
    for (int j = 0; j < R + 2 * PAD; ++j) {
        const int iy = oy0 - PAD + j;
        const bool ok = iy >= 0 && iy < H;
        float win[4 + 2 * PAD];
        tdw_row_window<PAD>(xp + (long)(ok ? iy : 0) * W, ok, ox0, W, lane, win);
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
            const int r = j - ky;                                   // input row j is tap row ky of output row j - ky
            if (r < 0 || r >= R) continue;
#pragma unroll
            for (int kx = 0; kx < KS; ++kx)
#pragma unroll
                for (int o = 0; o < 4; ++o) acc[r][o] = fmaf(win[o + kx], wk[ky * KS + kx], acc[r][o]);
        }
    }
    if constexpr (!MANY && !FLIP) {
        if (stat) {                                         // BatchNorm statistics of this workgroup's outputs (one channel): see tile_stats_store
            __shared__ double red[4][2];
            if (threadIdx.x < 8) red[threadIdx.x >> 1][threadIdx.x & 1] = 0;      // (a workgroup may have fewer than 4 waves)
            __syncthreads();
            double s1 = 0, s2 = 0;
            if (t < count) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const double a = acc[r][0], b = acc[r][1], c2 = acc[r][2], d = acc[r][3];
                    s1 += (a + b) + (c2 + d);
                    s2 += (a * a + b * b) + (c2 * c2 + d * d);
                }
            }
            for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_down(s1, o); s2 += __shfl_down(s2, o); }
            if (lane == 0) { red[threadIdx.x >> 6][0] = s1; red[threadIdx.x >> 6][1] = s2; }
            __syncthreads();
            if (threadIdx.x == 0) {
                const long nblocks = (long)(gridDim.x / C) * gridDim.y, block = (long)(plane / C) * gridDim.y + blockIdx.y;
                stat[(long)c * nblocks + block] = make_float2((float)((red[0][0] + red[1][0]) + (red[2][0] + red[3][0])),
                                                              (float)((red[0][1] + red[1][1]) + (red[2][1] + red[3][1])));
            }
        }
    }
    if constexpr (!MANY && FLIP) {
        if (red.part) {                                     // this IS dy of the layer below: its backward BatchNorm sums (see tpw4_mfma_kernel)
            __shared__ float rred[4][2];
            if (threadIdx.x < 8) rred[threadIdx.x >> 1][threadIdx.x & 1] = 0.f;
            __syncthreads();
            const float mean = red.stats[2 * c], inv = red.stats[2 * c + 1], gm = red.gamma[c], bt = red.beta[c];
            float s1 = 0.f, s2 = 0.f;
            if (t < count) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float4 z4 = *reinterpret_cast<const float4*>(red.z + (plane * H + oy0 + r) * W + ox0);
                    const float zz[4] = {z4.x, z4.y, z4.z, z4.w};
#pragma unroll
                    for (int o = 0; o < 4; ++o) {
                        float g = acc[r][o];
                        if (red.relu && !(tbn_affine(zz[o], mean, inv, gm, bt) > 0.f)) g = 0.f;
                        s1 += g;
                        s2 = fmaf(g, (zz[o] - mean) * inv, s2);
                    }
                }
            }
            for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_down(s1, o); s2 += __shfl_down(s2, o); }
            if (lane == 0) { rred[threadIdx.x >> 6][0] = s1; rred[threadIdx.x >> 6][1] = s2; }
            __syncthreads();
            if (threadIdx.x == 0) {
                const long nblocks = (long)(gridDim.x / C) * gridDim.y, block = (long)(plane / C) * gridDim.y + blockIdx.y;
                red.part[(long)c * nblocks + block] = make_float2((rred[0][0] + rred[1][0]) + (rred[2][0] + rred[3][0]),
                                                                  (rred[0][1] + rred[1][1]) + (rred[2][1] + rred[3][1]));
            }
        }
    }
    if (t >= count) return;
#pragma unroll
    for (int r = 0; r < R; ++r)
        *reinterpret_cast<float4*>(y + (plane * H + oy0 + r) * W + ox0) = make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
}

// the weight gradient of the same convolutions, same access pattern: per trip a thread takes 4 columns x R rows of dY (R float4) and the
// R + KS - 1 input rows (one float4 each + the lane exchange), KS*KS sums in registers; grid (chunks, C), a workgroup's trips stride over
// the (frame, row block, quad) list with a wave-uniform trip count.
template <int KS, int R>
__global__ void __launch_bounds__(256) tdw_wgrad_rows_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw, int N,
                                                             int C, int H, int W, long part_stride)
{
    constexpr int KK = KS * KS, PAD = (KS - 1) / 2;
    __shared__ float red[4][KK];
    const int c = blockIdx.y, lane = threadIdx.x & 63;
    const int per_row = W / 4, per_plane = (H / R) * per_row;
    const long total = (long)N * per_plane;
    float acc[KK];
#pragma unroll
    for (int i = 0; i < KK; ++i) acc[i] = 0.f;
    for (long base = (long)blockIdx.x * 256; base < total; base += (long)gridDim.x * 256) {
        const long g = base + threadIdx.x;
        const bool gv = g < total;
        const long gc = gv ? g : total - 1;
        const long n = gc / per_plane;
        const int t = (int)(gc - n * per_plane), rb = t / per_row, ox0 = (t - rb * per_row) * 4, oy0 = rb * R;
        const float* xp = x + (n * C + c) * (long)H * W;
        const float* gp = dy + (n * C + c) * (long)H * W + (long)oy0 * W + ox0;
        float4 g4[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            g4[r] = *reinterpret_cast<const float4*>(gp + (long)r * W);
            if (!gv) g4[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < R + 2 * PAD; ++j) {
            const int iy = oy0 - PAD + j;
            const bool ok = iy >= 0 && iy < H;
            float win[4 + 2 * PAD];
            tdw_row_window<PAD>(xp + (long)(ok ? iy : 0) * W, ok, ox0, W, lane, win);
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) {
                const int r = j - ky;
                if (r < 0 || r >= R) continue;
#pragma unroll
                for (int kx = 0; kx < KS; ++kx)
#pragma unroll
                    for (int o = 0; o < 4; ++o) acc[ky * KS + kx] = fmaf(((const float*)&g4[r])[o], win[o + kx], acc[ky * KS + kx]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < KK; ++i) {
        float v = acc[i];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
        if (lane == 0) red[threadIdx.x >> 6][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < KK)
        dw[(long)blockIdx.x * part_stride + (long)c * KK + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// ---- stride-1 depthwise convolution on planes whose width is not a multiple of 4 (the 8x10 maps of stride 32 at 256x320: the float4
// kernels above do not apply, and the one-thread-per-element fallback spent 57 us on an 18 MB tensor): a thread = one output ROW of one
// plane (W <= 16), the KS input rows in registers.  Same for the weight gradient, grid (chunks, C). ----
template <int KS, bool FLIP>
__global__ void __launch_bounds__(256) tdw_plane_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y, int C, int H,
                                                        int W, long nrows)
{
    constexpr int KK = KS * KS, PAD = (KS - 1) / 2, MW = 16;
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    if (r >= nrows) return;
    const long plane = r / H;
    const int oy = (int)(r - plane * H), c = (int)(plane % C);
    const float* xp = x + plane * H * W;
    float wk[KK];
#pragma unroll
    for (int i = 0; i < KK; ++i) wk[i] = w[(long)c * KK + (FLIP ? KK - 1 - i : i)];
    float acc[MW];
#pragma unroll
    for (int j = 0; j < MW; ++j) acc[j] = 0.f;
#pragma unroll
    for (int ky = 0; ky < KS; ++ky) {
        const int iy = oy - PAD + ky;
        if (iy < 0 || iy >= H) continue;
        const float* xr = xp + (long)iy * W;
        float row[MW + 2 * PAD];
#pragma unroll
        for (int j = 0; j < MW + 2 * PAD; ++j) row[j] = (j >= PAD && j - PAD < W) ? xr[j - PAD] : 0.f;
#pragma unroll
        for (int kx = 0; kx < KS; ++kx)
#pragma unroll
            for (int j = 0; j < MW; ++j) acc[j] = fmaf(row[j + kx], wk[ky * KS + kx], acc[j]);
    }
    float* yr = y + r * W;
#pragma unroll
    for (int j = 0; j < MW; ++j)
        if (j < W) yr[j] = acc[j];
}
template <int KS>
__global__ void __launch_bounds__(256) tdw_plane_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw, int N,
                                                              int C, int H, int W, long part_stride)
{
    constexpr int KK = KS * KS, PAD = (KS - 1) / 2, MW = 16;
    __shared__ float red[4][KK];
    const int c = blockIdx.y;
    const long rows = (long)N * H;
    float acc[KK];
#pragma unroll
    for (int i = 0; i < KK; ++i) acc[i] = 0.f;
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < rows; r += (long)gridDim.x * 256) {
        const long n = r / H;
        const int oy = (int)(r - n * H);
        const float* xp = x + (n * C + c) * (long)H * W;
        const float* gr = dy + ((n * C + c) * (long)H + oy) * W;
        float g[MW];
#pragma unroll
        for (int j = 0; j < MW; ++j) g[j] = j < W ? gr[j] : 0.f;
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
            const int iy = oy - PAD + ky;
            if (iy < 0 || iy >= H) continue;
            const float* xr = xp + (long)iy * W;
            float row[MW + 2 * PAD];
#pragma unroll
            for (int j = 0; j < MW + 2 * PAD; ++j) row[j] = (j >= PAD && j - PAD < W) ? xr[j - PAD] : 0.f;
#pragma unroll
            for (int kx = 0; kx < KS; ++kx)
#pragma unroll
                for (int j = 0; j < MW; ++j) acc[ky * KS + kx] = fmaf(g[j], row[j + kx], acc[ky * KS + kx]);
        }
    }
#pragma unroll
    for (int i = 0; i < KK; ++i) {
        float v = acc[i];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < KK)
        dw[(long)blockIdx.x * part_stride + (long)c * KK + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// backward-data of the depthwise 3x3 stride-2 pad-1 convolution: one thread = the 2x2 input block (2a.., 2b..), see tconv3s2_bwd_data_kernel
__global__ void tdw3s2_bwd_data_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx, int C, int Ho, int Wo)
{
    const int plane = blockIdx.x, c = plane % C;
    const int t = blockIdx.y * blockDim.x + threadIdx.x;
    if (t >= Ho * Wo) return;
    const int a = t / Wo, b = t - a * Wo;
    const bool vb = b + 1 < Wo, va = a + 1 < Ho;
    const float* d = dy + (long)plane * Ho * Wo + (long)a * Wo + b;
    const float d00 = d[0], d01 = vb ? d[1] : 0.f, d10 = va ? d[Wo] : 0.f, d11 = (va && vb) ? d[Wo + 1] : 0.f;
    const float* k = w + (long)c * 9;
    const int W = 2 * Wo;
    float* o = dx + ((long)plane * 2 * Ho + 2 * a) * W + 2 * b;
    *reinterpret_cast<float2*>(o) = make_float2(d00 * k[4], fmaf(d00, k[5], d01 * k[3]));
    *reinterpret_cast<float2*>(o + W) = make_float2(fmaf(d00, k[7], d10 * k[1]), fmaf(d00, k[8], fmaf(d01, k[6], fmaf(d10, k[2], d11 * k[0]))));
}

// where a conv kernel may leave BatchNorm's partial sums: behind the first MB of the scratch (BatchNorm's own chunk pairs)
static inline bool tstat_room(TStatPart* st, long count, int C)
{
    if (!st || !st->part) return false;
    static const bool off = getenv("YF_TSTAT_OFF") != nullptr;
    if (off || (size_t)count * C * sizeof(float2) > st->cap_bytes) return false;
    st->count = count;
    return true;
}
static inline bool tred_room(TBnRed* red, long count, int C)
{
    if (!red || !red->part || !red->z) return false;
    static const bool off = getenv("YF_TRED_OFF") != nullptr;
    if (off || (size_t)count * C * sizeof(float2) > red->cap_bytes) return false;
    red->count = count;
    return true;
}
static const bool tdw_rows_off = getenv("YF_TDW_ROWS_OFF") != nullptr;
template <int KS, int S, bool FLIP>
static void launch_tdw_conv(const float* x, const float* w, float* y, int N, int C, int H, int W, int Ho, int Wo, hipStream_t s, TStatPart* st = nullptr,
                            TBnRed* red = nullptr)
{
    if constexpr (S == 1) {
        // large maps: 4 rows per thread (see tdw_rows_kernel); the plane must still give a workgroup something to do
        if (!tdw_rows_off && H % 4 == 0 && (H / 4) * (W / 4) >= 64) {
            // workgroup = 64 .. 256 threads, whichever leaves the fewest idle (a 32x40 plane is 80 threads' worth: 256 would idle 69 % of them)
            const int count = (H / 4) * (W / 4);
            int bs = 256, waste = (count + 255) / 256 * 256 - count;
            for (int b = 192; b >= 64; b -= 64) {
                const int wst = (count + b - 1) / b * b - count;
                if (wst < waste) { waste = wst; bs = b; }
            }
            const int ny = (count + bs - 1) / bs;
            float2* sp = (!FLIP && tstat_room(st, (long)N * ny, C)) ? st->part : nullptr;
            TRedArgs ra{nullptr, nullptr, nullptr, nullptr, nullptr, 0};
            if (FLIP && tred_room(red, (long)N * ny, C)) ra = TRedArgs{red->z, red->stats, red->gamma, red->beta, red->part, red->relu};
            hipLaunchKernelGGL((tdw_rows_kernel<KS, FLIP, 4>), dim3(N * C, ny), dim3(bs), 0, s, x, w, y, C, H, W, 0L, sp, ra);
            return;
        }
        if (!tdw_rows_off && H % 4 == 0 && (long)N * C * (H / 4) * (W / 4) >= 16384) {      // small planes, many of them
            const long total = (long)N * C * (H / 4) * (W / 4);
            hipLaunchKernelGGL((tdw_rows_kernel<KS, FLIP, 4, true>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, w, y, C, H, W, (long)N * C);
            return;
        }
    }
    const int threads = Ho * (Wo / 4), bs = threads <= 64 ? 64 : 256;
    hipLaunchKernelGGL((tdw_conv_kernel<KS, S, FLIP>), dim3(N * C, (threads + bs - 1) / bs), dim3(bs), 0, s, x, w, y, C, H, W, Ho, Wo);
}

// ---- Conv2d backward with respect to the input ----
__global__ void __launch_bounds__(256) tconv_bwd_data_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx,
                                                             int N, int Cin, int H, int W, int Cout, int Ho, int Wo, int k, int stride, int depthwise)
{
    const long idx = (long)blockIdx.x * 256 + threadIdx.x, total = (long)N * Cin * H * W;
    if (idx >= total) return;
    const int ix = (int)(idx % W), iy = (int)((idx / W) % H), ci = (int)((idx / ((long)W * H)) % Cin), n = (int)(idx / ((long)W * H * Cin));
    const int pad = (k - 1) / 2;
    float s = 0.f;
    const int o0 = depthwise ? ci : 0, o1 = depthwise ? ci + 1 : Cout;
    for (int co = o0; co < o1; ++co) {
        const float* dp = dy + ((long)n * Cout + co) * Ho * Wo;
        const float* wp = w + ((long)co * (depthwise ? 1 : Cin) + (depthwise ? 0 : ci)) * k * k;
        for (int ky = 0; ky < k; ++ky) {
            const int ty = iy + pad - ky;
            if (ty < 0 || ty % stride) continue;
            const int oy = ty / stride;
            if (oy >= Ho) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int tx = ix + pad - kx;
                if (tx < 0 || tx % stride) continue;
                const int ox = tx / stride;
                if (ox >= Wo) continue;
                s = fmaf(dp[(long)oy * Wo + ox], wp[ky * k + kx], s);
            }
        }
    }
    dx[idx] = s;
}

// ---- Conv2d backward with respect to the weight: one workgroup per (weight element, chunk of the N*Ho*Wo reduction) ----
__global__ void __launch_bounds__(256) tconv_bwd_weight_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw,
                                                               int N, int Cin, int H, int W, int Cout, int Ho, int Wo, int k, int stride,
                                                               int depthwise, int nchunk)
{
    __shared__ float red[4];
    const int chunk = blockIdx.x % nchunk;
    const long widx = blockIdx.x / nchunk;              // (co, ci', ky, kx), ci' = 0 for depthwise
    const int kx = (int)(widx % k), ky = (int)((widx / k) % k);
    const int cig = depthwise ? 1 : Cin;
    const int ci_ = (int)((widx / ((long)k * k)) % cig), co = (int)(widx / ((long)k * k * cig));
    const int ci = depthwise ? co : ci_;
    const int pad = (k - 1) / 2;
    const long P = (long)N * Ho * Wo, per = (P + nchunk - 1) / nchunk, p0 = chunk * per, p1 = p0 + per < P ? p0 + per : P;
    float s = 0.f;
    for (long p = p0 + threadIdx.x; p < p1; p += 256) {
        const int ox = (int)(p % Wo), oy = (int)((p / Wo) % Ho), n = (int)(p / ((long)Wo * Ho));
        const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
        s = fmaf(x[(((long)n * Cin + ci) * H + iy) * W + ix], dy[(((long)n * Cout + co) * Ho + oy) * Wo + ox], s);
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&dw[widx], red[0] + red[1] + red[2] + red[3]);
}

// ---- dense conv weight gradient as a split-K GEMM: dW[co][r] = sum_p dy[co][p] * X[r][p], r = (ci, ky, kx) (im2col row), p = output
// pixel over the batch.  One workgroup = a 64 x 64 tile of (co, r) and a slice of p; 32 pixels at a time are staged in LDS, each
// thread accumulates 4 x 4 outputs and adds them to dW with atomics at the end.
__global__ void __launch_bounds__(256) tconv_bwd_weight_gemm_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw,
                                                                    int N, int Cin, int H, int W, int Cout, int Ho, int Wo, int k, int stride,
                                                                    long p_per)
{
    __shared__ float xs[64][33], ds[64][33];
    const int tid = threadIdx.x, tc = tid & 15, tr = tid >> 4;
    const int R = Cin * k * k, r0 = blockIdx.y * 64, c0 = blockIdx.z * 64, pad = (k - 1) / 2, kk = k * k;
    const long P = (long)N * Ho * Wo, HWo = (long)Ho * Wo;
    const long pb = (long)blockIdx.x * p_per, pe = pb + p_per < P ? pb + p_per : P;
    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
    const int px = tid & 31, row8 = tid >> 5;                     // loader: 8 rows x 32 pixels per pass, 8 passes
    for (long p0 = pb; p0 < pe; p0 += 32) {
        const long p = p0 + px;
        const bool pv = p < pe;
        int n = 0, oy = 0, ox = 0;
        if (pv) { n = (int)(p / HWo); const long rem = p - (long)n * HWo; oy = (int)(rem / Wo); ox = (int)(rem - (long)oy * Wo); }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = it * 8 + row8;
            float dv = 0.f, xv = 0.f;
            if (pv) {
                const int co = c0 + row;
                if (co < Cout) dv = dy[((long)n * Cout + co) * HWo + (long)oy * Wo + ox];
                const int r = r0 + row;
                if (r < R) {
                    const int ci = r / kk, t = r - ci * kk, ky = t / k, kx = t - ky * k;
                    const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
                    if (iy >= 0 && iy < H && ix >= 0 && ix < W) xv = x[(((long)n * Cin + ci) * H + iy) * W + ix];
                }
            }
            ds[row][px] = dv;
            xs[row][px] = xv;
        }
        __syncthreads();
#pragma unroll 8
        for (int q = 0; q < 32; ++q) {
            float dv[4], xv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) { dv[a] = ds[tr * 4 + a][q]; xv[a] = xs[tc * 4 + a][q]; }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = fmaf(dv[a], xv[b], acc[a][b]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int co = c0 + tr * 4 + a, r = r0 + tc * 4 + b;
            if (co < Cout && r < R && acc[a][b] != 0.f) atomicAdd(&dw[(long)co * R + r], acc[a][b]);
        }
}

// ---- ConvTranspose2d(k = 2, stride = 2, pad = 0), weight [Cin, Cout, 2, 2] ----
__global__ void __launch_bounds__(256) tdeconv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y, int N,
                                                          int Cin, int H, int W, int Cout)
{
    const int Ho = 2 * H, Wo = 2 * W;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x, total = (long)N * Cout * Ho * Wo;
    if (idx >= total) return;
    const int ox = (int)(idx % Wo), oy = (int)((idx / Wo) % Ho), co = (int)((idx / ((long)Wo * Ho)) % Cout), n = (int)(idx / ((long)Wo * Ho * Cout));
    const int iy = oy >> 1, ix = ox >> 1, dy_ = oy & 1, dx_ = ox & 1;
    float s = 0.f;
    for (int ci = 0; ci < Cin; ++ci) s = fmaf(x[(((long)n * Cin + ci) * H + iy) * W + ix], w[(((long)ci * Cout + co) * 2 + dy_) * 2 + dx_], s);
    y[idx] = s;
}

// one thread per input element: the fallback for small batches (few pixels: the GEMM form below has too few waves)
__global__ void __launch_bounds__(256) tdeconv_bwd_data_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx,
                                                               int N, int Cin, int H, int W, int Cout)
{
    const int Ho = 2 * H, Wo = 2 * W;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x, total = (long)N * Cin * H * W;
    if (idx >= total) return;
    const int ix = (int)(idx % W), iy = (int)((idx / W) % H), ci = (int)((idx / ((long)W * H)) % Cin), n = (int)(idx / ((long)W * H * Cin));
    float s = 0.f;
    for (int co = 0; co < Cout; ++co)
        for (int q = 0; q < 4; ++q)
            s = fmaf(dy[(((long)n * Cout + co) * Ho + 2 * iy + (q >> 1)) * Wo + 2 * ix + (q & 1)], w[((long)ci * Cout + co) * 4 + q], s);
    dx[idx] = s;
}

__global__ void __launch_bounds__(256) tdeconv_bwd_weight_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw,
                                                                 int N, int Cin, int H, int W, int Cout, int nchunk)
{
    __shared__ float red[4];
    const int Ho = 2 * H, Wo = 2 * W;
    const int chunk = blockIdx.x % nchunk;
    const long widx = blockIdx.x / nchunk;              // (ci, co, dy, dx)
    const int q = (int)(widx & 3), co = (int)((widx >> 2) % Cout), ci = (int)((widx >> 2) / Cout);
    const long P = (long)N * H * W, per = (P + nchunk - 1) / nchunk, p0 = chunk * per, p1 = p0 + per < P ? p0 + per : P;
    float s = 0.f;
    for (long p = p0 + threadIdx.x; p < p1; p += 256) {
        const int ix = (int)(p % W), iy = (int)((p / W) % H), n = (int)(p / ((long)W * H));
        s = fmaf(x[(((long)n * Cin + ci) * H + iy) * W + ix], dy[(((long)n * Cout + co) * Ho + 2 * iy + (q >> 1)) * Wo + 2 * ix + (q & 1)], s);
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&dw[widx], red[0] + red[1] + red[2] + red[3]);
}

// ---- BatchNorm2d, training mode (torch.nn.BatchNorm2d: eps 1e-5, momentum 0.1; running_var takes the UNBIASED batch variance) ----
// Two launches each way, no cross-workgroup synchronisation inside a kernel (a device-scope fence costs an L2 write-back per XCD):
//   1. partial sums over N*H*W per channel in double: grid (nchunk <= 256, C), every workgroup sums units of 256 V contiguous
//      elements (V = 4: one float4 per thread, when H*W % 4 == 0) and stores its pair into scratch[c][chunk];
//   2. the elementwise kernel, grid (blocks, C): each workgroup first adds its channel's partial pairs (lane l takes chunks l, l + 64,
//      ..., then a fixed shuffle tree: deterministic), then transforms its units; the first workgroup of a channel also writes the
//      per-channel results (stats + running statistics, or dgamma / dbeta).
#define TBN_MAXCHUNK 256
template <int V> struct tbn_vec;
template <> struct tbn_vec<1> { typedef float type; };
template <> struct tbn_vec<4> { typedef float4 type; };
template <int V> __device__ __forceinline__ float tbn_at(const typename tbn_vec<V>::type& v, int j) { return ((const float*)&v)[j]; }

// unit -> element mapping of the four kernels.  Per-frame units (256 V elements of ONE frame, the tail of a plane idle) suit the large
// maps; FLAT (V = 4) numbers the float4 of a channel across the frames, so a 16x20 or 8x10 plane does not leave 40-70 % of a workgroup idle.
template <int V, bool FLAT>
struct TbnMap {
    long per, total;                                                    // FLAT: float4 per plane, float4 per channel; else units per frame, -
    __device__ TbnMap(int N, long HW) : per(FLAT ? HW / V : (HW + 256 * V - 1) / (256 * V)), total(FLAT ? (long)N * (HW / V) : 0) {}
    __device__ long units(int N) const { return FLAT ? (total + 255) / 256 : (long)N * per; }
    __device__ bool at(long u, long HW, long& n, long& i) const
    {
        if constexpr (FLAT) {
            const long f = u * 256 + threadIdx.x;
            n = f / per; i = (f - n * per) * V;
            return f < total;
        } else {
            n = u / per; i = ((u - n * per) * 256 + threadIdx.x) * V;
            return i < HW;
        }
    }
};
static inline long tbn_units(int N, long HW, int V, bool flat) { return flat ? ((long)N * (HW / V) + 255) / 256 : (long)N * ((HW + 256 * V - 1) / (256 * V)); }

// y before the ReLU, in ONE fixed operation order: the backward recomputes it from z to get the ReLU mask (y > 0) without reading y
__device__ __forceinline__ float tbn_affine(float x, float mean, float invstd, float gamma, float beta)
{
    return __fmaf_rn(__fmul_rn(__fsub_rn(x, mean), invstd), gamma, beta);
}
__device__ __forceinline__ void tbn_block_store(double s, double t, double* __restrict__ part)
{
    __shared__ double r1[4], r2[4];
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_down(s, o); t += __shfl_down(t, o); }
    if ((threadIdx.x & 63) == 0) { r1[threadIdx.x >> 6] = s; r2[threadIdx.x >> 6] = t; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = r1[0] + r1[1] + r1[2] + r1[3];
        part[2 * blockIdx.x + 1] = r2[0] + r2[1] + r2[2] + r2[3];
    }
}
// the channel's two sums, in every thread of the workgroup
__device__ __forceinline__ void tbn_block_total(const double* __restrict__ part, int nchunk, double& s, double& t)
{
    __shared__ double tot[2];
    if (threadIdx.x < 64) {
        double a = 0, b = 0;
        double pa[4], pb[4];                                 // nchunk <= TBN_MAXCHUNK = 256: at most 4 per lane, requested together (a rolled
#pragma unroll                                               // loop waits for every pair before asking for the next: 4 round trips at the top of
        for (int u = 0; u < 4; ++u) {                        // every workgroup of the elementwise kernels)
            const int i = threadIdx.x + 64 * u;
            pa[u] = i < nchunk ? part[2 * i] : 0.0;
            pb[u] = i < nchunk ? part[2 * i + 1] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { a += pa[u]; b += pb[u]; }
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o); b += __shfl_down(b, o); }
        if (threadIdx.x == 0) { tot[0] = a; tot[1] = b; }
    }
    __syncthreads();
    s = tot[0]; t = tot[1];
}

template <int V, bool FLAT = false>
__global__ void __launch_bounds__(256) tbn_stats_kernel(const float* __restrict__ x, int N, int C, long HW, double* __restrict__ scratch)
{
    typedef typename tbn_vec<V>::type vec;
    const int c = blockIdx.y, nchunk = gridDim.x;
    const TbnMap<V, FLAT> map(N, HW);
    const long U = map.units(N);
    double s = 0, ss = 0;
    for (long u = blockIdx.x; u < U; u += nchunk) {
        long n, i;
        if (map.at(u, HW, n, i)) {
            const vec v = *reinterpret_cast<const vec*>(x + (n * C + c) * HW + i);
#pragma unroll
            for (int j = 0; j < V; ++j) { const double e = tbn_at<V>(v, j); s += e; ss += e * e; }
        }
    }
    tbn_block_store(s, ss, scratch + (long)c * TBN_MAXCHUNK * 2);
}

// the statistics from the pairs a conv kernel left per pixel block (tile_stats_store): grid (chunks, C), a chunk's share of the channel's
// pairs added in double -> the chunk pair the elementwise kernel expects from tbn_stats_kernel
__global__ void __launch_bounds__(256) tbn_stats_from_parts_kernel(const float2* __restrict__ part, long count, double* __restrict__ scratch)
{
    const int c = blockIdx.y;
    double s = 0, ss = 0;
    const long step = (long)gridDim.x * 256;
    long p = (long)blockIdx.x * 256 + threadIdx.x;
    for (; p + 3 * step < count; p += 4 * step) {            // four pairs requested at once
        const float2 v0 = part[(long)c * count + p], v1 = part[(long)c * count + p + step];
        const float2 v2 = part[(long)c * count + p + 2 * step], v3 = part[(long)c * count + p + 3 * step];
        s += (double)v0.x; ss += (double)v0.y; s += (double)v1.x; ss += (double)v1.y;
        s += (double)v2.x; ss += (double)v2.y; s += (double)v3.x; ss += (double)v3.y;
    }
    for (; p < count; p += step) {
        const float2 v = part[(long)c * count + p];
        s += (double)v.x; ss += (double)v.y;
    }
    tbn_block_store(s, ss, scratch + (long)c * TBN_MAXCHUNK * 2);
}

// stats[c] = {mean, invstd}
template <int V, bool FLAT = false>
__global__ void __launch_bounds__(256) tbn_apply_kernel(const float* __restrict__ x, const double* __restrict__ scratch, int nchunk,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ y,
                                                        int N, int C, long HW, int relu, float eps, float momentum, float* __restrict__ stats,
                                                        float* __restrict__ running_mean, float* __restrict__ running_var,
                                                        const float* __restrict__ res)
{
    typedef typename tbn_vec<V>::type vec;
    const int c = blockIdx.y;
    double s, ss;
    tbn_block_total(scratch + (long)c * TBN_MAXCHUNK * 2, nchunk, s, ss);
    const double P = (double)N * (double)HW, mean = s / P;
    double var = ss / P - mean * mean;
    if (var < 0) var = 0;
    const float fm = (float)mean, fi = (float)(1.0 / sqrt(var + (double)eps));
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        stats[2 * c] = fm;
        stats[2 * c + 1] = fi;
        if (running_mean) {
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * fm;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(var * P / (P > 1 ? P - 1 : 1));
        }
    }
    const float g = gamma[c], b = beta[c];
    const TbnMap<V, FLAT> map(N, HW);
    const long U = map.units(N);
    for (long u = blockIdx.x; u < U; u += gridDim.x) {
        long n, i;
        if (!map.at(u, HW, n, i)) continue;
        const long idx = (n * C + c) * HW + i;
        const vec xv = *reinterpret_cast<const vec*>(x + idx);
        vec o, rv = xv;
        if (res) rv = *reinterpret_cast<const vec*>(res + idx);   // out += residual (BasicResBlock), fused
#pragma unroll
        for (int j = 0; j < V; ++j) {
            float v = tbn_affine(tbn_at<V>(xv, j), fm, fi, g, b);
            if (relu) v = fmaxf(v, 0.f);
            ((float*)&o)[j] = res ? v + tbn_at<V>(rv, j) : v;
        }
        *reinterpret_cast<vec*>(y + idx) = o;
    }
}

// backward: dy_eff = dy * (y > 0) with ReLU; {sum dy_eff, sum dy_eff * xhat} = (dbeta, dgamma).  The mask is recomputed from z
// (tbn_affine, bit-identical to the forward's value): one tensor less to read in each of the two backward passes.
template <int V, bool FLAT = false>
__global__ void __launch_bounds__(256) tbn_bwd_reduce_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ stats,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta, int N, int C, long HW,
                                                             int relu, double* __restrict__ scratch)
{
    typedef typename tbn_vec<V>::type vec;
    const int c = blockIdx.y, nchunk = gridDim.x;
    const TbnMap<V, FLAT> map(N, HW);
    const long U = map.units(N);
    const float mean = stats[2 * c], invstd = stats[2 * c + 1], gm = gamma[c], bt = beta[c];
    double s = 0, sx = 0;
    for (long u = blockIdx.x; u < U; u += nchunk) {
        long n, i;
        if (map.at(u, HW, n, i)) {
            const long idx = (n * C + c) * HW + i;
            const vec gv = *reinterpret_cast<const vec*>(dy + idx), xv = *reinterpret_cast<const vec*>(x + idx);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                float g = tbn_at<V>(gv, j);
                const float xe = tbn_at<V>(xv, j);
                if (relu && !(tbn_affine(xe, mean, invstd, gm, bt) > 0.f)) g = 0.f;
                s += g; sx += (double)g * (double)((xe - mean) * invstd);
            }
        }
    }
    tbn_block_store(s, sx, scratch + (long)c * TBN_MAXCHUNK * 2);
}

// dx = gamma * invstd * (dy_eff - (dbeta + xhat * dgamma) / P)
template <int V, bool FLAT = false>
__global__ void __launch_bounds__(256) tbn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ stats,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const double* __restrict__ scratch, int nchunk, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, float* __restrict__ dx, int N, int C, long HW, int relu)
{
    typedef typename tbn_vec<V>::type vec;
    const int c = blockIdx.y;
    double s, sx;
    tbn_block_total(scratch + (long)c * TBN_MAXCHUNK * 2, nchunk, s, sx);
    const float db = (float)s, dg = (float)sx;
    if (blockIdx.x == 0 && threadIdx.x == 0) { dbeta[c] = db; dgamma[c] = dg; }
    const float fm = stats[2 * c], fi = stats[2 * c + 1], gm = gamma[c], bt = beta[c], gi = gm * fi, invP = 1.f / (float)((long)N * HW);
    const TbnMap<V, FLAT> map(N, HW);
    const long U = map.units(N);
    for (long u = blockIdx.x; u < U; u += gridDim.x) {
        long n, i;
        if (!map.at(u, HW, n, i)) continue;
        const long idx = (n * C + c) * HW + i;
        const vec gv = *reinterpret_cast<const vec*>(dy + idx), xv = *reinterpret_cast<const vec*>(x + idx);
        vec o;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            float g = tbn_at<V>(gv, j);
            const float xe = tbn_at<V>(xv, j);
            if (relu && !(tbn_affine(xe, fm, fi, gm, bt) > 0.f)) g = 0.f;
            const float xhat = (xe - fm) * fi;
            ((float*)&o)[j] = gi * (g - (db + xhat * dg) * invP);
        }
        *reinterpret_cast<vec*>(dx + idx) = o;
    }
}

// ---- small maps (N*H*W <= TBN_SMALL per channel; measured break-even ~10 k): statistics and the elementwise pass in ONE launch, one 1024-thread workgroup per
// channel -- at the reference's batch 16 the strides 16 and 32 (half of the layers), where a launch costs more than its work ----
#define TBN_SMALL 8192
__device__ __forceinline__ void tbn_block_total1024(double& s, double& t)
{
    __shared__ double r1[16], r2[16], tot[2];
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_down(s, o); t += __shfl_down(t, o); }
    if ((threadIdx.x & 63) == 0) { r1[threadIdx.x >> 6] = s; r2[threadIdx.x >> 6] = t; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0, b = 0;
        for (int i = 0; i < 16; ++i) { a += r1[i]; b += r2[i]; }
        tot[0] = a; tot[1] = b;
    }
    __syncthreads();
    s = tot[0]; t = tot[1];
}
__global__ void __launch_bounds__(1024) tbn_fwd_small_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ y, int N, int C, int HW, int relu, float eps, float momentum,
                                                             float* __restrict__ stats, float* __restrict__ running_mean,
                                                             float* __restrict__ running_var, const float* __restrict__ res)
{
    const int c = blockIdx.x, P = N * HW;
    double s = 0, ss = 0;
    for (int p = threadIdx.x; p < P; p += 1024) {
        const int n = p / HW, i = p - n * HW;
        const double v = x[((long)n * C + c) * HW + i];
        s += v; ss += v * v;
    }
    tbn_block_total1024(s, ss);
    const double Pd = (double)P, mean = s / Pd;
    double var = ss / Pd - mean * mean;
    if (var < 0) var = 0;
    const float fm = (float)mean, fi = (float)(1.0 / sqrt(var + (double)eps));
    if (threadIdx.x == 0) {
        stats[2 * c] = fm;
        stats[2 * c + 1] = fi;
        if (running_mean) {
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * fm;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(var * Pd / (Pd > 1 ? Pd - 1 : 1));
        }
    }
    const float g = gamma[c], b = beta[c];
    for (int p = threadIdx.x; p < P; p += 1024) {
        const int n = p / HW, i = p - n * HW;
        const long idx = ((long)n * C + c) * HW + i;
        float v = tbn_affine(x[idx], fm, fi, g, b);
        if (relu) v = fmaxf(v, 0.f);
        y[idx] = res ? v + res[idx] : v;
    }
}
__global__ void __launch_bounds__(1024) tbn_bwd_small_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ stats,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dx, int N,
                                                             int C, int HW, int relu)
{
    const int c = blockIdx.x, P = N * HW;
    const float fm = stats[2 * c], fi = stats[2 * c + 1], gm = gamma[c], bt = beta[c];
    double s = 0, sx = 0;
    for (int p = threadIdx.x; p < P; p += 1024) {
        const int n = p / HW, i = p - n * HW;
        const long idx = ((long)n * C + c) * HW + i;
        float g = dy[idx];
        const float xe = x[idx];
        if (relu && !(tbn_affine(xe, fm, fi, gm, bt) > 0.f)) g = 0.f;
        s += g; sx += (double)g * (double)((xe - fm) * fi);
    }
    tbn_block_total1024(s, sx);
    const float db = (float)s, dg = (float)sx, gi = gm * fi, invP = 1.f / (float)P;
    if (threadIdx.x == 0) { dbeta[c] = db; dgamma[c] = dg; }
    for (int p = threadIdx.x; p < P; p += 1024) {
        const int n = p / HW, i = p - n * HW;
        const long idx = ((long)n * C + c) * HW + i;
        float g = dy[idx];
        const float xe = x[idx];
        if (relu && !(tbn_affine(xe, fm, fi, gm, bt) > 0.f)) g = 0.f;
        dx[idx] = gi * (g - (db + (xe - fm) * fi * dg) * invP);
    }
}

// The same for H*W % 4 == 0 and up to 4096 UPT elements per channel: a thread keeps its UPT float4 in registers between the statistics
// and the elementwise pass, so z (and dy) are read ONCE -- at the reference's batch 16 this also takes the stride-8 layers (20480
// elements per channel) from two launches each way to one.
template <int UPT>
__global__ void __launch_bounds__(1024) tbn_fwd_small4_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float* __restrict__ y, int N, int C, int HW, int relu, float eps, float momentum,
                                                              float* __restrict__ stats, float* __restrict__ running_mean,
                                                              float* __restrict__ running_var, const float* __restrict__ res)
{
    const int c = blockIdx.x, hw4 = HW / 4, P4 = N * hw4;
    float4 v[UPT];
    long idx[UPT];
    double s = 0, ss = 0;
#pragma unroll
    for (int j = 0; j < UPT; ++j) {
        const int f = threadIdx.x + j * 1024;
        const int fc = f < P4 ? f : P4 - 1, n = fc / hw4;
        idx[j] = ((long)n * C + c) * HW + (long)(fc - n * hw4) * 4;
        v[j] = *reinterpret_cast<const float4*>(x + idx[j]);
        if (f < P4) {
            const double a = v[j].x, b = v[j].y, d = v[j].z, e = v[j].w;
            s += (a + b) + (d + e); ss += (a * a + b * b) + (d * d + e * e);
        }
    }
    tbn_block_total1024(s, ss);
    const double Pd = (double)N * (double)HW, mean = s / Pd;
    double var = ss / Pd - mean * mean;
    if (var < 0) var = 0;
    const float fm = (float)mean, fi = (float)(1.0 / sqrt(var + (double)eps));
    if (threadIdx.x == 0) {
        stats[2 * c] = fm;
        stats[2 * c + 1] = fi;
        if (running_mean) {
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * fm;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(var * Pd / (Pd > 1 ? Pd - 1 : 1));
        }
    }
    const float g = gamma[c], b = beta[c];
    float4 rr[UPT];                                          // the residuals requested together (idx is clamped: always a valid address)
#pragma unroll
    for (int j = 0; j < UPT; ++j) rr[j] = res ? *reinterpret_cast<const float4*>(res + idx[j]) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < UPT; ++j) {
        if (threadIdx.x + j * 1024 >= P4) continue;
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[e] = tbn_affine(((const float*)&v[j])[e], fm, fi, g, b);
            if (relu) o[e] = fmaxf(o[e], 0.f);
        }
        if (res) { o[0] += rr[j].x; o[1] += rr[j].y; o[2] += rr[j].z; o[3] += rr[j].w; }
        *reinterpret_cast<float4*>(y + idx[j]) = make_float4(o[0], o[1], o[2], o[3]);
    }
}
template <int UPT>
__global__ void __launch_bounds__(1024) tbn_bwd_small4_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ stats,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dx, int N,
                                                              int C, int HW, int relu)
{
    const int c = blockIdx.x, hw4 = HW / 4, P4 = N * hw4;
    const float fm = stats[2 * c], fi = stats[2 * c + 1], gm = gamma[c], bt = beta[c];
    float4 xv[UPT], gv[UPT];
    long idx[UPT];
    double s = 0, sx = 0;
#pragma unroll
    for (int j = 0; j < UPT; ++j) {
        const int f = threadIdx.x + j * 1024;
        const int fc = f < P4 ? f : P4 - 1, n = fc / hw4;
        idx[j] = ((long)n * C + c) * HW + (long)(fc - n * hw4) * 4;
        xv[j] = *reinterpret_cast<const float4*>(x + idx[j]);
        gv[j] = *reinterpret_cast<const float4*>(dy + idx[j]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float& g = ((float*)&gv[j])[e];
            const float xe = ((const float*)&xv[j])[e];
            if (relu && !(tbn_affine(xe, fm, fi, gm, bt) > 0.f)) g = 0.f;
            if (f < P4) { s += g; sx += (double)g * (double)((xe - fm) * fi); }
        }
    }
    tbn_block_total1024(s, sx);
    const float db = (float)s, dg = (float)sx, gi = gm * fi, invP = 1.f / (float)((long)N * HW);
    if (threadIdx.x == 0) { dbeta[c] = db; dgamma[c] = dg; }
#pragma unroll
    for (int j = 0; j < UPT; ++j) {
        if (threadIdx.x + j * 1024 >= P4) continue;
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = gi * (((const float*)&gv[j])[e] - (db + (((const float*)&xv[j])[e] - fm) * fi * dg) * invP);
        *reinterpret_cast<float4*>(dx + idx[j]) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// per-channel sum over N, H, W (bias gradient of the head convs)
__global__ void __launch_bounds__(256) tchan_sum_kernel(const float* __restrict__ dy, int N, int C, long HW, float* __restrict__ out)
{
    __shared__ double r1[4];
    const int c = blockIdx.x;
    const long upn = (HW + 255) / 256, U = (long)N * upn;
    double s = 0;
    for (long u = 0; u < U; ++u) {
        const long n = u / upn, i = (u - n * upn) * 256 + threadIdx.x;
        if (i < HW) s += dy[(n * C + c) * HW + i];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0) r1[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[c] = (float)(r1[0] + r1[1] + r1[2] + r1[3]);
}

// the same sum split over chunks of the (frame, pixel) list: partial sums in double to the scratch, added in chunk order by a second launch
__global__ void __launch_bounds__(256) tchan_sum_part_kernel(const float* __restrict__ dy, int N, int C, long HW, double* __restrict__ part)
{
    __shared__ double r1[4];
    const int c = blockIdx.y, nchunk = gridDim.x;
    const long hw4 = HW / 4, U = (long)N * hw4;                       // float4 units (HW % 4 == 0)
    double s = 0;
    for (long u = (long)blockIdx.x * 256 + threadIdx.x; u < U; u += (long)nchunk * 256) {
        const long n = u / hw4, i = (u - n * hw4) * 4;
        const float4 v = *reinterpret_cast<const float4*>(dy + (n * C + c) * HW + i);
        s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0) r1[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[(long)c * nchunk + blockIdx.x] = r1[0] + r1[1] + r1[2] + r1[3];
}
__global__ void tchan_sum_final_kernel(const double* __restrict__ part, int nchunk, int C, float* __restrict__ out)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0;
    for (int k = 0; k < nchunk; ++k) s += part[(long)c * nchunk + k];
    out[c] = (float)s;
}

// out = a + b (residual add, gradient accumulation); out may alias a
__global__ void __launch_bounds__(256) tadd_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long total)
{
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx < total) out[idx] = a[idx] + b[idx];
}

// channel slices of NCHW tensors: dst[n, dc0 + c, :, :] = src[n, sc0 + c, :, :] for c < C (torch.cat over channels and its backward)
__global__ void __launch_bounds__(256) tslice_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int C, long HW, int Cs, int sc0,
                                                     int Cd, int dc0)
{
    const long idx = (long)blockIdx.x * 256 + threadIdx.x, total = (long)N * C * HW;
    if (idx >= total) return;
    const long i = idx % HW, c = (idx / HW) % C, n = idx / (HW * C);
    dst[(n * Cd + dc0 + c) * HW + i] = src[(n * Cs + sc0 + c) * HW + i];
}

// torch.optim.Adam (no weight decay, no amsgrad), in the operation order of torch's single-tensor implementation:
//   m += (1 - b1) (g - m);  v = v b2 + (1 - b2) g g;  p += -(lr / (1 - b1^t)) * (m / (sqrt(v) / sqrt(1 - b2^t) + eps))
// the scalars are formed in double on the host and rounded once to float, like torch's Python-double scalars.
__global__ void __launch_bounds__(256) tadam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                    long total, float w1, float b2, float w2, float eps, float step_size, float bc2_sqrt)
{
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const float gi = g[idx];
    const float mi = m[idx] + w1 * (gi - m[idx]);
    const float vi = v[idx] * b2 + (w2 * gi) * gi;
    m[idx] = mi; v[idx] = vi;
    p[idx] += -step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
}

// every parameter tensor in one launch: tab[t] = {p, g, m, v, first block, elements}
struct TAdamEntry { float* p; const float* g; float* m; float* v; long block0; long n; };
__global__ void __launch_bounds__(256) tadam_multi_kernel(const TAdamEntry* __restrict__ tab, int nt, float w1, float b2, float w2, float eps,
                                                          float step_size, float bc2_sqrt)
{
    int lo = 0, hi = nt - 1;                                   // the tensor this workgroup belongs to
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (tab[mid].block0 <= (long)blockIdx.x) lo = mid; else hi = mid - 1; }
    const TAdamEntry e = tab[lo];
    const long idx = ((long)blockIdx.x - e.block0) * 256 + threadIdx.x;
    if (idx >= e.n) return;
    const float gi = e.g[idx];
    const float mi = e.m[idx] + w1 * (gi - e.m[idx]);
    const float vi = e.v[idx] * b2 + (w2 * gi) * gi;
    e.m[idx] = mi; e.v[idx] = vi;
    e.p[idx] += -step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
}

static inline unsigned nblk(long total) { return (unsigned)((total + 255) / 256); }

static void launch_tpw_gemm(const float* x, const float* a, const float* bias, const float* addend, float* y, long Q, long HW, int M, int K, long sm,
                            long sk, hipStream_t s, TStatPart* st = nullptr)
{
    if (st) st->count = 0;
    const int wm = tpw_waves_m(M), wq = 4 / wm;
    const unsigned my = (unsigned)((M + 16 * wm - 1) / (16 * wm));
    const long b4 = (Q + 64L * wq - 1) / (64L * wq), b1 = (Q + 16L * wq - 1) / (16L * wq);     // workgroups along the pixels, NT = 4 / 1
    static const bool old_only = getenv("YF_TPW_OLD") != nullptr;
    const int tiles = (M + 15) / 16;
    int mgroups = (tiles + 3) / 4, mt = (tiles + mgroups - 1) / mgroups;
    while (mt > 1 && (Q + 63) / 64 * mgroups < 512) {                  // few pixels: channel tiles on separate waves (the B re-reads hit L2)
        mt = (mt + 1) / 2;
        mgroups = (tiles + mt - 1) / mt;
    }
    const long wg = (Q + 255) / 256 * mgroups;
    // big A operand and enough pixel blocks per workgroup to pay for staging it: the weight-stationary form
    static const bool lds_off = getenv("YF_TPW_LDS_OFF") != nullptr;
    const bool want_stat = st && !bias && !addend && mt <= 2 && HW % 4 == 0 && K % 4 == 0 && Q % 4 == 0 && !old_only;
    if (!want_stat && !old_only && !lds_off && HW % 4 == 0 && K % 4 == 0 && Q % 4 == 0 && (long)M * K >= 8192 && Q >= 32768) {
        int rc = -1;
        if (mt == 1) rc = launch_tpw4_lds<1, false>(x, a, bias, addend, y, Q, HW, M, K, sm, sk, mgroups, 0, s);
        else if (mt == 2) rc = launch_tpw4_lds<2, false>(x, a, bias, addend, y, Q, HW, M, K, sm, sk, mgroups, 0, s);
        else if (mt == 3) rc = launch_tpw4_lds<3, false>(x, a, bias, addend, y, Q, HW, M, K, sm, sk, mgroups, 0, s);
        else rc = launch_tpw4_lds<4, false>(x, a, bias, addend, y, Q, HW, M, K, sm, sk, mgroups, 0, s);
        if (rc == 0) return;
    }
    if (!old_only && HW % 4 == 0 && K % 4 == 0 && Q % 4 == 0) {
        float2* sp = (want_stat && tstat_room(st, (Q + 255) / 256 * 4, M)) ? st->part : nullptr;
#define YF_PW4(MT_) hipLaunchKernelGGL(tpw4_mfma_kernel<MT_>, dim3((unsigned)wg), dim3(256), 0, s, x, a, bias, addend, y, Q, HW, M, K, sm, sk, mgroups, 0, sp)
        if (mt == 1) YF_PW4(1); else if (mt == 2) YF_PW4(2); else if (mt == 3) YF_PW4(3); else YF_PW4(4);
#undef YF_PW4
        return;
    }
    if (b4 * my >= 512)
        hipLaunchKernelGGL(tpw_mfma_kernel<4>, dim3((unsigned)(b4 * my)), dim3(256), 0, s, x, a, bias, addend, y, Q, HW, M, K, sm, sk);
    else
        hipLaunchKernelGGL(tpw_mfma_kernel<1>, dim3((unsigned)(b1 * my)), dim3(256), 0, s, x, a, bias, addend, y, Q, HW, M, K, sm, sk);
}
void launch_tconv_fwd(const float* x, const float* w, const float* bias, float* y, int N, int Cin, int H, int W, int Cout, int k, int stride,
                      int depthwise, hipStream_t s, TStatPart* st)
{
    const int pad = (k - 1) / 2, Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    if (st) st->count = 0;
    if (st && (long)N * Ho * Wo <= 4096 * 8) st = nullptr;              // BatchNorm's one-launch kernels read z once anyway
    if (!depthwise && k == 1 && stride == 1) {
        launch_tpw_gemm(x, w, bias, nullptr, y, (long)N * H * W, (long)H * W, Cout, Cin, (long)Cin, 1L, s, st);
        return;
    }
    static const bool s2_off = getenv("YF_TCONV3S2_OFF") != nullptr;
    if (!depthwise && k == 3 && stride == 2 && !s2_off && H % 2 == 0 && W % 8 == 0) {
        const long G = (long)N * Ho * (Wo / 4);
        if (Cin == 1 && Cout <= 8) {
            hipLaunchKernelGGL(tconv3s2_c1_kernel<8>, dim3(nblk(G)), dim3(256), 0, s, x, w, bias, y, N, H, W, Cout);
            return;
        }
        if (Cin % 4 == 0 && Cout <= 32) {
            const dim3 grid((unsigned)((G + 63) / 64));
            if (Cout <= 16) hipLaunchKernelGGL(tconv3s2_mfma_kernel<1>, grid, dim3(256), 0, s, x, w, bias, y, N, Cin, H, W, Cout);
            else hipLaunchKernelGGL(tconv3s2_mfma_kernel<2>, grid, dim3(256), 0, s, x, w, bias, y, N, Cin, H, W, Cout);
            return;
        }
    }
    if (!depthwise && k == 3) {
        hipLaunchKernelGGL(tconv_im2col_mfma_kernel<3>, dim3((unsigned)(((long)N * Ho * Wo + 63) / 64), (Cout + 63) / 64), dim3(256), 0, s, x, w, bias, y,
                           N, Cin, H, W, Ho, Wo, Cout, stride);
        return;
    }
    if (!depthwise) {
        const long Q = (long)N * Ho * Wo;
        if (Cout > 8)
            hipLaunchKernelGGL(tconv_mc_kernel<16>, dim3(nblk(Q), (Cout + 15) / 16), dim3(256), 0, s, x, w, bias, y, N, Cin, H, W, Cout, Ho, Wo, k,
                               stride, (long)Cin * k * k, (long)k * k);
        else
            hipLaunchKernelGGL(tconv_mc_kernel<8>, dim3(nblk(Q), (Cout + 7) / 8), dim3(256), 0, s, x, w, bias, y, N, Cin, H, W, Cout, Ho, Wo, k, stride,
                               (long)Cin * k * k, (long)k * k);
        return;
    }
    if (depthwise && !bias && stride == 1 && W % 4 != 0 && W <= 16 && (k == 3 || k == 5)) {
        const long nrows = (long)N * Cout * H;
        if (k == 3) hipLaunchKernelGGL((tdw_plane_kernel<3, false>), dim3(nblk(nrows)), dim3(256), 0, s, x, w, y, Cout, H, W, nrows);
        else hipLaunchKernelGGL((tdw_plane_kernel<5, false>), dim3(nblk(nrows)), dim3(256), 0, s, x, w, y, Cout, H, W, nrows);
        return;
    }
    if (depthwise && !bias && Wo % 4 == 0) {
        if (k == 3 && stride == 1) return launch_tdw_conv<3, 1, false>(x, w, y, N, Cout, H, W, Ho, Wo, s, st);
        if (k == 3 && stride == 2) return launch_tdw_conv<3, 2, false>(x, w, y, N, Cout, H, W, Ho, Wo, s);
        if (k == 5 && stride == 1) return launch_tdw_conv<5, 1, false>(x, w, y, N, Cout, H, W, Ho, Wo, s, st);
    }
    hipLaunchKernelGGL(tconv_fwd_kernel, dim3(nblk((long)N * Cout * Ho * Wo)), dim3(256), 0, s, x, w, bias, y, N, Cin, H, W, Cout, Ho, Wo, k, stride, depthwise);
}
// addend (optional, like dx): added to the result -- fused for the pointwise GEMM, a separate pass otherwise
void launch_tconv_bwd_data(const float* dy, const float* w, float* dx, int N, int Cin, int H, int W, int Cout, int k, int stride, int depthwise,
                           hipStream_t s, const float* addend, TBnRed* red)
{
    if (red) red->count = 0;
    if (red && (long)N * H * W <= 4096 * 8) red = nullptr;              // BatchNorm's one-launch backward reads dy and z once anyway
    if (addend && !(!depthwise && k == 1 && stride == 1)) {
        launch_tconv_bwd_data(dy, w, dx, N, Cin, H, W, Cout, k, stride, depthwise, s, nullptr, nullptr);
        launch_tadd(dx, addend, dx, (long)N * Cin * H * W, s);
        return;
    }
    const int pad = (k - 1) / 2, Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    if (!depthwise && k == 1 && stride == 1) {       // pointwise: dx[ci] = sum_co dy[co] w[co][ci] -- the same GEMM with the weight transposed
        // (the same epilogue in the pointwise GEMM: a z load, 4 constants and ~45 instructions for each of a lane's 4 MT rows cost what
        //  the reduction pass saved -- +60 / -58 us on the 8-channel layers at batch 256; only the depthwise kernel carries it)
        launch_tpw_gemm(dy, w, nullptr, addend, dx, (long)N * H * W, (long)H * W, Cin, Cout, 1L, (long)Cin, s);
        return;
    }
    static const bool s2m_off = getenv("YF_TCONV3S2_OFF") != nullptr;
    if (!depthwise && k == 3 && stride == 2 && H == 2 * Ho && W == 2 * Wo && !s2m_off && Cout % 4 == 0 && Wo % 4 == 0 && Cin <= 32 &&
        (long)N * Ho * (Wo / 4) >= 2048) {
        const dim3 grid((unsigned)(((long)N * Ho * (Wo / 4) + 63) / 64));
        if (Cin <= 16) hipLaunchKernelGGL(tconv3s2_bwd_mfma_kernel<1>, grid, dim3(256), 0, s, dy, w, dx, N, Cin, Cout, Ho, Wo);
        else hipLaunchKernelGGL(tconv3s2_bwd_mfma_kernel<2>, grid, dim3(256), 0, s, dy, w, dx, N, Cin, Cout, Ho, Wo);
        return;
    }
    if (!depthwise && k == 3 && stride == 2 && H == 2 * Ho && W == 2 * Wo) {
        hipLaunchKernelGGL(tconv3s2_bwd_data_kernel<8>, dim3(nblk((long)N * Ho * Wo), (Cin + 7) / 8), dim3(256), 0, s, dy, w, dx, N, Cin, Cout, Ho, Wo);
        return;
    }
    if (depthwise) {
        if (stride == 1 && W % 4 != 0 && W <= 16 && (k == 3 || k == 5)) {
            const long nrows = (long)N * Cin * H;
            if (k == 3) hipLaunchKernelGGL((tdw_plane_kernel<3, true>), dim3(nblk(nrows)), dim3(256), 0, s, dy, w, dx, Cin, H, W, nrows);
            else hipLaunchKernelGGL((tdw_plane_kernel<5, true>), dim3(nblk(nrows)), dim3(256), 0, s, dy, w, dx, Cin, H, W, nrows);
            return;
        }
        if (stride == 1 && W % 4 == 0 && k == 3) return launch_tdw_conv<3, 1, true>(dy, w, dx, N, Cin, H, W, H, W, s, nullptr, red);
        if (stride == 1 && W % 4 == 0 && k == 5) return launch_tdw_conv<5, 1, true>(dy, w, dx, N, Cin, H, W, H, W, s, nullptr, red);
        if (stride == 2 && k == 3 && H == 2 * Ho && W == 2 * Wo) {
            const int threads = Ho * Wo, bs = threads <= 64 ? 64 : 256;
            hipLaunchKernelGGL(tdw3s2_bwd_data_kernel, dim3(N * Cin, (threads + bs - 1) / bs), dim3(bs), 0, s, dy, w, dx, Cin, Ho, Wo);
            return;
        }
    }
    hipLaunchKernelGGL(tconv_bwd_data_kernel, dim3(nblk((long)N * Cin * H * W)), dim3(256), 0, s, dy, w, dx, N, Cin, H, W, Cout, Ho, Wo, k, stride, depthwise);
}
// waves per workgroup of the weight-gradient GEMM: fill the chip (>= 16 waves per CU) when the slab limit keeps the grid small and a
// slice is long enough to deal out
static inline int twgrad_waves(long workgroups, long q_per)
{
    static const int forced = getenv("YF_WGRAD_NW") ? atoi(getenv("YF_WGRAD_NW")) : 0;
    if (forced == 1 || forced == 4 || forced == 8) return forced;
    if (workgroups * 8 <= 8192 && q_per >= 8 * 64) return 8;
    if (workgroups * 4 <= 8192 && q_per >= 4 * 64) return 4;
    return 1;
}
// the slabs of the split reductions: nsplit <= what fits into the scratch
static inline void tsum_partials(const float* part, long nsplit, long nw, float* dw, hipStream_t s)
{
    int spl = 1;
    while (spl < 64 && spl * 8 <= nsplit) spl *= 2;                 // >= 8 slabs per lane
    hipLaunchKernelGGL(tsum_partials_kernel, dim3(nblk(nw * spl)), dim3(256), 0, s, part, (int)nsplit, nw, nw, dw, spl);
}
// the multi-tensor form: every layer's slabs stay alive in the defer region until ONE launch at the end of the pass adds them all
// (tsum_multi_kernel: the same lanes, order and arithmetic per output as tsum_partials_kernel -- bitwise the same sums)
__global__ void __launch_bounds__(256) tsum_multi_kernel(const TSumEntry* __restrict__ tab, int n, const float* __restrict__ slab, float* __restrict__ dst)
{
    // this block's entry = the last one that starts at or before it: one parallel probe of the (<= 256) entries, not a dependent scan
    const int e = __syncthreads_count((int)threadIdx.x < n && tab[threadIdx.x].blk0 <= (long)blockIdx.x) - 1;
    const TSumEntry E = tab[e];
    const float* part = slab + E.part_off;
    const long t = ((long)blockIdx.x - E.blk0) * 256 + threadIdx.x;
    const long i = t / E.spl;
    const int j = (int)(t - i * E.spl);
    float v = 0.f;
    if (i < E.nw) {
        int sidx = j;
        for (; sidx + 3 * E.spl < E.nsplit; sidx += 4 * E.spl) {          // four slabs requested at once, added in slab order
            const float p0 = part[(long)sidx * E.nw + i], p1 = part[(long)(sidx + E.spl) * E.nw + i];
            const float p2 = part[(long)(sidx + 2 * E.spl) * E.nw + i], p3 = part[(long)(sidx + 3 * E.spl) * E.nw + i];
            v += p0; v += p1; v += p2; v += p3;
        }
        for (; sidx < E.nsplit; sidx += E.spl) v += part[(long)sidx * E.nw + i];
    }
    for (int o = E.spl >> 1; o > 0; o >>= 1) v += __shfl_down(v, o);
    if (i < E.nw && j == 0) dst[E.dst_off + i] = v;
}
float* TSumDefer::take(long floats)
{
    const size_t need = ((size_t)floats + 63) & ~(size_t)63;
    if (!slab || used + need > cap_floats) return nullptr;
    float* p = slab + used;
    used += need;
    return p;
}
void TSumDefer::push(const float* part, long nsplit, long nw, float* dw)
{
    TSumEntry E;
    E.part_off = part - slab; E.dst_off = dw - dst_base; E.nw = nw; E.nsplit = (int)nsplit;
    int spl = 1;
    while (spl < 64 && spl * 8 <= nsplit) spl *= 2;
    E.spl = spl;
    E.blk0 = nblocks;
    nblocks += (nw * spl + 255) / 256;
    entries.push_back(E);
}
void launch_tsum_multi(const TSumEntry* d_tab, int n, long nblocks, const float* slab, float* dst, hipStream_t s)
{
    if (n > 0) hipLaunchKernelGGL(tsum_multi_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, d_tab, n, slab, dst);
}
// where a split reduction writes: the gradient itself without a split, a slab of the defer region, or the shared scratch
static inline float* tsum_out(void* scratch, long nsplit, long nw, float* dw, TSumDefer* d)
{
    if (nsplit <= 1) return dw;
    if (d) {
        float* p = d->take(nsplit * nw);
        if (p) return p;
    }
    return (float*)scratch;
}
static inline void tsum_finish(const float* out, void* scratch, long nsplit, long nw, float* dw, hipStream_t s, TSumDefer* d)
{
    if (nsplit <= 1) return;
    if (d && out != (const float*)scratch) d->push(out, nsplit, nw, dw);
    else tsum_partials(out, nsplit, nw, dw, s);
}
void launch_tconv_bwd_weight(const float* x, const float* dy, float* dw, int N, int Cin, int H, int W, int Cout, int k, int stride, int depthwise,
                             void* scratch, size_t scratch_bytes, hipStream_t s, TSumDefer* defer)
{
    const int pad = (k - 1) / 2, Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    const long nw = (long)Cout * (depthwise ? 1 : Cin) * k * k, P = (long)N * Ho * Wo;
    const long fit = scratch ? (long)(scratch_bytes / ((size_t)nw * sizeof(float))) : 0;      // slabs that fit
    static const bool s2_off = getenv("YF_TCONV3S2_OFF") != nullptr;
    if (!depthwise && k == 3 && stride == 2 && !s2_off && H % 2 == 0 && W % 8 == 0 && fit >= 1) {
        const long G = (long)N * Ho * (Wo / 4);
        if (Cin == 1 && Cout <= 8 && G >= 4096) {
            long chunks = G / 2048;                                     // >= 8 groups per thread
            if (chunks > 1024) chunks = 1024;
            if (chunks > fit) chunks = fit;
            float* out = tsum_out(scratch, chunks, nw, dw, defer);
            hipLaunchKernelGGL(tconv3s2_c1_wgrad_kernel<8>, dim3((unsigned)chunks), dim3(256), 0, s, x, dy, out, N, H, W, Cout, nw);
            tsum_finish(out, scratch, chunks, nw, dw, s, defer);
            return;
        }
        if (Cin % 4 == 0 && Cin <= 32 && Cout <= 32 && G >= 2048) {
            long nsplit = G / 32;                                       // >= 8 steps per wave
            if (nsplit > 1024) nsplit = 1024;
            if (nsplit > fit) nsplit = fit;
            const int n_cu = device_cu_count(current_device());        // equal slices, one workgroup each: a whole number per CU
            if (n_cu > 0 && nsplit > n_cu) nsplit -= nsplit % n_cu;
            long g_per = (G + nsplit - 1) / nsplit;
            g_per = (g_per + 3) / 4 * 4;
            nsplit = (G + g_per - 1) / g_per;
            float* out = tsum_out(scratch, nsplit, nw, dw, defer);
            hipLaunchKernelGGL(tconv3s2_wgrad_mfma_kernel, dim3((unsigned)nsplit), dim3(256), 0, s, x, dy, out, N, Cin, H, W, Cout, g_per, nw);
            tsum_finish(out, scratch, nsplit, nw, dw, s, defer);
            return;
        }
    }
    // 4 consecutive output pixels per lane, in one frame
    if (!depthwise && (k == 3 || (k == 1 && stride == 1)) && ((long)Ho * Wo) % 4 == 0) {
        const int R = Cin * k * k, tiles = ((Cout + 15) / 16) * ((R + 63) / 64);
        // many pixels: 4 waves per workgroup share a slice (added through LDS) rather than 4 slices -- the same waves in flight, a
        // quarter of the slabs to write and to add up afterwards
        static const bool forced_nw = getenv("YF_WGRAD_NW") != nullptr;
        const long pw = (!forced_nw && P >= 65536) ? 4 : 1;
        long nsplit = (P + 128 * pw - 1) / (128 * pw);                  // >= 8 MFMA steps per wave ...
        while (nsplit * tiles * pw > 8192 && nsplit > 1) nsplit = (nsplit + 1) / 2;   // ... and a bounded grid
        if (nsplit > 1024) nsplit = 1024;
        if (nsplit > fit) nsplit = fit < 1 ? 1 : fit;
        long q_per = (P + nsplit - 1) / nsplit;
        q_per = (q_per + 15) / 16 * 16;
        nsplit = (P + q_per - 1) / q_per;
        float* out = tsum_out(scratch, nsplit, nw, dw, defer);
        const dim3 grid((unsigned)(nsplit * tiles));
        int nwv = twgrad_waves(nsplit * tiles, q_per);
        if (pw == 4 && nwv < 4 && q_per >= 4 * 64) nwv = 4;
#define YF_WG(KS_, NW_) hipLaunchKernelGGL((tconv_wgrad_mfma_kernel<KS_, NW_>), grid, dim3(64 * NW_), 0, s, x, dy, out, N, Cin, H, W, Cout, Ho, Wo, stride, q_per, nw)
        if (k == 1) { if (nwv == 8) YF_WG(1, 8); else if (nwv == 4) YF_WG(1, 4); else YF_WG(1, 1); }
        else { if (nwv == 8) YF_WG(3, 8); else if (nwv == 4) YF_WG(3, 4); else YF_WG(3, 1); }
#undef YF_WG
        tsum_finish(out, scratch, nsplit, nw, dw, s, defer);
        return;
    }
    if (depthwise && (k == 3 || k == 5)) {
        long chunks = (P + 2047) / 2048;                                // ~8 pixels per thread
        while (chunks * Cout > 4096 && chunks > 1) chunks = (chunks + 1) / 2;
        const bool rows = !tdw_rows_off && stride == 1 && W % 4 == 0 && H % 4 == 0 && (long)N * (H / 4) * (W / 4) >= 2048;
        if (rows && (long)N * (H / 4) * (W / 4) / 512 < chunks) chunks = (long)N * (H / 4) * (W / 4) / 512;   // 16 outputs per thread and trip
        if (chunks > fit) chunks = fit < 1 ? 1 : fit;
        float* out = tsum_out(scratch, chunks, nw, dw, defer);
        const dim3 grid((unsigned)chunks, Cout);
        if (rows) {
            if (k == 3) hipLaunchKernelGGL((tdw_wgrad_rows_kernel<3, 4>), grid, dim3(256), 0, s, x, dy, out, N, Cout, H, W, nw);
            else hipLaunchKernelGGL((tdw_wgrad_rows_kernel<5, 4>), grid, dim3(256), 0, s, x, dy, out, N, Cout, H, W, nw);
            tsum_finish(out, scratch, chunks, nw, dw, s, defer);
            return;
        }
        if (stride == 1 && W % 4 != 0 && W <= 16) {
            if (k == 3) hipLaunchKernelGGL(tdw_plane_wgrad_kernel<3>, grid, dim3(256), 0, s, x, dy, out, N, Cout, H, W, nw);
            else hipLaunchKernelGGL(tdw_plane_wgrad_kernel<5>, grid, dim3(256), 0, s, x, dy, out, N, Cout, H, W, nw);
            tsum_finish(out, scratch, chunks, nw, dw, s, defer);
            return;
        }
        if (k == 3)
            hipLaunchKernelGGL(tdw_wgrad_kernel<3>, grid, dim3(256), 0, s, x, dy, out, N, Cout, H, W, Ho, Wo, stride, nw);
        else
            hipLaunchKernelGGL(tdw_wgrad_kernel<5>, grid, dim3(256), 0, s, x, dy, out, N, Cout, H, W, Ho, Wo, stride, nw);
        tsum_finish(out, scratch, chunks, nw, dw, s, defer);
        return;
    }
    (void)hipMemsetAsync(dw, 0, (size_t)nw * sizeof(float), s);        // the fallbacks below accumulate with atomics
    if (!depthwise) {
        const int R = Cin * k * k, tiles = ((R + 63) / 64) * ((Cout + 63) / 64);
        long nsplit = (P + 255) / 256;                                  // >= 8 staged tiles per workgroup ...
        while (nsplit * tiles > 2048 && nsplit > 1) nsplit = (nsplit + 1) / 2;   // ... and a bounded grid
        long p_per = (P + nsplit - 1) / nsplit;
        p_per = (p_per + 31) / 32 * 32;
        nsplit = (P + p_per - 1) / p_per;
        hipLaunchKernelGGL(tconv_bwd_weight_gemm_kernel, dim3((unsigned)nsplit, (R + 63) / 64, (Cout + 63) / 64), dim3(256), 0, s, x, dy, dw, N, Cin, H,
                           W, Cout, Ho, Wo, k, stride, p_per);
        return;
    }
    int nchunk = (int)((P + 4095) / 4096);                       // ~16 reduction elements per thread
    while ((long)nchunk * nw > 262144 && nchunk > 1) nchunk /= 2;   // bound the grid
    hipLaunchKernelGGL(tconv_bwd_weight_kernel, dim3((unsigned)(nw * nchunk)), dim3(256), 0, s, x, dy, dw, N, Cin, H, W, Cout, Ho, Wo, k, stride,
                       depthwise, nchunk);
}
// backward-data + weight gradient of a pointwise layer in one launch (tpw_bwd_dual_kernel); false: not applicable, launch them separately
bool launch_tpw_bwd_dual(const float* x, const float* dz, const float* w, float* dw, float* dx, const float* addend, int N, int Cin, int H, int W,
                         int Cout, void* scratch, size_t scratch_bytes, hipStream_t s, TSumDefer* defer)
{
    static const bool off = getenv("YF_TPW_DUAL_OFF") != nullptr;
    const long HW = (long)H * W, Q = (long)N * HW;
    // up to 2 M pixels per batch (batch 256: strides 4 and up; measured 18.11 -> 17.94 -> 17.80 ms for limits of 128 k / 400 k / 2 M, no
    // further gain without a limit): below that the two kernels mostly wait, and side by side they wait once
    static const long qmax = getenv("YF_TPW_DUAL_QMAX") ? atol(getenv("YF_TPW_DUAL_QMAX")) : 2000000;
    if (off || HW % 4 || Cout % 4 || Q > qmax || Q < 256) return false;
    const long nw = (long)Cout * Cin;
    const long fit = scratch ? (long)(scratch_bytes / ((size_t)nw * sizeof(float))) : 0;
    // the data gradient's grid (launch_tpw_gemm's choice without the weight-stationary variant)
    const int mtiles = (Cin + 15) / 16;
    int mgroups = (mtiles + 3) / 4, mt = (mtiles + mgroups - 1) / mgroups;
    while (mt > 1 && (Q + 63) / 64 * mgroups < 512) { mt = (mt + 1) / 2; mgroups = (mtiles + mt - 1) / mt; }
    const long nA = (Q + 255) / 256 * mgroups;
    // the weight gradient's: 4 waves per slice
    const int tiles = ((Cout + 15) / 16) * ((Cin + 63) / 64);
    long nsplit = (Q + 511) / 512;
    while (nsplit * tiles * 4 > 8192 && nsplit > 1) nsplit = (nsplit + 1) / 2;
    if (nsplit > 1024) nsplit = 1024;
    if (nsplit > fit) nsplit = fit < 1 ? 1 : fit;
    long q_per = (Q + nsplit - 1) / nsplit;
    q_per = (q_per + 15) / 16 * 16;
    nsplit = (Q + q_per - 1) / q_per;
    float* out = tsum_out(scratch, nsplit, nw, dw, defer);
    TPwBwdArgs a{dz, w, addend, dx, x, out, Q, HW, N, Cin, Cout, H, W, mgroups, (unsigned)nA, q_per, nw};
    const dim3 grid((unsigned)(nA + nsplit * tiles));
    if (mt == 1) hipLaunchKernelGGL(tpw_bwd_dual_kernel<1>, grid, dim3(256), 0, s, a);
    else if (mt == 2) hipLaunchKernelGGL(tpw_bwd_dual_kernel<2>, grid, dim3(256), 0, s, a);
    else if (mt == 3) hipLaunchKernelGGL(tpw_bwd_dual_kernel<3>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(tpw_bwd_dual_kernel<4>, grid, dim3(256), 0, s, a);
    tsum_finish(out, scratch, nsplit, nw, dw, s, defer);
    return true;
}
void launch_tdeconv_fwd(const float* x, const float* w, float* y, int N, int Cin, int H, int W, int Cout, hipStream_t s)
{
    static const bool off = getenv("YF_TDECONV_OLD") != nullptr;
    const long Q = (long)N * H * W, HW = (long)H * W;
    if (!off && Q >= 256 && HW % 4 == 0 && Cin % 4 == 0) {             // the GEMM form (tpw4_mfma_kernel<.., DECONV>): M = 4 Cout rows
        const int M = 4 * Cout, tiles = (M + 15) / 16;
        int mgroups = (tiles + 3) / 4, mt = (tiles + mgroups - 1) / mgroups;
        while (mt > 1 && (Q + 63) / 64 * mgroups < 512) { mt = (mt + 1) / 2; mgroups = (tiles + mt - 1) / mt; }
        const dim3 grid((unsigned)((Q + 255) / 256 * mgroups));
#define YF_DC4(MT_) hipLaunchKernelGGL((tpw4_mfma_kernel<MT_, true>), grid, dim3(256), 0, s, x, w, (const float*)nullptr, (const float*)nullptr, y, Q, HW, M, Cin, 1L, (long)M, mgroups, W)
        if (mt == 1) YF_DC4(1); else if (mt == 2) YF_DC4(2); else if (mt == 3) YF_DC4(3); else YF_DC4(4);
#undef YF_DC4
        return;
    }
    hipLaunchKernelGGL(tdeconv_fwd_kernel, dim3(nblk((long)N * Cout * 4 * H * W)), dim3(256), 0, s, x, w, y, N, Cin, H, W, Cout);
}
void launch_tdeconv_bwd_data(const float* dy, const float* w, float* dx, int N, int Cin, int H, int W, int Cout, hipStream_t s)
{
    static const bool off = getenv("YF_TDECONV_OLD") != nullptr;
    if ((long)N * H * W < (off ? 8192 : 256)) {      // (the old gather GEMM at 16 x 8 x 10 pixels: 134 us against 25 us for this one)
        hipLaunchKernelGGL(tdeconv_bwd_data_kernel, dim3(nblk((long)N * Cin * H * W)), dim3(256), 0, s, dy, w, dx, N, Cin, H, W, Cout);
        return;
    }
    if (!off) {
        const long Q = (long)N * H * W;
        const int tiles = (Cin + 15) / 16;
        int mgroups = (tiles + 2) / 3, mt = (tiles + mgroups - 1) / mgroups;
        while (mt > 1 && (Q + 63) / 64 * mgroups < 512) { mt = (mt + 1) / 2; mgroups = (tiles + mt - 1) / mt; }
        const dim3 grid((unsigned)((Q + 255) / 256 * mgroups));
#define YF_DB(MT_) hipLaunchKernelGGL(tdeconv_bwd_mfma_kernel<MT_>, grid, dim3(256), 0, s, dy, w, dx, Q, H, W, Cin, Cout, mgroups)
        if (mt == 1) YF_DB(1); else if (mt == 2) YF_DB(2); else YF_DB(3);
#undef YF_DB
        return;
    }
    // dx[ci][p] = sum over (co, a, b) of dY[co][2 iy + a][2 ix + b] w[ci][co][a][b]: a 2x2 stride-2 pad-0 convolution of dY with the weight
    // read as [Cin][(co, a, b)] -- the im2col GEMM
    hipLaunchKernelGGL(tconv_im2col_mfma_kernel<2>, dim3((unsigned)(((long)N * H * W + 63) / 64), (Cin + 63) / 64), dim3(256), 0, s, dy, w,
                       (const float*)nullptr, dx, N, Cout, 2 * H, 2 * W, H, W, Cin, 2);
}
void launch_tdeconv_bwd_weight(const float* x, const float* dy, float* dw, int N, int Cin, int H, int W, int Cout, void* scratch, size_t scratch_bytes,
                               hipStream_t s, TSumDefer* defer)
{
    const long nw = (long)Cin * Cout * 4, P = (long)N * H * W;
    if (((long)H * W) % 4 == 0) {                 // the conv weight-gradient GEMM with the operands swapped (see tconv_wgrad_mfma_kernel)
        const long fit = scratch ? (long)(scratch_bytes / ((size_t)nw * sizeof(float))) : 0;
        const int R = Cout * 4, tiles = ((Cin + 15) / 16) * ((R + 63) / 64);
        long nsplit = (P + 127) / 128;
        while (nsplit * tiles > 8192 && nsplit > 1) nsplit = (nsplit + 1) / 2;
        if (nsplit > 1024) nsplit = 1024;
        if (nsplit > fit) nsplit = fit < 1 ? 1 : fit;
        long q_per = (P + nsplit - 1) / nsplit;
        q_per = (q_per + 15) / 16 * 16;
        nsplit = (P + q_per - 1) / q_per;
        float* out = tsum_out(scratch, nsplit, nw, dw, defer);
        const int nwv = twgrad_waves(nsplit * tiles, q_per);
        const dim3 grid((unsigned)(nsplit * tiles));
        if (nwv == 8) hipLaunchKernelGGL((tconv_wgrad_mfma_kernel<2, 8>), grid, dim3(512), 0, s, dy, x, out, N, Cout, 2 * H, 2 * W, Cin, H, W, 2, q_per, nw);
        else if (nwv == 4) hipLaunchKernelGGL((tconv_wgrad_mfma_kernel<2, 4>), grid, dim3(256), 0, s, dy, x, out, N, Cout, 2 * H, 2 * W, Cin, H, W, 2, q_per, nw);
        else hipLaunchKernelGGL((tconv_wgrad_mfma_kernel<2, 1>), grid, dim3(64), 0, s, dy, x, out, N, Cout, 2 * H, 2 * W, Cin, H, W, 2, q_per, nw);
        tsum_finish(out, scratch, nsplit, nw, dw, s, defer);
        return;
    }
    int nchunk = (int)((P + 4095) / 4096);
    while ((long)nchunk * nw > 262144 && nchunk > 1) nchunk /= 2;
    (void)hipMemsetAsync(dw, 0, (size_t)nw * sizeof(float), s);
    hipLaunchKernelGGL(tdeconv_bwd_weight_kernel, dim3((unsigned)(nw * nchunk)), dim3(256), 0, s, x, dy, dw, N, Cin, H, W, Cout, nchunk);
}
// one scratch for the split reductions of a stream: BatchNorm partial pairs (256 KB) or weight-gradient slabs (all of it)
size_t train_scratch_bytes() { return (size_t)32 << 20; }
static inline unsigned tbn_apply_blocks(int N, int C, long HW, int V, bool flat)
{
    const long U = tbn_units(N, HW, V, flat);
    long n = 8192 / C;
    if (n > U) n = U;
    return n < 1 ? 1u : (unsigned)n;
}
static inline int tbn_chunks(int N, int C, long HW, int V, bool flat)
{
    const long U = tbn_units(N, HW, V, flat);
    long n = 8192 / C;
    if (n > TBN_MAXCHUNK) n = TBN_MAXCHUNK;
    if (n > U) n = U;
    return n < 1 ? 1 : (int)n;
}
// scratch: >= 1 MB of device memory (partial sums; needs no initialisation); C <= 256
// residual (optional, like y): y = bn(x) [relu] + residual
void launch_tbn_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var, float* stats, float* y, int N,
                    int C, long HW, int relu, void* scratch, hipStream_t s, const float* residual, const TStatPart* st)
{
    static const bool small4_off = getenv("YF_TBN_SMALL4_OFF") != nullptr;
    if (!small4_off && HW % 4 == 0 && (long)N * HW <= 4096 * 8) {
        const long P4 = (long)N * HW / 4;
#define YF_BNF(U_) hipLaunchKernelGGL(tbn_fwd_small4_kernel<U_>, dim3(C), dim3(1024), 0, s, x, gamma, beta, y, N, C, (int)HW, relu, 1e-5f, 0.1f, stats, running_mean, running_var, residual)
        if (P4 <= 1024) YF_BNF(1); else if (P4 <= 2048) YF_BNF(2); else if (P4 <= 4096) YF_BNF(4); else YF_BNF(8);
#undef YF_BNF
        return;
    }
    if ((long)N * HW <= TBN_SMALL) {
        hipLaunchKernelGGL(tbn_fwd_small_kernel, dim3(C), dim3(1024), 0, s, x, gamma, beta, y, N, C, (int)HW, relu, 1e-5f, 0.1f, stats, running_mean,
                           running_var, residual);
        return;
    }
    static const bool flat_off = getenv("YF_TBN_FLAT_OFF") != nullptr;
    const bool flat = !flat_off && HW % 4 == 0 && HW % 1024 != 0;      // float4 numbered across the frames (small / ragged planes)
    const bool parts = st && st->count > 0;                             // the conv left the partial sums: no pass over z for them
    int pchunks = parts ? (int)((st->count + 1023) / 1024) : 0;        // >= 4 pairs per thread
    if (pchunks > 64) pchunks = 64;
    const int V = (flat || (HW % 4 == 0 && HW >= 1024)) ? 4 : 1, nchunk = parts ? pchunks : tbn_chunks(N, C, HW, V, flat);
    const dim3 g1(nchunk, C), g2(tbn_apply_blocks(N, C, HW, V, flat), C);
    if (parts) hipLaunchKernelGGL(tbn_stats_from_parts_kernel, dim3(pchunks, C), dim3(256), 0, s, (const float2*)st->part, st->count, (double*)scratch);
    if (flat) {
        if (!parts) hipLaunchKernelGGL((tbn_stats_kernel<4, true>), g1, dim3(256), 0, s, x, N, C, HW, (double*)scratch);
        hipLaunchKernelGGL((tbn_apply_kernel<4, true>), g2, dim3(256), 0, s, x, (const double*)scratch, nchunk, gamma, beta, y, N, C, HW, relu, 1e-5f, 0.1f,
                           stats, running_mean, running_var, residual);
    } else if (V == 4) {
        if (!parts) hipLaunchKernelGGL(tbn_stats_kernel<4>, g1, dim3(256), 0, s, x, N, C, HW, (double*)scratch);
        hipLaunchKernelGGL(tbn_apply_kernel<4>, g2, dim3(256), 0, s, x, (const double*)scratch, nchunk, gamma, beta, y, N, C, HW, relu, 1e-5f, 0.1f,
                           stats, running_mean, running_var, residual);
    } else {
        hipLaunchKernelGGL(tbn_stats_kernel<1>, g1, dim3(256), 0, s, x, N, C, HW, (double*)scratch);
        hipLaunchKernelGGL(tbn_apply_kernel<1>, g2, dim3(256), 0, s, x, (const double*)scratch, nchunk, gamma, beta, y, N, C, HW, relu, 1e-5f, 0.1f,
                           stats, running_mean, running_var, residual);
    }
}
void launch_tbn_bwd(const float* x, const float* dy, const float* stats, const float* gamma, const float* beta, float* dgamma, float* dbeta, float* dx,
                    int N, int C, long HW, int relu, void* scratch, hipStream_t s, const TBnRed* red)
{
    static const bool small4_off = getenv("YF_TBN_SMALL4_OFF") != nullptr;
    if (!small4_off && HW % 4 == 0 && (long)N * HW <= 4096 * 8) {
        const long P4 = (long)N * HW / 4;
#define YF_BNB(U_) hipLaunchKernelGGL(tbn_bwd_small4_kernel<U_>, dim3(C), dim3(1024), 0, s, x, dy, stats, gamma, beta, dgamma, dbeta, dx, N, C, (int)HW, relu)
        if (P4 <= 1024) YF_BNB(1); else if (P4 <= 2048) YF_BNB(2); else if (P4 <= 4096) YF_BNB(4); else YF_BNB(8);
#undef YF_BNB
        return;
    }
    if ((long)N * HW <= TBN_SMALL) {
        hipLaunchKernelGGL(tbn_bwd_small_kernel, dim3(C), dim3(1024), 0, s, x, dy, stats, gamma, beta, dgamma, dbeta, dx, N, C, (int)HW, relu);
        return;
    }
    static const bool flat_off = getenv("YF_TBN_FLAT_OFF") != nullptr;
    const bool flat = !flat_off && HW % 4 == 0 && HW % 1024 != 0;
    const bool parts = red && red->count > 0;                          // the data-gradient kernel above left the pairs: no reduction pass
    int pchunks = parts ? (int)((red->count + 1023) / 1024) : 0;
    if (pchunks > 64) pchunks = 64;
    const int V = (flat || (HW % 4 == 0 && HW >= 1024)) ? 4 : 1, nchunk = parts ? pchunks : tbn_chunks(N, C, HW, V, flat);
    const dim3 g1(nchunk, C), g2(tbn_apply_blocks(N, C, HW, V, flat), C);
    if (parts) {
        hipLaunchKernelGGL(tbn_stats_from_parts_kernel, dim3(pchunks, C), dim3(256), 0, s, (const float2*)red->part, red->count, (double*)scratch);
        if (flat)
            hipLaunchKernelGGL((tbn_bwd_apply_kernel<4, true>), g2, dim3(256), 0, s, x, dy, stats, gamma, beta, (const double*)scratch, nchunk, dgamma, dbeta,
                               dx, N, C, HW, relu);
        else if (V == 4)
            hipLaunchKernelGGL(tbn_bwd_apply_kernel<4>, g2, dim3(256), 0, s, x, dy, stats, gamma, beta, (const double*)scratch, nchunk, dgamma, dbeta, dx, N,
                               C, HW, relu);
        else
            hipLaunchKernelGGL(tbn_bwd_apply_kernel<1>, g2, dim3(256), 0, s, x, dy, stats, gamma, beta, (const double*)scratch, nchunk, dgamma, dbeta, dx, N,
                               C, HW, relu);
        return;
    }
    if (flat) {
        hipLaunchKernelGGL((tbn_bwd_reduce_kernel<4, true>), g1, dim3(256), 0, s, x, dy, stats, gamma, beta, N, C, HW, relu, (double*)scratch);
        hipLaunchKernelGGL((tbn_bwd_apply_kernel<4, true>), g2, dim3(256), 0, s, x, dy, stats, gamma, beta, (const double*)scratch, nchunk, dgamma, dbeta,
                           dx, N, C, HW, relu);
    } else if (V == 4) {
        hipLaunchKernelGGL(tbn_bwd_reduce_kernel<4>, g1, dim3(256), 0, s, x, dy, stats, gamma, beta, N, C, HW, relu, (double*)scratch);
        hipLaunchKernelGGL(tbn_bwd_apply_kernel<4>, g2, dim3(256), 0, s, x, dy, stats, gamma, beta, (const double*)scratch, nchunk, dgamma, dbeta, dx, N,
                           C, HW, relu);
    } else {
        hipLaunchKernelGGL(tbn_bwd_reduce_kernel<1>, g1, dim3(256), 0, s, x, dy, stats, gamma, beta, N, C, HW, relu, (double*)scratch);
        hipLaunchKernelGGL(tbn_bwd_apply_kernel<1>, g2, dim3(256), 0, s, x, dy, stats, gamma, beta, (const double*)scratch, nchunk, dgamma, dbeta, dx, N,
                           C, HW, relu);
    }
}
void launch_tchan_sum(const float* dy, float* out, int N, int C, long HW, hipStream_t s, void* scratch)
{
    const long U = (long)N * HW / 4;
    if (scratch && HW % 4 == 0 && U >= 16384) {                        // enough to spread over the chip: <= 64 chunks of >= 8 trips
        int nchunk = (int)(U / 2048);
        if (nchunk > 64) nchunk = 64;
        hipLaunchKernelGGL(tchan_sum_part_kernel, dim3(nchunk, C), dim3(256), 0, s, dy, N, C, HW, (double*)scratch);
        hipLaunchKernelGGL(tchan_sum_final_kernel, dim3((C + 63) / 64), dim3(64), 0, s, (const double*)scratch, nchunk, C, out);
        return;
    }
    hipLaunchKernelGGL(tchan_sum_kernel, dim3(C), dim3(256), 0, s, dy, N, C, HW, out);
}
void launch_tadd(const float* a, const float* b, float* out, long total, hipStream_t s)
{
    hipLaunchKernelGGL(tadd_kernel, dim3(nblk(total)), dim3(256), 0, s, a, b, out, total);
}
void launch_tslice(const float* src, float* dst, int N, int C, long HW, int Cs, int sc0, int Cd, int dc0, hipStream_t s)
{
    hipLaunchKernelGGL(tslice_kernel, dim3(nblk((long)N * C * HW)), dim3(256), 0, s, src, dst, N, C, HW, Cs, sc0, Cd, dc0);
}
void launch_tadam(float* p, const float* g, float* m, float* v, long total, double lr, double b1, double b2, double eps, int step, hipStream_t s)
{
    const double bc1 = 1.0 - pow(b1, (double)step), bc2 = 1.0 - pow(b2, (double)step);
    hipLaunchKernelGGL(tadam_kernel, dim3(nblk(total)), dim3(256), 0, s, p, g, m, v, total, (float)(1.0 - b1), (float)b2, (float)(1.0 - b2), (float)eps,
                       (float)(lr / bc1), (float)sqrt(bc2));
}

// host arrays of device pointers -> (one table upload +) one launch.  d_table: at least nt * 48 bytes of device memory.
// h_table: nt * 48 bytes of PINNED host memory owned by the caller (one per optimizer), or null.  With it the table is written there and
// uploaded with a truly asynchronous copy, and only when `upload` is set -- the caller sets it when the pointer set changed (p, m, v are
// stable; g is too when the flat gradient buffer is reused) and must not call again with upload = 1 before the previous upload has
// executed (training.Adam waits on an event).  Without it (null) the table is staged from pageable memory on every call, which the
// runtime completes before returning -- correct, but it can hold the host until the stream has drained.
int launch_tadam_multi(int nt, float* const* p, const float* const* g, float* const* m, float* const* v, const long* sizes, double lr, double b1,
                       double b2, double eps, int step, void* d_table, void* h_table, int upload, hipStream_t s)
{
    static thread_local std::vector<TAdamEntry> tab;
    TAdamEntry* host = static_cast<TAdamEntry*>(h_table);
    if (!host) { tab.resize(nt); host = tab.data(); upload = 1; }
    long blocks = 0;
    for (int t = 0; t < nt; ++t) {
        if (upload) host[t] = TAdamEntry{p[t], g[t], m[t], v[t], blocks, sizes[t]};
        blocks += (sizes[t] + 255) / 256;
    }
    if (upload && hipMemcpyAsync(d_table, host, (size_t)nt * sizeof(TAdamEntry), hipMemcpyHostToDevice, s) != hipSuccess) return -1;
    const double bc1 = 1.0 - pow(b1, (double)step), bc2 = 1.0 - pow(b2, (double)step);
    hipLaunchKernelGGL(tadam_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const TAdamEntry*)d_table, nt, (float)(1.0 - b1), (float)b2,
                       (float)(1.0 - b2), (float)eps, (float)(lr / bc1), (float)sqrt(bc2));
    return 0;
}

}  // namespace yf
