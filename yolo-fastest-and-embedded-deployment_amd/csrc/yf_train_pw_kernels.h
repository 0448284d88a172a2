// yf_train_pw_kernels.h -- pointwise convolutions as GEMMs on the fp32 matrix pipe: forward / backward-data (tpw_mfma, tpw4_mfma, tpw4_lds), the split weight-gradient GEMM (twgrad_body: pointwise, dense k x k, deconv) and the two in one launch (tpw_bwd_dual)
// Part of the training-step operators: yf_train_kernels.hip includes the family headers into ONE translation unit, INSIDE namespace yf, so the
// kernels keep their internal linkage and the launchers in that file see all of them.  Device code: include from there only.
#pragma once

// ---- pointwise convolution as a GEMM on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32), operands straight from global memory:
//   Y[m][q] = sum_k A[m][k] X[k][q],  q = pixel over the batch (frame n = q / HW), X and Y NCHW.
// forward: A = weight [Cout][Cin] (sm = Cin, sk = 1); backward-data: A = weight^T (m = ci, k = co: sm = 1, sk = Cin), X = dY.
// One wave = 16 output channels x 64 pixels (four 16x16 tiles sharing the A fragment); a workgroup = 4 waves on 4 channel tiles.
// MFMA operand layout: A lane l = (row l % 16, k l / 16); B lane l = (k l / 16, col l % 16); D lane l = rows 4 (l / 16) + i, col l % 16.
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
