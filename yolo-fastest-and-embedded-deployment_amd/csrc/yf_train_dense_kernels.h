// yf_train_dense_kernels.h -- dense k x k convolutions and the 2x2 stride-2 transposed convolution: the generic one-thread-per-output kernels (fallback for any shape), im2col GEMM, the stride-2 3x3 kernels of conv0 / conv1_9 (forward, data and weight gradients), deconv
// Part of the training-step operators: yf_train_kernels.hip includes the family headers into ONE translation unit, INSIDE namespace yf, so the
// kernels keep their internal linkage and the launchers in that file see all of them.  Device code: include from there only.
#pragma once

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
