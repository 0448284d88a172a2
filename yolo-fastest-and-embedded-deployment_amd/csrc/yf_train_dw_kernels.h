// yf_train_dw_kernels.h -- depthwise 3x3 / 5x5 convolutions: forward, stride-1 / stride-2 backward-data, weight gradients, and their launchers
// Part of the training-step operators: yf_train_kernels.hip includes the family headers into ONE translation unit, INSIDE namespace yf, so the
// kernels keep their internal linkage and the launchers in that file see all of them.  Device code: include from there only.
#pragma once

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
                const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, cq[4] = {c4.x, c4.y, c4.z, c4.w}, rq[4] = {r4.x, r4.y, r4.z, r4.w};
                const float gq[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
                for (int j = 0; j < PAD; ++j) { win[j] = lq[4 - PAD + j]; win[PAD + 4 + j] = rq[j]; }
#pragma unroll
                for (int j = 0; j < 4; ++j) win[PAD + j] = cq[j];
#pragma unroll
                for (int kx = 0; kx < KS; ++kx)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[ky * KS + kx] = fmaf(gq[j], win[j + kx], acc[ky * KS + kx]);
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
            // (components by name: indexing the float4s through a pointer left the 5x5 instantiation with 32 bytes of scratch)
            const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, cq[4] = {c4.x, c4.y, c4.z, c4.w}, rq[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
            for (int j = 0; j < PAD; ++j) { win[j] = lq[4 - PAD + j]; win[PAD + 4 + j] = rq[j]; }
#pragma unroll
            for (int j = 0; j < 4; ++j) win[PAD + j] = cq[j];
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
