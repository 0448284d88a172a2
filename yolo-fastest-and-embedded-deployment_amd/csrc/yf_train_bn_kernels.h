// yf_train_bn_kernels.h -- BatchNorm2d in train mode: statistics, apply, backward reduce / apply, the one-launch kernels of the small maps
// Part of the training-step operators: yf_train_kernels.hip includes the family headers into ONE translation unit, INSIDE namespace yf, so the
// kernels keep their internal linkage and the launchers in that file see all of them.  Device code: include from there only.
#pragma once

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
