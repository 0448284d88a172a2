// yf_train_kernels.hip -- operators of the reference's TRAINING step (second slice): forward in train mode
// and backward of every layer type of YoloFastest (src/model_training/model/yolo_fastest.py:16-66), and the optimizer update of
// src/model_training/train.py:84 (Adam), on NCHW fp32 tensors like the reference's.
//
//   conv_norm_relu / conv_norm :16-38   Conv2d(bias=False) -> BatchNorm2d(train: batch statistics) -> [ReLU]
//   deconv_norm_relu           :42-48   ConvTranspose2d(k=2, s=2) -> BN -> ReLU
//   BasicResBlock              :52-66   three units + residual add
//   heads                      :136,146 Conv2d 1x1 with bias
//
// CORRECTNESS-FIRST kernels: one thread per output element with coalesced innermost-x access, reductions by block + atomics.
// They are NOT the tuned inference kernels (those fold BN, which training cannot: batch statistics) and are priced as such in
// DESIGN.md.  Everything is stream-ordered; no allocation, no synchronisation.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

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
// one workgroup per channel: mean and biased variance over N*H*W in double; stats[c] = {mean, invstd}
__global__ void __launch_bounds__(256) tbn_stats_kernel(const float* __restrict__ x, int N, int C, long HW, float eps, float momentum,
                                                        float* __restrict__ stats, float* __restrict__ running_mean, float* __restrict__ running_var)
{
    __shared__ double r1[4], r2[4];
    const int c = blockIdx.x;
    const long P = (long)N * HW;
    double s = 0, ss = 0;
    for (long p = threadIdx.x; p < P; p += 256) {
        const long n = p / HW, i = p - n * HW;
        const double v = x[(n * C + c) * HW + i];
        s += v; ss += v * v;
    }
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_down(s, o); ss += __shfl_down(ss, o); }
    if ((threadIdx.x & 63) == 0) { r1[threadIdx.x >> 6] = s; r2[threadIdx.x >> 6] = ss; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double sum = r1[0] + r1[1] + r1[2] + r1[3], sq = r2[0] + r2[1] + r2[2] + r2[3];
        const double mean = sum / (double)P;
        double var = sq / (double)P - mean * mean;
        if (var < 0) var = 0;
        stats[2 * c] = (float)mean;
        stats[2 * c + 1] = (float)(1.0 / sqrt(var + (double)eps));
        if (running_mean) {
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(var * (double)P / (double)(P > 1 ? P - 1 : 1));
        }
    }
}

__global__ void __launch_bounds__(256) tbn_apply_kernel(const float* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ y, long total, int C, long HW, int relu)
{
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)((idx / HW) % C);
    float v = (x[idx] - stats[2 * c]) * stats[2 * c + 1] * gamma[c] + beta[c];
    y[idx] = relu ? fmaxf(v, 0.f) : v;
}

// backward: dy_eff = dy * (y > 0) with ReLU; sums[c] = {sum dy_eff, sum dy_eff * xhat}  (= dbeta, dgamma)
__global__ void __launch_bounds__(256) tbn_bwd_reduce_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dy,
                                                             const float* __restrict__ stats, int N, int C, long HW, int relu,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta)
{
    __shared__ double r1[4], r2[4];
    const int c = blockIdx.x;
    const long P = (long)N * HW;
    const float mean = stats[2 * c], invstd = stats[2 * c + 1];
    double s = 0, sx = 0;
    for (long p = threadIdx.x; p < P; p += 256) {
        const long n = p / HW, i = p - n * HW, idx = (n * C + c) * HW + i;
        float g = dy[idx];
        if (relu && !(y[idx] > 0.f)) g = 0.f;
        s += g; sx += (double)g * (double)((x[idx] - mean) * invstd);
    }
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_down(s, o); sx += __shfl_down(sx, o); }
    if ((threadIdx.x & 63) == 0) { r1[threadIdx.x >> 6] = s; r2[threadIdx.x >> 6] = sx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        dbeta[c] = (float)(r1[0] + r1[1] + r1[2] + r1[3]);
        dgamma[c] = (float)(r2[0] + r2[1] + r2[2] + r2[3]);
    }
}

// dx = gamma * invstd * (dy_eff - (dbeta + xhat * dgamma) / P)
__global__ void __launch_bounds__(256) tbn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dy,
                                                            const float* __restrict__ stats, const float* __restrict__ gamma,
                                                            const float* __restrict__ dgamma, const float* __restrict__ dbeta, float* __restrict__ dx,
                                                            long total, int C, long HW, long P, int relu)
{
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)((idx / HW) % C);
    float g = dy[idx];
    if (relu && !(y[idx] > 0.f)) g = 0.f;
    const float xhat = (x[idx] - stats[2 * c]) * stats[2 * c + 1];
    dx[idx] = gamma[c] * stats[2 * c + 1] * (g - (dbeta[c] + xhat * dgamma[c]) / (float)P);
}

// per-channel sum over N, H, W (bias gradient of the head convs)
__global__ void __launch_bounds__(256) tchan_sum_kernel(const float* __restrict__ dy, int N, int C, long HW, float* __restrict__ out)
{
    __shared__ double r1[4];
    const int c = blockIdx.x;
    const long P = (long)N * HW;
    double s = 0;
    for (long p = threadIdx.x; p < P; p += 256) {
        const long n = p / HW, i = p - n * HW;
        s += dy[(n * C + c) * HW + i];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0) r1[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[c] = (float)(r1[0] + r1[1] + r1[2] + r1[3]);
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

static inline unsigned nblk(long total) { return (unsigned)((total + 255) / 256); }

void launch_tconv_fwd(const float* x, const float* w, const float* bias, float* y, int N, int Cin, int H, int W, int Cout, int k, int stride,
                      int depthwise, hipStream_t s)
{
    const int pad = (k - 1) / 2, Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    hipLaunchKernelGGL(tconv_fwd_kernel, dim3(nblk((long)N * Cout * Ho * Wo)), dim3(256), 0, s, x, w, bias, y, N, Cin, H, W, Cout, Ho, Wo, k, stride, depthwise);
}
void launch_tconv_bwd_data(const float* dy, const float* w, float* dx, int N, int Cin, int H, int W, int Cout, int k, int stride, int depthwise,
                           hipStream_t s)
{
    const int pad = (k - 1) / 2, Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    hipLaunchKernelGGL(tconv_bwd_data_kernel, dim3(nblk((long)N * Cin * H * W)), dim3(256), 0, s, dy, w, dx, N, Cin, H, W, Cout, Ho, Wo, k, stride, depthwise);
}
void launch_tconv_bwd_weight(const float* x, const float* dy, float* dw, int N, int Cin, int H, int W, int Cout, int k, int stride, int depthwise,
                             hipStream_t s)
{
    const int pad = (k - 1) / 2, Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    const long nw = (long)Cout * (depthwise ? 1 : Cin) * k * k, P = (long)N * Ho * Wo;
    int nchunk = (int)((P + 4095) / 4096);                       // ~16 reduction elements per thread
    while ((long)nchunk * nw > 262144 && nchunk > 1) nchunk /= 2;   // bound the grid
    (void)hipMemsetAsync(dw, 0, (size_t)nw * sizeof(float), s);
    hipLaunchKernelGGL(tconv_bwd_weight_kernel, dim3((unsigned)(nw * nchunk)), dim3(256), 0, s, x, dy, dw, N, Cin, H, W, Cout, Ho, Wo, k, stride,
                       depthwise, nchunk);
}
void launch_tdeconv_fwd(const float* x, const float* w, float* y, int N, int Cin, int H, int W, int Cout, hipStream_t s)
{
    hipLaunchKernelGGL(tdeconv_fwd_kernel, dim3(nblk((long)N * Cout * 4 * H * W)), dim3(256), 0, s, x, w, y, N, Cin, H, W, Cout);
}
void launch_tdeconv_bwd_data(const float* dy, const float* w, float* dx, int N, int Cin, int H, int W, int Cout, hipStream_t s)
{
    hipLaunchKernelGGL(tdeconv_bwd_data_kernel, dim3(nblk((long)N * Cin * H * W)), dim3(256), 0, s, dy, w, dx, N, Cin, H, W, Cout);
}
void launch_tdeconv_bwd_weight(const float* x, const float* dy, float* dw, int N, int Cin, int H, int W, int Cout, hipStream_t s)
{
    const long nw = (long)Cin * Cout * 4, P = (long)N * H * W;
    int nchunk = (int)((P + 4095) / 4096);
    while ((long)nchunk * nw > 262144 && nchunk > 1) nchunk /= 2;
    (void)hipMemsetAsync(dw, 0, (size_t)nw * sizeof(float), s);
    hipLaunchKernelGGL(tdeconv_bwd_weight_kernel, dim3((unsigned)(nw * nchunk)), dim3(256), 0, s, x, dy, dw, N, Cin, H, W, Cout, nchunk);
}
void launch_tbn_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var, float* stats, float* y, int N,
                    int C, long HW, int relu, hipStream_t s)
{
    hipLaunchKernelGGL(tbn_stats_kernel, dim3(C), dim3(256), 0, s, x, N, C, HW, 1e-5f, 0.1f, stats, running_mean, running_var);
    hipLaunchKernelGGL(tbn_apply_kernel, dim3(nblk((long)N * C * HW)), dim3(256), 0, s, x, stats, gamma, beta, y, (long)N * C * HW, C, HW, relu);
}
void launch_tbn_bwd(const float* x, const float* y, const float* dy, const float* stats, const float* gamma, float* dgamma, float* dbeta, float* dx,
                    int N, int C, long HW, int relu, hipStream_t s)
{
    hipLaunchKernelGGL(tbn_bwd_reduce_kernel, dim3(C), dim3(256), 0, s, x, y, dy, stats, N, C, HW, relu, dgamma, dbeta);
    hipLaunchKernelGGL(tbn_bwd_apply_kernel, dim3(nblk((long)N * C * HW)), dim3(256), 0, s, x, y, dy, stats, gamma, dgamma, dbeta, dx,
                       (long)N * C * HW, C, HW, (long)N * HW, relu);
}
void launch_tchan_sum(const float* dy, float* out, int N, int C, long HW, hipStream_t s)
{
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

}  // namespace yf
