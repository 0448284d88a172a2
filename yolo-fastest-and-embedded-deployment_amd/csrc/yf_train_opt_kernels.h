// yf_train_opt_kernels.h -- sums of split partial results, channel sums (bias gradients), add / slice, Adam
// Part of the training-step operators: yf_train_kernels.hip includes the family headers into ONE translation unit, INSIDE namespace yf, so the
// kernels keep their internal linkage and the launchers in that file see all of them.  Device code: include from there only.
#pragma once

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
