// ubench.hip -- micro-benchmarks of instruction streams (developer tool): how fast can a wave stream FMAs whose
// second operand is a wave-uniform weight fetched through the scalar path, vs. weights already in VGPRs?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// P pixels per lane, 8 inputs x 8 outputs per "layer step": 64 weights -> 64*P FMAs per lane
template <int P, int WSET>
__global__ void __launch_bounds__(256) sgpr_fma(const float* __restrict__ w, float* out, int iters)
{
    float x[P][8], acc[P][8];
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int k = 0; k < 8; ++k) { x[p][k] = threadIdx.x * 0.001f + k + p; acc[p][k] = 0.f; }
    for (int it = 0; it < iters; ++it) {
        const float* wc = w + (it % WSET) * 64;  // wave-uniform -> s_load
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float wv = wc[k * 8 + j];
#pragma unroll
                for (int p = 0; p < P; ++p) acc[p][j] = fmaf(x[p][k], wv, acc[p][j]);
            }
#pragma unroll
        for (int p = 0; p < P; ++p) x[p][it & 7] += acc[p][it & 7] * 1e-9f;  // keep a dependency
    }
    float s = 0;
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int j = 0; j < 8; ++j) s += acc[p][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// same arithmetic, weights as per-lane VGPR values loaded once (no scalar traffic in the loop)
template <int P>
__global__ void __launch_bounds__(256) vgpr_fma(const float* __restrict__ w, float* out, int iters)
{
    float x[P][8], acc[P][8], wv[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) wv[i] = w[i * 64 + (threadIdx.x & 63)];
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int k = 0; k < 8; ++k) { x[p][k] = threadIdx.x * 0.001f + k + p; acc[p][k] = 0.f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int p = 0; p < P; ++p) acc[p][j] = fmaf(x[p][k], wv[k * 8 + j], acc[p][j]);
#pragma unroll
        for (int p = 0; p < P; ++p) x[p][it & 7] += acc[p][it & 7] * 1e-9f;
    }
    float s = 0;
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int j = 0; j < 8; ++j) s += acc[p][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F> static float time_us(F&& f, int reps = 10)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms * 1000.f / reps;
}

int main()
{
    float *w, *out; CK(hipMalloc(&w, 1 << 20)); CK(hipMemset(w, 0, 1 << 20)); CK(hipMalloc(&out, 4 << 20));
    const int iters = 2000;
    for (int blocks : {256, 512, 1024, 2048}) {
        auto rep = [&](const char* tag, int P, float us) {
            double macs = (double)blocks * 256 * iters * 64.0 * P;
            printf("blocks=%5d (%d waves/SIMD) %-28s %9.1f us  %6.2f TMAC/s\n", blocks, blocks / 256, tag, us, macs / us * 1e-6);
        };
        rep("sgpr P=1 wset 16 (4 KB)", 1, time_us([&] { hipLaunchKernelGGL((sgpr_fma<1, 16>), dim3(blocks), dim3(256), 0, 0, w, out, iters); }));
        rep("sgpr P=2 wset 16", 2, time_us([&] { hipLaunchKernelGGL((sgpr_fma<2, 16>), dim3(blocks), dim3(256), 0, 0, w, out, iters); }));
        rep("sgpr P=4 wset 16", 4, time_us([&] { hipLaunchKernelGGL((sgpr_fma<4, 16>), dim3(blocks), dim3(256), 0, 0, w, out, iters); }));
        rep("sgpr P=8 wset 16", 8, time_us([&] { hipLaunchKernelGGL((sgpr_fma<8, 16>), dim3(blocks), dim3(256), 0, 0, w, out, iters); }));
        rep("sgpr P=4 wset 256 (64 KB)", 4, time_us([&] { hipLaunchKernelGGL((sgpr_fma<4, 256>), dim3(blocks), dim3(256), 0, 0, w, out, iters); }));
        rep("vgpr P=1", 1, time_us([&] { hipLaunchKernelGGL((vgpr_fma<1>), dim3(blocks), dim3(256), 0, 0, w, out, iters); }));
        rep("vgpr P=4", 4, time_us([&] { hipLaunchKernelGGL((vgpr_fma<4>), dim3(blocks), dim3(256), 0, 0, w, out, iters); }));
    }
    return 0;
}
