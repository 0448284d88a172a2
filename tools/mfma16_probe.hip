// mfma16_probe.hip -- developer probe for the split-operand ("f16x3") fp16 MFMA path on gfx950:
//   1. does v_mfma_f32_16x16x16_f16 keep fp16 SUBNORMAL inputs (hi/lo halves of small activations are subnormal)?
//   2. accuracy of a = hi + lo (both fp16), products hi*hi + hi*lo + lo*hi, fp32 accumulate, against fp64;
//   3. issue rate of 16x16x16_f16 / 16x16x32_f16 / 16x16x4_f32 (one wave per SIMD, 8 independent accumulators);
//   4. does fp16 MFMA overlap VALU work of the same wave's SIMD partner (two waves/SIMD: one MFMA-only, one FMA-only)?
//   hipcc -O3 --offload-arch=gfx950 tools/mfma16_probe.hip -o tools/kb_mfma16 && tools/kb_mfma16
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void denorm(float av, float bv, float* d)
{
    const f16x4 a = {(_Float16)av, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
    const f16x4 b = {(_Float16)bv, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, acc, 0, 0, 0);
    d[threadIdx.x] = acc[0];
}

// D[16x16] = A[16xK] * B[Kx16], K = 64, three ways: f32 MFMA, single fp16, split fp16 (3 MFMAs)
__global__ void split(const float* A, const float* B, float* out)   // A [16][64], B [64][16] row-major
{
    const int l = threadIdx.x, r = l & 15, q = l >> 4;
    f32x4 c32 = {0, 0, 0, 0}, c16 = {0, 0, 0, 0}, cx3 = {0, 0, 0, 0};
    for (int kb = 0; kb < 4; ++kb) {
        float a[4], b[4];
        for (int j = 0; j < 4; ++j) { a[j] = A[r * 64 + kb * 16 + 4 * q + j]; b[j] = B[(kb * 16 + 4 * q + j) * 16 + r]; }
        // f32: k-step j holds channel 4q + j in lane group q (a k permutation, the same on both operands)
        for (int j = 0; j < 4; ++j) c32 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], c32, 0, 0, 0);
        f16x4 ah, al, bh, bl;
        for (int j = 0; j < 4; ++j) {
            ah[j] = (_Float16)a[j]; al[j] = (_Float16)(a[j] - (float)ah[j]);
            bh[j] = (_Float16)b[j]; bl[j] = (_Float16)(b[j] - (float)bh[j]);
        }
        c16 = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bh, c16, 0, 0, 0);
        cx3 = __builtin_amdgcn_mfma_f32_16x16x16f16(al, bh, cx3, 0, 0, 0);   // small terms first
        cx3 = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bl, cx3, 0, 0, 0);
        cx3 = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bh, cx3, 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) {   // D: col r, row 4q + i
        out[0 * 256 + (4 * q + i) * 16 + r] = c32[i];
        out[1 * 256 + (4 * q + i) * 16 + r] = c16[i];
        out[2 * 256 + (4 * q + i) * 16 + r] = cx3[i];
    }
}

template <int KIND>
__global__ void __launch_bounds__(512) rate(float* out, int iters, int mixed)
{
    const int wave = threadIdx.x >> 6;
    const float af = threadIdx.x * 0.001f, bf = 1.f + threadIdx.x * 0.002f;
    const f16x4 a4 = {(_Float16)af, (_Float16)bf, (_Float16)af, (_Float16)bf};
    const f16x8 a8 = {(_Float16)af, (_Float16)bf, (_Float16)af, (_Float16)bf, (_Float16)af, (_Float16)bf, (_Float16)af, (_Float16)bf};
    f32x4 c[8];
    for (int i = 0; i < 8; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = af + i;
    unsigned long long t0 = __builtin_readcyclecounter();
    if (mixed && wave >= 4) {   // the SIMD partner: VALU only (8 independent fma chains)
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = fmaf(v[i], bf, af);
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) c[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, a4, c[i], 0, 0, 0);
                else if (KIND == 1) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, a8, c[i], 0, 0, 0);
                else c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, c[i], 0, 0, 0);
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3] + v[i];
    out[16 + blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[wave] = (float)(t1 - t0) / (iters * 8.f);
}

int main()
{
    float *d;
    CK(hipMalloc(&d, 1 << 20));
    float h[1024];
    // 1. subnormal inputs: 2^-20 (fp16 subnormal) * 1024 -> 2^-10 if kept, 0 if flushed
    denorm<<<1, 64>>>(ldexpf(1.f, -20), 1024.f, d);
    CK(hipMemcpy(h, d, 64 * 4, hipMemcpyDeviceToHost));
    printf("subnormal A (2^-20) x 1024: got %g (kept: %g, flushed: 0)\n", h[0], ldexp(1.0, -10));
    denorm<<<1, 64>>>(1024.f, ldexpf(1.f, -22), d);
    CK(hipMemcpy(h, d, 64 * 4, hipMemcpyDeviceToHost));
    printf("1024 x subnormal B (2^-22): got %g (kept: %g)\n", h[0], ldexp(1.0, -12));
    // 2. accuracy
    float A[16 * 64], B[64 * 16], *dA, *dB;
    srand(1);
    for (int i = 0; i < 1024; ++i) {
        A[i] = (float)(rand() / (double)RAND_MAX) * (i % 7 == 0 ? 30.f : 1.f) * (i % 3 ? 1.f : 1e-3f);   // post-ReLU-like, wide range
        B[i] = (float)(rand() / (double)RAND_MAX - 0.5) * (i % 5 == 0 ? 4.f : 0.3f);
    }
    CK(hipMalloc(&dA, sizeof A)); CK(hipMalloc(&dB, sizeof B));
    CK(hipMemcpy(dA, A, sizeof A, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B, sizeof B, hipMemcpyHostToDevice));
    split<<<1, 64>>>(dA, dB, d);
    CK(hipMemcpy(h, d, 768 * 4, hipMemcpyDeviceToHost));
    double e32 = 0, e16 = 0, ex3 = 0, mag = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double s = 0;
            for (int k = 0; k < 64; ++k) s += (double)A[i * 64 + k] * (double)B[k * 16 + j];
            e32 = fmax(e32, fabs(h[i * 16 + j] - s)); e16 = fmax(e16, fabs(h[256 + i * 16 + j] - s)); ex3 = fmax(ex3, fabs(h[512 + i * 16 + j] - s));
            mag = fmax(mag, fabs(s));
        }
    printf("K = 64 GEMM vs fp64 (max |result| %.3g): max err f32 MFMA %.3g, single fp16 %.3g, split fp16 x3 %.3g\n", mag, e32, e16, ex3);
    // 3./4. rates
    const char* names[3] = {"16x16x16_f16", "16x16x32_f16", "16x16x4_f32"};
    for (int mixed = 0; mixed < 2; ++mixed)
        for (int kind = 0; kind < 3; ++kind) {
            const int threads = mixed ? 512 : 256, iters = 20000;
            if (kind == 0) rate<0><<<1, threads>>>(d, iters, mixed);
            else if (kind == 1) rate<1><<<1, threads>>>(d, iters, mixed);
            else rate<2><<<1, threads>>>(d, iters, mixed);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h, d, 16 * 4, hipMemcpyDeviceToHost));
            if (!mixed) printf("%s alone (1 wave/SIMD): %.1f cycles per MFMA\n", names[kind], h[0]);
            else printf("%s beside a VALU-only partner wave: %.1f cycles per MFMA; partner: %.1f cycles per v_fma_f32 (alone: ~4)\n", names[kind], h[0], h[4]);
        }
    return 0;
}
