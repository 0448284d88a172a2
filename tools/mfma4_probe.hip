// mfma4_probe.hip -- developer probe: operand layout and issue rate of v_mfma_f32_4x4x1_16B_f32 on gfx950.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma4_probe.hip -o tools/kb_mfma4 && tools/kb_mfma4
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void layout(const float* a, const float* b, float* d)
{
    const int l = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) d[l * 4 + i] = acc[i];
}

template <int KIND>
__global__ void rate(float* out, int iters)
{
    const float a = threadIdx.x * 0.001f, b = 1.f + threadIdx.x * 0.002f;
    f32x4 c[8];
    for (int i = 0; i < 8; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) c[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c[i], 0, 0, 0);
            else c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[i], 0, 0, 0);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0) / (iters * 8.f);
}

int main()
{
    float ha[64], hb[64], hd[256];
    for (int i = 0; i < 64; ++i) { ha[i] = 1.f + i; hb[i] = 100.f + 3.f * i; }
    float *a, *b, *d;
    CK(hipMalloc(&a, 256)); CK(hipMalloc(&b, 256)); CK(hipMalloc(&d, 1024));
    CK(hipMemcpy(a, ha, 256, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb, 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, a, b, d);
    CK(hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost));
    // hypothesis: lane l = (block bl = l / 4, j = l % 4); D[l][i] = A[lane 4*bl + i] * B[lane l]
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 4; ++i) {
            float want = ha[(l / 4) * 4 + i] * hb[l];
            if (hd[l * 4 + i] != want) { if (bad < 8) printf("lane %d reg %d: got %g want %g\n", l, i, hd[l * 4 + i], want); ++bad; }
        }
    printf("layout hypothesis (D[lane (b,j)][i] = A_b[i] * B_b[j]): %s\n", bad ? "WRONG" : "confirmed");
    float* o; CK(hipMalloc(&o, 256 * 256 * 4 * 4));
    for (int kind = 0; kind < 2; ++kind) {
        for (int waves = 1; waves <= 4; waves *= 2) {
            if (kind == 0) hipLaunchKernelGGL(rate<0>, dim3(256), dim3(64 * 4 * waves), 0, 0, o, 2000);
            else hipLaunchKernelGGL(rate<1>, dim3(256), dim3(64 * 4 * waves), 0, 0, o, 2000);
            CK(hipDeviceSynchronize());
            float r; CK(hipMemcpy(&r, o, 4, hipMemcpyDeviceToHost));
            printf("%s: %d waves/SIMD: %.1f shader-clock ticks per MFMA per wave (s_memtime/readcyclecounter units)\n", kind ? "16x16x4" : "4x4x1  ", waves, r);
        }
    }
    return 0;
}
