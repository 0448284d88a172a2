// hi/lo split of fp32 into two fp16 values: the C form (convert, subtract, convert) against v_fma_mixlo/hi_f16.  tools/mix_probe.hip (round 6)
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I yolo-fastest-and-embedded-deployment_amd/csrc -o tools/mix_probe.bin tools/mix_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, uint32_t* c_lo, uint32_t* a_lo, uint32_t* hi_out, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    const float a0 = in[2 * i], a1 = in[2 * i + 1];
    const f32x2 a = {a0, a1};
    const f16x2 hi = __builtin_convertvector(a, f16x2);
    const f16x2 lo_c = __builtin_convertvector(a - __builtin_convertvector(hi, f32x2), f16x2);
    unsigned r, h = __builtin_bit_cast(unsigned, hi);
    asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(a0));
    asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(h), "v"(a1));
    c_lo[i] = __builtin_bit_cast(unsigned, lo_c); a_lo[i] = r; hi_out[i] = h;
}
int main()
{
    const int n = 1 << 22;
    float* h = (float*)malloc(n * 4);
    uint32_t seed = 12345;
    for (int i = 0; i < n; ++i) {
        seed = seed * 1664525u + 1013904223u;
        uint32_t bits = seed;
        if (i % 4 == 0) { float v = ldexpf((float)((seed >> 8) & 0xffff) / 65536.f + 1.f, -(int)(seed % 40)); if (seed & 1) v = -v; h[i] = v; }
        else if (i % 4 == 1) { memcpy(&h[i], &bits, 4); if (!isfinite(h[i]) || fabsf(h[i]) > 6e4f) h[i] = 0.37f; }
        else if (i % 4 == 2) h[i] = (float)((int)(seed >> 16) - 32768) / 256.f;
        else h[i] = (float)((int)(seed >> 12) % 4096) * 1e-6f;
    }
    h[0] = 0.f; h[1] = -0.f; h[2] = 6.1e-5f; h[3] = -6.1e-5f; h[4] = 1e-7f; h[5] = 65504.f;
    float* d; uint32_t *c, *a, *hh;
    hipMalloc(&d, n * 4); hipMalloc(&c, n * 2); hipMalloc(&a, n * 2); hipMalloc(&hh, n * 2);
    hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    k<<<n / 2 / 256, 256>>>(d, c, a, hh, n);
    uint32_t* hc = (uint32_t*)malloc(n * 2); uint32_t* ha = (uint32_t*)malloc(n * 2);
    hipMemcpy(hc, c, n * 2, hipMemcpyDeviceToHost); hipMemcpy(ha, a, n * 2, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int i = 0; i < n / 2; ++i)
        if (hc[i] != ha[i]) { if (bad < 12) printf("in %.9g %.9g: C lo %08x  mix lo %08x\n", h[2 * i], h[2 * i + 1], hc[i], ha[i]); ++bad; }
    printf("pairs %d, differing %ld\n", n / 2, bad);
    return 0;
}
