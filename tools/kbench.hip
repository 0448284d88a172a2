// kbench.hip -- stand-alone micro-benchmark of kernel template variants (developer tool, not product).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-fast-math -ffp-contract=off -I yolo-fastest-and-embedded-deployment_amd/csrc \
//         tools/kbench.hip -o gpurun_out/kbench && gpurun_out/kbench
// Times each variant with HIP events over `reps` launches on random data (batch 256 of the 320x256 config).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "yf_fused_kernels.hip"
#include "yf_k19_kernels.hip"
#include "yf_mfma_kernels.hip"
#include "yf_conv_kernels.hip"
#include "yf_mres_kernels.hip"
#include "yf_mdw_kernels.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static float* dev_rand(size_t n, float scale = 1.f)
{
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = scale * ((rand() & 0xffff) / 32768.f - 1.f);
    float* d;
    CK(hipMalloc(&d, n * 4));
    CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    return d;
}

template <typename F>
static float time_us(F&& launch, int reps = 20)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    CK(hipGetLastError());
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1000.f / reps;
}

using namespace yf;

template <int CIN, int CEXP, int COUT, int S, bool RES, bool PRE, int TYB, int TXB, int BH, int BW, int EC, int CG, int PE = 1, bool XL = true>
static void bench_fb(const char* tag, int N, int H, int W)
{
    FbArgs a{};
    const int Ho = H / S, Wo = W / S;
    a.in = dev_rand(PRE ? (size_t)N * 4 * H * W : (size_t)N * H * W * CIN);
    a.w0 = dev_rand(72, 0.3f); a.b0 = dev_rand(8, 0.1f);
    a.wp = dev_rand(fb_packed_floats(CIN, CEXP, COUT, EC) + 64, 0.3f);
    float* out; CK(hipMalloc(&out, (size_t)N * Ho * Wo * COUT * 4)); a.out = out;
    a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo;
    a.tiles_y = (Ho + TYB * BH - 1) / (TYB * BH); a.tiles_x = (Wo + TXB * BW - 1) / (TXB * BW);
    dim3 grid(N * a.tiles_y * a.tiles_x);
#ifdef YF_STAMP
    CK(hipMalloc(&a.dbg, 64)); CK(hipMemset(a.dbg, 0, 64));
#endif
    float us = time_us([&] { hipLaunchKernelGGL((fused_block_kernel<CIN, CEXP, COUT, S, RES, false, PRE, TYB, TXB, BH, BW, EC, CG, PE, XL, float>), grid, dim3(TYB * TXB), 0, 0, a); });
#ifdef YF_STAMP
    { unsigned long long h[6]; CK(hipMemcpy(h, a.dbg, 48, hipMemcpyDeviceToHost)); double tot = 0; for (int i = 0; i < 6; ++i) tot += (double)h[i];
      printf("    stamps%%: loop-top %.1f | expand %.1f | barrier1 %.1f | dw+project %.1f | barrier2 %.1f | epilogue %.1f   (avg cycles/wave %.0f)\n",
             100 * h[0] / tot, 100 * h[1] / tot, 100 * h[2] / tot, 100 * h[3] / tot, 100 * h[4] / tot, 100 * h[5] / tot, tot / 23.0 / (grid.x * (TYB * TXB / 64.0))); }
#endif
    double macs = (double)N * (PRE ? H * W * 72.0 : 0) + (double)N * H * W * CIN * CEXP + (double)N * Ho * Wo * CEXP * (9 + COUT);
    printf("%-44s thr=%4d tile=%2dx%-2d EC=%2d CG=%2d PE=%d XL=%d grid=%6u  %8.1f us  %6.2f TMAC/s\n", tag, TYB * TXB, TYB * BH, TXB * BW, EC, CG, PE, (int)XL, grid.x, us, macs / us * 1e-6);
}

template <int K1, int N, int MT, bool RELU, bool RES>
static void bench_mfma(const char* tag, long M)
{
    PwArgs a{};
    a.in1 = dev_rand((size_t)M * K1); a.in2 = nullptr;
    a.w = dev_rand(mfma_packed_floats(K1, 0, N), 0.2f); a.b = dev_rand(N, 0.1f);
    a.res = RES ? dev_rand((size_t)M * N) : nullptr;
    float* out; CK(hipMalloc(&out, (size_t)M * N * 4)); a.out = out;
    a.npix = M; a.HW = 320; a.W = 20;
    const long waves = (M + 16 * MT - 1) / (16 * MT);
    dim3 grid((unsigned)((waves + 3) / 4));
    float us = time_us([&] { hipLaunchKernelGGL((pw_mfma_kernel<K1, 0, N, MT, RELU, RES, 0, float>), grid, dim3(256), 0, 0, a); });
    printf("%-44s M=%7ld MT=%d grid=%6u  %8.1f us  %6.2f TMAC/s  out %.1f GB/s\n", tag, M, MT, grid.x, us, (double)M * K1 * N / us * 1e-6, (double)M * N * 4 / us * 1e-3);
}

template <int CIN, int CEXP, int COUT, bool RES, int TH, int TW, int NWAVE>
static void bench_mres(const char* tag, int N, int H, int W)
{
    MresArgs a{};
    a.in = dev_rand((size_t)N * H * W * CIN);
    a.wp = dev_rand(mres_packed_floats(CIN, CEXP, COUT) + 4096, 0.2f);
    float* out; CK(hipMalloc(&out, (size_t)N * H * W * COUT * 4)); a.out = out;
    a.H = H; a.W = W;
    a.tiles_y = (H + TH - 1) / TH; a.tiles_x = (W + TW - 1) / TW;
    constexpr int MTR = ((TH + 2) * (TW + 2) + 15) / 16;
    constexpr size_t lds = ((size_t)MTR * 16 * (CIN + 4) + 16 * mres_epl(MTR) +
                            ((((CEXP + 15) / 16) * mres_chunk_floats(CIN, COUT) + COUT + 3) & ~3)) * sizeof(float);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(mres_kernel<CIN, CEXP, COUT, RES, 1, TH, TW, NWAVE, float>),
                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid(N * a.tiles_y * a.tiles_x);
#ifdef YF_MRES_STAMP
    { unsigned long long z[16] = {0}; CK(hipMemcpyToSymbol(HIP_SYMBOL(yf_mres_dbg), z, sizeof z)); }
#endif
    float us = time_us([&] { hipLaunchKernelGGL((mres_kernel<CIN, CEXP, COUT, RES, 1, TH, TW, NWAVE, float>), grid, dim3(NWAVE * 64), lds, 0, a); });
    double macs = (double)N * H * W * CEXP * (CIN + 9 + COUT);
    printf("%-40s tile=%2dx%-2d waves=%d lds=%6zu grid=%6u  %8.1f us  %6.2f TMAC/s\n", tag, TH, TW, NWAVE, lds, grid.x, us, macs / us * 1e-6);
#ifdef YF_MRES_STAMP
    { unsigned long long h[16]; CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(yf_mres_dbg), sizeof h));
      for (int w = 0; w < 2; ++w) { double tot = 0; for (int i = 0; i < 8; ++i) tot += (double)h[w * 8 + i];
        printf("    %s wave cycles/WG: stage %.0f | frags %.0f | chunk weights %.0f | expand %.0f | barrier1 %.0f | dw+project %.0f | barrier2 %.0f | epilogue %.0f | total %.0f\n",
               w ? "last " : "first", h[w*8+0] / 23.0 / grid.x, h[w*8+1] / 23.0 / grid.x, h[w*8+2] / 23.0 / grid.x, h[w*8+3] / 23.0 / grid.x, h[w*8+4] / 23.0 / grid.x,
               h[w*8+5] / 23.0 / grid.x, h[w*8+6] / 23.0 / grid.x, h[w*8+7] / 23.0 / grid.x, tot / 23.0 / grid.x); } }
#endif
}

template <int CIN, int CEXP, int COUT, bool RES, int TH, int TW, int NWP, int NWC>
static void bench_mres_pc(const char* tag, int N, int H, int W)
{
    MresArgs a{};
    a.in = dev_rand((size_t)N * H * W * CIN);
    a.wp = dev_rand(mres_packed_floats(CIN, CEXP, COUT) + 4096, 0.2f);
    float* out; CK(hipMalloc(&out, (size_t)N * H * W * COUT * 4)); a.out = out;
    a.H = H; a.W = W;
    a.tiles_y = (H + TH - 1) / TH; a.tiles_x = (W + TW - 1) / TW;
    constexpr int MTR = ((TH + 2) * (TW + 2) + 15) / 16;
    constexpr size_t lds = ((size_t)MTR * 16 * (CIN + 4) + 2 * 16 * mres_epl(MTR) +
                            ((((CEXP + 15) / 16) * mres_chunk_floats(CIN, COUT) + COUT + 3) & ~3)) * sizeof(float);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(mres_pc_kernel<CIN, CEXP, COUT, RES, TH, TW, NWP, NWC, float>),
                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid(N * a.tiles_y * a.tiles_x);
    float us = time_us([&] { hipLaunchKernelGGL((mres_pc_kernel<CIN, CEXP, COUT, RES, TH, TW, NWP, NWC, float>), grid, dim3((NWP + NWC) * 64), lds, 0, a); });
    double macs = (double)N * H * W * CEXP * (CIN + 9 + COUT);
    printf("%-40s tile=%2dx%-2d prod=%d cons=%d lds=%6zu grid=%6u  %8.1f us  %6.2f TMAC/s\n", tag, TH, TW, NWP, NWC, lds, grid.x, us, macs / us * 1e-6);
}

int main(int argc, char** argv)
{
    const int N = 256;
    const char* which = argc > 1 ? argv[1] : "all";
    auto on = [&](const char* k) { return !strcmp(which, "all") || !strcmp(which, k); };
    if (on("s4")) {
        printf("--- res2_x: 8/32 residual block at 64x80 ---\n");
        bench_fb<8, 32, 8, 1, true, false, 32, 8, 2, 2, 8, 8, 1, false>("64x16 tile 2x2 (old, no X staging)", N, 64, 80);
        bench_fb<8, 32, 8, 1, true, false, 16, 20, 1, 2, 8, 8, 1, true>("16x40 tile 1x2", N, 64, 80);
        bench_fb<8, 32, 8, 1, true, false, 16, 20, 1, 2, 8, 8, 2, true>("16x40 tile 1x2 PE2", N, 64, 80);
        bench_fb<8, 32, 8, 1, true, false, 16, 16, 1, 2, 8, 8, 1, true>("16x32 tile 1x2", N, 64, 80);
        bench_fb<8, 32, 8, 1, true, false, 16, 16, 1, 1, 8, 8, 1, true>("16x16 tile 1x1", N, 64, 80);
        bench_fb<8, 32, 8, 1, true, false, 8, 16, 2, 2, 8, 8, 1, true>("16x32 tile 2x2 (128 thr)", N, 64, 80);
        bench_fb<8, 32, 8, 1, true, false, 8, 8, 2, 2, 8, 8, 1, true>("16x16 tile 2x2 (64 thr)", N, 64, 80);
    }
    if (on("fbprod")) {
        printf("--- VALU block kernels, production shapes ---\n");
        bench_fb<8, 8, 4, 1, false, true, 16, 16, 2, 2, 8, 8, 1, false>("stem 32x32 2x2", N, 128, 160);
        bench_fb<4, 8, 4, 1, true, false, 16, 16, 1, 2, 8, 8, 1, false>("res1_1 16x32 1x2", N, 128, 160);
        bench_fb<8, 32, 8, 1, true, false, 16, 16, 2, 2, 8, 8, 1, false>("res2 32x32 2x2", N, 64, 80);
        bench_fb<8, 32, 8, 1, true, false, 16, 16, 2, 2, 8, 8, 1, false>("res2 32x32 2x2, 64x64 frames (no partial tiles)", N, 64, 64);
        bench_fb<8, 32, 8, 1, true, false, 16, 16, 2, 2, 8, 8, 1, false>("res2 32x32 2x2, 64x96 frames", N, 64, 96);
    }
    if (on("xl0")) {
        printf("--- tile shapes WITHOUT input staging (the production form), after the SGPR-spill fix ---\n");
        bench_fb<8, 8, 4, 1, false, true, 16, 16, 2, 2, 8, 8, 1, false>("stem 32x32 2x2 (production)", N, 128, 160);
        bench_fb<8, 8, 4, 1, false, true, 16, 16, 1, 2, 8, 8, 1, false>("stem 16x32 1x2", N, 128, 160);
        bench_fb<8, 8, 4, 1, false, true, 16, 16, 2, 1, 8, 8, 1, false>("stem 32x16 2x1", N, 128, 160);
        bench_fb<8, 8, 4, 1, false, true, 16, 16, 1, 1, 8, 8, 1, false>("stem 16x16 1x1", N, 128, 160);
        bench_fb<8, 8, 4, 1, false, true, 8, 32, 2, 2, 8, 8, 1, false>("stem 16x64 2x2", N, 128, 160);
        bench_fb<8, 8, 4, 1, false, true, 16, 16, 2, 4, 8, 8, 1, false>("stem 32x64 2x4", N, 128, 160);
        bench_fb<4, 8, 4, 1, true, false, 16, 16, 2, 2, 8, 8, 1, false>("res1_1 32x32 2x2 (production)", N, 128, 160);
        bench_fb<4, 8, 4, 1, true, false, 16, 16, 1, 2, 8, 8, 1, false>("res1_1 16x32 1x2", N, 128, 160);
        bench_fb<4, 8, 4, 1, true, false, 16, 16, 2, 4, 8, 8, 1, false>("res1_1 32x64 2x4", N, 128, 160);
        bench_fb<4, 8, 4, 1, true, false, 8, 32, 2, 2, 8, 8, 1, false>("res1_1 16x64 2x2", N, 128, 160);
        bench_fb<4, 8, 4, 1, true, false, 16, 16, 4, 2, 8, 8, 1, false>("res1_1 64x32 4x2", N, 128, 160);
        bench_fb<8, 32, 8, 1, true, false, 32, 8, 2, 2, 8, 8, 1, false>("res2 64x16 2x2 (production)", N, 64, 80);
        bench_fb<8, 32, 8, 1, true, false, 16, 16, 2, 2, 8, 8, 1, false>("res2 32x32 2x2", N, 64, 80);
        bench_fb<8, 32, 8, 1, true, false, 16, 20, 2, 2, 8, 8, 1, false>("res2 32x40 2x2", N, 64, 80);
        bench_fb<8, 32, 8, 1, true, false, 16, 20, 1, 2, 8, 8, 1, false>("res2 16x40 1x2", N, 64, 80);
        bench_fb<8, 32, 8, 1, true, false, 16, 20, 2, 1, 8, 8, 1, false>("res2 32x20 2x1", N, 64, 80);
        bench_fb<8, 32, 8, 1, true, false, 16, 20, 1, 1, 8, 8, 1, false>("res2 16x20 1x1", N, 64, 80);
        bench_fb<8, 32, 8, 1, true, false, 16, 20, 2, 4, 8, 8, 1, false>("res2 32x80 2x4", N, 64, 80);
        bench_fb<8, 32, 8, 2, false, false, 16, 20, 1, 1, 8, 8, 1, false>("conv2_2 s2 16x20 1x1 (production)", N, 64, 80);
        bench_fb<8, 32, 8, 2, false, false, 16, 20, 1, 2, 8, 8, 1, false>("conv2_2 s2 16x40 1x2", N, 64, 80);
        bench_fb<8, 48, 8, 1, true, false, 16, 20, 1, 2, 8, 8, 2, false>("res3_1 16x40 1x2 PE2 (production)", N, 32, 40);
        bench_fb<8, 48, 8, 1, true, false, 16, 20, 1, 1, 8, 8, 2, false>("res3_1 16x20 1x1 PE2", N, 32, 40);
        bench_fb<8, 48, 8, 1, true, false, 16, 20, 2, 2, 8, 8, 2, false>("res3_1 32x40 2x2 PE2", N, 32, 40);
        bench_fb<8, 48, 8, 1, true, false, 16, 20, 1, 2, 8, 8, 1, false>("res3_1 16x40 1x2 PE1", N, 32, 40);
    }
    if (on("stem")) {
        printf("--- stem conv0+conv1_2/1_3/1_4 and res1_1 at 128x160 ---\n");
        bench_fb<8, 8, 4, 1, false, true, 16, 16, 2, 2, 8, 8, 1, true>("stem 32x32 2x2", N, 128, 160);
        bench_fb<8, 8, 4, 1, false, true, 16, 16, 2, 2, 8, 8, 2, true>("stem 32x32 2x2 PE2", N, 128, 160);
        bench_fb<8, 8, 4, 1, false, true, 16, 16, 1, 2, 8, 8, 1, true>("stem 16x32 1x2", N, 128, 160);
        bench_fb<8, 8, 4, 1, false, true, 16, 16, 1, 1, 8, 8, 1, true>("stem 16x16 1x1", N, 128, 160);
        bench_fb<4, 8, 4, 1, true, false, 16, 16, 2, 2, 8, 8, 1, false>("res1_1 32x32 2x2 (old)", N, 128, 160);
        bench_fb<4, 8, 4, 1, true, false, 16, 16, 2, 2, 8, 8, 1, true>("res1_1 32x32 2x2", N, 128, 160);
        bench_fb<4, 8, 4, 1, true, false, 16, 16, 2, 2, 8, 8, 4, true>("res1_1 32x32 2x2 PE4", N, 128, 160);
        bench_fb<4, 8, 4, 1, true, false, 16, 16, 1, 2, 8, 8, 1, true>("res1_1 16x32 1x2", N, 128, 160);
    }
    if (on("s8")) {
        printf("--- 8/48 ---\n");
        bench_fb<8, 48, 8, 1, true, false, 16, 20, 1, 2, 8, 8, 2, false>("8/48 PE2 (no X staging)", N, 32, 40);
        bench_fb<8, 48, 8, 1, true, false, 16, 20, 1, 2, 8, 8, 1, true>("8/48 PE1", N, 32, 40);
        bench_fb<8, 48, 8, 1, true, false, 16, 20, 1, 2, 8, 8, 2, true>("8/48 PE2", N, 32, 40);
        bench_fb<8, 48, 8, 1, true, false, 16, 20, 1, 1, 8, 8, 2, true>("8/48 16x20 1x1 PE2", N, 32, 40);
    }
    if (on("k19")) {
        K19Args a{};
        a.in = dev_rand((size_t)N * 128 * 160 * 4 + 8192) + 4096;  // k19m reads a guard band around its input
        a.w8 = dev_rand(96, 0.3f); a.b8 = dev_rand(24, 0.1f); a.w9 = dev_rand(9 * 24 * 24, 0.1f); a.b9 = dev_rand(24, 0.1f);
        a.w21 = dev_rand(24 * 8, 0.2f); a.b21 = dev_rand(8, 0.1f);
        float* out; CK(hipMalloc(&out, (size_t)N * 64 * 80 * 8 * 4)); a.out = out;
        a.H = 128; a.W = 160; a.Ho = 64; a.Wo = 80;
        double macs = (double)N * (128.0 * 160 * 96 + 64.0 * 80 * (5184 + 192));
        a.tiles_y = 4; a.tiles_x = 5;
        float us = time_us([&] { hipLaunchKernelGGL(k19_kernel<float>, dim3(N * 20), dim3(256), 0, 0, a); });
        printf("k19 1 px/lane  16x16 tile   %8.1f us  %6.2f TMAC/s\n", us, macs / us * 1e-6);
        // matrix-core version (random fragments: timing only)
        a.wp = dev_rand(k19_packed_floats(false) + 64, 0.1f);
        a.tiles_y = 8; a.tiles_x = 5; a.n_frames = N;
        auto run = [&](auto kern, int dt, unsigned grid, const char* tag) {
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)k19m_lds_bytes(dt)));
            float t = time_us([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(512), k19m_lds_bytes(dt), 0, a); });
            printf("k19m %-28s grid %4u %8.1f us  %6.2f TMAC/s\n", tag, grid, t, macs / t * 1e-6);
        };
        for (unsigned grid : {128u, 256u, 512u}) {
            run(&k19m_kernel<float, 0>, DT_F32, grid, "f32");
            run(&k19m_kernel<half_t, 0>, DT_F16, grid, "f16");
        }
        run(&k19m_kernel<float, 1>, DT_F32, 256, "f32 no phase 1");
        run(&k19m_kernel<float, 2>, DT_F32, 256, "f32 no phase-2 MFMAs");
        run(&k19m_kernel<float, 3>, DT_F32, 256, "f32 neither");
        run(&k19m_kernel<half_t, 1>, DT_F16, 256, "f16 no phase 1");
        run(&k19m_kernel<half_t, 2>, DT_F16, 256, "f16 no phase-2 MFMAs");
        run(&k19m_kernel<half_t, 3>, DT_F16, 256, "f16 neither");
    }
    if (on("mrespc")) {
        printf("--- MFMA residual blocks, producer/consumer waves ---\n");
        bench_mres<16, 96, 16, true, 16, 20, 8>("16/96 s8 (2-barrier, 8 waves)", N, 32, 40);
        bench_mres_pc<16, 96, 16, true, 16, 20, 4, 4>("16/96 s8", N, 32, 40);
        bench_mres_pc<16, 96, 16, true, 16, 20, 3, 5>("16/96 s8", N, 32, 40);
        bench_mres_pc<16, 96, 16, true, 16, 20, 2, 5>("16/96 s8", N, 32, 40);
        bench_mres_pc<16, 96, 16, true, 16, 20, 4, 10>("16/96 s8", N, 32, 40);
        bench_mres_pc<16, 96, 16, true, 8, 20, 2, 5>("16/96 s8", N, 32, 40);
        bench_mres<24, 136, 24, true, 16, 20, 8>("24/136 s16 (2-barrier, 8 waves)", N, 16, 20);
        bench_mres_pc<24, 136, 24, true, 16, 20, 4, 4>("24/136 s16", N, 16, 20);
        bench_mres_pc<24, 136, 24, true, 16, 20, 3, 5>("24/136 s16", N, 16, 20);
        bench_mres_pc<24, 136, 24, true, 16, 20, 5, 10>("24/136 s16", N, 16, 20);
        bench_mres_pc<24, 136, 24, true, 16, 20, 6, 10>("24/136 s16", N, 16, 20);
        bench_mres<48, 224, 48, true, 8, 10, 8>("48/224 s32 (2-barrier, 8 waves)", N, 8, 10);
        bench_mres_pc<48, 224, 48, true, 8, 10, 4, 4>("48/224 s32", N, 8, 10);
        bench_mres_pc<48, 224, 48, true, 8, 10, 4, 5>("48/224 s32", N, 8, 10);
        bench_mres_pc<48, 224, 48, true, 8, 10, 8, 5>("48/224 s32", N, 8, 10);
        bench_mres_pc<48, 224, 48, true, 8, 10, 10, 5>("48/224 s32", N, 8, 10);
        bench_mres_pc<48, 224, 48, true, 8, 10, 11, 5>("48/224 s32", N, 8, 10);
        bench_mres_pc<48, 224, 48, true, 8, 10, 8, 8>("48/224 s32", N, 8, 10);
        bench_mres_pc<48, 224, 48, true, 8, 10, 6, 5>("48/224 s32", N, 8, 10);
        bench_mres_pc<24, 136, 24, true, 16, 20, 6, 10>("24/136 s16", N, 16, 20);
        bench_mres_pc<24, 136, 24, true, 16, 20, 8, 8>("24/136 s16", N, 16, 20);
        bench_mres_pc<24, 136, 24, true, 16, 20, 10, 5>("24/136 s16", N, 16, 20);
        bench_mres_pc<24, 136, 24, true, 16, 20, 8, 5>("24/136 s16", N, 16, 20);
        bench_mres_pc<24, 136, 24, true, 16, 20, 7, 7>("24/136 s16", N, 16, 20);
        bench_mres_pc<24, 136, 24, true, 8, 20, 4, 5>("24/136 s16", N, 16, 20);
        bench_mres_pc<24, 136, 24, true, 8, 20, 8, 5>("24/136 s16", N, 16, 20);
        bench_mres_pc<48, 224, 48, true, 8, 10, 3, 5>("48/224 s32", N, 8, 10);
        bench_mres<8, 48, 16, false, 16, 20, 4>("8/48/16 s8 (2-barrier, 4 waves)", N, 32, 40);
        bench_mres_pc<8, 48, 16, false, 16, 20, 2, 5>("8/48/16 s8", N, 32, 40);
        bench_mres_pc<8, 48, 16, false, 16, 20, 3, 5>("8/48/16 s8", N, 32, 40);
        bench_mres_pc<8, 32, 8, true, 16, 20, 2, 5>("8/32 s4", N, 64, 80);
        bench_mres_pc<8, 48, 8, true, 16, 20, 2, 5>("8/48 s8", N, 32, 40);
    }
    if (on("mres2")) {
        printf("--- MFMA blocks, small tiles / many waves ---\n");
        bench_mres<8, 32, 8, true, 8, 10, 8>("8/32 s4", N, 64, 80);
        bench_mres<8, 32, 8, true, 8, 20, 8>("8/32 s4", N, 64, 80);
        bench_mres<8, 32, 8, true, 16, 10, 8>("8/32 s4", N, 64, 80);
        bench_mres<8, 32, 8, true, 8, 10, 4>("8/32 s4", N, 64, 80);
        bench_mres<8, 32, 8, true, 8, 20, 4>("8/32 s4", N, 64, 80);
        bench_mres<8, 48, 8, true, 8, 10, 8>("8/48 s8", N, 32, 40);
        bench_mres<8, 48, 8, true, 8, 20, 8>("8/48 s8", N, 32, 40);
        bench_mres<8, 48, 8, true, 8, 10, 4>("8/48 s8", N, 32, 40);
        bench_mres<8, 48, 8, true, 16, 20, 8>("8/48 s8", N, 32, 40);
        bench_mres<8, 48, 16, false, 8, 10, 8>("8/48/16 s8", N, 32, 40);
        bench_mres<8, 48, 16, false, 8, 20, 8>("8/48/16 s8", N, 32, 40);
        bench_mres<8, 48, 16, false, 16, 20, 8>("8/48/16 s8", N, 32, 40);
        bench_mres<16, 96, 16, true, 8, 10, 8>("16/96 s8", N, 32, 40);
        bench_mres<16, 96, 16, true, 8, 20, 8>("16/96 s8", N, 32, 40);
        bench_mres<16, 96, 16, true, 16, 10, 8>("16/96 s8", N, 32, 40);
        bench_mres<16, 96, 16, true, 8, 10, 4>("16/96 s8", N, 32, 40);
        bench_mres<24, 136, 24, true, 8, 10, 8>("24/136 s16", N, 16, 20);
        bench_mres<24, 136, 24, true, 8, 20, 8>("24/136 s16", N, 16, 20);
    }
    if (on("mresp")) {
        printf("--- MFMA blocks, production shapes (YF_MRES_DBG=%d) ---\n", YF_MRES_DBG);
        bench_mres<8, 48, 8, true, 16, 20, 8>("8/48 s8", N, 32, 40);
        bench_mres<16, 96, 16, true, 16, 20, 8>("16/96 s8", N, 32, 40);
        bench_mres<16, 96, 16, true, 16, 20, 5>("16/96 s8 5 waves", N, 32, 40);
        bench_mres<16, 96, 16, true, 16, 20, 10>("16/96 s8 10 waves", N, 32, 40);
    }
    if (on("mres")) {
        printf("--- MFMA residual blocks ---\n");
        bench_mres<16, 96, 16, true, 16, 20, 4>("16/96 s8", N, 32, 40);
        bench_mres<16, 96, 16, true, 16, 20, 5>("16/96 s8 (25 / 20 tiles: even)", N, 32, 40);
        bench_mres<16, 96, 16, true, 16, 20, 8>("16/96 s8", N, 32, 40);
        bench_mres<16, 96, 16, true, 16, 20, 10>("16/96 s8", N, 32, 40);
        bench_mres<16, 96, 16, true, 32, 20, 8>("16/96 s8", N, 32, 40);
        bench_mres<16, 96, 16, true, 16, 40, 8>("16/96 s8", N, 32, 40);
        bench_mres<16, 96, 16, true, 8, 20, 4>("16/96 s8", N, 32, 40);
        bench_mres<16, 96, 16, true, 8, 40, 4>("16/96 s8", N, 32, 40);
        bench_mres<24, 136, 24, true, 16, 20, 4>("24/136 s16", N, 16, 20);
        bench_mres<24, 136, 24, true, 16, 20, 8>("24/136 s16", N, 16, 20);
        bench_mres<24, 136, 24, true, 8, 20, 4>("24/136 s16", N, 16, 20);
        bench_mres<24, 136, 24, true, 8, 20, 2>("24/136 s16", N, 16, 20);
        bench_mres<48, 224, 48, true, 8, 10, 4>("48/224 s32", N, 8, 10);
        bench_mres<48, 224, 48, true, 8, 10, 8>("48/224 s32", N, 8, 10);
        bench_mres<48, 224, 48, true, 8, 10, 2>("48/224 s32", N, 8, 10);
        bench_mres<8, 48, 16, false, 16, 20, 4>("8/48/16 s8", N, 32, 40);
        bench_mres<8, 48, 16, false, 16, 40, 8>("8/48/16 s8", N, 32, 40);
        bench_mres<8, 32, 8, true, 16, 20, 4>("8/32 s4", N, 64, 80);
        bench_mres<8, 32, 8, true, 16, 40, 8>("8/32 s4", N, 64, 80);
    }
    if (on("mdw")) {
        printf("--- dw5x5 -> 1x1 (-> head) kernels, 256 frames of 16x20 ---\n");
        auto bench_mdw = [&](int c, int n, int headn, int H, int W, const char* tag) {
            MdwArgs a{};
            a.in = dev_rand((size_t)N * H * W * c);
            a.wp = dev_rand(mdw_packed_floats(c, n, headn) + 64, 0.1f);
            float* out; CK(hipMalloc(&out, (size_t)N * H * W * (headn ? 24 : n) * 4)); a.out = out;
            a.H = H; a.W = W;
#ifdef YF_MDW_STAMP
            unsigned long long z[8] = {0}; CK(hipMemcpyToSymbol(HIP_SYMBOL(yf_mdw_dbg), z, sizeof z));
#endif
            float us = time_us([&] { launch_mdw(c, n, headn, a, N, 0, DT_F32); });
            printf("mdw %-32s %8.1f us\n", tag, us);
#ifdef YF_MDW_STAMP
            unsigned long long h[8]; CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(yf_mdw_dbg), sizeof h));
            double tot = 0; for (int i = 0; i < 7; ++i) tot += (double)h[i];
            printf("    stamps%%: prologue %.1f | fill %.1f | barrier1 %.1f | depthwise %.1f | 1x1 MFMA %.1f | barrier2 %.1f | epilogue %.1f  (cycles/WG %.0f)\n",
                   100 * h[0] / tot, 100 * h[1] / tot, 100 * h[2] / tot, 100 * h[3] / tot, 100 * h[4] / tot, 100 * h[5] / tot, 100 * h[6] / tot, tot / 23.0 / N);
#endif
        };
        bench_mdw(96, 96, 0, 16, 20, "96->96 s16");
        bench_mdw(96, 96, 24, 16, 20, "96->96->head s16");
        bench_mdw(96, 128, 0, 8, 10, "96->128 s32");
        bench_mdw(128, 128, 24, 8, 10, "128->128->head s32");
    }
    if (on("ws")) {
        printf("--- weight-stationary GEMM, conv4_1_1 shape (232 -> 96, 81920 rows) ---\n");
        PwArgs a{};
        const long M = 81920;
        a.in1 = dev_rand((size_t)M * 136); a.in2 = dev_rand((size_t)M * 96);
        a.w = dev_rand(mfma_packed_floats(136, 96, 96), 0.2f); a.b = dev_rand(96, 0.1f);
        float* out; CK(hipMalloc(&out, (size_t)M * 96 * 4)); a.out = out;
        a.npix = M; a.HW = 320; a.W = 20;
        auto run = [&](auto kern, int threads, unsigned grid, const char* tag) {
            float t = time_us([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, a); });
            printf("ws %-36s grid %4u x %4d  %8.1f us  %6.2f TMAC/s\n", tag, grid, threads, t, (double)M * 232 * 96 / t * 1e-6);
        };
        run(&pw_ws_kernel<136, 96, 96, 1, 2, true, 0, 0>, 768, 256, "UPW1 RS2");
        run(&pw_ws_kernel<136, 96, 96, 1, 2, true, 0, 1>, 768, 256, "UPW1 RS2 no A refill");
        run(&pw_ws_kernel<136, 96, 96, 1, 2, true, 0, 2>, 768, 256, "UPW1 RS2 no stores");
        run(&pw_ws_kernel<136, 96, 96, 1, 2, true, 0, 3>, 768, 256, "UPW1 RS2 neither");
        run(&pw_ws_kernel<136, 96, 96, 2, 4, true, 0, 0>, 768, 256, "UPW2 RS4");
        run(&pw_ws_kernel<136, 96, 96, 2, 4, true, 0, 3>, 768, 256, "UPW2 RS4 neither");
        run(&pw_ws_kernel<136, 96, 96, 3, 2, true, 0, 0>, 256, 256, "UPW3 RS2 (4 waves)");
        run(&pw_ws_kernel<136, 96, 96, 3, 2, true, 0, 0>, 256, 512, "UPW3 RS2 (4 waves) 2x grid");
    }
    if (on("mfma")) {
        printf("--- MFMA pointwise GEMMs ---\n");
        bench_mfma<48, 224, 1, true, false>("48->224 s32", 20480);
        bench_mfma<48, 224, 2, true, false>("48->224 s32", 20480);
        bench_mfma<224, 48, 1, false, true>("224->48 s32 +res", 20480);
        bench_mfma<224, 48, 2, false, true>("224->48 s32 +res", 20480);
        bench_mfma<24, 136, 1, true, false>("24->136 s16", 81920);
        bench_mfma<24, 136, 2, true, false>("24->136 s16", 81920);
        bench_mfma<24, 136, 4, true, false>("24->136 s16", 81920);
        bench_mfma<136, 24, 1, false, true>("136->24 s16 +res", 81920);
        bench_mfma<136, 24, 2, false, true>("136->24 s16 +res", 81920);
        bench_mfma<136, 24, 4, false, true>("136->24 s16 +res", 81920);
        bench_mfma<96, 96, 1, false, false>("96->96 s16", 81920);
        bench_mfma<96, 96, 2, false, false>("96->96 s16", 81920);
        bench_mfma<96, 96, 4, false, false>("96->96 s16", 81920);
    }
    return 0;
}
