// yf_mres_kernels.hip -- BasicResBlock on the matrix cores, fully on chip.
//
//   x -> pw-expand(+ReLU) -> dw3x3(+ReLU) -> pw-project (+ x)       src/model_training/model/yolo_fastest.py:52-66
//
// One workgroup = one TH x TW tile of one frame (whole frame at strides 16/32).  Both pointwise convs are
// v_mfma_f32_16x16x4_f32 GEMMs (exact fp32) whose B operands (weights) sit in VGPR fragments, loaded once per
// 16-channel chunk from a host-packed stream; only the depthwise conv runs on the VALU.  Nothing wide touches HBM:
//
//   HBM --16-B loads--> LDS X[region px][CIN]      (halo'd input tile, also the residual)
//   X --A fragments (kept in VGPRs for the whole kernel)--> MFMA expand --C frag: 4 consecutive pixels of one
//   channel--> bias, ReLU, zero outside the image --4 conflict-free ds_write_b32--> LDS E[4 ch-groups][region px][4 ch]
//   E --9 ds_read_b128 per pixel (4 channels each)--> depthwise FMA chain + ReLU --is directly the A fragment of-->
//   MFMA project (accumulators live in VGPRs across all chunks) --> + bias + residual(X) --> HBM
//
// Fragment conventions (16x16x4 f32): lane l = (r = l & 15, q = l >> 4).
//   A: row r, k = q (per k-step);  B: k = q, col r;  C/D: col r, rows 4q + reg.
//   expand : rows = region pixels (linear index rp = ry*RW + rx), cols = the chunk's 16 channels;
//   project: rows = output pixels (linear index op = oy*TW + ox), k = chunk channel 4q + j for k-step j, cols = cout.
// The k permutations are folded into the host-packed B fragments (mres_pack_weights).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "yf_kernels.h"

namespace yf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ constexpr int mres_ksteps(int K) { return (K / 16) * 4 + ((K % 16) ? 2 : 0); }
__host__ __device__ constexpr int mres_chunk_floats(int cin, int cout)
{
    return mres_ksteps(cin) * 64 + 16 + 9 * 16 + 16 + 4 * ((cout + 15) / 16) * 64;
}

template <int CIN, int CEXP, int COUT, bool RES, int TH, int TW, int NWAVE>
__global__ void __launch_bounds__(NWAVE * 64) mres_kernel(MresArgs a)
{
    constexpr int RH = TH + 2, RW = TW + 2, NRP = RH * RW;
    constexpr int MTR = (NRP + 15) / 16, MTO = (TH * TW) / 16;
    constexpr int MTRW = (MTR + NWAVE - 1) / NWAVE, MTOW = (MTO + NWAVE - 1) / NWAVE;
    constexpr int XP = CIN + 4;                           // X row pitch: conflict-free b128/b64 fragment reads
    constexpr int EPL = ((MTR * 16 + 7) / 8) * 8 + 1;     // pixels per 4-channel plane, == 1 (mod 8): conflict-free writes
    constexpr int KS1 = mres_ksteps(CIN), NB1 = CIN / 16, NT2 = (COUT + 15) / 16, NCH = (CEXP + 15) / 16;
    constexpr int OFF_B1 = KS1 * 64, OFF_WD = OFF_B1 + 16, OFF_BD = OFF_WD + 144, OFF_W2 = OFF_BD + 16;
    constexpr int CHUNK = OFF_W2 + 4 * NT2 * 64;
    static_assert((TH * TW) % 16 == 0 && CIN % 8 == 0 && COUT % 4 == 0, "shape");
    static_assert(!RES || CIN == COUT, "residual needs same shape");
    static_assert(CHUNK == mres_chunk_floats(CIN, COUT), "pack layout");
    static_assert(MTRW * 4 <= 32, "in-image mask bits");
    extern __shared__ __attribute__((aligned(16))) float mres_smem[];
    float* X = mres_smem;                 // [MTR*16][XP]
    float* E = mres_smem + MTR * 16 * XP; // [4][EPL][4]
    float* WL = E + 16 * EPL;             // the block's whole weight stream (NCH chunks + b2), staged once
    constexpr int WFLOATS = (NCH * CHUNK + COUT + 3) & ~3;

    const int b = blockIdx.x;
    const int tx = b % a.tiles_x, ty = (b / a.tiles_x) % a.tiles_y, n = b / (a.tiles_x * a.tiles_y);
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;

    // ---- stage the weight stream (coalesced 16-B copies; every wave reads its fragments from LDS afterwards) ----
    for (int i = threadIdx.x * 4; i < WFLOATS; i += NWAVE * 64 * 4)
        *reinterpret_cast<float4*>(&WL[i]) = *reinterpret_cast<const float4*>(a.wp + i);
    // ---- stage the halo'd input tile (zeros outside the image / beyond the region) ----
    {
        constexpr int C4 = CIN / 4;
        const float* __restrict__ src = a.in + (long)n * a.H * a.W * CIN;
        for (int idx = threadIdx.x; idx < MTR * 16 * C4; idx += NWAVE * 64) {
            const int rp = idx / C4, c4 = idx - rp * C4;
            const int ry = rp / RW, rx = rp - ry * RW;
            const int iy = oy0 - 1 + ry, ix = ox0 - 1 + rx;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rp < NRP && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
                v = *reinterpret_cast<const float4*>(src + ((long)iy * a.W + ix) * CIN + c4 * 4);
            *reinterpret_cast<float4*>(&X[rp * XP + c4 * 4]) = v;
        }
    }
    __syncthreads();

    // ---- expansion A fragments (constant over chunks) and the in-image mask of this lane's 4 C rows ----
    float a1[MTRW][KS1];
    unsigned inmask = 0;
#pragma unroll
    for (int i = 0; i < MTRW; ++i) {
        const int mt = wave + i * NWAVE;
        const int row = (mt < MTR ? mt : 0) * 16 + r;
#pragma unroll
        for (int kb = 0; kb < NB1; ++kb) {
            const float4 t = *reinterpret_cast<const float4*>(&X[row * XP + kb * 16 + 4 * q]);
            a1[i][kb * 4 + 0] = t.x; a1[i][kb * 4 + 1] = t.y; a1[i][kb * 4 + 2] = t.z; a1[i][kb * 4 + 3] = t.w;
        }
        if constexpr (CIN % 16 != 0) {
            const float2 t = *reinterpret_cast<const float2*>(&X[row * XP + NB1 * 16 + 2 * q]);
            a1[i][NB1 * 4 + 0] = t.x; a1[i][NB1 * 4 + 1] = t.y;
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int rp = mt * 16 + 4 * q + reg;
            const int ry = rp / RW, rx = rp - ry * RW;
            const int iy = oy0 - 1 + ry, ix = ox0 - 1 + rx;
            if (mt < MTR && rp < NRP && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) inmask |= 1u << (i * 4 + reg);
        }
    }
    // ---- projection accumulators and the E offsets of this lane's output pixel (as A-fragment row r) ----
    f32x4 acc[MTOW][NT2];
    int rp0[MTOW];
#pragma unroll
    for (int i = 0; i < MTOW; ++i) {
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) acc[i][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int mo = wave + i * NWAVE;
        const int op = (mo < MTO ? mo : 0) * 16 + r;
        const int oy = op / TW, ox = op - oy * TW;
        rp0[i] = (oy + 1) * RW + ox + 1;
    }

#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
        const float* wc = WL + c * CHUNK;
        float w1f[KS1];
#pragma unroll
        for (int s = 0; s < KS1; ++s) w1f[s] = wc[s * 64 + lane];
        const float b1 = wc[OFF_B1 + r];
        float4 wd[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) wd[t] = *reinterpret_cast<const float4*>(wc + OFF_WD + t * 16 + 4 * q);
        const float4 bd = *reinterpret_cast<const float4*>(wc + OFF_BD + 4 * q);
        float w2f[4][NT2];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int nt = 0; nt < NT2; ++nt) w2f[j][nt] = wc[OFF_W2 + (j * NT2 + nt) * 64 + lane];

        // ---- expand: E[ch r][pixels mt*16 + 4q .. +3] ----
#pragma unroll
        for (int i = 0; i < MTRW; ++i) {
            const int mt = wave + i * NWAVE;
            if (mt < MTR) {
                f32x4 cf = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KS1; ++s) cf = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i][s], w1f[s], cf, 0, 0, 0);
                float* dst = E + ((r >> 2) * EPL + mt * 16 + 4 * q) * 4 + (r & 3);  // channel r of pixels 4q..4q+3
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    dst[reg * 4] = (inmask >> (i * 4 + reg)) & 1 ? fmaxf(cf[reg] + b1, 0.f) : 0.f;
            }
        }
        __syncthreads();
        // ---- depthwise 3x3 of channels 4q..4q+3 at output pixel r  ==  A fragment of the projection ----
#pragma unroll
        for (int i = 0; i < MTOW; ++i) {
            const int mo = wave + i * NWAVE;
            if (mo < MTO) {
                const float4* e = reinterpret_cast<const float4*>(E) + q * EPL + rp0[i];
                float d[4] = {bd.x, bd.y, bd.z, bd.w};
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const float4 v = e[(ky - 1) * RW + (kx - 1)];
                        const float4 w = wd[ky * 3 + kx];
                        d[0] = fmaf(v.x, w.x, d[0]);
                        d[1] = fmaf(v.y, w.y, d[1]);
                        d[2] = fmaf(v.z, w.z, d[2]);
                        d[3] = fmaf(v.w, w.w, d[3]);
                    }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float dj = fmaxf(d[j], 0.f);
#pragma unroll
                    for (int nt = 0; nt < NT2; ++nt)
                        acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dj, w2f[j][nt], acc[i][nt], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // ---- epilogue: + bias (+ residual from X), NHWC store; lane holds cout = nt*16 + r of pixels 4q + reg ----
    const float* b2 = WL + NCH * CHUNK;
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt) {
        const int col = nt * 16 + r;
        if (col >= COUT) continue;
        const float bias = b2[col];
#pragma unroll
        for (int i = 0; i < MTOW; ++i) {
            const int mo = wave + i * NWAVE;
            if (mo >= MTO) continue;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int op = mo * 16 + 4 * q + reg;
                const int oy = op / TW, ox = op - oy * TW;
                const int gy = oy0 + oy, gx = ox0 + ox;
                if (gy >= a.H || gx >= a.W) continue;
                float v = acc[i][nt][reg] + bias;
                if constexpr (RES) v += X[((oy + 1) * RW + ox + 1) * XP + col];
                a.out[(((long)n * a.H + gy) * a.W + gx) * COUT + col] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// mres_pc_kernel: the same block with the workgroup's waves split into NWP *producers* (expansion MFMAs of chunk c+1
// into one of two E buffers) and NWC *consumers* (depthwise + projection of chunk c from the other buffer): the matrix
// pipe and the LDS/VALU pipes work on different chunks at the same time and there is ONE barrier per chunk instead of
// two.  The two roles run in different branches of a wave-uniform condition (disjoint register live ranges); both execute
// exactly NCH + 1 barriers.
// ------------------------------------------------------------------------------------------------
template <int CIN, int CEXP, int COUT, bool RES, int TH, int TW, int NWP, int NWC>
__global__ void __launch_bounds__((NWP + NWC) * 64) mres_pc_kernel(MresArgs a)
{
    constexpr int NWAVE = NWP + NWC;
    constexpr int RH = TH + 2, RW = TW + 2, NRP = RH * RW;
    constexpr int MTR = (NRP + 15) / 16, MTO = (TH * TW) / 16;
    constexpr int MTRW = (MTR + NWP - 1) / NWP, MTOW = (MTO + NWC - 1) / NWC;
    constexpr int XP = CIN + 4;
    constexpr int EPL = ((MTR * 16 + 7) / 8) * 8 + 1;
    constexpr int KS1 = mres_ksteps(CIN), NB1 = CIN / 16, NT2 = (COUT + 15) / 16, NCH = (CEXP + 15) / 16;
    constexpr int OFF_B1 = KS1 * 64, OFF_WD = OFF_B1 + 16, OFF_BD = OFF_WD + 144, OFF_W2 = OFF_BD + 16;
    constexpr int CHUNK = OFF_W2 + 4 * NT2 * 64;
    static_assert((TH * TW) % 16 == 0 && CIN % 8 == 0 && COUT % 4 == 0, "shape");
    static_assert(!RES || CIN == COUT, "residual needs same shape");
    static_assert(MTRW * 4 <= 64, "in-image mask bits");
    extern __shared__ __attribute__((aligned(16))) float mres_smem[];
    float* X = mres_smem;                  // [MTR*16][XP]
    float* E = mres_smem + MTR * 16 * XP;  // [2][4][EPL][4]
    float* WL = E + 2 * 16 * EPL;          // weight stream
    constexpr int WFLOATS = (NCH * CHUNK + COUT + 3) & ~3;

    const int b = blockIdx.x;
    const int tx = b % a.tiles_x, ty = (b / a.tiles_x) % a.tiles_y, n = b / (a.tiles_x * a.tiles_y);
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;

    for (int i = threadIdx.x * 4; i < WFLOATS; i += NWAVE * 64 * 4)
        *reinterpret_cast<float4*>(&WL[i]) = *reinterpret_cast<const float4*>(a.wp + i);
    {
        constexpr int C4 = CIN / 4;
        const float* __restrict__ src = a.in + (long)n * a.H * a.W * CIN;
        for (int idx = threadIdx.x; idx < MTR * 16 * C4; idx += NWAVE * 64) {
            const int rp = idx / C4, c4 = idx - rp * C4;
            const int ry = rp / RW, rx = rp - ry * RW;
            const int iy = oy0 - 1 + ry, ix = ox0 - 1 + rx;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rp < NRP && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
                v = *reinterpret_cast<const float4*>(src + ((long)iy * a.W + ix) * CIN + c4 * 4);
            *reinterpret_cast<float4*>(&X[rp * XP + c4 * 4]) = v;
        }
    }
    __syncthreads();

    if (wave < NWP) {
        // ================= producer: expansion of chunk s into E[s & 1] =================
        float a1[MTRW][KS1];
        unsigned long long inmask = 0;
#pragma unroll
        for (int i = 0; i < MTRW; ++i) {
            const int mt = wave + i * NWP;
            const int row = (mt < MTR ? mt : 0) * 16 + r;
#pragma unroll
            for (int kb = 0; kb < NB1; ++kb) {
                const float4 t = *reinterpret_cast<const float4*>(&X[row * XP + kb * 16 + 4 * q]);
                a1[i][kb * 4 + 0] = t.x; a1[i][kb * 4 + 1] = t.y; a1[i][kb * 4 + 2] = t.z; a1[i][kb * 4 + 3] = t.w;
            }
            if constexpr (CIN % 16 != 0) {
                const float2 t = *reinterpret_cast<const float2*>(&X[row * XP + NB1 * 16 + 2 * q]);
                a1[i][NB1 * 4 + 0] = t.x; a1[i][NB1 * 4 + 1] = t.y;
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int rp = mt * 16 + 4 * q + reg;
                const int ry = rp / RW, rx = rp - ry * RW;
                const int iy = oy0 - 1 + ry, ix = ox0 - 1 + rx;
                if (mt < MTR && rp < NRP && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) inmask |= 1ull << (i * 4 + reg);
            }
        }
#pragma unroll 1
        for (int s = 0; s <= NCH; ++s) {
            if (s < NCH) {
                const float* wc = WL + s * CHUNK;
                float* Eb = E + (s & 1) * 16 * EPL;
                float w1f[KS1];
#pragma unroll
                for (int k = 0; k < KS1; ++k) w1f[k] = wc[k * 64 + lane];
                const float b1 = wc[OFF_B1 + r];
#pragma unroll
                for (int i = 0; i < MTRW; ++i) {
                    const int mt = wave + i * NWP;
                    if (mt < MTR) {
                        f32x4 cf = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int k = 0; k < KS1; ++k) cf = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i][k], w1f[k], cf, 0, 0, 0);
                        float* dst = Eb + ((r >> 2) * EPL + mt * 16 + 4 * q) * 4 + (r & 3);
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg)
                            dst[reg * 4] = (inmask >> (i * 4 + reg)) & 1 ? fmaxf(cf[reg] + b1, 0.f) : 0.f;
                    }
                }
            }
            __syncthreads();
        }
    } else {
        // ================= consumer: depthwise + projection of chunk s - 1 from E[(s - 1) & 1] =================
        const int cw = wave - NWP;
        f32x4 acc[MTOW][NT2];
        int rp0[MTOW];
#pragma unroll
        for (int i = 0; i < MTOW; ++i) {
#pragma unroll
            for (int nt = 0; nt < NT2; ++nt) acc[i][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int mo = cw + i * NWC;
            const int op = (mo < MTO ? mo : 0) * 16 + r;
            const int oy = op / TW, ox = op - oy * TW;
            rp0[i] = (oy + 1) * RW + ox + 1;
        }
#pragma unroll 1
        for (int s = 0; s <= NCH; ++s) {
            if (s >= 1) {
                const float* wc = WL + (s - 1) * CHUNK;
                const float* Eb = E + ((s - 1) & 1) * 16 * EPL;
                float4 wd[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) wd[t] = *reinterpret_cast<const float4*>(wc + OFF_WD + t * 16 + 4 * q);
                const float4 bd = *reinterpret_cast<const float4*>(wc + OFF_BD + 4 * q);
                float w2f[4][NT2];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int nt = 0; nt < NT2; ++nt) w2f[j][nt] = wc[OFF_W2 + (j * NT2 + nt) * 64 + lane];
#pragma unroll
                for (int i = 0; i < MTOW; ++i) {
                    const int mo = cw + i * NWC;
                    if (mo < MTO) {
                        const float4* e = reinterpret_cast<const float4*>(Eb) + q * EPL + rp0[i];
                        float d[4] = {bd.x, bd.y, bd.z, bd.w};
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                            for (int kx = 0; kx < 3; ++kx) {
                                const float4 v = e[(ky - 1) * RW + (kx - 1)];
                                const float4 w = wd[ky * 3 + kx];
                                d[0] = fmaf(v.x, w.x, d[0]); d[1] = fmaf(v.y, w.y, d[1]);
                                d[2] = fmaf(v.z, w.z, d[2]); d[3] = fmaf(v.w, w.w, d[3]);
                            }
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float dj = fmaxf(d[j], 0.f);
#pragma unroll
                            for (int nt = 0; nt < NT2; ++nt)
                                acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dj, w2f[j][nt], acc[i][nt], 0, 0, 0);
                        }
                    }
                }
            }
            __syncthreads();
        }
        const float* b2 = WL + NCH * CHUNK;
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) {
            const int col = nt * 16 + r;
            if (col >= COUT) continue;
            const float bias = b2[col];
#pragma unroll
            for (int i = 0; i < MTOW; ++i) {
                const int mo = cw + i * NWC;
                if (mo >= MTO) continue;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int op = mo * 16 + 4 * q + reg;
                    const int oy = op / TW, ox = op - oy * TW;
                    const int gy = oy0 + oy, gx = ox0 + ox;
                    if (gy >= a.H || gx >= a.W) continue;
                    float v = acc[i][nt][reg] + bias;
                    if constexpr (RES) v += X[((oy + 1) * RW + ox + 1) * XP + col];
                    a.out[(((long)n * a.H + gy) * a.W + gx) * COUT + col] = v;
                }
            }
        }
    }
}

template <int CIN, int CEXP, int COUT, bool RES, int TH, int TW, int NWP, int NWC>
static int launch_mres_pc_t(MresArgs a, int N, hipStream_t s)
{
    a.tiles_y = (a.H + TH - 1) / TH;
    a.tiles_x = (a.W + TW - 1) / TW;
    constexpr int MTR = ((TH + 2) * (TW + 2) + 15) / 16;
    constexpr size_t lds = ((size_t)MTR * 16 * (CIN + 4) + 2 * 16 * (((MTR * 16 + 7) / 8) * 8 + 1) +
                            ((((CEXP + 15) / 16) * mres_chunk_floats(CIN, COUT) + COUT + 3) & ~3)) * sizeof(float);
    static_assert(lds <= 160 * 1024, "LDS");
    static bool attr_done = false;
    if (lds > 64 * 1024 && !attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(mres_pc_kernel<CIN, CEXP, COUT, RES, TH, TW, NWP, NWC>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return -2;
        attr_done = true;
    }
    hipLaunchKernelGGL((mres_pc_kernel<CIN, CEXP, COUT, RES, TH, TW, NWP, NWC>), dim3((unsigned)(N * a.tiles_y * a.tiles_x)),
                       dim3((NWP + NWC) * 64), lds, s, a);
    return 0;
}

template <int CIN, int CEXP, int COUT, bool RES, int TH, int TW, int NWAVE>
static int launch_mres_t(MresArgs a, int N, hipStream_t s)
{
    a.tiles_y = (a.H + TH - 1) / TH;
    a.tiles_x = (a.W + TW - 1) / TW;
    constexpr int MTR = ((TH + 2) * (TW + 2) + 15) / 16;
    constexpr size_t lds = ((size_t)MTR * 16 * (CIN + 4) + 16 * (((MTR * 16 + 7) / 8) * 8 + 1) +
                            ((((CEXP + 15) / 16) * mres_chunk_floats(CIN, COUT) + COUT + 3) & ~3)) * sizeof(float);
    static_assert(lds <= 160 * 1024, "LDS");
    static bool attr_done = false;
    if (lds > 64 * 1024 && !attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(mres_kernel<CIN, CEXP, COUT, RES, TH, TW, NWAVE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return -2;
        attr_done = true;
    }
    hipLaunchKernelGGL((mres_kernel<CIN, CEXP, COUT, RES, TH, TW, NWAVE>), dim3((unsigned)(N * a.tiles_y * a.tiles_x)),
                       dim3(NWAVE * 64), lds, s, a);
    return 0;
}

//      (cin, cexp, cout, residual, TH, TW, producer waves (0 = two-barrier kernel), waves / consumer waves)
// Producer/consumer pays where one workgroup owns the CU anyway (strides 16, 32); at stride 8 its second E buffer
// halves the workgroups per CU and it is slower (tools/kbench.hip mrespc).
#define YF_MRES_SHAPES(MR)                                                            \
    MR(8, 48, 16, false, 16, 20, 0, 4)   /* conv3_2/3_3/3_4          @ H/8  */         \
    MR(16, 96, 16, true, 16, 20, 0, 8)   /* res3_3 .. res3_6         @ H/8  */         \
    MR(24, 136, 24, true, 16, 20, 6, 10) /* res4_1 .. res4_4         @ H/16 */         \
    MR(48, 224, 48, true, 8, 10, 4, 5)   /* res5_1 .. res5_5         @ H/32 */

template <int CIN, int CEXP, int COUT, bool RES, int TH, int TW, int NWP, int NW>
static int launch_mres_any(const MresArgs& a, int N, hipStream_t s)
{
    if constexpr (NWP == 0) return launch_mres_t<CIN, CEXP, COUT, RES, TH, TW, NW>(a, N, s);
    else return launch_mres_pc_t<CIN, CEXP, COUT, RES, TH, TW, NWP, NW>(a, N, s);
}

int launch_mres(int cin, int cexp, int cout, bool res, const MresArgs& a, int N, hipStream_t s)
{
#define MR(ci, ce, co, rs, th, tw, np, nw) \
    if (cin == ci && cexp == ce && cout == co && res == rs) return launch_mres_any<ci, ce, co, rs, th, tw, np, nw>(a, N, s);
    YF_MRES_SHAPES(MR)
#undef MR
    return -1;
}

bool mres_has_kernel(int cin, int cexp, int cout, bool res)
{
#define MR(ci, ce, co, rs, th, tw, np, nw) \
    if (cin == ci && cexp == ce && cout == co && res == rs) return true;
    YF_MRES_SHAPES(MR)
#undef MR
    return false;
}

// Host-side weight stream of one block: NCH chunks of [W1 frags | b1 | wd 9x16 | bd | W2 frags], then b2.
size_t mres_packed_floats(int cin, int cexp, int cout) { return (((size_t)((cexp + 15) / 16) * mres_chunk_floats(cin, cout) + cout) + 3) & ~(size_t)3; }

void mres_pack_weights(const float* w1 /*[cin][cexp]*/, const float* b1, const float* wd /*[9][cexp]*/, const float* bd,
                       const float* w2 /*[cexp][cout]*/, const float* b2, int cin, int cexp, int cout, float* out)
{
    const int KS1 = mres_ksteps(cin), NB1 = cin / 16, NT2 = (cout + 15) / 16, NCH = (cexp + 15) / 16;
    const int CH = mres_chunk_floats(cin, cout);
    for (int c = 0; c < NCH; ++c) {
        float* o = out + (size_t)c * CH;
        auto ch_ok = [&](int ch) { return c * 16 + ch < cexp; };
        for (int s = 0; s < KS1; ++s)
            for (int lane = 0; lane < 64; ++lane) {
                const int q = lane >> 4, nn = lane & 15;
                const int kb = s / 4, j = s % 4;
                const int k = kb < NB1 ? kb * 16 + 4 * q + j : NB1 * 16 + 2 * q + j;  // trailing 8-block: j in {0,1}
                o[s * 64 + lane] = ch_ok(nn) ? w1[(size_t)k * cexp + c * 16 + nn] : 0.f;
            }
        o += KS1 * 64;
        for (int ch = 0; ch < 16; ++ch) o[ch] = ch_ok(ch) ? b1[c * 16 + ch] : 0.f;
        o += 16;
        for (int t = 0; t < 9; ++t)
            for (int ch = 0; ch < 16; ++ch) o[t * 16 + ch] = ch_ok(ch) ? wd[(size_t)t * cexp + c * 16 + ch] : 0.f;
        o += 144;
        for (int ch = 0; ch < 16; ++ch) o[ch] = ch_ok(ch) ? bd[c * 16 + ch] : 0.f;
        o += 16;
        for (int j = 0; j < 4; ++j)
            for (int nt = 0; nt < NT2; ++nt)
                for (int lane = 0; lane < 64; ++lane) {
                    const int q = lane >> 4, nn = nt * 16 + (lane & 15), ch = 4 * q + j;
                    o[(j * NT2 + nt) * 64 + lane] = (ch_ok(ch) && nn < cout) ? w2[(size_t)(c * 16 + ch) * cout + nn] : 0.f;
                }
    }
    for (int i = 0; i < cout; ++i) out[(size_t)NCH * CH + i] = b2[i];
}

}  // namespace yf
