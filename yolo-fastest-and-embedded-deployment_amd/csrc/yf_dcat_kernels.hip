// yf_dcat_kernels.hip -- the large head's entry in ONE launch (fusion level 2):
//
//     deconv5_1 = ConvTranspose2d(96, 96, k=2, s=2) + BN + ReLU on conv5_2            src/model_training/model/yolo_fastest.py:138, :208
//     conv4_1_1 = 1x1 conv 232 -> 96 + BN + ReLU over torch.cat((conv4_2, deconv5_1), 1)                          :141, :209-211
//
// A 2x2 stride-2 transposed conv is four independent 96x96 GEMMs, one per output-pixel parity ("quadrant" (dy, dx)): output pixel
// (2y + dy, 2x + dx) sees input pixel (y, x) through tap (dy, dx) only.  So a 16-pixel M-tile of conv5_2 pixels yields, per
// quadrant, the 96 deconv channels of 16 OUTPUT pixels -- and with the weights as the MFMA's A operand a lane (r, q) ends up holding
// channels nt*16 + 4q .. +3 of output pixel r for every n-tile nt, which is exactly the B fragment conv4_1_1's GEMM wants for its
// k-block nt.  The deconv result therefore never leaves the registers: relu(acc + bias) of n-tile kb IS conv4_1_1's operand for the
// k-steps of channels 136 + kb*16 .. +15; the 136 conv4_2 channels of the same 16 output pixels come straight from HBM.
//
// Workgroup = 4 waves, one per quadrant (one wave per SIMD; the MFMA work of a frame, 9840 MFMAs, is the floor: 78.7 k pipe cycles):
//   * a wave keeps its quadrant's deconv weights in registers for the whole launch (144 fragments);
//   * conv4_1_1's weights (89 KB) are staged in LDS once per workgroup, packed so that one ds_read_b128 feeds the four k-steps of a
//     16-channel block of one n-tile;
//   * work item = up to five M-tiles (80 conv5_2 pixels = one stride-32 frame of the 320x256 net; larger frames are several items);
//     the grid is persistent over the items.  Operand fragments of the next M-tile are requested while the current one is computed
//     (the "A ring" of pw_ws_kernel).  No barrier after the weight staging.
// Arithmetic and its order are those of the two launches this replaces (pw_ws_kernel, OMODE 2 and the two-source GEMM): acc = 0,
// k-steps in order, + bias, ReLU -- bitwise the same conv4_1_1 tensor.  DT_F32 here; DT_F16X3: dcat_x3_kernel, DT_F16: dcat_h_kernel below.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "yf_kernels.h"

namespace yf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int DC_CIN = 96, DC_N = 96, DC_SKIP = 136;     // deconv 96 -> 96; conv4_1_1 over 136 + 96 channels -> 96
constexpr int DC_NT = DC_N / 16;                         // 6 n-tiles (both GEMMs)
constexpr int DC_KB_D = DC_CIN / 16;                     // 6 k-blocks of the deconv
constexpr int DC_KB_S = DC_SKIP / 16;                    // 8 full k-blocks of the skip source, then 8 channels = 2 k-steps
constexpr int DC_OFF_TAIL = DC_KB_S * DC_NT * 64 * 4;    // floats: [kb][nt][lane][4 j]
constexpr int DC_OFF_K2 = DC_OFF_TAIL + 2 * DC_NT * 64;  //         [j][nt][lane]
constexpr int DC_OFF_BD = DC_OFF_K2 + DC_KB_D * DC_NT * 64 * 4;
constexpr int DC_OFF_BC = DC_OFF_BD + DC_N;
constexpr int DC_WFLOATS = DC_OFF_BC + DC_N;             // 22464 floats = 89856 B
constexpr int DC_MT = 5;                                 // M-tiles per work item (small batches: 1, so that a frame spreads over five workgroups)
}  // namespace

struct DcatArgs {
    const float* x;     // conv5_2, NHWC [N, h, w, 96]
    const float* skip;  // conv4_2, NHWC [N, 2h, 2w, 136]
    const float* wd;    // deconv fragments: 4 quadrants x mfma_pack_weights(96, 0, 96) = [k-step][n-tile][lane]
    const float* wc;    // conv4_1_1 stream for LDS (dcat_pack_weights), biases of both layers at its end
    float* out;         // conv4_1_1, NHWC [N, 2h, 2w, 96]
    int h, w;           // conv5_2 frame
    int nitems, items_per_frame;
};

template <int MT>   // M-tiles per work item
__global__ void __launch_bounds__(256) dcat_kernel(DcatArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float dc_smem[];
    float* WL = dc_smem;
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const int qd = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), dy = qd >> 1, dx = qd & 1;

    LdsStage<DC_WFLOATS, 256> stage;   // conv4_1_1's stream: requested first, written to LDS after the deconv weights are requested too
    stage.issue(a.wc);
    // ---- this wave's deconv weights: 6 n-tiles x 24 k-steps, register-resident for the whole launch ----
    float dw[DC_NT][DC_KB_D * 4];
    {
        const float* w = a.wd + (size_t)qd * (DC_KB_D * 4) * DC_NT * 64 + lane;
#pragma unroll
        for (int s = 0; s < DC_KB_D * 4; ++s)
#pragma unroll
            for (int nt = 0; nt < DC_NT; ++nt) dw[nt][s] = w[(s * DC_NT + nt) * 64];
    }
    stage.commit(WL);
    __syncthreads();
    const float4* WL4 = reinterpret_cast<const float4*>(WL);

    const int npx = a.h * a.w, ow = 2 * a.w;
    const long ntile = (long)((a.nitems - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x) * MT;   // this workgroup's M-tiles
    // M-tile t of this workgroup: item blockIdx.x + (t / 5) * gridDim.x, tile t % 5 -> operand pointers of this lane's pixel
    struct Tile { const float* sp; const float* cp; long o; bool ok; };
    auto tile = [&](long t) {
        const long it = (long)blockIdx.x + (t / MT) * gridDim.x;
        const int n = (int)(it / a.items_per_frame), chunk = (int)(it - (long)n * a.items_per_frame);
        const int p = chunk * (MT * 16) + (int)(t % MT) * 16 + r;
        Tile T;
        T.ok = p < npx;
        const int pc = T.ok ? p : npx - 1;   // clamp the loads, guard the stores
        const int y = pc / a.w, x = pc - y * a.w;
        const long opix = (long)n * 4 * npx + (long)(2 * y + dy) * ow + 2 * x + dx;
        T.sp = a.x + ((long)n * npx + pc) * DC_CIN + 4 * q;
        T.cp = a.skip + opix * DC_SKIP;
        T.o = opix * DC_N + 4 * q;
        return T;
    };

    if (ntile <= 0) return;
    float4 sf[DC_KB_D], cf[DC_KB_S];
    float2 ct;
    Tile cur = tile(0);
#pragma unroll
    for (int kb = 0; kb < DC_KB_D; ++kb) sf[kb] = *reinterpret_cast<const float4*>(cur.sp + kb * 16);
#pragma unroll
    for (int kb = 0; kb < DC_KB_S; ++kb) cf[kb] = *reinterpret_cast<const float4*>(cur.cp + kb * 16 + 4 * q);
    ct = *reinterpret_cast<const float2*>(cur.cp + DC_KB_S * 16 + 2 * q);

#pragma unroll 1
    for (long t = 0; t < ntile; ++t) {
        const Tile nxt = tile(t + 1 < ntile ? t + 1 : t);
        // ---- deconv of this quadrant: 6 n-tiles x 24 k-steps; each fragment piece is re-requested for the next M-tile once consumed ----
        f32x4 dacc[DC_NT];
#pragma unroll
        for (int nt = 0; nt < DC_NT; ++nt) dacc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < DC_KB_D; ++kb) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int nt = 0; nt < DC_NT; ++nt)
                    dacc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dw[nt][kb * 4 + j], ((const float*)&sf[kb])[j], dacc[nt], 0, 0, 0);
            sf[kb] = *reinterpret_cast<const float4*>(nxt.sp + kb * 16);
        }
        // + bias, ReLU: lane (r, q) now holds deconv channels nt*16 + 4q .. +3 of ITS output pixel = conv4_1_1's fragment of k-block nt
        float dv[DC_NT][4];
#pragma unroll
        for (int nt = 0; nt < DC_NT; ++nt) {
            const float4 b = *reinterpret_cast<const float4*>(WL + DC_OFF_BD + nt * 16 + 4 * q);
            dv[nt][0] = fmaxf(dacc[nt][0] + b.x, 0.f); dv[nt][1] = fmaxf(dacc[nt][1] + b.y, 0.f);
            dv[nt][2] = fmaxf(dacc[nt][2] + b.z, 0.f); dv[nt][3] = fmaxf(dacc[nt][3] + b.w, 0.f);
        }
        // ---- conv4_1_1: source 1 = conv4_2 (8 blocks + 2 k-steps), source 2 = the deconv result in registers (6 blocks).  15 weight
        // groups from LDS, software-pipelined one group ahead; a scheduling barrier per group keeps the compiler from hoisting ALL the
        // LDS reads to the top (it did: 512 VGPRs and 70 spilled) ----
        f32x4 cacc[DC_NT];
#pragma unroll
        for (int nt = 0; nt < DC_NT; ++nt) cacc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        constexpr int NG = DC_KB_S + 1 + DC_KB_D;   // 8 skip blocks, the 8-channel tail, 6 deconv blocks
        auto ldw = [&](int g, float4 (&w4)[DC_NT]) {
            if (g < DC_KB_S) {
#pragma unroll
                for (int nt = 0; nt < DC_NT; ++nt) w4[nt] = WL4[(g * DC_NT + nt) * 64 + lane];
            } else if (g == DC_KB_S) {
#pragma unroll
                for (int nt = 0; nt < DC_NT; ++nt) {
                    w4[nt].x = WL[DC_OFF_TAIL + nt * 64 + lane];
                    w4[nt].y = WL[DC_OFF_TAIL + (DC_NT + nt) * 64 + lane];
                }
            } else {
#pragma unroll
                for (int nt = 0; nt < DC_NT; ++nt) w4[nt] = WL4[DC_OFF_K2 / 4 + ((g - DC_KB_S - 1) * DC_NT + nt) * 64 + lane];
            }
        };
        float4 wbuf[2][DC_NT];
        ldw(0, wbuf[0]);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) ldw(g + 1, wbuf[(g + 1) & 1]);
            const float4(&w4)[DC_NT] = wbuf[g & 1];
            if (g < DC_KB_S) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int nt = 0; nt < DC_NT; ++nt)
                        cacc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(((const float*)&w4[nt])[j], ((const float*)&cf[g])[j], cacc[nt], 0, 0, 0);
                cf[g] = *reinterpret_cast<const float4*>(nxt.cp + g * 16 + 4 * q);
            } else if (g == DC_KB_S) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int nt = 0; nt < DC_NT; ++nt)
                        cacc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(((const float*)&w4[nt])[j], ((const float*)&ct)[j], cacc[nt], 0, 0, 0);
                ct = *reinterpret_cast<const float2*>(nxt.cp + DC_KB_S * 16 + 2 * q);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int nt = 0; nt < DC_NT; ++nt)
                        cacc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(((const float*)&w4[nt])[j], dv[g - DC_KB_S - 1][j], cacc[nt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (cur.ok) {
#pragma unroll
            for (int nt = 0; nt < DC_NT; ++nt) {
                const float4 b = *reinterpret_cast<const float4*>(WL + DC_OFF_BC + nt * 16 + 4 * q);
                *reinterpret_cast<float4*>(a.out + cur.o + nt * 16) =
                    make_float4(fmaxf(cacc[nt][0] + b.x, 0.f), fmaxf(cacc[nt][1] + b.y, 0.f), fmaxf(cacc[nt][2] + b.z, 0.f), fmaxf(cacc[nt][3] + b.w, 0.f));
            }
        }
        cur = nxt;
    }
}

// ------------------------------------------------------------------------------------------------
// dcat_x3_kernel: the same launch for DT_F16X3 engines (fp32 storage, split-operand fp16 MFMAs; yf_kernels.h): every operand a = hi + lo
// (two fp16 halves, lo = rne(a - hi)), a k-block of 16 channels issues w_lo a_hi + w_hi a_lo + w_hi a_hi on v_mfma_f32_16x16x16_f16 -- the
// order of pw_ws_x3_kernel, whose two launches this replaces bit for bit.  A wave keeps its quadrant's deconv weights as hi and lo
// fragments (the same 144 registers as the fp32 fragments); conv4_1_1's stream in LDS holds, per (16-channel block, n-tile, lane), the
// 16-byte record [hi4 | lo4], so one ds_read_b128 still feeds a whole k-block; activations are split where they are consumed (the
// conv4_2 fragments once per M-tile, the deconv result once, in registers).
// ------------------------------------------------------------------------------------------------
namespace {
constexpr int DX_NG = DC_KB_S + 1 + DC_KB_D;                 // 15 k-blocks of conv4_1_1: 8 skip, the 8-channel tail, 6 deconv
constexpr int DX_OFF_BD = DX_NG * DC_NT * 64 * 4;            // floats: [block][nt][lane][hi4 | lo4] = 4 floats per lane
constexpr int DX_OFF_BC = DX_OFF_BD + DC_N;
constexpr int DX_WFLOATS = DX_OFF_BC + DC_N;                 // 23232 floats = 92928 B
}  // namespace

template <int MT>
__global__ void __launch_bounds__(256) dcat_x3_kernel(DcatArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float dc_smem[];
    float* WL = dc_smem;
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const int qd = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), dy = qd >> 1, dx = qd & 1;

    LdsStage<DX_WFLOATS, 256> stage;
    stage.issue(a.wc);
    // this wave's deconv weights (mfma_pack_weights_x3 per quadrant: [hi: k-block][n-tile][lane] f16x4, then lo the same)
    f16x4 dh[DC_NT][DC_KB_D], dl[DC_NT][DC_KB_D];
    {
        const f16x4* w = reinterpret_cast<const f16x4*>(a.wd) + (size_t)qd * 2 * DC_KB_D * DC_NT * 64 + lane;
#pragma unroll
        for (int kb = 0; kb < DC_KB_D; ++kb)
#pragma unroll
            for (int nt = 0; nt < DC_NT; ++nt) {
                dh[nt][kb] = w[(kb * DC_NT + nt) * 64];
                dl[nt][kb] = w[((DC_KB_D + kb) * DC_NT + nt) * 64];
            }
    }
    stage.commit(WL);
    __syncthreads();
    const float4* WL4 = reinterpret_cast<const float4*>(WL);

    const int npx = a.h * a.w, ow = 2 * a.w;
    const long ntile = (long)((a.nitems - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x) * MT;
    struct Tile { const float* sp; const float* cp; long o; bool ok; };
    auto tile = [&](long t) {
        const long it = (long)blockIdx.x + (t / MT) * gridDim.x;
        const int n = (int)(it / a.items_per_frame), chunk = (int)(it - (long)n * a.items_per_frame);
        const int p = chunk * (MT * 16) + (int)(t % MT) * 16 + r;
        Tile T;
        T.ok = p < npx;
        const int pc = T.ok ? p : npx - 1;
        const int y = pc / a.w, x = pc - y * a.w;
        const long opix = (long)n * 4 * npx + (long)(2 * y + dy) * ow + 2 * x + dx;
        T.sp = a.x + ((long)n * npx + pc) * DC_CIN + 4 * q;
        T.cp = a.skip + opix * DC_SKIP;
        T.o = opix * DC_N + 4 * q;
        return T;
    };
    if (ntile <= 0) return;
    float4 sf[DC_KB_D], cf[DC_KB_S];
    float2 ct;
    Tile cur = tile(0);
#pragma unroll
    for (int kb = 0; kb < DC_KB_D; ++kb) sf[kb] = *reinterpret_cast<const float4*>(cur.sp + kb * 16);
#pragma unroll
    for (int kb = 0; kb < DC_KB_S; ++kb) cf[kb] = *reinterpret_cast<const float4*>(cur.cp + kb * 16 + 4 * q);
    ct = *reinterpret_cast<const float2*>(cur.cp + DC_KB_S * 16 + 2 * q);

    // w_lo a_hi, w_hi a_lo, w_hi a_hi for all six n-tiles of one k-block (pw_ws_x3_kernel's mac)
    auto mac6 = [&](f32x4 (&acc)[DC_NT], const f16x4 (&wh)[DC_NT], const f16x4 (&wl)[DC_NT], float x0, float x1, float x2, float x3) {
        f16x4 ah, al;
        split_f16x4(x0, x1, x2, x3, ah, al);
#pragma unroll
        for (int nt = 0; nt < DC_NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x16f16(wl[nt], ah, acc[nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < DC_NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x16f16(wh[nt], al, acc[nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < DC_NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x16f16(wh[nt], ah, acc[nt], 0, 0, 0);
    };

#pragma unroll 1
    for (long t = 0; t < ntile; ++t) {
        const Tile nxt = tile(t + 1 < ntile ? t + 1 : t);
        f32x4 dacc[DC_NT];
#pragma unroll
        for (int nt = 0; nt < DC_NT; ++nt) dacc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < DC_KB_D; ++kb) {
            f16x4 wh[DC_NT], wl[DC_NT];
#pragma unroll
            for (int nt = 0; nt < DC_NT; ++nt) { wh[nt] = dh[nt][kb]; wl[nt] = dl[nt][kb]; }
            mac6(dacc, wh, wl, sf[kb].x, sf[kb].y, sf[kb].z, sf[kb].w);
            sf[kb] = *reinterpret_cast<const float4*>(nxt.sp + kb * 16);
        }
        float dv[DC_NT][4];
#pragma unroll
        for (int nt = 0; nt < DC_NT; ++nt) {
            const float4 b = *reinterpret_cast<const float4*>(WL + DX_OFF_BD + nt * 16 + 4 * q);
            dv[nt][0] = fmaxf(dacc[nt][0] + b.x, 0.f); dv[nt][1] = fmaxf(dacc[nt][1] + b.y, 0.f);
            dv[nt][2] = fmaxf(dacc[nt][2] + b.z, 0.f); dv[nt][3] = fmaxf(dacc[nt][3] + b.w, 0.f);
        }
        f32x4 cacc[DC_NT];
#pragma unroll
        for (int nt = 0; nt < DC_NT; ++nt) cacc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        float4 wbuf[2][DC_NT];
#pragma unroll
        for (int nt = 0; nt < DC_NT; ++nt) wbuf[0][nt] = WL4[nt * 64 + lane];
#pragma unroll
        for (int g = 0; g < DX_NG; ++g) {
            if (g + 1 < DX_NG) {
#pragma unroll
                for (int nt = 0; nt < DC_NT; ++nt) wbuf[(g + 1) & 1][nt] = WL4[((g + 1) * DC_NT + nt) * 64 + lane];
            }
            f16x4 wh[DC_NT], wl[DC_NT];
#pragma unroll
            for (int nt = 0; nt < DC_NT; ++nt) {
                const float4 w = wbuf[g & 1][nt];
                wh[nt] = __builtin_bit_cast(f16x4, make_float2(w.x, w.y));
                wl[nt] = __builtin_bit_cast(f16x4, make_float2(w.z, w.w));
            }
            if (g < DC_KB_S) {
                mac6(cacc, wh, wl, cf[g].x, cf[g].y, cf[g].z, cf[g].w);
                cf[g] = *reinterpret_cast<const float4*>(nxt.cp + g * 16 + 4 * q);
            } else if (g == DC_KB_S) {
                mac6(cacc, wh, wl, ct.x, ct.y, 0.f, 0.f);
                ct = *reinterpret_cast<const float2*>(nxt.cp + DC_KB_S * 16 + 2 * q);
            } else {
                const int kb = g - DC_KB_S - 1;
                mac6(cacc, wh, wl, dv[kb][0], dv[kb][1], dv[kb][2], dv[kb][3]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (cur.ok) {
#pragma unroll
            for (int nt = 0; nt < DC_NT; ++nt) {
                const float4 b = *reinterpret_cast<const float4*>(WL + DX_OFF_BC + nt * 16 + 4 * q);
                *reinterpret_cast<float4*>(a.out + cur.o + nt * 16) =
                    make_float4(fmaxf(cacc[nt][0] + b.x, 0.f), fmaxf(cacc[nt][1] + b.y, 0.f), fmaxf(cacc[nt][2] + b.z, 0.f), fmaxf(cacc[nt][3] + b.w, 0.f));
            }
        }
        cur = nxt;
    }
}

// ------------------------------------------------------------------------------------------------
// dcat_h_kernel: the same launch for DT_F16 engines (fp16 storage, single fp16 MFMA operands): the two launches it replaces (pw_ws_kernel
// <half_t>, OMODE 2, and the two-source GEMM) round deconv5_1 to fp16 on its way through HBM; here the ReLU'd deconv result is rounded to
// fp16 once, in registers, as conv4_1_1's operand -- the same values, so the conv4_1_1 tensor is bitwise the one of the two launches
// (k-steps in the same order).  A wave holds its quadrant's 36 deconv fragments (72 VGPRs), the conv4_1_1 stream in LDS is one 8-byte
// record per (16-channel block, n-tile, lane) (47 KB), operands are the 8-byte fragments as they lie in HBM: no conversions on the way in.
// ------------------------------------------------------------------------------------------------
namespace {
constexpr int DH_OFF_BD = DX_NG * DC_NT * 64 * 2;            // floats: [block][nt][lane] f16x4 = 2 floats per lane
constexpr int DH_OFF_BC = DH_OFF_BD + DC_N;
constexpr int DH_WFLOATS = DH_OFF_BC + DC_N;                 // 11712 floats = 46848 B
}  // namespace

template <int MT>
__global__ void __launch_bounds__(256) dcat_h_kernel(DcatArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float dc_smem[];
    float* WL = dc_smem;
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const int qd = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), dy = qd >> 1, dx = qd & 1;
    const half_t* const xh = reinterpret_cast<const half_t*>(a.x);
    const half_t* const skh = reinterpret_cast<const half_t*>(a.skip);
    half_t* const outh = reinterpret_cast<half_t*>(a.out);

    LdsStage<DH_WFLOATS, 256> stage;
    stage.issue(a.wc);
    f16x4 dw[DC_NT][DC_KB_D];   // this wave's deconv weights (mfma_pack_weights_f16 per quadrant: [k-block][n-tile][lane] f16x4)
    {
        const f16x4* w = reinterpret_cast<const f16x4*>(a.wd) + (size_t)qd * DC_KB_D * DC_NT * 64 + lane;
#pragma unroll
        for (int kb = 0; kb < DC_KB_D; ++kb)
#pragma unroll
            for (int nt = 0; nt < DC_NT; ++nt) dw[nt][kb] = w[(kb * DC_NT + nt) * 64];
    }
    stage.commit(WL);
    __syncthreads();
    const f16x4* WL2 = reinterpret_cast<const f16x4*>(WL);

    const int npx = a.h * a.w, ow = 2 * a.w;
    const long ntile = (long)((a.nitems - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x) * MT;
    struct Tile { const half_t* sp; const half_t* cp; long o; bool ok; };
    auto tile = [&](long t) {
        const long it = (long)blockIdx.x + (t / MT) * gridDim.x;
        const int n = (int)(it / a.items_per_frame), chunk = (int)(it - (long)n * a.items_per_frame);
        const int p = chunk * (MT * 16) + (int)(t % MT) * 16 + r;
        Tile T;
        T.ok = p < npx;
        const int pc = T.ok ? p : npx - 1;
        const int y = pc / a.w, x = pc - y * a.w;
        const long opix = (long)n * 4 * npx + (long)(2 * y + dy) * ow + 2 * x + dx;
        T.sp = xh + ((long)n * npx + pc) * DC_CIN + 4 * q;
        T.cp = skh + opix * DC_SKIP;
        T.o = opix * DC_N + 4 * q;
        return T;
    };
    if (ntile <= 0) return;
    f16x4 sf[DC_KB_D], cf[DC_KB_S];
    f16x2 ct;
    Tile cur = tile(0);
#pragma unroll
    for (int kb = 0; kb < DC_KB_D; ++kb) sf[kb] = *reinterpret_cast<const f16x4*>(cur.sp + kb * 16);
#pragma unroll
    for (int kb = 0; kb < DC_KB_S; ++kb) cf[kb] = *reinterpret_cast<const f16x4*>(cur.cp + kb * 16 + 4 * q);
    ct = *reinterpret_cast<const f16x2*>(cur.cp + DC_KB_S * 16 + 2 * q);
    const f16x4 zero4 = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};

#pragma unroll 1
    for (long t = 0; t < ntile; ++t) {
        const Tile nxt = tile(t + 1 < ntile ? t + 1 : t);
        f32x4 dacc[DC_NT];
#pragma unroll
        for (int nt = 0; nt < DC_NT; ++nt) dacc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < DC_KB_D; ++kb) {
#pragma unroll
            for (int nt = 0; nt < DC_NT; ++nt) dacc[nt] = __builtin_amdgcn_mfma_f32_16x16x16f16(dw[nt][kb], sf[kb], dacc[nt], 0, 0, 0);
            sf[kb] = *reinterpret_cast<const f16x4*>(nxt.sp + kb * 16);
        }
        f16x4 dv[DC_NT];   // relu(deconv + bias), rounded to fp16 as the two-launch plan stores it
#pragma unroll
        for (int nt = 0; nt < DC_NT; ++nt) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(WL + DH_OFF_BD + nt * 16 + 4 * q);
            dv[nt] = __builtin_elementwise_max(__builtin_convertvector(dacc[nt] + b, f16x4), zero4);
        }
        f32x4 cacc[DC_NT];
#pragma unroll
        for (int nt = 0; nt < DC_NT; ++nt) cacc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        f16x4 wbuf[2][DC_NT];
#pragma unroll
        for (int nt = 0; nt < DC_NT; ++nt) wbuf[0][nt] = WL2[nt * 64 + lane];
#pragma unroll
        for (int g = 0; g < DX_NG; ++g) {
            if (g + 1 < DX_NG) {
#pragma unroll
                for (int nt = 0; nt < DC_NT; ++nt) wbuf[(g + 1) & 1][nt] = WL2[((g + 1) * DC_NT + nt) * 64 + lane];
            }
            f16x4 op;
            if (g < DC_KB_S) {
                op = cf[g];
                cf[g] = *reinterpret_cast<const f16x4*>(nxt.cp + g * 16 + 4 * q);
            } else if (g == DC_KB_S) {
                op = f16x4{ct[0], ct[1], (half_t)0.f, (half_t)0.f};
                ct = *reinterpret_cast<const f16x2*>(nxt.cp + DC_KB_S * 16 + 2 * q);
            } else {
                op = dv[g - DC_KB_S - 1];
            }
#pragma unroll
            for (int nt = 0; nt < DC_NT; ++nt) cacc[nt] = __builtin_amdgcn_mfma_f32_16x16x16f16(wbuf[g & 1][nt], op, cacc[nt], 0, 0, 0);
        }
        if (cur.ok) {
#pragma unroll
            for (int nt = 0; nt < DC_NT; ++nt) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(WL + DH_OFF_BC + nt * 16 + 4 * q);
                *reinterpret_cast<f16x4*>(outh + cur.o + nt * 16) = __builtin_elementwise_max(__builtin_convertvector(cacc[nt] + b, f16x4), zero4);
            }
        }
        cur = nxt;
    }
}

size_t dcat_packed_floats_f16() { return (size_t)DH_WFLOATS; }

void dcat_pack_weights_f16(const float* w, const float* bd, const float* bc, float* out)
{
    uint16_t* o16 = reinterpret_cast<uint16_t*>(out);
    for (int g = 0; g < DX_NG; ++g)
        for (int nt = 0; nt < DC_NT; ++nt)
            for (int lane = 0; lane < 64; ++lane) {
                const int q = lane >> 4, c = nt * 16 + (lane & 15);
                for (int j = 0; j < 4; ++j) {
                    int k = -1;   // row of w, or none (a zero k-slot)
                    if (g < DC_KB_S) k = g * 16 + 4 * q + j;
                    else if (g == DC_KB_S) k = j < 2 ? DC_KB_S * 16 + 2 * q + j : -1;
                    else k = DC_SKIP + (g - DC_KB_S - 1) * 16 + 4 * q + j;
                    o16[((size_t)(g * DC_NT + nt) * 64 + lane) * 4 + j] = f32_to_f16_bits(k >= 0 ? w[(size_t)k * DC_N + c] : 0.f);
                }
            }
    for (int i = 0; i < DC_N; ++i) { out[DH_OFF_BD + i] = bd[i]; out[DH_OFF_BC + i] = bc[i]; }
}

size_t dcat_packed_floats_x3() { return (size_t)DX_WFLOATS; }

void dcat_pack_weights_x3(const float* w, const float* bd, const float* bc, float* out)
{
    uint16_t* o16 = reinterpret_cast<uint16_t*>(out);
    for (int g = 0; g < DX_NG; ++g)
        for (int nt = 0; nt < DC_NT; ++nt)
            for (int lane = 0; lane < 64; ++lane) {
                const int q = lane >> 4, c = nt * 16 + (lane & 15);
                for (int j = 0; j < 4; ++j) {
                    int k = -1;   // row of w, or none (a zero k-slot)
                    if (g < DC_KB_S) k = g * 16 + 4 * q + j;
                    else if (g == DC_KB_S) k = j < 2 ? DC_KB_S * 16 + 2 * q + j : -1;
                    else k = DC_SKIP + (g - DC_KB_S - 1) * 16 + 4 * q + j;
                    const float v = k >= 0 ? w[(size_t)k * DC_N + c] : 0.f;
                    const size_t rec = ((size_t)(g * DC_NT + nt) * 64 + lane) * 8;   // 8 halves: hi4 | lo4
                    o16[rec + j] = f32_to_f16_bits(v);
                    o16[rec + 4 + j] = f16_lo_bits(v);
                }
            }
    for (int i = 0; i < DC_N; ++i) { out[DX_OFF_BD + i] = bd[i]; out[DX_OFF_BC + i] = bc[i]; }
}

size_t dcat_packed_floats() { return (size_t)DC_WFLOATS; }

// w: conv4_1_1's folded weights [232][96] (rows 0..135 = conv4_2 channels, 136..231 = deconv5_1 channels), bd / bc: the two biases
void dcat_pack_weights(const float* w, const float* bd, const float* bc, float* out)
{
    for (int lane = 0; lane < 64; ++lane) {
        const int q = lane >> 4, r = lane & 15;
        for (int nt = 0; nt < DC_NT; ++nt) {
            const int c = nt * 16 + r;
            for (int kb = 0; kb < DC_KB_S; ++kb)
                for (int j = 0; j < 4; ++j) out[((kb * DC_NT + nt) * 64 + lane) * 4 + j] = w[(size_t)(kb * 16 + 4 * q + j) * DC_N + c];
            for (int j = 0; j < 2; ++j) out[DC_OFF_TAIL + (j * DC_NT + nt) * 64 + lane] = w[(size_t)(DC_KB_S * 16 + 2 * q + j) * DC_N + c];
            for (int kb = 0; kb < DC_KB_D; ++kb)
                for (int j = 0; j < 4; ++j)
                    out[DC_OFF_K2 + ((kb * DC_NT + nt) * 64 + lane) * 4 + j] = w[(size_t)(DC_SKIP + kb * 16 + 4 * q + j) * DC_N + c];
        }
    }
    for (int i = 0; i < DC_N; ++i) { out[DC_OFF_BD + i] = bd[i]; out[DC_OFF_BC + i] = bc[i]; }
}

bool dcat_has_kernel(int cin, int cskip, int cout) { return cin == DC_CIN && cskip == DC_SKIP && cout == DC_N; }

int launch_dcat(const float* x, const float* skip, const float* wd, const float* wc, float* out, int h, int w, int Nf, hipStream_t s, int dtype)
{
    static bool attr_done[3][2][YF_MAX_DEVICES] = {};
    const int dev = current_device();
    const int n_cu = device_cu_count(dev);
    if (dev < 0 || n_cu <= 0) return -2;
    if (dtype != DT_F32 && dtype != DT_F16X3 && dtype != DT_F16) return -1;
    const bool x3 = dtype == DT_F16X3, h16 = dtype == DT_F16;
    const size_t lds = (size_t)(x3 ? DX_WFLOATS : h16 ? DH_WFLOATS : DC_WFLOATS) * sizeof(float);
    static_assert((size_t)DX_WFLOATS * sizeof(float) <= 160 * 1024 && (size_t)DC_WFLOATS * sizeof(float) <= 160 * 1024, "LDS");
    // Small batches: an item of five M-tiles is 33 us of MFMA time on ONE CU; when the five-tile items would leave more
    // than half of the CUs idle, an item is ONE M-tile -- a frame of the 320x256 net then runs on five workgroups.  Same tiles, same
    // arithmetic, same bits (an M-tile does not see how the tiles are grouped).
    const int items5 = Nf * ((h * w + DC_MT * 16 - 1) / (DC_MT * 16));
    const bool small = 2 * items5 <= n_cu;
    const void* fn = x3 ? (small ? reinterpret_cast<const void*>(dcat_x3_kernel<1>) : reinterpret_cast<const void*>(dcat_x3_kernel<DC_MT>))
                   : h16 ? (small ? reinterpret_cast<const void*>(dcat_h_kernel<1>) : reinterpret_cast<const void*>(dcat_h_kernel<DC_MT>))
                        : (small ? reinterpret_cast<const void*>(dcat_kernel<1>) : reinterpret_cast<const void*>(dcat_kernel<DC_MT>));
    if (!attr_done[x3 ? 1 : h16 ? 2 : 0][small][dev]) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -2;
        attr_done[x3 ? 1 : h16 ? 2 : 0][small][dev] = true;
    }
    DcatArgs a{x, skip, wd, wc, out, h, w, 0, 0};
    const int mt = small ? 1 : DC_MT;
    a.items_per_frame = (h * w + mt * 16 - 1) / (mt * 16);
    a.nitems = Nf * a.items_per_frame;
    // persistent grid, every workgroup the same number of items
    const int rounds = (a.nitems + n_cu - 1) / n_cu, grid = (a.nitems + rounds - 1) / rounds;
    if (h16) { if (small) hipLaunchKernelGGL(dcat_h_kernel<1>, dim3((unsigned)grid), dim3(256), lds, s, a); else hipLaunchKernelGGL(dcat_h_kernel<DC_MT>, dim3((unsigned)grid), dim3(256), lds, s, a); }
    else if (x3) { if (small) hipLaunchKernelGGL(dcat_x3_kernel<1>, dim3((unsigned)grid), dim3(256), lds, s, a); else hipLaunchKernelGGL(dcat_x3_kernel<DC_MT>, dim3((unsigned)grid), dim3(256), lds, s, a); }
    else { if (small) hipLaunchKernelGGL(dcat_kernel<1>, dim3((unsigned)grid), dim3(256), lds, s, a); else hipLaunchKernelGGL(dcat_kernel<DC_MT>, dim3((unsigned)grid), dim3(256), lds, s, a); }
    return 0;
}

}  // namespace yf
