// yf_fused_kernels.hip -- block-fused kernels: the wide (expanded) tensors of the net never leave the CU.
//
// fused_block_kernel:  pw-expand(+ReLU) -> dw3x3(+ReLU) -> pw-project (+residual) (+ReLU)
//   = the reference's BasicResBlock (src/model_training/model/yolo_fastest.py:52-66) and the un-named
//   bottlenecks conv1_2/1_3/1_4, conv2_2/2_3/3_1, conv3_2/3_3/3_4, conv3_5/3_6/4_1, conv4_2/4_3/5_1 (:80-118);
//   with PRE it also evaluates conv0 (dense 3x3 s2 on the 1-channel input, :78) in front of the expansion.
// k19_kernel:          conv1_8 (pw 4->24 +ReLU) -> conv1_9 (dense 3x3 s2 24->24 +ReLU) -> conv2_1 (pw 24->8)   (:86-89)
//
// One workgroup = one spatial tile of one frame.  Data flow per tile:
//   HBM --(narrow NHWC input tile + halo, 16-B loads)--> registers --expand--> LDS (channel-planar chunk of EC
//   expanded channels over the halo'd region) --sliding window--> registers (dw) --> project accumulators in
//   registers --> HBM (narrow NHWC output).  Padding semantics: the depthwise conv pads ITS input, i.e. the
//   expanded tensor is zero outside the image (not relu(bias)).
// All weights are wave-uniform and travel through the scalar path (s_load -> SGPR operand of v_fma_f32).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "yf_kernels.h"

namespace yf {

__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

template <int CIN, int CEXP, int COUT, int S, bool RES, bool RELU_OUT, bool PRE, int TYB, int TXB, int BH, int BW,
          int EC, int CG>
__global__ void __launch_bounds__(TYB* TXB) fused_block_kernel(FbArgs a)
{
    constexpr int NT = TYB * TXB, NW = NT / 64;
    constexpr int TH = TYB * BH, TW = TXB * BW;
    constexpr int RH = (TH - 1) * S + 3, RW = (TW - 1) * S + 3;
    constexpr int RWP = (RW + 3) & ~3;            // row pitch (floats), 16-B aligned rows
    constexpr int PLANE = RH * RWP;               // one channel plane
    constexpr int NRP = RH * RW, NPB = (NRP + 63) / 64, NCG = EC / CG, NITEM = NPB * NCG;
    constexpr int WR = (BH - 1) * S + 3, WC = (BW - 1) * S + 3;  // dw window of one thread's output block
    static_assert(NT % 64 == 0 && CEXP % EC == 0 && EC % CG == 0 && COUT % 4 == 0, "shape");
    static_assert(!RES || (CIN == COUT && S == 1 && !PRE), "residual needs same shape");
    static_assert(CIN % 4 == 0 || PRE, "NHWC 16-B loads");
    __shared__ __attribute__((aligned(16))) float E[EC * PLANE];

    const int b = blockIdx.x;
    const int tx = b % a.tiles_x, ty = (b / a.tiles_x) % a.tiles_y, n = b / (a.tiles_x * a.tiles_y);
    const int oy0 = ty * TH, ox0 = tx * TW, iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
    const int wave = wave_id(), lane = threadIdx.x & 63;
    const int tyb = threadIdx.x / TXB, txb = threadIdx.x % TXB;

    float acc[BH * BW][COUT];
#pragma unroll
    for (int p = 0; p < BH * BW; ++p)
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[p][co] = 0.f;

    for (int ch = 0; ch < CEXP / EC; ++ch) {
        // ---------------- expansion of the halo'd region into LDS ----------------
        for (int item = wave; item < NITEM; item += NW) {
            const int pb = item % NPB, cg = item / NPB;
            const int c0 = ch * EC + cg * CG;  // wave-uniform
            const int rp = pb * 64 + lane;
            const int ry = rp / RW, rx = rp - ry * RW;
            const int iy = iy0 + ry, ix = ix0 + rx;
            const bool inreg = rp < NRP;
            const bool inimg = inreg && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            float x[CIN];
            if constexpr (PRE) {
                // conv0: 3x3 stride 2 pad 1 on the 1-channel net input [N, 2H, 2W] (+ReLU)
                const float* __restrict__ src = a.in + (long)n * (4L * a.H * a.W);
                float v[9];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        int yy = 2 * iy - 1 + ky, xx = 2 * ix - 1 + kx;
                        bool ok = inimg && yy >= 0 && yy < 2 * a.H && xx >= 0 && xx < 2 * a.W;
                        v[ky * 3 + kx] = ok ? src[(long)yy * (2 * a.W) + xx] : 0.f;
                    }
#pragma unroll
                for (int c = 0; c < CIN; ++c) {
                    float s = a.b0[c];
#pragma unroll
                    for (int t = 0; t < 9; ++t) s = fmaf(v[t], a.w0[t * CIN + c], s);
                    x[c] = fmaxf(s, 0.f);
                }
            } else {
                const float* __restrict__ src = a.in + (((long)n * a.H + (inimg ? iy : 0)) * a.W + (inimg ? ix : 0)) * CIN;
#pragma unroll
                for (int k = 0; k < CIN; k += 4) {
                    float4 t = *reinterpret_cast<const float4*>(src + k);
                    x[k] = t.x; x[k + 1] = t.y; x[k + 2] = t.z; x[k + 3] = t.w;
                }
            }
            float e[CG];
#pragma unroll
            for (int j = 0; j < CG; ++j) e[j] = a.b1[c0 + j];
#pragma unroll
            for (int k = 0; k < CIN; ++k)
#pragma unroll
                for (int j = 0; j < CG; ++j) e[j] = fmaf(x[k], a.w1[k * CEXP + c0 + j], e[j]);
            if (inreg) {
                float* dst = E + (cg * CG) * PLANE + ry * RWP + rx;
#pragma unroll
                for (int j = 0; j < CG; ++j) dst[j * PLANE] = inimg ? fmaxf(e[j], 0.f) : 0.f;
            }
        }
        __syncthreads();
        // ---------------- depthwise 3x3 from LDS + projection into registers ----------------
#pragma unroll
        for (int c = 0; c < EC; ++c) {
            const int cc = ch * EC + c;
            const float* Ec = E + c * PLANE + (tyb * BH * S) * RWP + txb * BW * S;
            float win[WR][WC];
#pragma unroll
            for (int r = 0; r < WR; ++r) {
                if constexpr ((BW * S) % 4 == 0 && WC >= 4) {
                    float4 t = *reinterpret_cast<const float4*>(Ec + r * RWP);
                    win[r][0] = t.x; win[r][1] = t.y; win[r][2] = t.z; win[r][3] = t.w;
#pragma unroll
                    for (int q = 4; q < WC; ++q) win[r][q] = Ec[r * RWP + q];
                } else if constexpr ((BW * S) % 2 == 0) {
#pragma unroll
                    for (int q = 0; q + 1 < WC; q += 2) {
                        float2 t = *reinterpret_cast<const float2*>(Ec + r * RWP + q);
                        win[r][q] = t.x; win[r][q + 1] = t.y;
                    }
                    if constexpr (WC % 2) win[r][WC - 1] = Ec[r * RWP + WC - 1];
                } else {
#pragma unroll
                    for (int q = 0; q < WC; ++q) win[r][q] = Ec[r * RWP + q];
                }
            }
            float wd[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) wd[t] = a.wd[t * CEXP + cc];
            const float bd = a.bd[cc];
#pragma unroll
            for (int by = 0; by < BH; ++by)
#pragma unroll
                for (int bx = 0; bx < BW; ++bx) {
                    float d = bd;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) d = fmaf(win[by * S + ky][bx * S + kx], wd[ky * 3 + kx], d);
                    d = fmaxf(d, 0.f);
#pragma unroll
                    for (int co = 0; co < COUT; ++co)
                        acc[by * BW + bx][co] = fmaf(d, a.w2[cc * COUT + co], acc[by * BW + bx][co]);
                }
        }
        __syncthreads();
    }
    // ---------------- epilogue: bias (+ residual) (+ ReLU), NHWC store ----------------
#pragma unroll
    for (int by = 0; by < BH; ++by)
#pragma unroll
        for (int bx = 0; bx < BW; ++bx) {
            const int oy = oy0 + tyb * BH + by, ox = ox0 + txb * BW + bx;
            if (oy >= a.Ho || ox >= a.Wo) continue;
            const long opix = ((long)n * a.Ho + oy) * a.Wo + ox;
            float* o = a.out + opix * COUT;
#pragma unroll
            for (int co = 0; co < COUT; co += 4) {
                float4 v = make_float4(acc[by * BW + bx][co] + a.b2[co], acc[by * BW + bx][co + 1] + a.b2[co + 1],
                                       acc[by * BW + bx][co + 2] + a.b2[co + 2], acc[by * BW + bx][co + 3] + a.b2[co + 3]);
                if constexpr (RES) {
                    float4 r = *reinterpret_cast<const float4*>(a.in + opix * CIN + co);
                    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                }
                if constexpr (RELU_OUT) {
                    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                }
                *reinterpret_cast<float4*>(o + co) = v;
            }
        }
}

// ------------------------------------------------------------------------------------------------
// conv1_8 (pw 4->24, ReLU) -> conv1_9 (dense 3x3 stride 2 pad 1, 24->24, ReLU) -> conv2_1 (pw 24->8, linear)
//   tile: 16x16 output pixels (stride-4 resolution), 256 threads, one output pixel per thread.
//   conv1_8's output over the (33x33) halo'd region goes to LDS in two halves of 12 channels, split into
//   even-column and odd-column planes ("space to depth") so that lanes on consecutive output columns read
//   consecutive 48-B pixel records: conflict-free ds_read_b128.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k19_kernel(K19Args a)
{
    constexpr int T = 16, RH = 2 * T + 1, RWE = T + 1, RWO = T;  // even cols 0,2,..,32 (17); odd cols 1,..,31 (16)
    constexpr int CH = 12;                                       // channels per half
    constexpr int NE = RH * RWE, NO = RH * RWO, NR = NE + NO;    // records (pixels) in the even / odd plane
    __shared__ __attribute__((aligned(16))) float A[NR * CH];

    const int b = blockIdx.x;
    const int tx = b % a.tiles_x, ty = (b / a.tiles_x) % a.tiles_y, n = b / (a.tiles_x * a.tiles_y);
    const int oy0 = ty * T, ox0 = tx * T;
    const int iy0 = 2 * oy0 - 1, ix0 = 2 * ox0 - 1;  // region origin in stride-2 coordinates
    const int tyb = threadIdx.x >> 4, txb = threadIdx.x & 15;

    float acc[24];
#pragma unroll
    for (int c = 0; c < 24; ++c) acc[c] = a.b9[c];

    for (int half = 0; half < 2; ++half) {
        // conv1_8 half: region pixels, record index r in [0, NR): first the even plane, then the odd plane
        for (int r = threadIdx.x; r < NR; r += 256) {
            int ry, rx;
            if (r < NE) { ry = r / RWE; rx = 2 * (r - ry * RWE); }
            else { int q = r - NE; ry = q / RWO; rx = 2 * (q - ry * RWO) + 1; }
            const int iy = iy0 + ry, ix = ix0 + rx;
            const bool inimg = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (inimg) t = *reinterpret_cast<const float4*>(a.in + (((long)n * a.H + iy) * a.W + ix) * 4);
            float* dst = A + r * CH;
#pragma unroll
            for (int j = 0; j < CH; j += 4) {
                float o[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = half * CH + j + q;
                    float s = a.b8[c];
                    s = fmaf(t.x, a.w8[0 * 24 + c], s); s = fmaf(t.y, a.w8[1 * 24 + c], s);
                    s = fmaf(t.z, a.w8[2 * 24 + c], s); s = fmaf(t.w, a.w8[3 * 24 + c], s);
                    o[q] = inimg ? fmaxf(s, 0.f) : 0.f;  // conv1_9 pads conv1_8's OUTPUT with zeros
                }
                *reinterpret_cast<float4*>(dst + j) = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
        __syncthreads();
        // conv1_9 partial sums over this half's 12 input channels
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ry = 2 * tyb + ky;
                // column 2*txb + kx: kx=0 -> even plane idx txb; kx=1 -> odd plane idx txb; kx=2 -> even plane idx txb+1
                const int rec = (kx == 1) ? NE + ry * RWO + txb : ry * RWE + txb + (kx >> 1);
                const float* src = A + rec * CH;
                const float* w = a.w9 + ((ky * 3 + kx) * 24 + half * CH) * 24;
#pragma unroll
                for (int j = 0; j < CH; j += 4) {
                    float4 x = *reinterpret_cast<const float4*>(src + j);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float xv = ((const float*)&x)[q];
#pragma unroll
                        for (int c = 0; c < 24; ++c) acc[c] = fmaf(xv, w[(j + q) * 24 + c], acc[c]);
                    }
                }
            }
        }
        __syncthreads();
    }
    // ReLU, conv2_1 (24 -> 8, linear), store
    const int oy = oy0 + tyb, ox = ox0 + txb;
    if (oy >= a.Ho || ox >= a.Wo) return;
    float o8[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) o8[c] = a.b21[c];
#pragma unroll
    for (int k = 0; k < 24; ++k) {
        const float v = fmaxf(acc[k], 0.f);
#pragma unroll
        for (int c = 0; c < 8; ++c) o8[c] = fmaf(v, a.w21[k * 8 + c], o8[c]);
    }
    float* o = a.out + (((long)n * a.Ho + oy) * a.Wo + ox) * 8;
    *reinterpret_cast<float4*>(o) = make_float4(o8[0], o8[1], o8[2], o8[3]);
    *reinterpret_cast<float4*>(o + 4) = make_float4(o8[4], o8[5], o8[6], o8[7]);
}

// ------------------------------------------------------------------------------------------------
// Launchers
// ------------------------------------------------------------------------------------------------
template <int CIN, int CEXP, int COUT, int S, bool RES, bool RELU_OUT, bool PRE, int TYB, int TXB, int BH, int BW, int EC, int CG>
static int launch_fb_t(FbArgs a, int N, hipStream_t s)
{
    a.tiles_y = (a.Ho + TYB * BH - 1) / (TYB * BH);
    a.tiles_x = (a.Wo + TXB * BW - 1) / (TXB * BW);
    dim3 grid((unsigned)(N * a.tiles_y * a.tiles_x));
    hipLaunchKernelGGL((fused_block_kernel<CIN, CEXP, COUT, S, RES, RELU_OUT, PRE, TYB, TXB, BH, BW, EC, CG>), grid,
                       dim3(TYB * TXB), 0, s, a);
    return 0;
}

int launch_fused_block(int cin, int cexp, int cout, int stride, bool res, bool relu_out, bool pre, const FbArgs& a, int N,
                       hipStream_t s)
{
#define FB(ci, ce, co, st, rs, ro, pr, tyb, txb, bh, bw, ec, cg)                                                  \
    if (cin == ci && cexp == ce && cout == co && stride == st && res == rs && relu_out == ro && pre == pr)         \
        return launch_fb_t<ci, ce, co, st, rs, ro, pr, tyb, txb, bh, bw, ec, cg>(a, N, s);
    //  cin cexp cout S  res    relu   pre    TYB TXB BH BW EC CG
    FB(8, 8, 4, 1, false, false, true, 16, 16, 2, 2, 8, 8)     // conv0 + conv1_2/1_3/1_4      @ H/2
    FB(4, 8, 4, 1, true, false, false, 16, 16, 2, 2, 8, 8)     // res1_1                        @ H/2
    FB(8, 32, 8, 1, true, false, false, 32, 8, 2, 2, 8, 8)     // res2_1, res2_2                @ H/4
    FB(8, 32, 8, 2, false, false, false, 16, 20, 1, 1, 8, 8)   // conv2_2/2_3/3_1               H/4 -> H/8
    FB(8, 48, 8, 1, true, false, false, 16, 20, 1, 2, 8, 8)    // res3_1, res3_2                @ H/8
    FB(8, 48, 16, 1, false, false, false, 16, 20, 1, 2, 8, 8)  // conv3_2/3_3/3_4               @ H/8
    FB(16, 96, 16, 1, true, false, false, 16, 20, 1, 2, 8, 8)  // res3_3 .. res3_6              @ H/8
    FB(16, 96, 24, 2, false, false, false, 16, 20, 1, 1, 8, 8) // conv3_5/3_6/4_1               H/8 -> H/16
    FB(24, 136, 24, 1, true, false, false, 16, 20, 1, 1, 8, 8) // res4_1 .. res4_4              @ H/16
#undef FB
    return -1;
}

int launch_k19(K19Args a, int N, hipStream_t s)
{
    a.tiles_y = (a.Ho + 15) / 16;
    a.tiles_x = (a.Wo + 15) / 16;
    hipLaunchKernelGGL(k19_kernel, dim3((unsigned)(N * a.tiles_y * a.tiles_x)), dim3(256), 0, s, a);
    return 0;
}

}  // namespace yf
