// yf_fused_kernels.hip -- block-fused kernels: the wide (expanded) tensors of the net never leave the CU.
//
// fused_block_kernel:  pw-expand(+ReLU) -> dw3x3(+ReLU) -> pw-project (+residual) (+ReLU)
//   = the reference's BasicResBlock (src/model_training/model/yolo_fastest.py:52-66) and the un-named
//   bottlenecks conv1_2/1_3/1_4, conv2_2/2_3/3_1, conv3_2/3_3/3_4, conv3_5/3_6/4_1, conv4_2/4_3/5_1 (:80-118);
//   with PRE it also evaluates conv0 (dense 3x3 s2 on the 1-channel input, :78) in front of the expansion.
// k19_kernel:          conv1_8 (pw 4->24 +ReLU) -> conv1_9 (dense 3x3 s2 24->24 +ReLU) -> conv2_1 (pw 24->8)   (:86-89)
//
// One workgroup = one spatial tile of one frame.  Data flow per tile:
//   HBM --(narrow NHWC input tile + halo, 16-B loads)--> registers --expand--> LDS (chunk of EC, channel pairs interleaved:
//   expanded channels over the halo'd region) --sliding window--> registers (dw) --> project accumulators in
//   registers --> HBM (narrow NHWC output).  Padding semantics: the depthwise conv pads ITS input, i.e. the
//   expanded tensor is zero outside the image (not relu(bias)).
// All weights are wave-uniform and travel through the scalar path (s_load -> SGPR operand of v_fma_f32).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "yf_kernels.h"

#ifndef YF_FB_DW_UNROLL
#define YF_FB_DW_UNROLL 1
#endif

// Output pixels per thread (BH x BW) of the two stride-2 / stride-4 VALU blocks: these kernels are LATENCY-bound (47-55 % of their wave
// time parked, profiles/r02_wave_time_breakdown.txt) and their residency is set by the E tile in LDS; 16x32-pixel tiles (1 x 2 per
// thread, 20.7 KB, 7 workgroups per CU) beat 32x32 (2 x 2, 39 KB, 4 per CU) although the halo share of the expansion grows from 13
// to 20 %: stem 70.9 -> 66.8 us, res2 blocks 58 -> 54 us (A/B, tools/ops_ab.sh).
#ifndef YF_STEM_BH
#define YF_STEM_BH 1
#endif
#ifndef YF_STEM_BW
#define YF_STEM_BW 2
#endif
#ifndef YF_RES2_BH
#define YF_RES2_BH 2   // res2: 32x16 tiles (2 x 1 per thread) are another 6 % faster than 16x32 (54 -> 50 us)
#endif
#ifndef YF_RES2_BW
#define YF_RES2_BW 1
#endif
#ifndef YF_FB_MED3
#define YF_FB_MED3 1
#endif
#ifndef YF_FB_PK
#define YF_FB_PK 1   // expansion / projection / conv0 FMAs over output-channel PAIRS as v_pk_fma_f32 (scalar weight pair x broadcast value)
#endif

namespace yf {

typedef float fb_f32x2 __attribute__((ext_vector_type(2)));
typedef float __attribute__((address_space(4))) cfloat;               // constant address space: scalar (s_load) reads of uniform data
typedef fb_f32x2 __attribute__((address_space(4))) cfloat2;

__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }


// Diagnostic build only (-DYF_STAMP, tools/kbench.hip): per-phase shader-clock sums, written to a buffer nothing else reads.
#ifndef YF_FB_HOIST
#define YF_FB_HOIST 2   // >= 1: the res2 pair keeps its region's input channels in registers across the expansion chunks; 2: single-chunk
#endif                  // blocks (res1_1) request all their items' inputs together as well (A/B: DESIGN.md section 4)
#ifdef YF_STAMP
__device__ __forceinline__ unsigned long long yf_stamp()
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define YF_STAMP_DECL unsigned long long st_[6] = {0, 0, 0, 0, 0, 0}, st_t_ = yf_stamp();
#define YF_STAMP_AT(i) { unsigned long long n_ = yf_stamp(); st_[i] += n_ - st_t_; st_t_ = n_; }
#define YF_STAMP_FLUSH(dbg) if (dbg && (threadIdx.x & 63) == 0) { for (int i_ = 0; i_ < 6; ++i_) atomicAdd(&dbg[i_], st_[i_]); }
#else
#define YF_STAMP_DECL
#define YF_STAMP_AT(i)
#define YF_STAMP_FLUSH(dbg)
#endif

// Weight stream of one block (host-packed, fb_pack_weights): per EC-channel chunk, contiguous
//   [W1 chunk CIN x EC | b1 EC | wd 9 x EC | bd EC | W2 chunk EC x COUT], then b2[COUT]
// so that a wave fetches a chunk's weights with a few s_load_dwordx16 instead of strided single-dword loads.
__host__ __device__ constexpr int fb_chunk_floats(int cin, int cout, int ec) { return cin * ec + ec + 9 * ec + ec + ec * cout; }

template <int CIN, int CEXP, int COUT, int S, bool RES, bool RELU_OUT, bool PRE, int TYB, int TXB, int BH, int BW,
          int EC, int CG, int PE, bool XL, typename T, int C0 = 1>
__global__ void __launch_bounds__(TYB* TXB) fused_block_kernel(FbArgs a)
{
    // C0 (PRE only): input channels of conv0 = io_params input_channel (yolo_fastest.py:78): 1 (gray), 3 (cv2's BGR), or 2 / 4 (NCHW
    // planes; u8 frames: HWC, channel order reversed like `img[:, :, ::-1]`, detect.py:119)
    static_assert(C0 == 1 || (PRE && !XL && C0 >= 2 && C0 <= 4), "conv0 on 1 .. 4 input channels");
    constexpr int NT = TYB * TXB, NW = NT / 64;
    constexpr int O_B1 = CIN * EC, O_WD = O_B1 + EC, O_BD = O_WD + 9 * EC, O_W2 = O_BD + EC, CHF = O_W2 + EC * COUT;
    static_assert(CHF == fb_chunk_floats(CIN, COUT, EC), "pack layout");
    constexpr int TH = TYB * BH, TW = TXB * BW;
    constexpr int RH = (TH - 1) * S + 3, RW = (TW - 1) * S + 3;
    constexpr int RWP = (RW + 3) & ~3;            // row pitch (floats), 16-B aligned rows
    constexpr int PLANE = RH * RWP;               // floats per channel (E interleaves channel pairs: [EC/2][RH][RWP][2])
    constexpr int NRP = RH * RW, NPB = (NRP + 64 * PE - 1) / (64 * PE), NCG = EC / CG, NITEM = NPB * NCG;
    constexpr int WR = (BH - 1) * S + 3, WC = (BW - 1) * S + 3;  // dw window of one thread's output block
    static_assert(NT % 64 == 0 && CEXP % EC == 0 && EC % CG == 0 && COUT % 4 == 0, "shape");
    static_assert(!RES || (CIN == COUT && S == 1 && !PRE), "residual needs same shape");
    static_assert(CIN % 4 == 0 || PRE, "NHWC 16-B loads");
    // input tile staged once: X[region px][CIN] (pitch XP: conflict-free 16-B reads); PRE: the raw 1-channel window
    constexpr int XP = PRE ? 0 : (CIN == 4 ? 4 : CIN + 4);
    constexpr int IRH = 2 * RH + 1, IRW = 2 * RW + 1, IRWP = (IRW + 3) & ~3;  // PRE: input rows/cols feeding the region
    constexpr int XFLOATS = !XL ? 4 : PRE ? IRH * IRWP : NRP * XP;  // XL = false: inputs come straight from HBM/L2
    __shared__ __attribute__((aligned(16))) float E[EC * PLANE];
    __shared__ __attribute__((aligned(16))) float X[XFLOATS];
    // u8 input (PRE): (v - 128) / 255 of the 256 possible pixel values (a 2x2 box mean is an integer 0..255 too).  The IEEE division is
    // ~10 VALU instructions and sat in the loop nine times per region pixel (stem 98 us from u8 against 67 us from f32, batch 256); one
    // division per thread here and an LDS read per tap give the same bits (torch's `(img - 128.0) / 255.0`, detect.py:124)
    __shared__ float LUT[PRE ? 256 : 1];

    const int b = xcd_tile(blockIdx.x, gridDim.x);
    const int tx = b % a.tiles_x, ty = (b / a.tiles_x) % a.tiles_y, n = b / (a.tiles_x * a.tiles_y);
    const int oy0 = ty * TH, ox0 = tx * TW, iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
    const int wave = wave_id(), lane = threadIdx.x & 63;
    const int tyb = threadIdx.x / TXB, txb = threadIdx.x % TXB;

    if constexpr (PRE) {
        if (a.in_u8)
            for (int i = threadIdx.x; i < 256; i += NT) LUT[i] = ((float)i - 128.0f) / 255.0f;
    }
    // ---------------- stage the input tile (one exposed HBM latency per workgroup) ----------------
    if constexpr (!XL) {
    } else if constexpr (PRE) {
        const float* __restrict__ src = a.in + (long)n * (4L * a.H * a.W);
        const int yy0 = 2 * iy0 - 1, xx0 = 2 * ix0 - 1;
        for (int idx = threadIdx.x; idx < IRH * IRW; idx += NT) {
            const int r_ = idx / IRW, c_ = idx - r_ * IRW;
            const int yy = yy0 + r_, xx = xx0 + c_;
            X[r_ * IRWP + c_] = (yy >= 0 && yy < 2 * a.H && xx >= 0 && xx < 2 * a.W) ? src[(long)yy * (2 * a.W) + xx] : 0.f;
        }
    } else {
        constexpr int C4 = CIN / 4;
        const T* __restrict__ src = reinterpret_cast<const T*>(a.in) + (long)n * a.H * a.W * CIN;
        for (int idx = threadIdx.x; idx < NRP * C4; idx += NT) {
            const int rp = idx / C4, c4 = idx - rp * C4;
            const int ry_ = rp / RW, rx_ = rp - ry_ * RW;
            const int iy = iy0 + ry_, ix = ix0 + rx_;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) v = ld4<T>(src + ((long)iy * a.W + ix) * CIN + c4 * 4);
            *reinterpret_cast<float4*>(&X[rp * XP + c4 * 4]) = v;
        }
    }
    __syncthreads();

    fb_f32x2 acc2[BH * BW][COUT / 2];   // projection accumulators, output-channel pairs
#pragma unroll
    for (int p = 0; p < BH * BW; ++p)
#pragma unroll
        for (int co = 0; co < COUT / 2; ++co) acc2[p][co] = fb_f32x2{0.f, 0.f};

    // HOIST (blocks whose items are whole pixel blocks: res1_1, the res2 pair): a region pixel's input channels do not depend on the
    // chunk, yet the item loop below re-loaded them in every chunk -- and every load was one exposed L2 round trip per item and chunk
    // (load, wait, 16 / 32 packed FMAs, next item).  With HOIST the wave's MAXI items are loaded ONCE, all requests in flight together,
    // before the chunk loop, and stay in registers (MAXI x CIN VGPRs).
    constexpr int MAXI = (NITEM + NW - 1) / NW;
    constexpr bool HOIST = YF_FB_HOIST && !PRE && !XL && (CEXP / EC > 1 || YF_FB_HOIST > 1) && NCG == 1 && PE == 1 && MAXI * CIN <= 32;
    float xh[HOIST ? MAXI : 1][CIN];
    int hdst[HOIST ? MAXI : 1];       // ... and so are the item's E offset (-1: no pixel) and its ReLU limit (+inf inside the image, 0 outside):
    float hlim[HOIST ? MAXI : 1];     // 13 of the 64 VALU instructions of an item and chunk were this index arithmetic
    if constexpr (HOIST) {
#pragma unroll
        for (int it = 0; it < MAXI; ++it) {
            const int rp = (wave + it * NW) * 64 + lane;      // item = pixel block (NCG == 1, PE == 1); beyond the region: pixel 0, unused
            const int ryh = (rp < NRP ? rp : 0) / RW, rxh = (rp < NRP ? rp : 0) - ryh * RW;
            const int iy = iy0 + ryh, ix = ix0 + rxh;
            const bool in = rp < NRP && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            hdst[it] = rp < NRP ? (ryh * RWP + rxh) * 2 : -1;
            hlim[it] = in ? __builtin_inff() : 0.f;
            const T* __restrict__ srch = reinterpret_cast<const T*>(a.in) + (((long)n * a.H + (in ? iy : 0)) * a.W + (in ? ix : 0)) * CIN;
#pragma unroll
            for (int k = 0; k < CIN; k += 4) {
                const float4 t = ld4<T>(srch + k);
                xh[it][k] = t.x; xh[it][k + 1] = t.y; xh[it][k + 2] = t.z; xh[it][k + 3] = t.w;
            }
        }
    }

    YF_STAMP_DECL
    for (int ch = 0; ch < CEXP / EC; ++ch) {
        YF_STAMP_AT(0)
        const cfloat* __restrict__ wc = (const cfloat*)(a.wp + ch * CHF);  // this chunk's weights (wave-uniform, constant address space -> scalar loads)
        // ---------------- expansion of the halo'd region into LDS: PE pixels per lane per item ----------------
#pragma unroll
        for (int it = 0; it < (HOIST ? MAXI : 1); ++it)   // HOIST: item it of this wave, unrolled so that xh[it] is a register; else the plain item loop
        for (int item = HOIST ? wave + it * NW : wave; item < (HOIST ? (wave + it * NW < NITEM ? wave + it * NW + 1 : 0) : NITEM); item += NW) {
            const int pb = item % NPB, cg = item / NPB;
            float x[PE][CIN];
            int ry[PE], rx[PE];
            bool inreg[PE], inimg[PE];
#pragma unroll
            for (int p = 0; p < PE; ++p) {
                const int rp = pb * (64 * PE) + p * 64 + lane;
                ry[p] = rp / RW; rx[p] = rp - ry[p] * RW;
                const int iy = iy0 + ry[p], ix = ix0 + rx[p];
                inreg[p] = rp < NRP;
                inimg[p] = inreg[p] && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
                const int rpc = inreg[p] ? rp : 0;
                if constexpr (PRE) {
                    // conv0: 3x3 stride 2 pad 1 on the C0-channel net input (+ReLU); v[(ky * 3 + kx) * C0 + ci]
                    float v[9 * C0];
                    if constexpr (XL) {  // window rows 2*ry.., cols 2*rx.. of the staged input
                        const float* win0 = X + (2 * (rpc / RW)) * IRWP + 2 * (rpc % RW);
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                            for (int kx = 0; kx < 3; ++kx) v[ky * 3 + kx] = win0[ky * IRWP + kx];
                    } else if (a.in_u8) {
                        // Detect_YOLO.__pre_process fused into the load (src/detect.py:115-124): u8 gray frame, optional
                        // exact-2x box mean (a+b+c+d+2)>>2, then (v-128)/255; conv0 zero-pads the NORMALISED tensor
                        // (C0 = 3: the frame is cv2's HWC BGR; net channel ci is source channel 2 - ci, detect.py:119)
                        // Branch-free and in two passes: ALL byte loads of the 9 C0 taps first (from addresses clamped into the frame, so no
                        // load sits under a bounds branch), the table look-ups after, zeros selected in at the end.  Written tap by tap
                        // (bounds branch, load, look up) the compiler waited for every load before its look-up: nine dependent memory
                        // round trips per region pixel (stem 84 us from u8 against 66 us from f32).
                        const int sw = (a.u8_down2 ? 4 * a.W : 2 * a.W) * C0;
                        const uint8_t* __restrict__ src = a.in_u8 + (long)n * (a.u8_down2 ? 16L : 4L) * a.H * a.W * C0;
                        int yo[3], xo[3];
                        unsigned rows = 0, cols = 0;
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            const int yy = 2 * iy - 1 + k, xx = 2 * ix - 1 + k;
                            const int yc = min(max(yy, 0), 2 * a.H - 1), xc = min(max(xx, 0), 2 * a.W - 1);
                            yo[k] = (a.u8_down2 ? 2 * yc : yc) * sw;
                            xo[k] = (a.u8_down2 ? 2 * xc : xc) * C0;
                            if (inimg[p] && yy == yc) rows |= 0x7u << (3 * k);      // tap row k inside the frame
                            if (xx == xc) cols |= 0x49u << k;                        // tap column k inside the frame
                        }
                        const unsigned okm = rows & cols;   // bit ky * 3 + kx: the tap reads the image (else conv0's zero padding)
                        if (a.u8_down2) {
                            unsigned r0[9 * C0], r1[9 * C0];   // the two source rows of a tap's 2x2 block: two pixels each
#pragma unroll
                            for (int t = 0; t < 9 * C0; ++t) {
                                const uint8_t* q = src + yo[t / C0 / 3] + xo[(t / C0) % 3] + (C0 - 1 - t % C0);
                                if constexpr (C0 == 1) {   // the two pixels of a row are adjacent bytes: one 16-bit load
                                    r0[t] = *reinterpret_cast<const unsigned short*>(q);
                                    r1[t] = *reinterpret_cast<const unsigned short*>(q + sw);
                                } else {
                                    r0[t] = (unsigned)q[0] | ((unsigned)q[C0] << 8);
                                    r1[t] = (unsigned)q[sw] | ((unsigned)q[sw + C0] << 8);
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int t = 0; t < 9 * C0; ++t) {
                                const float lv = LUT[((r0[t] & 255u) + (r0[t] >> 8) + (r1[t] & 255u) + (r1[t] >> 8) + 2u) >> 2];
                                v[t] = (okm >> (t / C0)) & 1u ? lv : 0.f;
                            }
                        } else {
                            unsigned r0[9 * C0];
#pragma unroll
                            for (int t = 0; t < 9 * C0; ++t) r0[t] = src[yo[t / C0 / 3] + xo[(t / C0) % 3] + (C0 - 1 - t % C0)];
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int t = 0; t < 9 * C0; ++t) {
                                const float lv = LUT[r0[t]];
                                v[t] = (okm >> (t / C0)) & 1u ? lv : 0.f;
                            }
                        }
                    } else {
                        const long plane = 4L * a.H * a.W;   // NCHW input: C0 planes per frame
                        const float* __restrict__ src = a.in + (long)n * C0 * plane;
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                            for (int kx = 0; kx < 3; ++kx) {
                                int yy = 2 * iy - 1 + ky, xx = 2 * ix - 1 + kx;
                                bool ok = inimg[p] && yy >= 0 && yy < 2 * a.H && xx >= 0 && xx < 2 * a.W;
#pragma unroll
                                for (int ci = 0; ci < C0; ++ci) v[(ky * 3 + kx) * C0 + ci] = ok ? src[ci * plane + (long)yy * (2 * a.W) + xx] : 0.f;
                            }
                    }
#if YF_FB_PK
#pragma unroll
                    for (int c = 0; c < CIN; c += 2) {
                        fb_f32x2 s2 = *(const cfloat2*)(const cfloat*)(a.b0 + c);
#pragma unroll
                        for (int t = 0; t < 9 * C0; ++t)
                            s2 = __builtin_elementwise_fma(fb_f32x2{v[t], v[t]}, *(const cfloat2*)(const cfloat*)(a.w0 + t * CIN + c), s2);
                        x[p][c] = fmaxf(s2[0], 0.f); x[p][c + 1] = fmaxf(s2[1], 0.f);
                    }
#else
#pragma unroll
                    for (int c = 0; c < CIN; ++c) {
                        float s = a.b0[c];
#pragma unroll
                        for (int t = 0; t < 9 * C0; ++t) s = fmaf(v[t], a.w0[t * CIN + c], s);
                        x[p][c] = fmaxf(s, 0.f);
                    }
#endif
                } else if constexpr (XL) {
#pragma unroll
                    for (int k = 0; k < CIN; k += 4) {
                        float4 t = *reinterpret_cast<const float4*>(&X[rpc * XP + k]);
                        x[p][k] = t.x; x[p][k + 1] = t.y; x[p][k + 2] = t.z; x[p][k + 3] = t.w;
                    }
                } else if constexpr (HOIST) {
#pragma unroll
                    for (int k = 0; k < CIN; ++k) x[p][k] = xh[it][k];
                } else {
                    const T* __restrict__ src = reinterpret_cast<const T*>(a.in) + (((long)n * a.H + (inimg[p] ? iy : 0)) * a.W + (inimg[p] ? ix : 0)) * CIN;
#pragma unroll
                    for (int k = 0; k < CIN; k += 4) {
                        float4 t = ld4<T>(src + k);
                        x[p][k] = t.x; x[p][k + 1] = t.y; x[p][k + 2] = t.z; x[p][k + 3] = t.w;
                    }
                }
            }
            float e[PE][CG];
#if YF_FB_PK
            static_assert(CG % 2 == 0 && EC % 2 == 0, "channel pairs");
            fb_f32x2 e2[PE][CG / 2];
#pragma unroll
            for (int j = 0; j < CG; j += 2) {
                const fb_f32x2 bv = *(const cfloat2*)(wc + O_B1 + cg * CG + j);
#pragma unroll
                for (int p = 0; p < PE; ++p) e2[p][j / 2] = bv;
            }
#pragma unroll
            for (int k = 0; k < CIN; ++k)
#pragma unroll
                for (int j = 0; j < CG; j += 2) {
                    const fb_f32x2 wv = *(const cfloat2*)(wc + k * EC + cg * CG + j);
#pragma unroll
                    for (int p = 0; p < PE; ++p) e2[p][j / 2] = __builtin_elementwise_fma(fb_f32x2{x[p][k], x[p][k]}, wv, e2[p][j / 2]);
                }
#pragma unroll
            for (int p = 0; p < PE; ++p)
#pragma unroll
                for (int j = 0; j < CG; ++j) e[p][j] = e2[p][j / 2][j & 1];
#else
#pragma unroll
            for (int j = 0; j < CG; ++j) {
                const float bv = wc[O_B1 + cg * CG + j];
#pragma unroll
                for (int p = 0; p < PE; ++p) e[p][j] = bv;
            }
#pragma unroll
            for (int k = 0; k < CIN; ++k)
#pragma unroll
                for (int j = 0; j < CG; ++j) {
                    const float wv = wc[k * EC + cg * CG + j];
#pragma unroll
                    for (int p = 0; p < PE; ++p) e[p][j] = fmaf(x[p][k], wv, e[p][j]);
                }
#endif
            // E holds channel PAIRS interleaved: [EC/2][RH][RWP][2] -- one 8-byte write per pair here, and the depthwise below runs
            // its taps on both channels of a pair with one v_pk_fma_f32
#pragma unroll
            for (int p = 0; p < PE; ++p)
                if (HOIST ? hdst[it] >= 0 : inreg[p]) {
                    float* dst = E + (cg * CG) * PLANE + (HOIST ? hdst[it] : (ry[p] * RWP + rx[p]) * 2);
                    // ReLU and the zero outside the image in ONE instruction per value: median(x, 0, lim), lim = +inf inside, 0 outside
                    // (v_max + v_cndmask before: 2 of every ~7 VALU instructions of the res2 expansion: res2_1 / res2_2 40.2 -> 36.7 us; not in the
                    //  stem, which it slows from 57.7 to 62.1 us -- A/B, tools/ops_abn.sh)
                    const float lim = HOIST ? hlim[it] : inimg[p] ? __builtin_inff() : 0.f;
#pragma unroll
                    for (int j = 0; j < CG; j += 2)
                        *reinterpret_cast<float2*>(dst + j * PLANE) =
                            (YF_FB_MED3 && !PRE) ? make_float2(__builtin_amdgcn_fmed3f(e[p][j], 0.f, lim), __builtin_amdgcn_fmed3f(e[p][j + 1], 0.f, lim))
                                       : make_float2(inimg[p] ? fmaxf(e[p][j], 0.f) : 0.f, inimg[p] ? fmaxf(e[p][j + 1], 0.f) : 0.f);
                }
        }
        YF_STAMP_AT(1)
        __syncthreads();
        YF_STAMP_AT(2)
        // ---------------- depthwise 3x3 from LDS + projection into registers ----------------
        // NOT fully unrolled: unrolled, the scalar loads of all EC channels' weights are hoisted to the top and, with the
        // expansion's, exceed the SGPR file -- the compiler then spills SGPRs to VGPR lanes and 30-50 % of the VALU
        // instructions of the stride-2 kernels were v_readlane / v_writelane (tools/isa_stats.py)
#pragma unroll YF_FB_DW_UNROLL
        for (int c = 0; c < EC; c += 2) {
            const float* Ec = E + c * PLANE + ((tyb * BH * S) * RWP + txb * BW * S) * 2;
            fb_f32x2 win[WR][WC];   // (channel c, channel c + 1) of every window pixel
#pragma unroll
            for (int r = 0; r < WR; ++r) {
                if constexpr ((BW * S) % 2 == 0) {   // 16-byte aligned: two pixels x two channels per read
#pragma unroll
                    for (int q = 0; q + 1 < WC; q += 2) {
                        const float4 t = *reinterpret_cast<const float4*>(Ec + (r * RWP + q) * 2);
                        win[r][q] = fb_f32x2{t.x, t.y}; win[r][q + 1] = fb_f32x2{t.z, t.w};
                    }
                    if constexpr (WC % 2) win[r][WC - 1] = *reinterpret_cast<const fb_f32x2*>(Ec + (r * RWP + WC - 1) * 2);
                } else {
#pragma unroll
                    for (int q = 0; q < WC; ++q) win[r][q] = *reinterpret_cast<const fb_f32x2*>(Ec + (r * RWP + q) * 2);
                }
            }
            fb_f32x2 wd[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) wd[t] = *(const cfloat2*)(wc + O_WD + t * EC + c);
            const fb_f32x2 bd = *(const cfloat2*)(wc + O_BD + c);
#pragma unroll
            for (int by = 0; by < BH; ++by)
#pragma unroll
                for (int bx = 0; bx < BW; ++bx) {
                    fb_f32x2 d2 = bd;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) d2 = __builtin_elementwise_fma(win[by * S + ky][bx * S + kx], wd[ky * 3 + kx], d2);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float d = fmaxf(d2[h], 0.f);
#pragma unroll
                        for (int co = 0; co < COUT; co += 2)
                            acc2[by * BW + bx][co / 2] = __builtin_elementwise_fma(fb_f32x2{d, d}, *(const cfloat2*)(wc + O_W2 + (c + h) * COUT + co),
                                                                                    acc2[by * BW + bx][co / 2]);
                    }
                }
        }
        YF_STAMP_AT(3)
        __syncthreads();
        YF_STAMP_AT(4)
    }
    // ---------------- epilogue: bias (+ residual) (+ ReLU), NHWC store ----------------
    const cfloat* __restrict__ b2 = (const cfloat*)(a.wp + (CEXP / EC) * CHF);
#pragma unroll
    for (int by = 0; by < BH; ++by)
#pragma unroll
        for (int bx = 0; bx < BW; ++bx) {
            const int oy = oy0 + tyb * BH + by, ox = ox0 + txb * BW + bx;
            if (oy >= a.Ho || ox >= a.Wo) continue;
            const long opix = ((long)n * a.Ho + oy) * a.Wo + ox;
            T* o = reinterpret_cast<T*>(a.out) + opix * COUT;
#pragma unroll
            for (int co = 0; co < COUT; co += 4) {
                float4 v = make_float4(acc2[by * BW + bx][co / 2][0] + b2[co], acc2[by * BW + bx][co / 2][1] + b2[co + 1],
                                       acc2[by * BW + bx][co / 2 + 1][0] + b2[co + 2], acc2[by * BW + bx][co / 2 + 1][1] + b2[co + 3]);
                if constexpr (RES) {  // the residual is the centre of the staged tile
                    float4 r = XL ? *reinterpret_cast<const float4*>(&X[((tyb * BH + by + 1) * RW + txb * BW + bx + 1) * XP + co])
                                  : ld4<T>(reinterpret_cast<const T*>(a.in) + opix * CIN + co);
                    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                }
                if constexpr (RELU_OUT) {
                    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                }
                st4<T>(o + co, v);
            }
        }
    YF_STAMP_AT(5)
    YF_STAMP_FLUSH(a.dbg)
}

// ------------------------------------------------------------------------------------------------
// conv1_8 (pw 4->24, ReLU) -> conv1_9 (dense 3x3 stride 2 pad 1, 24->24, ReLU) -> conv2_1 (pw 24->8, linear)
//   tile: 16x16 output pixels (stride-4 resolution), 256 threads, one output pixel per thread.
//   Measured alternatives that did NOT help (tools/kbench.hip, profiles/): two pixels per lane (each scalar weight feeding
//   two FMAs) and v_pk_fma_f32 over pixel pairs or output-channel pairs all land at the same ~250 us / 30 TMAC/s although the
//   packed forms halve the VALU instruction count -- the kernel is not VALU-issue-bound (fully unrolled straight-line code,
//   instruction fetch is the suspect).  Later finding (tools/isa_mix.py): this fully unrolled form also spills ~900 SGPR values to
//   VGPR lanes (v_readlane / v_writelane are a third of its VALU instructions).  Superseded by k19m_kernel (yf_k19_kernels.hip);
//   kept as the VALU reference point of tools/kbench.hip.
//   conv1_8's output over the (33x33) halo'd region goes to LDS in two halves of 12 channels, split into
//   even-column and odd-column planes ("space to depth") so that lanes on consecutive output columns read
//   consecutive 48-B pixel records: conflict-free ds_read_b128.
// ------------------------------------------------------------------------------------------------
template <typename TT>
__global__ void __launch_bounds__(256) k19_kernel(K19Args a)
{
    constexpr int T = 16, RH = 2 * T + 1, RWE = T + 1, RWO = T;  // even cols 0,2,..,32 (17); odd cols 1,..,31 (16)
    constexpr int CH = 12;                                       // channels per half
    constexpr int NE = RH * RWE, NO = RH * RWO, NR = NE + NO;    // records (pixels) in the even / odd plane
    __shared__ __attribute__((aligned(16))) float A[NR * CH];

    const int b = xcd_tile(blockIdx.x, gridDim.x);
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
            if (inimg) t = ld4<TT>(reinterpret_cast<const TT*>(a.in) + (((long)n * a.H + iy) * a.W + ix) * 4);
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
    TT* o = reinterpret_cast<TT*>(a.out) + (((long)n * a.Ho + oy) * a.Wo + ox) * 8;
    st4<TT>(o, make_float4(o8[0], o8[1], o8[2], o8[3]));
    st4<TT>(o + 4, make_float4(o8[4], o8[5], o8[6], o8[7]));
}

// ------------------------------------------------------------------------------------------------
// Launchers
// ------------------------------------------------------------------------------------------------
template <int CIN, int CEXP, int COUT, int S, bool RES, bool RELU_OUT, bool PRE, int TYB, int TXB, int BH, int BW, int EC, int CG, int PE, bool XL, typename T, int C0 = 1>
static int launch_fb_t(FbArgs a, int N, hipStream_t s)
{
    a.tiles_y = (a.Ho + TYB * BH - 1) / (TYB * BH);
    a.tiles_x = (a.Wo + TXB * BW - 1) / (TXB * BW);
    dim3 grid((unsigned)(N * a.tiles_y * a.tiles_x));
    hipLaunchKernelGGL((fused_block_kernel<CIN, CEXP, COUT, S, RES, RELU_OUT, PRE, TYB, TXB, BH, BW, EC, CG, PE, XL, T, C0>), grid,
                       dim3(TYB * TXB), 0, s, a);
    return 0;
}

//     cin cexp cout S  res    relu   pre    TYB TXB BH BW EC CG PE XL      (XL: stage the input tile in LDS -- measured
//     slower at these shapes: it costs occupancy and LDS bandwidth, the L2-served loads were already hidden)
#define YF_FB_SHAPES(FB)                                                                                                 \
    FB(8, 8, 4, 1, false, false, true, 16, 16, YF_STEM_BH, YF_STEM_BW, 8, 8, 1, false)     /* conv0 + conv1_2/1_3/1_4      @ H/2  */          \
    FB(8, 8, 4, 1, false, false, false, 16, 16, 1, 2, 8, 8, 1, false)    /* conv1_2/1_3/1_4 behind a separate conv0 (input_channel > 4) */ \
    FB(4, 8, 4, 1, true, false, false, 16, 16, 1, 2, 8, 8, 1, false)     /* res1_1                        @ H/2  */          \
    FB(8, 32, 8, 1, true, false, false, 16, 16, YF_RES2_BH, YF_RES2_BW, 8, 8, 1, false)    /* res2_1, res2_2                @ H/4  */          \
    FB(8, 32, 8, 2, false, false, false, 16, 20, 1, 1, 8, 8, 1, false)   /* conv2_2/2_3/3_1               H/4 -> H/8 */      \
    FB(8, 48, 8, 1, true, false, false, 16, 20, 1, 2, 8, 8, 2, false)    /* res3_1, res3_2                @ H/8  */          \
    FB(8, 48, 16, 1, false, false, false, 16, 20, 1, 2, 8, 8, 2, false)  /* conv3_2/3_3/3_4 (fallback)    @ H/8  */          \
    FB(16, 96, 16, 1, true, false, false, 16, 20, 1, 2, 8, 8, 1, false)  /* res3_3 .. res3_6 (fallback)   @ H/8  */          \
    FB(16, 96, 24, 2, false, false, false, 16, 20, 1, 1, 8, 8, 1, false) /* conv3_5/3_6/4_1               H/8 -> H/16 */     \
    FB(24, 136, 24, 1, true, false, false, 16, 20, 1, 1, 8, 8, 1, false) /* res4_1 .. res4_4 (fallback)   @ H/16 */

int launch_fused_block(int cin, int cexp, int cout, int stride, bool res, bool relu_out, int pre_c0, const FbArgs& a, int N,
                       hipStream_t s, int dtype)
{
    const bool pre = pre_c0 != 0;
    if (pre && (pre_c0 < 1 || pre_c0 > 4)) return -1;
    if (pre_c0 > 1) {   // the stem of a multi-channel model: the same tile shape, conv0 over C0 input planes
        if (!(cin == 8 && cexp == 8 && cout == 4 && stride == 1 && !res && !relu_out)) return -1;
#define YF_STEM_C0(C0_)                                                                                                                            \
        if (pre_c0 == C0_)                                                                                                                          \
            return dtype == DT_F16 ? launch_fb_t<8, 8, 4, 1, false, false, true, 16, 16, YF_STEM_BH, YF_STEM_BW, 8, 8, 1, false, half_t, C0_>(a, N, s) \
                                   : launch_fb_t<8, 8, 4, 1, false, false, true, 16, 16, YF_STEM_BH, YF_STEM_BW, 8, 8, 1, false, float, C0_>(a, N, s);
        YF_STEM_C0(2) YF_STEM_C0(3) YF_STEM_C0(4)
#undef YF_STEM_C0
        return -1;
    }
    // Small batches (round 5): the res2 blocks (stride 4: 80x64 pixels per frame of the 320x256 net) on 16x16 tiles instead of 32x16 when the
    // larger tiling would leave more than half of the CUs idle -- one output pixel per lane instead of two; a pixel's arithmetic is the same.
    if (cin == 8 && cexp == 32 && cout == 8 && stride == 1 && res && !relu_out && !pre) {
        const int n_cu = device_cu_count(current_device());
        const long big = (long)N * ((a.Ho + 16 * YF_RES2_BH - 1) / (16 * YF_RES2_BH)) * ((a.Wo + 16 * YF_RES2_BW - 1) / (16 * YF_RES2_BW));
        static const bool off = getenv("YF_MRES_SMALL_OFF") != nullptr;
        if (!off && n_cu > 0 && 2 * big <= n_cu)
            return dtype == DT_F16 ? launch_fb_t<8, 32, 8, 1, true, false, false, 16, 16, 1, 1, 8, 8, 1, false, half_t>(a, N, s)
                                   : launch_fb_t<8, 32, 8, 1, true, false, false, 16, 16, 1, 1, 8, 8, 1, false, float>(a, N, s);
    }
#define FB(ci, ce, co, st, rs, ro, pr, tyb, txb, bh, bw, ec, cg, pe, xl)                                          \
    if (cin == ci && cexp == ce && cout == co && stride == st && res == rs && relu_out == ro && pre == pr)         \
        return dtype == DT_F16 ? launch_fb_t<ci, ce, co, st, rs, ro, pr, tyb, txb, bh, bw, ec, cg, pe, xl, half_t>(a, N, s)   \
                               : launch_fb_t<ci, ce, co, st, rs, ro, pr, tyb, txb, bh, bw, ec, cg, pe, xl, float>(a, N, s);
    YF_FB_SHAPES(FB)
#undef FB
    return -1;
}

int fb_chunk_channels(int cin, int cexp, int cout, int stride, bool res, bool relu_out, int pre_c0)
{
    const bool pre = pre_c0 != 0;
#define FB(ci, ce, co, st, rs, ro, pr, tyb, txb, bh, bw, ec, cg, pe, xl) \
    if (cin == ci && cexp == ce && cout == co && stride == st && res == rs && relu_out == ro && pre == pr) return ec;
    YF_FB_SHAPES(FB)
#undef FB
    return -1;
}

size_t fb_packed_floats(int cin, int cexp, int cout, int ec) { return (size_t)(cexp / ec) * fb_chunk_floats(cin, cout, ec) + cout; }

void fb_pack_weights(const float* w1 /*[cin][cexp]*/, const float* b1, const float* wd /*[9][cexp]*/, const float* bd,
                     const float* w2 /*[cexp][cout]*/, const float* b2, int cin, int cexp, int cout, int ec, float* out)
{
    const int CHF = fb_chunk_floats(cin, cout, ec);
    for (int ch = 0; ch < cexp / ec; ++ch) {
        float* o = out + (size_t)ch * CHF;
        for (int k = 0; k < cin; ++k)
            for (int j = 0; j < ec; ++j) o[k * ec + j] = w1[(size_t)k * cexp + ch * ec + j];
        o += cin * ec;
        for (int j = 0; j < ec; ++j) o[j] = b1[ch * ec + j];
        o += ec;
        for (int t = 0; t < 9; ++t)
            for (int j = 0; j < ec; ++j) o[t * ec + j] = wd[(size_t)t * cexp + ch * ec + j];
        o += 9 * ec;
        for (int j = 0; j < ec; ++j) o[j] = bd[ch * ec + j];
        o += ec;
        for (int j = 0; j < ec; ++j)
            for (int co = 0; co < cout; ++co) o[j * cout + co] = w2[(size_t)(ch * ec + j) * cout + co];
    }
    for (int co = 0; co < cout; ++co) out[(size_t)(cexp / ec) * CHF + co] = b2[co];
}

int launch_k19(K19Args a, int N, hipStream_t s, int dtype)
{
    a.tiles_y = (a.Ho + 15) / 16;
    a.tiles_x = (a.Wo + 15) / 16;
    if (dtype == DT_F16) hipLaunchKernelGGL(k19_kernel<half_t>, dim3((unsigned)(N * a.tiles_y * a.tiles_x)), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k19_kernel<float>, dim3((unsigned)(N * a.tiles_y * a.tiles_x)), dim3(256), 0, s, a);
    return 0;
}

}  // namespace yf
