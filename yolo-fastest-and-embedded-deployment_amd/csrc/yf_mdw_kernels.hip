// yf_mdw_kernels.hip -- the detection-head pairs  dw5x5(+ReLU) -> 1x1 conv  [-> 1x1 head conv]  on chip.
//
//   conv5_3 -> conv5_4,   conv5_5 -> conv5_6 -> head_5      (small head, src/model_training/model/yolo_fastest.py:132-136,200-206)
//   conv4_1_2 -> conv4_1_3,   conv4_1_4 -> conv4_1_5 -> head_4   (large head, :141-146, :211-216)
//
// One workgroup = one TH x TW tile of one frame (the whole frame at the shipped sizes).  Per 16-channel chunk of the
// input: HBM --16-B loads, one chunk prefetched in registers--> LDS E[4 ch-groups][region px][4 ch] --25 ds_read_b128
// per pixel--> depthwise FMA chains (+bias, ReLU) == A fragments of  v_mfma_f32_16x16x4_f32  against the 1x1 conv's
// weights (fragments from the LDS-staged weight stream, as the MFMA's A operand); accumulators live in VGPRs across chunks.
// With HEADN the finished fragments (+bias) are directly the B operand of the head GEMM (chained in registers) and the
// head's logits are stored NCHW like the reference's output; otherwise the result is stored NHWC, 16 bytes per lane.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "yf_kernels.h"

namespace yf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Diagnostic build only (-DYF_MDW_STAMP, tools/kbench.hip mdw): per-phase shader-clock sums of wave 0 of every workgroup.
#ifdef YF_MDW_STAMP
__device__ unsigned long long yf_mdw_dbg[8];
__device__ __forceinline__ unsigned long long mdw_clock()
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define MDW_STAMP_DECL unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t_ = mdw_clock();
#define MDW_STAMP(i) { unsigned long long n_ = mdw_clock(); st_[i] += n_ - st_t_; st_t_ = n_; }
#define MDW_STAMP_FLUSH if (threadIdx.x == 0) { for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&yf_mdw_dbg[i_], st_[i_]); }
#else
#define MDW_STAMP_DECL
#define MDW_STAMP(i)
#define MDW_STAMP_FLUSH
#endif

// h16: B fragments are f16x4 per lane (one v_mfma_f32_16x16x16_f16 per 16-channel chunk and n-tile); dw weights / biases stay fp32
// wmode WM_F16X3 (split operands): every fp16 fragment block is followed by its lo halves (same layout)
__host__ __device__ constexpr int mdw_chunk_floats(int n, int wmode = WM_F32)
{
    return 25 * 16 + 16 + (wmode != WM_F32 ? (wmode == WM_F16X3 ? 2 : 1) * (n / 16) * 128 : 4 * (n / 16) * 64);
}
// head conv: its output channels headn = num_anchors * (5 + num_cls) (yolo_fastest.py:76,138,148) are padded to PAIRS of 16-column
// MFMA tiles (nthp tiles): the shipped 24 and anything up to 32 is one pair -- the same stream as ever
__host__ __device__ constexpr int mdw_head_tiles(int headn) { return headn ? 2 * ((headn + 31) / 32) : 0; }
__host__ __device__ constexpr int mdw_stream_floats(int c, int n, int headn, int wmode = WM_F32)
{
    int f = (c / 16) * mdw_chunk_floats(n, wmode) + n;
    const int nthp = mdw_head_tiles(headn);
    if (headn) f += (wmode != WM_F32 ? (wmode == WM_F16X3 ? 2 : 1) * (n / 16) * nthp * 128 : (n / 4) * nthp * 64) + nthp * 16;
    return (f + 3) & ~3;
}

typedef float mdw_f32x2 __attribute__((ext_vector_type(2)));

// Pixels per 4-channel plane of E, chosen for the 25-tap window READS (ten times the fill writes): a ds_read_b128 is serviced
// in lane groups that mix two channel quads -- lanes r in {0-3, 12-15} of quad q with r in {4-11} of quad q + 1
// (MI355X_MICROARCH.md, LDS).  Lanes on consecutive pixels (one tile per wave) want the pitch == 0 (mod 16 records) so the two
// quads land on complementary banks; lanes on every second pixel (ADJ: two adjacent pixels per lane) want it == 1 (mod 16) so
// quad q + 1 takes the odd records.  (The old pitch, == 2 mod 8, was conflict-free for the fills and 2-way conflicted on every
// window read.)
#ifndef YF_MDW_EPL_ADD
#define YF_MDW_EPL_ADD 0
#endif
// the small head's 1x1 convs (128 output channels) in fp32: per-chunk partial sums (see mdw_kernel, PSUM)
__host__ __device__ constexpr bool mdw_psum(int n, int wmode) { return n == 128 && wmode == WM_F32; }
__host__ __device__ constexpr int mdw_epl(int th, int tw, int nwave)
{
    const int nrp = (th + 4) * (tw + 4), mto = th * tw / 16, mtow = (mto + nwave - 1) / nwave;
    const bool adj = mto % nwave == 0 && mtow > 1 && tw % mtow == 0;
    return ((nrp + 15) / 16) * 16 + (adj ? 1 : 0) + YF_MDW_EPL_ADD;
}

// HM (head mode): 0 = no head conv; 1 = head conv of up to 32 channels (one pair of MFMA column tiles, fragments staged in LDS with the
// rest of the stream: the shipped 3 x (5 + 3) = 24 and every num_out <= 32); 2 = any width: a run-time loop over pairs of column
// tiles whose fragments come straight from global memory / L2 (they would not fit the LDS beside the stream: 255 channels = 128 KB).
// a.headn is the real channel count in both modes.
template <int C, int N, int HM, int TH, int TW, int NWAVE, typename TT>
__global__ void __launch_bounds__(NWAVE * 64) mdw_kernel(MdwArgs a)
{
    constexpr bool X3 = is_x3<TT>::value;       // fp32 storage, split-operand fp16 MFMAs (yf_kernels.h DT_F16X3)
    constexpr bool H16 = sizeof(TT) == 2 || X3;  // the 1x1 convs run on v_mfma_f32_16x16x16_f16
    constexpr int WM = X3 ? 2 : 1;
    constexpr int NTHR = NWAVE * 64;
    constexpr int RH = TH + 4, RW = TW + 4, NRP = RH * RW;
    constexpr int MTO = (TH * TW) / 16, MTOW = (MTO + NWAVE - 1) / NWAVE;
    constexpr bool EVEN = MTO % NWAVE == 0;  // every wave owns MTOW tiles: no wave-uniform branches in the chunk loop (with them the
                                             // fp16 build copied all accumulators around every branch: 1300 v_mov of 1900 VALU instructions)
    constexpr int EPL = mdw_epl(TH, TW, NWAVE);   // pixels per 4-channel plane (see mdw_epl)
    constexpr int NT = N / 16, NCH = C / 16;
    constexpr int OFF_BD = 400, OFF_W = 416, CHUNK = 416 + (H16 ? WM * NT * 128 : 4 * NT * 64);
    constexpr int OFF_BPW = NCH * CHUNK, OFF_HW = OFF_BPW + N, KSH = N / 4, NTH = 2;
    constexpr int OFF_HB = OFF_HW + (H16 ? WM * (N / 16) * NTH * 128 : KSH * NTH * 64);
    constexpr int WFLOATS = mdw_stream_floats(C, N, HM == 1 ? 32 : 0, wmode_of<TT>());   // what is staged in LDS
    constexpr int NLD = (NRP * 4 + NTHR - 1) / NTHR;  // float4 loads per thread per chunk
    static_assert((TH * TW) % 16 == 0 && C % 16 == 0 && N % 16 == 0 && CHUNK == mdw_chunk_floats(N, wmode_of<TT>()), "shape");
    extern __shared__ __attribute__((aligned(16))) float mdw_smem[];
    float* E = mdw_smem;           // [4][EPL][4]
    float* WL = E + 16 * EPL;      // weight stream

    const int b = blockIdx.x;
    const int tx = b % a.tiles_x, ty = (b / a.tiles_x) % a.tiles_y, n = b / (a.tiles_x * a.tiles_y);
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const TT* __restrict__ src = reinterpret_cast<const TT*>(a.in) + (long)n * a.H * a.W * C;

    MDW_STAMP_DECL
#ifndef YF_MDW_ROLLED_STAGING
    stage_to_lds<WFLOATS, NTHR>(WL, a.wp);
#else   // round 2's staging loop (A/B builds only): one exposed L2 round trip per iteration
    for (int i = threadIdx.x * 4; i < WFLOATS; i += NTHR * 4)
        *reinterpret_cast<float4*>(&WL[i]) = *reinterpret_cast<const float4*>(a.wp + i);
#endif

    // chunk-invariant addressing of this thread's NLD fill items: item id -> (region pixel, channel quad)
    int goff[NLD];  // frame-relative pixel index, -1 = outside the image (zero fill), -2 = no item
#pragma unroll
    for (int m = 0; m < NLD; ++m) {
        const int id = threadIdx.x + m * NTHR;
        const int px = id >> 2;
        const int ry = px / RW, rx = px - ry * RW;
        const int iy = oy0 - 2 + ry, ix = ox0 - 2 + rx;
        goff[m] = px >= NRP ? -2 : (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) ? iy * a.W + ix : -1;
    }
    float4 pf[NLD];
    auto prefetch = [&](int c) {
#pragma unroll
        for (int m = 0; m < NLD; ++m) {
            const int id = threadIdx.x + m * NTHR;
            pf[m] = goff[m] >= 0 ? ld4<TT>(src + (long)goff[m] * C + c * 16 + 4 * (id & 3))
                                 : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    prefetch(0);

    // Which output pixel row r of the wave's i-th M-tile is.  ADJ: the MTOW tiles of a wave interleave, so that a lane's pixels
    // are horizontally ADJACENT (x, x + 1, ..): their 5x5 windows overlap and a window row costs 4 + MTOW LDS reads instead of
    // 5 MTOW (the depthwise phase is the kernel's largest and it is LDS-bound: 43 % of the time by the phase stamps).
    constexpr bool ADJ = EVEN && MTOW > 1 && TW % MTOW == 0;
    auto out_pixel = [&](int w, int i, int rr) {
        if constexpr (ADJ) return (w * 16 + rr) * MTOW + i;
        const int mo = w + i * NWAVE;
        return (mo < MTO ? mo : 0) * 16 + rr;
    };
    f32x4 acc[MTOW][NT];
    int rp0[MTOW];
#pragma unroll
    for (int i = 0; i < MTOW; ++i) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[i][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int op = out_pixel(wave, i, r);
        const int oy = op / TW, ox = op - oy * TW;
        rp0[i] = (oy + 2) * RW + ox + 2;
    }

    MDW_STAMP(0)   // prologue: weight staging requests, addressing, first prefetch
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
        // ---- fill E with this chunk (prefetched), then request the next chunk ----
#pragma unroll
        for (int m = 0; m < NLD; ++m) {
            const int id = threadIdx.x + m * NTHR;
            if (goff[m] != -2) *reinterpret_cast<float4*>(&E[((id & 3) * EPL + (id >> 2)) * 4]) = pf[m];
        }
        MDW_STAMP(1)   // E fill (waits for the prefetched chunk)
        __syncthreads();
        MDW_STAMP(2)   // barrier 1
        if (c + 1 < NCH) prefetch(c + 1);
        // ---- depthwise 5x5 of channels 4q..4q+3 at this lane's output pixels (taps outer: one weight read per tap) ----
        const float* wc = WL + c * CHUNK;
        const float4 bd = *reinterpret_cast<const float4*>(wc + OFF_BD + 4 * q);
        mdw_f32x2 d2[MTOW][2];   // channel pairs (4q, 4q+1), (4q+2, 4q+3): the taps are v_pk_fma_f32
#pragma unroll
        for (int i = 0; i < MTOW; ++i) { d2[i][0] = mdw_f32x2{bd.x, bd.y}; d2[i][1] = mdw_f32x2{bd.z, bd.w}; }
        const float4* e4 = reinterpret_cast<const float4*>(E) + q * EPL;
        // One window row at a time: all of the row's LDS reads (activations AND its five tap weights) are issued together, then
        // its FMAs.  (Reading each tap's weight right before its 8 FMAs exposed an LDS round trip per tap: 230 cycles per tap,
        // 43 % of the kernel by the phase stamps; double-buffering whole rows spills at 10 waves per workgroup.)
        constexpr int NV = ADJ ? 4 + MTOW : 5 * MTOW;   // activation reads per window row
#pragma unroll
        for (int ky = 0; ky < 5; ++ky) {
            float4 vc[NV], wr[5];
#pragma unroll
            for (int kx = 0; kx < 5; ++kx) wr[kx] = *reinterpret_cast<const float4*>(wc + (ky * 5 + kx) * 16 + 4 * q);
            if constexpr (ADJ) {
#pragma unroll
                for (int j = 0; j < NV; ++j) vc[j] = e4[rp0[0] + (ky - 2) * RW + (j - 2)];
            } else {
#pragma unroll
                for (int i = 0; i < MTOW; ++i)
#pragma unroll
                    for (int kx = 0; kx < 5; ++kx) vc[i * 5 + kx] = e4[rp0[i] + (ky - 2) * RW + (kx - 2)];   // (clamped tile for idle slots)
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kx = 0; kx < 5; ++kx)
#pragma unroll
                for (int i = 0; i < MTOW; ++i) {
                    const float4 v = ADJ ? vc[kx + i] : vc[i * 5 + kx];
                    // two channels per instruction (v_pk_fma_f32): the same fused multiply-add per element
                    d2[i][0] = __builtin_elementwise_fma(mdw_f32x2{v.x, v.y}, mdw_f32x2{wr[kx].x, wr[kx].y}, d2[i][0]);
                    d2[i][1] = __builtin_elementwise_fma(mdw_f32x2{v.z, v.w}, mdw_f32x2{wr[kx].z, wr[kx].w}, d2[i][1]);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        MDW_STAMP(3)   // depthwise
        // ---- 1x1 conv: A = relu(d) (k-step j = channel 4q+j), B fragments from the staged stream ----
        if constexpr (X3) {
            f16x4 w2h[NT], w2l[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                w2h[nt] = reinterpret_cast<const f16x4*>(wc + OFF_W)[nt * 64 + lane];
                w2l[nt] = reinterpret_cast<const f16x4*>(wc + OFF_W)[(NT + nt) * 64 + lane];
            }
#pragma unroll
            for (int i = 0; i < MTOW; ++i) {
                if (EVEN || wave + i * NWAVE < MTO) {
                    f16x4 dh, dl;
                    split_f16x4(fmaxf(d2[i][0][0], 0.f), fmaxf(d2[i][0][1], 0.f), fmaxf(d2[i][1][0], 0.f), fmaxf(d2[i][1][1], 0.f), dh, dl);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x16f16(w2l[nt], dh, acc[i][nt], 0, 0, 0);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x16f16(w2h[nt], dl, acc[i][nt], 0, 0, 0);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x16f16(w2h[nt], dh, acc[i][nt], 0, 0, 0);
                }
            }
        } else if constexpr (H16) {
            f16x4 w2h[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) w2h[nt] = reinterpret_cast<const f16x4*>(wc + OFF_W)[nt * 64 + lane];
#pragma unroll
            for (int i = 0; i < MTOW; ++i) {
                if (EVEN || wave + i * NWAVE < MTO) {
                    const f16x4 dh = f16x4{(half_t)fmaxf(d2[i][0][0], 0.f), (half_t)fmaxf(d2[i][0][1], 0.f), (half_t)fmaxf(d2[i][1][0], 0.f),
                                           (half_t)fmaxf(d2[i][1][1], 0.f)};
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x16f16(w2h[nt], dh, acc[i][nt], 0, 0, 0);
                }
            }
        } else if constexpr (mdw_psum(N, wmode_of<TT>())) {
            // PSUM (round 6, the small head's two 1x1 convs in fp32): one partial sum per 16-channel chunk, added to the running total -- the
            // association of mdw2_esplit_kernel (chunk c on its own workgroup), so both forms give the same bits (yf_mres_kernels.hip, PSUM)
            f32x4 part[MTOW][NT];
#pragma unroll
            for (int i = 0; i < MTOW; ++i)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) part[i][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float w2f[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) w2f[nt] = wc[OFF_W + (j * NT + nt) * 64 + lane];
#pragma unroll
                for (int i = 0; i < MTOW; ++i) {
                    if (EVEN || wave + i * NWAVE < MTO) {
                        const float dj = __int_as_float(max(__float_as_int(d2[i][j >> 1][j & 1]), 0));
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) part[i][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2f[nt], dj, part[i][nt], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < MTOW; ++i)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[i][nt] += part[i][nt];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float w2f[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) w2f[nt] = wc[OFF_W + (j * NT + nt) * 64 + lane];
#pragma unroll
                for (int i = 0; i < MTOW; ++i) {
                    if (EVEN || wave + i * NWAVE < MTO) {
                        const float dj = __int_as_float(max(__float_as_int(d2[i][j >> 1][j & 1]), 0));   // ReLU of a non-NaN as one v_max_i32
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2f[nt], dj, acc[i][nt], 0, 0, 0);
                    }
                }
            }
        }
        MDW_STAMP(4)   // 1x1 conv MFMAs
        __syncthreads();
        MDW_STAMP(5)   // barrier 2
    }

    // ---- epilogue.  The 1x1 conv ran with its weights as the MFMA's A operand and the depthwise result as B (the same
    // fragments, swapped), so lane (r, q) holds output channels nt*16 + 4q .. +3 of pixel mo*16 + r.  Without a head that is one
    // 16-byte NHWC store per lane and n-tile; with a head it IS the B operand of the head GEMM (k-step (nt, reg) <-> channel
    // nt*16 + 4q + reg, which is how the head's fragments are packed anyway), so the head conv chains in registers -- no LDS
    // transposition -- and its logits come out as lane (pixel r, head channels 4q .. +3): NCHW stores of 16 consecutive pixels ----
    const float* bpw = WL + OFF_BPW;
#pragma unroll
    for (int i = 0; i < MTOW; ++i) {  // (unrolled: a runtime-indexed acc[] would live in scratch)
        const int mo = wave + i * NWAVE;
        if (!EVEN && mo >= MTO) continue;
        const int op = out_pixel(wave, i, r);
        const int oy = op / TW, ox = op - oy * TW;
        const int gy = oy0 + oy, gx = ox0 + ox;
        const bool inside = gy < a.H && gx < a.W;
        if constexpr (HM == 0) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float4 bias = *reinterpret_cast<const float4*>(bpw + nt * 16 + 4 * q);
                if (inside)
                    st4<TT>(reinterpret_cast<TT*>(a.out) + (((long)n * a.H + gy) * a.W + gx) * N + nt * 16 + 4 * q,
                            make_float4(acc[i][nt][0] + bias.x, acc[i][nt][1] + bias.y, acc[i][nt][2] + bias.z, acc[i][nt][3] + bias.w));
            }
        } else if constexpr (HM == 1) {
            const float* hw = WL + OFF_HW;
            const float* hb = WL + OFF_HB;
            f32x4 h[NTH] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float4 bias = *reinterpret_cast<const float4*>(bpw + nt * 16 + 4 * q);
                const float hv[4] = {acc[i][nt][0] + bias.x, acc[i][nt][1] + bias.y, acc[i][nt][2] + bias.z, acc[i][nt][3] + bias.w};
                if constexpr (X3) {
                    f16x4 bh, bl;
                    split_f16x4(hv[0], hv[1], hv[2], hv[3], bh, bl);
                    const f16x4* hwh = reinterpret_cast<const f16x4*>(hw);
                    const f16x4* hwl = hwh + (N / 16) * NTH * 64;
#pragma unroll
                    for (int nth = 0; nth < NTH; ++nth) {
                        h[nth] = __builtin_amdgcn_mfma_f32_16x16x16f16(hwl[(nt * NTH + nth) * 64 + lane], bh, h[nth], 0, 0, 0);
                        h[nth] = __builtin_amdgcn_mfma_f32_16x16x16f16(hwh[(nt * NTH + nth) * 64 + lane], bl, h[nth], 0, 0, 0);
                        h[nth] = __builtin_amdgcn_mfma_f32_16x16x16f16(hwh[(nt * NTH + nth) * 64 + lane], bh, h[nth], 0, 0, 0);
                    }
                } else if constexpr (H16) {
                    const f16x4 bh = f16x4{(half_t)hv[0], (half_t)hv[1], (half_t)hv[2], (half_t)hv[3]};
#pragma unroll
                    for (int nth = 0; nth < NTH; ++nth)
                        h[nth] = __builtin_amdgcn_mfma_f32_16x16x16f16(reinterpret_cast<const f16x4*>(hw)[(nt * NTH + nth) * 64 + lane], bh, h[nth], 0, 0, 0);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int nth = 0; nth < NTH; ++nth)
                            h[nth] = __builtin_amdgcn_mfma_f32_16x16x4f32(hw[((nt * 4 + j) * NTH + nth) * 64 + lane], hv[j], h[nth], 0, 0, 0);
                }
            }
#pragma unroll
            for (int nth = 0; nth < NTH; ++nth)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int hc = nth * 16 + 4 * q + reg;
                    if (hc < a.headn && inside) a.out[(((long)n * a.headn + hc) * a.H + gy) * a.W + gx] = h[nth][reg] + hb[hc];  // NCHW
                }
        }
    }
    if constexpr (HM == 2) {
        // any head width: pairs of column tiles in a run-time loop, the fragments of a pair from global memory / L2.  The pair loop is
        // the OUTER one: a pair's fragments (all k-blocks) are requested together and serve every M-tile of the wave (with the tile
        // loop outside, each tile re-read them, k-block by k-block: 122 us for the 255-channel head_4 launch against 40 us for 24
        // channels).  Same k order per output channel as mode 1.
        const int nthp = mdw_head_tiles(a.headn);
        const float* hwg = a.wp + OFF_HW;
        const float* hbg = hwg + (H16 ? WM * (N / 16) * nthp * 128 : KSH * nthp * 64);
        int gyv[MTOW], gxv[MTOW];
        bool ins[MTOW];
#pragma unroll
        for (int i = 0; i < MTOW; ++i) {
            const int op = out_pixel(wave, i, r);
            const int oy = op / TW, ox = op - oy * TW;
            gyv[i] = oy0 + oy; gxv[i] = ox0 + ox;
            ins[i] = (EVEN || wave + i * NWAVE < MTO) && gyv[i] < a.H && gxv[i] < a.W;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float4 bias = *reinterpret_cast<const float4*>(bpw + nt * 16 + 4 * q);
                acc[i][nt][0] += bias.x; acc[i][nt][1] += bias.y; acc[i][nt][2] += bias.z; acc[i][nt][3] += bias.w;
            }
        }
#pragma unroll 1
        for (int pr = 0; pr < nthp; pr += 2) {
            if constexpr (X3) {
                const f16x4* hwh = reinterpret_cast<const f16x4*>(hwg);
                const f16x4* hwl = hwh + (N / 16) * nthp * 64;
                f16x4 fh[NT][2], fl[NT][2];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int nth = 0; nth < 2; ++nth) { fh[nt][nth] = hwh[(nt * nthp + pr + nth) * 64 + lane]; fl[nt][nth] = hwl[(nt * nthp + pr + nth) * 64 + lane]; }
#pragma unroll
                for (int i = 0; i < MTOW; ++i) {
                    if (!EVEN && wave + i * NWAVE >= MTO) continue;
                    f32x4 h[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        f16x4 bh, bl;
                        split_f16x4(acc[i][nt][0], acc[i][nt][1], acc[i][nt][2], acc[i][nt][3], bh, bl);
#pragma unroll
                        for (int nth = 0; nth < 2; ++nth) {
                            h[nth] = __builtin_amdgcn_mfma_f32_16x16x16f16(fl[nt][nth], bh, h[nth], 0, 0, 0);
                            h[nth] = __builtin_amdgcn_mfma_f32_16x16x16f16(fh[nt][nth], bl, h[nth], 0, 0, 0);
                            h[nth] = __builtin_amdgcn_mfma_f32_16x16x16f16(fh[nt][nth], bh, h[nth], 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int nth = 0; nth < 2; ++nth)
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) {
                            const int hc = (pr + nth) * 16 + 4 * q + reg;
                            if (hc < a.headn && ins[i]) a.out[(((long)n * a.headn + hc) * a.H + gyv[i]) * a.W + gxv[i]] = h[nth][reg] + hbg[hc];  // NCHW
                        }
                }
            } else if constexpr (H16) {
                const f16x4* hwh = reinterpret_cast<const f16x4*>(hwg);
                f16x4 fh[NT][2];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int nth = 0; nth < 2; ++nth) fh[nt][nth] = hwh[(nt * nthp + pr + nth) * 64 + lane];
#pragma unroll
                for (int i = 0; i < MTOW; ++i) {
                    if (!EVEN && wave + i * NWAVE >= MTO) continue;
                    f32x4 h[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const f16x4 bh = f16x4{(half_t)acc[i][nt][0], (half_t)acc[i][nt][1], (half_t)acc[i][nt][2], (half_t)acc[i][nt][3]};
#pragma unroll
                        for (int nth = 0; nth < 2; ++nth) h[nth] = __builtin_amdgcn_mfma_f32_16x16x16f16(fh[nt][nth], bh, h[nth], 0, 0, 0);
                    }
#pragma unroll
                    for (int nth = 0; nth < 2; ++nth)
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) {
                            const int hc = (pr + nth) * 16 + 4 * q + reg;
                            if (hc < a.headn && ins[i]) a.out[(((long)n * a.headn + hc) * a.H + gyv[i]) * a.W + gxv[i]] = h[nth][reg] + hbg[hc];  // NCHW
                        }
                }
            } else {
                float f[NT][4][2];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int nth = 0; nth < 2; ++nth) f[nt][j][nth] = hwg[((nt * 4 + j) * nthp + pr + nth) * 64 + lane];
#pragma unroll
                for (int i = 0; i < MTOW; ++i) {
                    if (!EVEN && wave + i * NWAVE >= MTO) continue;
                    f32x4 h[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int nth = 0; nth < 2; ++nth) h[nth] = __builtin_amdgcn_mfma_f32_16x16x4f32(f[nt][j][nth], acc[i][nt][j], h[nth], 0, 0, 0);
#pragma unroll
                    for (int nth = 0; nth < 2; ++nth)
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) {
                            const int hc = (pr + nth) * 16 + 4 * q + reg;
                            if (hc < a.headn && ins[i]) a.out[(((long)n * a.headn + hc) * a.H + gyv[i]) * a.W + gxv[i]] = h[nth][reg] + hbg[hc];  // NCHW
                        }
                }
            }
        }
    }
    MDW_STAMP(6)   // epilogue
    MDW_STAMP_FLUSH
}

// ------------------------------------------------------------------------------------------------
// mdw2_kernel: the SMALL head's two pairs in one launch (fusion level 2)
//     conv5_3 (dw5x5+ReLU) -> conv5_4 (1x1)   ->   conv5_5 (dw5x5+ReLU) -> conv5_6 (1x1) -> head_5     yolo_fastest.py:200-206
// for frames that fit ONE tile (the stride-32 frame of the 320x256 net is 8 x 10 pixels), so that stage 2's 5x5 windows need no
// pixels of another workgroup.  Stage 1 is mdw_kernel's chunk loop; its result (+ bias, zero outside the image = stage 2's zero
// padding) goes to LDS as Y[16-ch chunk][4-ch group][pixel][4] instead of HBM.  Stage 2 is the same chunk loop with the E fill taken
// from Y (one 16-byte LDS copy per thread and chunk; E's halo ring is zero from stage 1's last fill) and the head conv chained in
// registers.  What the single launch removes: the second launch's fill/drain, the conv5_4 tensor's HBM round trip, and stage 2's
// exposed weight staging -- the first SPLIT chunks of stage 2's weight stream are staged beside stage 1's at kernel start, the rest
// is requested into registers before stage 1's epilogue and lands in stage 1's (then dead) weight region behind one barrier.
// Arithmetic and its order are mdw_kernel's: the heads are bitwise those of the two-launch plan.
// ------------------------------------------------------------------------------------------------
struct Mdw2Args {
    const float* in;    // NHWC [N,H,W,C1]
    const float* wp1;   // stage 1 stream: mdw_pack_weights(C1, N1, no head)
    const float* wp2;   // stage 2 stream: mdw_pack_weights(N1, N2, HEADN)
    float* out;         // NCHW [N,headn,H,W]
    int H, W;
    int headn;          // the head conv's output channels (<= 32: one pair of MFMA column tiles)
};

__host__ __device__ constexpr int mdw2_split(int c1, int n1, int n2, int headn, int wmode)
{
    // chunks of stage 2's stream staged at kernel start: the fewest that let the REST fit into stage 1's weight region
    int s = 0;
    while (mdw_stream_floats(n1, n2, headn, wmode) - s * mdw_chunk_floats(n2, wmode) > mdw_stream_floats(c1, n1, 0, wmode)) ++s;
    return s;
}
__host__ __device__ constexpr size_t mdw2_lds_floats(int c1, int n1, int n2, int headn, int th, int tw, int nwave, int wmode)
{
    return (size_t)16 * mdw_epl(th, tw, nwave) + (size_t)n1 * th * tw + mdw_stream_floats(c1, n1, 0, wmode) +
           (size_t)mdw2_split(c1, n1, n2, headn, wmode) * mdw_chunk_floats(n2, wmode);
}

// depthwise 5x5 of channels 4q..4q+3 at this lane's pixel + the 1x1 conv's MFMAs of one 16-channel chunk (one M-tile per wave)
template <int N, int RW, typename TT>
__device__ __forceinline__ void mdw_chunk_1tile(const float* E, int epl, const float* wc, int lane, int rp0, f32x4 (&acc)[N / 16])
{
    constexpr bool X3 = is_x3<TT>::value;
    constexpr bool H16 = sizeof(TT) == 2 || X3;
    constexpr int NT = N / 16, OFF_BD = 400, OFF_W = 416;
    const int q = lane >> 4;
    const float4 bd = *reinterpret_cast<const float4*>(wc + OFF_BD + 4 * q);
    mdw_f32x2 d2[2] = {mdw_f32x2{bd.x, bd.y}, mdw_f32x2{bd.z, bd.w}};
    const float4* e4 = reinterpret_cast<const float4*>(E) + q * epl;
#pragma unroll
    for (int ky = 0; ky < 5; ++ky) {
        float4 vc[5], wr[5];
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) wr[kx] = *reinterpret_cast<const float4*>(wc + (ky * 5 + kx) * 16 + 4 * q);
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) vc[kx] = e4[rp0 + (ky - 2) * RW + (kx - 2)];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) {
            const float4 v = vc[kx];
            d2[0] = __builtin_elementwise_fma(mdw_f32x2{v.x, v.y}, mdw_f32x2{wr[kx].x, wr[kx].y}, d2[0]);
            d2[1] = __builtin_elementwise_fma(mdw_f32x2{v.z, v.w}, mdw_f32x2{wr[kx].z, wr[kx].w}, d2[1]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (X3) {
        f16x4 dh, dl;
        split_f16x4(fmaxf(d2[0][0], 0.f), fmaxf(d2[0][1], 0.f), fmaxf(d2[1][0], 0.f), fmaxf(d2[1][1], 0.f), dh, dl);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x16f16(reinterpret_cast<const f16x4*>(wc + OFF_W)[(NT + nt) * 64 + lane], dh, acc[nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x16f16(reinterpret_cast<const f16x4*>(wc + OFF_W)[nt * 64 + lane], dl, acc[nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x16f16(reinterpret_cast<const f16x4*>(wc + OFF_W)[nt * 64 + lane], dh, acc[nt], 0, 0, 0);
    } else if constexpr (H16) {
        const f16x4 dh = f16x4{(half_t)fmaxf(d2[0][0], 0.f), (half_t)fmaxf(d2[0][1], 0.f), (half_t)fmaxf(d2[1][0], 0.f), (half_t)fmaxf(d2[1][1], 0.f)};
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x16f16(reinterpret_cast<const f16x4*>(wc + OFF_W)[nt * 64 + lane], dh, acc[nt], 0, 0, 0);
    } else {
        constexpr bool PSUM = mdw_psum(N, WM_F32);      // per-chunk partial sums: mdw_kernel, PSUM
        f32x4 part[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) part[nt] = PSUM ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[nt];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float w2f[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) w2f[nt] = wc[OFF_W + (j * NT + nt) * 64 + lane];
            const float dj = __int_as_float(max(__float_as_int(d2[j >> 1][j & 1]), 0));   // ReLU of a non-NaN as one v_max_i32
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) part[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2f[nt], dj, part[nt], 0, 0, 0);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = PSUM ? acc[nt] + part[nt] : part[nt];
    }
}

template <int C1, int N1, int N2, int TH, int TW, int NWAVE, typename TT>
__global__ void __launch_bounds__(NWAVE * 64) mdw2_kernel(Mdw2Args a)
{
    constexpr int HEADN = 32;   // stream layout of any head of up to 32 channels; a.headn is the real count
    constexpr bool X3 = is_x3<TT>::value;
    constexpr bool H16 = sizeof(TT) == 2 || X3;
    constexpr int WMODE = wmode_of<TT>(), WM = X3 ? 2 : 1;
    constexpr int NTHR = NWAVE * 64;
    constexpr int RH = TH + 4, RW = TW + 4, NRP = RH * RW, NPX = TH * TW;
    constexpr int MTO = NPX / 16;
    static_assert(MTO == NWAVE && NPX % 16 == 0, "one M-tile per wave");
    constexpr int EPL = mdw_epl(TH, TW, NWAVE);
    constexpr int NT1 = N1 / 16, NCH1 = C1 / 16, NT2 = N2 / 16, NCH2 = N1 / 16;
    constexpr int CHUNK1 = mdw_chunk_floats(N1, WMODE), CHUNK2 = mdw_chunk_floats(N2, WMODE);
    constexpr int WF1 = mdw_stream_floats(C1, N1, 0, WMODE), WF2 = mdw_stream_floats(N1, N2, HEADN, WMODE);
    constexpr int SPLIT = mdw2_split(C1, N1, N2, HEADN, WMODE), PART_A = SPLIT * CHUNK2, PART_B = WF2 - PART_A;
    static_assert(PART_B <= WF1 && SPLIT <= NCH2 && PART_A % 4 == 0 && PART_B % 4 == 0, "stage 2's stream must fit");
    constexpr int OFF_BPW1 = NCH1 * CHUNK1;
    // stage 2's stream, as addressed from its chunk SPLIT on (region A): [chunks SPLIT.. | b_pw | head fragments | head bias]
    constexpr int OFF_BPW2 = (NCH2 - SPLIT) * CHUNK2, OFF_HW = OFF_BPW2 + N2, KSH = N2 / 4, NTH = 2;
    constexpr int OFF_HB = OFF_HW + (H16 ? WM * (N2 / 16) * NTH * 128 : KSH * NTH * 64);
    constexpr int NLD = (NRP * 4 + NTHR - 1) / NTHR;
    constexpr int NB4 = (PART_B / 4 + NTHR - 1) / NTHR;
    extern __shared__ __attribute__((aligned(16))) float mdw_smem[];
    float* E = mdw_smem;                 // [4][EPL][4]
    float* Y = E + 16 * EPL;             // [NCH2][4][NPX][4]: stage 1's result
    float* WA = Y + N1 * NPX;            // stage 1's stream, later the rest of stage 2's
    float* WB = WA + WF1;                // the first SPLIT chunks of stage 2's stream

    const int n = blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const TT* __restrict__ src = reinterpret_cast<const TT*>(a.in) + (long)n * a.H * a.W * C1;

    {   // both weight regions in one batch of loads
        LdsStage<WF1, NTHR> sa;
        LdsStage<PART_A, NTHR> sb;
        sa.issue(a.wp1);
        sb.issue(a.wp2);
        sa.commit(WA);
        sb.commit(WB);
    }

    int goff[NLD];  // frame-relative pixel index, -1 = outside the image (zero fill), -2 = no item
#pragma unroll
    for (int m = 0; m < NLD; ++m) {
        const int id = threadIdx.x + m * NTHR;
        const int px = id >> 2;
        const int ry = px / RW, rx = px - ry * RW;
        const int iy = ry - 2, ix = rx - 2;
        goff[m] = px >= NRP ? -2 : (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) ? iy * a.W + ix : -1;
    }
    float4 pf[NLD];
    auto prefetch = [&](int c) {
#pragma unroll
        for (int m = 0; m < NLD; ++m) {
            const int id = threadIdx.x + m * NTHR;
            pf[m] = goff[m] >= 0 ? ld4<TT>(src + (long)goff[m] * C1 + c * 16 + 4 * (id & 3)) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    prefetch(0);

    const int op = wave * 16 + r, oy = op / TW, ox = op - oy * TW;   // this lane's output pixel (both stages)
    const int rp0 = (oy + 2) * RW + ox + 2;
    const bool inside = oy < a.H && ox < a.W;

    // ================= stage 1: dw5x5 (C1) -> 1x1 (N1) =================
    f32x4 acc1[NT1];
#pragma unroll
    for (int nt = 0; nt < NT1; ++nt) acc1[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int c = 0; c < NCH1; ++c) {
#pragma unroll
        for (int m = 0; m < NLD; ++m) {
            const int id = threadIdx.x + m * NTHR;
            if (goff[m] != -2) *reinterpret_cast<float4*>(&E[((id & 3) * EPL + (id >> 2)) * 4]) = pf[m];
        }
        __syncthreads();
        if (c + 1 < NCH1) prefetch(c + 1);
        mdw_chunk_1tile<N1, RW, TT>(E, EPL, WA + c * CHUNK1, lane, rp0, acc1);
        __syncthreads();
    }
    // the rest of stage 2's stream: requested now, written over stage 1's stream once every wave is past its epilogue
    f32x4 wv[NB4];
#pragma unroll
    for (int i = 0; i < NB4; ++i) {
        const int idx = threadIdx.x + i * NTHR;
        wv[i] = *reinterpret_cast<const f32x4*>(a.wp2 + PART_A + 4 * (idx < PART_B / 4 ? idx : PART_B / 4 - 1));
    }
    // stage 1's epilogue: + bias -> Y (lane (r, q) holds channels nt*16 + 4q .. +3 of its pixel = one record of stage 2's chunk nt);
    // pixels outside the image are stage 2's zero padding
    {
        const float* bpw = WA + OFF_BPW1;
#pragma unroll
        for (int nt = 0; nt < NT1; ++nt) {
            const float4 bias = *reinterpret_cast<const float4*>(bpw + nt * 16 + 4 * q);
            const float4 v = inside ? make_float4(acc1[nt][0] + bias.x, acc1[nt][1] + bias.y, acc1[nt][2] + bias.z, acc1[nt][3] + bias.w)
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(&Y[((nt * 4 + q) * NPX + op) * 4]) = v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NB4; ++i) {
        const int idx = threadIdx.x + i * NTHR;
        if (idx < PART_B / 4) *reinterpret_cast<f32x4*>(&WA[4 * idx]) = wv[i];
    }
    // E fill of a stage-2 chunk: interior pixels from Y (the halo ring of E is zero since stage 1's last fill: tile == frame)
    auto fill2 = [&](int c) {
        for (int idx = threadIdx.x; idx < 4 * NPX; idx += NTHR) {
            const int qq = idx / NPX, p = idx - qq * NPX;
            const int py = p / TW, px = p - py * TW;
            *reinterpret_cast<float4*>(&E[(qq * EPL + (py + 2) * RW + px + 2) * 4]) = *reinterpret_cast<const float4*>(&Y[((c * 4 + qq) * NPX + p) * 4]);
        }
    };
    fill2(0);
    __syncthreads();

    // ================= stage 2: dw5x5 (N1) -> 1x1 (N2) -> head =================
    f32x4 acc2[NT2];
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt) acc2[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int c = 0; c < NCH2; ++c) {
        const float* wc = c < SPLIT ? WB + c * CHUNK2 : WA + (c - SPLIT) * CHUNK2;
        mdw_chunk_1tile<N2, RW, TT>(E, EPL, wc, lane, rp0, acc2);
        __syncthreads();
        if (c + 1 < NCH2) {
            fill2(c + 1);
            __syncthreads();
        }
    }
    // head conv chained in registers (mdw_kernel's epilogue): logits NCHW
    {
        const float* bpw = WA + OFF_BPW2;
        const float* hw = WA + OFF_HW;
        const float* hb = WA + OFF_HB;
        f32x4 h[NTH] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) {
            const float4 bias = *reinterpret_cast<const float4*>(bpw + nt * 16 + 4 * q);
            const float hv[4] = {acc2[nt][0] + bias.x, acc2[nt][1] + bias.y, acc2[nt][2] + bias.z, acc2[nt][3] + bias.w};
            if constexpr (X3) {
                f16x4 bh, bl;
                split_f16x4(hv[0], hv[1], hv[2], hv[3], bh, bl);
                const f16x4* hwh = reinterpret_cast<const f16x4*>(hw);
                const f16x4* hwl = hwh + (N2 / 16) * NTH * 64;
#pragma unroll
                for (int nth = 0; nth < NTH; ++nth) {
                    h[nth] = __builtin_amdgcn_mfma_f32_16x16x16f16(hwl[(nt * NTH + nth) * 64 + lane], bh, h[nth], 0, 0, 0);
                    h[nth] = __builtin_amdgcn_mfma_f32_16x16x16f16(hwh[(nt * NTH + nth) * 64 + lane], bl, h[nth], 0, 0, 0);
                    h[nth] = __builtin_amdgcn_mfma_f32_16x16x16f16(hwh[(nt * NTH + nth) * 64 + lane], bh, h[nth], 0, 0, 0);
                }
            } else if constexpr (H16) {
                const f16x4 bh = f16x4{(half_t)hv[0], (half_t)hv[1], (half_t)hv[2], (half_t)hv[3]};
#pragma unroll
                for (int nth = 0; nth < NTH; ++nth)
                    h[nth] = __builtin_amdgcn_mfma_f32_16x16x16f16(reinterpret_cast<const f16x4*>(hw)[(nt * NTH + nth) * 64 + lane], bh, h[nth], 0, 0, 0);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int nth = 0; nth < NTH; ++nth)
                        h[nth] = __builtin_amdgcn_mfma_f32_16x16x4f32(hw[((nt * 4 + j) * NTH + nth) * 64 + lane], hv[j], h[nth], 0, 0, 0);
            }
        }
#pragma unroll
        for (int nth = 0; nth < NTH; ++nth)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int hc = nth * 16 + 4 * q + reg;
                if (hc < a.headn && inside) a.out[(((long)n * a.headn + hc) * a.H + oy) * a.W + ox] = h[nth][reg] + hb[hc];  // NCHW
            }
    }
}

template <int C1, int N1, int N2, int TH, int TW, int NWAVE, typename T>
static int launch_mdw2_t(const Mdw2Args& a, int Nf, hipStream_t s)
{
    if (a.H > TH || a.W > TW || a.headn < 1 || a.headn > 32) return -4;   // the chain needs tile == frame and one pair of head tiles
    constexpr size_t lds = mdw2_lds_floats(C1, N1, N2, 32, TH, TW, NWAVE, wmode_of<T>()) * sizeof(float);
    static_assert(lds <= 160 * 1024, "LDS");
    static bool attr_done[YF_MAX_DEVICES] = {};
    const int dev = current_device();
    if (dev < 0) return -2;
    if (lds > 64 * 1024 && !attr_done[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(mdw2_kernel<C1, N1, N2, TH, TW, NWAVE, T>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return -2;
        attr_done[dev] = true;
    }
    hipLaunchKernelGGL((mdw2_kernel<C1, N1, N2, TH, TW, NWAVE, T>), dim3((unsigned)Nf), dim3(NWAVE * 64), lds, s, a);
    return 0;
}

// conv5_3 -> conv5_4 -> conv5_5 -> conv5_6 -> head_5 on H x W frames in one launch?
bool mdw2_can_chain(int c1, int n1, int n2, int headn, int H, int W)
{
    return c1 == 96 && n1 == 128 && n2 == 128 && headn >= 1 && headn <= 32 && H <= 8 && W <= 10;
}

// ------------------------------------------------------------------------------------------------
// mdw2_esplit_kernel (round 5, batch-1 latency): the small head -- conv5_3 (dw5x5) -> conv5_4 (1x1) -> conv5_5 (dw5x5) -> conv5_6 (1x1) ->
// head_5 (yolo_fastest.py:127-138) -- when only a few 8x10 frames are in flight.  One workgroup per frame takes 33 us on ONE CU.  Both
// dw -> 1x1 stages are sums over 16-channel chunks of their input (6 and 8 of them), so each chunk goes to its own workgroup, the
// PARTIAL 1x1 results go to HBM and the launch boundary is the exchange (yf_mres_kernels.hip mres_esplit_kernel, same scheme):
//   phase 0: chunk c of stage 1 (dw5x5 of 16 of conv5_2's channels + ReLU, partial conv5_4)                 N x 6 workgroups
//   phase 1: channels 16c .. 16c+15 of conv5_4 = sum of the six partials + bias (zero outside the frame: conv5_5's padding), dw5x5 + ReLU,
//            partial conv5_6                                                                               N x 8 workgroups
//   phase 2: conv5_6 = sum of the eight partials + bias, head conv chained through the accumulator layout, logits NCHW         N x 5 workgroups of one wave
// The bits of mdw2_kernel / mdw_kernel since round 6: their 1x1 sums are formed per chunk and added in chunk order as well (PSUM).
// fp32 engines, heads of up to 32 channels, frames of exactly 8x10.
// ------------------------------------------------------------------------------------------------
struct Mdw2EsplitArgs {
    const float* in;    // conv5_2, NHWC [N, 80, 96]
    const float* wp1;   // stage 1 stream (mdw_pack_weights: 6 chunks, then conv5_4's bias)
    const float* wp2;   // stage 2 stream (8 chunks, conv5_6's bias, head fragments, head bias)
    float* out;         // logits NCHW [N, headn, 8, 10]
    float* slab1;       // [N][6][80][128]
    float* slab2;       // [N][8][80][128]
    int headn, phase;
};

__global__ void __launch_bounds__(320) mdw2_esplit_kernel(Mdw2EsplitArgs a)
{
    constexpr int TH = 8, TW = 10, NPX = 80, C1 = 96, N1 = 128, N2 = 128, NCH1 = C1 / 16, NCH2 = N1 / 16, NT = 8;
    constexpr int RW = TW + 4, EPL = (((TH + 4) * RW + 15) / 16) * 16, CH = mdw_chunk_floats(128, WM_F32);
    static_assert(CH == 400 + 16 + 4 * NT * 64, "chunk layout");
    __shared__ __attribute__((aligned(16))) float E[4 * EPL * 4];
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    // phase 2 runs one wave per workgroup (a frame's five M-tiles on five CUs: each wave pulls 64 KB of partial sums through its CU's L2 port)
    const int wave = a.phase == 2 ? (int)(blockIdx.x % 5) : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int p = wave * 16 + r, py = p / TW, px = p - py * TW, rp = (py + 2) * RW + px + 2;
    if (a.phase == 2) {
        // ---- conv5_6 (+ bias) of this lane's pixel, 16 channels at a time, straight into the head conv's k-steps ----
        const int n = blockIdx.x / 5;
        const float* sl = a.slab2 + (long)n * NCH2 * NPX * N2 + (long)p * N2 + 4 * q;
        const float* bp = a.wp2 + NCH2 * CH;
        const float* hf = bp + N2;                                   // head fragments [(s)(2 tiles)(lane)]
        f32x4 hacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll 2
        for (int kb = 0; kb < N2 / 16; ++kb) {
            float4 y = *reinterpret_cast<const float4*>(sl + kb * 16);
#pragma unroll
            for (int c = 1; c < NCH2; ++c) {
                const float4 t = *reinterpret_cast<const float4*>(sl + (long)c * NPX * N2 + kb * 16);
                y.x += t.x; y.y += t.y; y.z += t.z; y.w += t.w;
            }
            const float4 b = *reinterpret_cast<const float4*>(bp + kb * 16 + 4 * q);
            const float yv[4] = {y.x + b.x, y.y + b.y, y.z + b.z, y.w + b.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int ht = 0; ht < 2; ++ht)
                    hacc[ht] = __builtin_amdgcn_mfma_f32_16x16x4f32(hf[((kb * 4 + j) * 2 + ht) * 64 + lane], yv[j], hacc[ht], 0, 0, 0);
        }
        const float* hb = hf + (N2 / 4) * 2 * 64;
#pragma unroll
        for (int ht = 0; ht < 2; ++ht)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ch = ht * 16 + 4 * q + i;
                if (ch < a.headn) a.out[(((long)n * a.headn + ch) * TH + py) * TW + px] = hacc[ht][i] + hb[ch];
            }
        return;
    }
    const int nch = a.phase == 0 ? NCH1 : NCH2;
    const int n = blockIdx.x / nch, c = blockIdx.x - n * nch;
    for (int i = threadIdx.x; i < 4 * EPL; i += 320) reinterpret_cast<float4*>(E)[i] = make_float4(0.f, 0.f, 0.f, 0.f);   // the 2-pixel ring: padding
    __syncthreads();
    // ---- this workgroup's 16 input channels of the frame -> E (one record per lane: pixel p, channels 16c + 4q .. +3) ----
    float4 v;
    if (a.phase == 0) {
        v = *reinterpret_cast<const float4*>(a.in + ((long)n * NPX + p) * C1 + 16 * c + 4 * q);
    } else {
        const float* sl = a.slab1 + (long)n * NCH1 * NPX * N1 + (long)p * N1 + 16 * c + 4 * q;
        v = *reinterpret_cast<const float4*>(sl);
#pragma unroll
        for (int s = 1; s < NCH1; ++s) {
            const float4 t = *reinterpret_cast<const float4*>(sl + (long)s * NPX * N1);
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        const float4 b = *reinterpret_cast<const float4*>(a.wp1 + NCH1 * CH + 16 * c + 4 * q);   // conv5_4's bias (no ReLU: yolo_fastest.py:131)
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    }
    *reinterpret_cast<float4*>(E + (q * EPL + rp) * 4) = v;
    const float* wc = (a.phase == 0 ? a.wp1 : a.wp2) + c * CH;
    float wf[4][NT];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wf[j][nt] = wc[416 + (j * NT + nt) * 64 + lane];
    const float4 bd = *reinterpret_cast<const float4*>(wc + 400 + 4 * q);
    __syncthreads();
    const float4* e = reinterpret_cast<const float4*>(E) + q * EPL + rp;
    float d[4] = {bd.x, bd.y, bd.z, bd.w};
#pragma unroll
    for (int t = 0; t < 25; ++t) {
        const float4 x = e[(t / 5 - 2) * RW + (t % 5 - 2)];
        const float4 w = *reinterpret_cast<const float4*>(wc + t * 16 + 4 * q);
        d[0] = fmaf(x.x, w.x, d[0]); d[1] = fmaf(x.y, w.y, d[1]); d[2] = fmaf(x.z, w.z, d[2]); d[3] = fmaf(x.w, w.w, d[3]);
    }
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float dj = fmaxf(d[j], 0.f);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j][nt], dj, acc[nt], 0, 0, 0);
    }
    float* so = (a.phase == 0 ? a.slab1 + ((long)n * NCH1 + c) * NPX * N1 : a.slab2 + ((long)n * NCH2 + c) * NPX * N2) + (long)p * 128 + 4 * q;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) *reinterpret_cast<float4*>(so + nt * 16) = make_float4(acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]);
}

enum { MDW2_ESPLIT_MAX_FRAMES = 9 };
size_t mdw2_esplit_scratch_floats() { return (size_t)MDW2_ESPLIT_MAX_FRAMES * (6 + 8) * 80 * 128; }
bool mdw2_esplit_ok(int H, int W, int headn, int Nf, int dtype, const float* scratch)
{
    static const bool off = getenv("YF_MRES_SMALL_OFF") != nullptr || getenv("YF_MDW2_ESPLIT_OFF") != nullptr;
    return !off && scratch && dtype == DT_F32 && H == 8 && W == 10 && headn >= 1 && headn <= 32 && Nf <= MDW2_ESPLIT_MAX_FRAMES;
}

int launch_mdw2(const float* in, const float* wp1, const float* wp2, float* out, int H, int W, int headn, int Nf, hipStream_t s, int dtype, float* scratch)
{
    if (mdw2_esplit_ok(H, W, headn, Nf, dtype, scratch)) {
        Mdw2EsplitArgs e{in, wp1, wp2, out, scratch, scratch + (size_t)Nf * 6 * 80 * 128, headn, 0};
        for (int ph = 0; ph < 3; ++ph) {
            e.phase = ph;
            hipLaunchKernelGGL(mdw2_esplit_kernel, dim3((unsigned)(Nf * (ph == 0 ? 6 : ph == 1 ? 8 : 5))), dim3(ph == 2 ? 64 : 320), 0, s, e);
        }
        return 0;
    }
    const Mdw2Args a{in, wp1, wp2, out, H, W, headn};
    return dtype == DT_F16 ? launch_mdw2_t<96, 128, 128, 8, 10, 5, half_t>(a, Nf, s)
         : dtype == DT_F16X3 ? launch_mdw2_t<96, 128, 128, 8, 10, 5, x3_t>(a, Nf, s) : launch_mdw2_t<96, 128, 128, 8, 10, 5, float>(a, Nf, s);
}

template <int C, int N, int HM, int TH, int TW, int NWAVE, typename T>
static int launch_mdw_t(MdwArgs a, int Nf, hipStream_t s)
{
    a.tiles_y = (a.H + TH - 1) / TH;
    a.tiles_x = (a.W + TW - 1) / TW;
    constexpr size_t lds = ((size_t)16 * mdw_epl(TH, TW, NWAVE) + mdw_stream_floats(C, N, HM == 1 ? 32 : 0, wmode_of<T>())) * sizeof(float);
    static_assert(lds <= 160 * 1024, "LDS");
    static bool attr_done[YF_MAX_DEVICES] = {};
    const int dev = current_device();
    if (dev < 0) return -2;
    if (lds > 64 * 1024 && !attr_done[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(mdw_kernel<C, N, HM, TH, TW, NWAVE, T>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return -2;
        attr_done[dev] = true;
    }
    hipLaunchKernelGGL((mdw_kernel<C, N, HM, TH, TW, NWAVE, T>), dim3((unsigned)(Nf * a.tiles_y * a.tiles_x)), dim3(NWAVE * 64),
                       lds, s, a);
    return 0;
}

#ifndef YF_MDW_LARGE_NW
#define YF_MDW_LARGE_NW 10
#endif
//      (c, n, head mode, TH, TW, waves)      head mode: 0 none, 1 = up to 32 head channels (the shipped 24), 2 = any width
#define YF_MDW_SHAPES(MD)                                            \
    MD(96, 128, 0, 8, 10, 5)    /* conv5_3 -> conv5_4            @ H/32 */ \
    MD(128, 128, 1, 8, 10, 5)   /* conv5_5 -> conv5_6 -> head_5  @ H/32 */ \
    MD(128, 128, 2, 8, 10, 5)                                              \
    MD(96, 96, 0, 16, 20, YF_MDW_LARGE_NW)   /* conv4_1_2 -> conv4_1_3        @ H/16 */ \
    MD(96, 96, 1, 16, 20, YF_MDW_LARGE_NW)   /* conv4_1_4 -> conv4_1_5 -> head_4      */ \
    MD(96, 96, 2, 16, 20, YF_MDW_LARGE_NW)

// Small batches: the stride-16 launches own a whole 16x20 frame per workgroup -- at batch 1 ONE CU works for ~30 us.
// When those tiles would leave more than half of the CUs idle, the same kernel runs on 8x10 tiles with 5 waves (the stride-32 shapes'
// geometry): four workgroups per 16x20 frame.  A pixel's arithmetic does not depend on the tile it is computed in: same bits.
#define YF_MDW_SMALL_SHAPES(MD) \
    MD(96, 96, 0, 8, 10, 5)     \
    MD(96, 96, 1, 8, 10, 5)     \
    MD(96, 96, 2, 8, 10, 5)

static int mdw_head_mode(int headn) { return headn <= 0 ? 0 : headn <= 32 ? 1 : 2; }

int launch_mdw(int c, int n, int headn, const MdwArgs& a0, int Nf, hipStream_t s, int dtype)
{
    MdwArgs a = a0;
    a.headn = headn;
    const int hm = mdw_head_mode(headn);
#define MD(cc, nn, hh, th, tw, nw)                                                               \
    if (c == cc && n == nn && hm == hh)                                                           \
        return dtype == DT_F16 ? launch_mdw_t<cc, nn, hh, th, tw, nw, half_t>(a, Nf, s)           \
             : dtype == DT_F16X3 ? launch_mdw_t<cc, nn, hh, th, tw, nw, x3_t>(a, Nf, s) : launch_mdw_t<cc, nn, hh, th, tw, nw, float>(a, Nf, s);
    {
        const int n_cu = device_cu_count(current_device());
        const long big_tiles = (long)Nf * ((a.H + 15) / 16) * ((a.W + 19) / 20);
        // (at full batches the 8x10 tiles lose: 640x512 batch 128, fp16 storage 46 -> 63 us and 44 -> 71 us, fp32 63 -> 73 and 68 -> 95)
        if (a.H > 8 && n_cu > 0 && 2 * big_tiles <= n_cu) { YF_MDW_SMALL_SHAPES(MD) }
    }
    YF_MDW_SHAPES(MD)
#undef MD
    return -1;
}

bool mdw_has_kernel(int c, int n, int headn)
{
    const int hm = mdw_head_mode(headn);
#define MD(cc, nn, hh, th, tw, nw) \
    if (c == cc && n == nn && hm == hh) return true;
    YF_MDW_SHAPES(MD)
#undef MD
    return false;
}

size_t mdw_packed_floats(int c, int n, int headn, int wmode) { return (size_t)mdw_stream_floats(c, n, headn, wmode); }

// Weight stream: NCH chunks of [wd 25x16 | bd 16 | W frags], then b_pw[n], then (head) frags, b_head[32].
//   fp32: W frags 4 x NT x 64 floats (one per k-step), head frags (n/4) x 2 x 64;  h16: f16x4 per lane: NT x 128 / (n/16) x 2 x 128 floats
void mdw_pack_weights(const float* wd /*[25][c]*/, const float* bd, const float* w /*[c][n]*/, const float* b, const float* hw /*[n][headn]*/,
                      const float* hb, int c, int n, int headn, float* out, int wmode)
{
    const bool h16 = wmode != WM_F32, x3 = wmode == WM_F16X3;
    const int NT = n / 16, NCH = c / 16, CH = mdw_chunk_floats(n, wmode);
    for (int ch = 0; ch < NCH; ++ch) {
        float* o = out + (size_t)ch * CH;
        for (int t = 0; t < 25; ++t)
            for (int k = 0; k < 16; ++k) o[t * 16 + k] = wd[(size_t)t * c + ch * 16 + k];
        for (int k = 0; k < 16; ++k) o[400 + k] = bd[ch * 16 + k];
        auto w_at = [&](int j, int nt, int lane) { return w[(size_t)(ch * 16 + 4 * (lane >> 4) + j) * n + nt * 16 + (lane & 15)]; };
        if (h16) {
            uint16_t* o16 = reinterpret_cast<uint16_t*>(o + 416);
            for (int nt = 0; nt < NT; ++nt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 4; ++j) {
                        o16[(nt * 64 + lane) * 4 + j] = f32_to_f16_bits(w_at(j, nt, lane));
                        if (x3) o16[((NT + nt) * 64 + lane) * 4 + j] = f16_lo_bits(w_at(j, nt, lane));
                    }
        } else {
            for (int j = 0; j < 4; ++j)
                for (int nt = 0; nt < NT; ++nt)
                    for (int lane = 0; lane < 64; ++lane) o[416 + (j * NT + nt) * 64 + lane] = w_at(j, nt, lane);
        }
    }
    float* o = out + (size_t)NCH * CH;
    for (int i = 0; i < n; ++i) o[i] = b[i];
    o += n;
    if (headn) {
        const int nthp = mdw_head_tiles(headn);   // column tiles, padded to pairs (2 for every head of up to 32 channels)
        auto h_at = [&](int s, int nt, int lane) -> float {
            const int k = (s / 4) * 16 + 4 * (lane >> 4) + (s % 4), col = nt * 16 + (lane & 15);
            return col < headn ? hw[(size_t)k * headn + col] : 0.f;
        };
        if (h16) {
            uint16_t* o16 = reinterpret_cast<uint16_t*>(o);
            for (int kb = 0; kb < n / 16; ++kb)
                for (int nt = 0; nt < nthp; ++nt)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 4; ++j) {
                            o16[((kb * nthp + nt) * 64 + lane) * 4 + j] = f32_to_f16_bits(h_at(kb * 4 + j, nt, lane));
                            if (x3) o16[(((n / 16 + kb) * nthp + nt) * 64 + lane) * 4 + j] = f16_lo_bits(h_at(kb * 4 + j, nt, lane));
                        }
            o += (x3 ? 2 : 1) * (n / 16) * nthp * 128;
        } else {
            for (int s = 0; s < n / 4; ++s)
                for (int nt = 0; nt < nthp; ++nt)
                    for (int lane = 0; lane < 64; ++lane) o[(s * nthp + nt) * 64 + lane] = h_at(s, nt, lane);
            o += (n / 4) * nthp * 64;
        }
        for (int i = 0; i < nthp * 16; ++i) o[i] = i < headn ? hb[i] : 0.f;
    }
}

}  // namespace yf
