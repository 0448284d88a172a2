// yf_mres_kernels.hip -- BasicResBlock on the matrix cores, fully on chip.
//
//   x -> pw-expand(+ReLU) -> dw3x3(+ReLU) -> pw-project (+ x)       src/model_training/model/yolo_fastest.py:52-66
//
// One workgroup = one TH x TW tile of one frame (whole frame at strides 16/32).  Both pointwise convs are
// v_mfma_f32_16x16x4_f32 GEMMs (exact fp32; v_mfma_f32_16x16x16_f16 with fp16 storage) with the WEIGHTS as the MFMA's A operand
// and the activations as B, so a lane's four result registers are four consecutive channels of ONE pixel: 16-byte LDS records
// and 16-byte HBM stores, and the depthwise result feeds the projection without leaving registers.  Only the depthwise conv
// runs on the VALU (packed fp32 FMAs over channel pairs).  Nothing wide touches HBM:
//
//   HBM --16-B loads, all issued before the first LDS write--> LDS X[region px][CIN]   (halo'd input tile, also the residual)
//   X --activation fragments (kept in VGPRs for the whole kernel)--> MFMA expand --lane: 4 consecutive channels of one region
//   pixel--> bias, ReLU + zero outside the image (one v_med3_f32) --one ds_write_b128--> LDS E[4-ch group][region px][4 ch]
//   E --9 ds_read_b128 per pixel, issued together--> depthwise taps (v_pk_fma_f32) + ReLU --is directly the B fragment of-->
//   MFMA project (accumulators live in VGPRs across all chunks) --> + bias + residual(X) --> HBM (or back into X: chains)
//
// Fragment conventions (16x16x4 f32): lane l = (r = l & 15, q = l >> 4).
//   A: row r, k = q (per k-step);  B: k = q, col r;  C/D: col r, rows 4q + reg.
//   expand : A rows = the chunk's 16 channels (weights), B cols = region pixels (linear index rp = ry*RW + rx);
//   project: A rows = output channels (weights), k = chunk channel 4q + j for k-step j, B cols = output pixels (op = oy*TW + ox).
// The k permutations are folded into the host-packed weight fragments (mres_pack_weights).
// Variants: stride 2 (the un-named bottleneck triples; the conv4_2 triple also WRITES its expanded tensor and applies conv5_1's
// ReLU), producer/consumer waves (mres_pc_kernel, strides 16/32), and chains of blocks in one launch where the tile is the frame.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "yf_kernels.h"

#ifndef YF_MRES_DBG
#define YF_MRES_DBG 0   // tools/kbench.hip only: 1 = one dw window address, 2 / 4 = one k-step in expansion / projection, 8 = one dw tap
#endif

namespace yf {

// Diagnostic build only (-DYF_MRES_STAMP, tools/kbench.hip mresp): per-phase shader-clock sums of the first ([0..7]) and the last
// ([8..15]) wave of every workgroup.
#ifdef YF_MRES_STAMP
__device__ unsigned long long yf_mres_dbg[16];
__device__ __forceinline__ unsigned long long mres_clock()
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define MRES_STAMP_DECL unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t_ = mres_clock();
#define MRES_STAMP(i) { unsigned long long n_ = mres_clock(); st_[i] += n_ - st_t_; st_t_ = n_; }
#define MRES_STAMP_FLUSH(NW) if ((threadIdx.x & 63) == 0 && (wave == 0 || wave == NW - 1)) { for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&yf_mres_dbg[(wave ? 8 : 0) + i_], st_[i_]); }
#else
#define MRES_STAMP_DECL
#define MRES_STAMP(i)
#define MRES_STAMP_FLUSH(NW)
#endif


typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifndef YF_RES5_CPP
#define YF_RES5_CPP 1   // chunks per barrier phase of the stride-32 chain (A/B: 2)
#endif
#ifndef YF_MRES_WAVES_ATTR
#define YF_MRES_WAVES_ATTR   // A/B builds: e.g. -DYF_MRES_WAVES_ATTR='__attribute__((amdgpu_waves_per_eu(6,6)))'
#endif
#ifndef YF_MRES_BATCH9_MIN
#define YF_MRES_BATCH9_MIN 32   // blocks with more expanded channels than this issue the nine depthwise window reads of a pixel together
#endif
#ifndef YF_MRES_FRAME
#define YF_MRES_FRAME 1   // frame-sized tiles of the producer/consumer kernel expand the interior pixels only (0: round 2's behaviour, for A/B builds)
#endif
#ifndef YF_MRES_Q4
#define YF_MRES_Q4 0   // 1: fp32 blocks with 8 output channels project on v_mfma_f32_4x4x1_16B_f32 (no half-empty 16-wide N tile).  Built and measured in
                       // round 5, OFF: half the matrix-pipe cycles of the projection, but every lane group keeps its own partial sums
                       // (8 accumulator registers per tile instead of 4): 120 -> 136 VGPRs in the 8/48/8 blocks (one resident workgroup fewer) and
                       // 78 -> 100 in the stride-2 8/32/8 triple (two instead of three): res3_1 20.3 -> 26.9 us, res3_2 19.4 -> 26.0, conv2_2 triple
                       // 30.1 -> 37.5 (tools/ops_abn.sh, three interleaved rounds; same heads to rounding, parity tests green)
#endif
#ifndef YF_MRES_PK
#define YF_MRES_PK 1   // depthwise taps as v_pk_fma_f32 (two channels per instruction; same fused multiply-add per element)
#endif
// ReLU of a value that is not NaN: one v_max_i32 (fmaxf() on an FMA / MFMA result costs a canonicalising v_add first)
__device__ __forceinline__ float relu_bits(float x) { return __int_as_float(max(__float_as_int(x), 0)); }

// Pixels per 4-channel plane of E.  A ds_read_b128 is serviced in lane groups that MIX two of the kernel's channel quads
// ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS): with the plane pitch a multiple of 16 records the two quads' pixels
// (r in {0-3, 12-15} of quad q, r in {4-11} of quad q + 1) land on complementary banks.  (The old pitch, == 1 mod 8, was made
// for 4-byte E writes; the 16-byte record writes of today are conflict-free at any pitch.)
#ifndef YF_MRES_EPL_PAD
#define YF_MRES_EPL_PAD 0
#endif
// Stride-2 blocks read every second pixel: quad q + 1 then wants the ODD records (pitch == 1 mod 16).
__host__ __device__ constexpr int mres_epl(int mtr, int s = 1) { return mtr * 16 + (s == 2 ? 1 : 0) + YF_MRES_EPL_PAD; }
__host__ __device__ constexpr int mres_ksteps(int K) { return (K / 16) * 4 + ((K % 16) ? 2 : 0); }
// wmode WM_F16X3: the hi fragments of a matrix are followed by its lo fragments (same layout)
__host__ __device__ constexpr int mres_chunk_floats(int cin, int cout, int wmode = WM_F32)
{
    return wmode != WM_F32 ? (wmode == WM_F16X3 ? 2 : 1) * ((mres_ksteps(cin) + 3) / 4) * 128 + 16 + 9 * 16 + 16 +
                                 (wmode == WM_F16X3 ? 2 : 1) * ((cout + 15) / 16) * 128
                           : mres_ksteps(cin) * 64 + 16 + 9 * 16 + 16 + ((YF_MRES_Q4 && cout == 8) ? 2 * 64 * 4 : 4 * ((cout + 15) / 16) * 64);
}

// Stage the block's weight stream and the halo'd input tile (zeros outside the image / beyond the region) in LDS.  ALL global loads
// are issued before the first LDS write: as rolled loops with the load under a bounds branch, every iteration exposed its own
// L2/HBM round trip -- six to seven serial trips were 40-50 % of a workgroup's life (tools/kbench.hip mresp, -DYF_MRES_STAMP).
template <int CIN, int S, int RW, int NRP, int MTR, int XP, int WFLOATS, int NTHR, typename T>
__device__ __forceinline__ void mres_stage(const MresArgs& a, int n, int oy0, int ox0, float* X, float* WL)
{
    constexpr int C4 = CIN / 4, NX = MTR * 16 * C4, NXI = (NX + NTHR - 1) / NTHR, NW4 = WFLOATS / 4, NWI = (NW4 + NTHR - 1) / NTHR;
    const T* __restrict__ src = reinterpret_cast<const T*>(a.in) + (long)n * a.H * a.W * CIN;
    float4 wv[NWI], xv[NXI];
#pragma unroll
    for (int i = 0; i < NWI; ++i) {
        const int idx = threadIdx.x + i * NTHR;
        wv[i] = (YF_MRES_DBG & 32) ? make_float4(0.01f, 0.02f, 0.03f, 0.04f) : *reinterpret_cast<const float4*>(a.wp + 4 * (idx < NW4 ? idx : NW4 - 1));
    }
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
        const int idx = threadIdx.x + i * NTHR;
        const int rp = idx / C4, c4 = idx - rp * C4;
        const int ry = rp / RW, rx = rp - ry * RW;
        const int iy = oy0 * S - 1 + ry, ix = ox0 * S - 1 + rx;
        const bool ok = rp < NRP && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        xv[i] = (YF_MRES_DBG & 16) ? make_float4(0.1f, 0.2f, 0.3f, 0.4f) : ld4<T>(src + (ok ? (iy * a.W + ix) * CIN + c4 * 4 : 0));
        if (!ok) xv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < NWI; ++i) {
        const int idx = threadIdx.x + i * NTHR;
        if (idx < NW4) *reinterpret_cast<float4*>(&WL[4 * idx]) = wv[i];
    }
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
        const int idx = threadIdx.x + i * NTHR;
        const int rp = idx / C4, c4 = idx - rp * C4;
        if (idx < NX) *reinterpret_cast<float4*>(&X[rp * XP + c4 * 4]) = xv[i];
    }
}

// The one block whose EXPANDED tensor is needed later (conv4_2 feeds the large head's concat, yolo_fastest.py:209): the kernel also
// writes it (MresArgs::out_exp).
__host__ __device__ constexpr bool mres_writes_expansion(int cin, int cexp, int cout, int s) { return cin == 24 && cexp == 136 && cout == 48 && s == 2; }
// ... and the one whose projection is a conv_norm_relu (conv5_1, yolo_fastest.py:124); every other block projects linearly (:62, :86-118)
__host__ __device__ constexpr bool mres_relu_out(int cin, int cexp, int cout, int s) { return mres_writes_expansion(cin, cexp, cout, s); }

// Output pixel (oy, ox) of row r of depthwise / projection M-tile mo.  Row-major 16-pixel runs cross a tile row in most M-tiles and
// the step of RW - TW = 2 records at the crossing puts two lane pairs of a ds_read_b128 lane group on the same banks.  Where the
// E row pitch is 6 (mod 16) records (RW = 22: the 16x20 tiles) an M-tile of 2 columns x 8 rows has all 16 lanes on different
// record slots: rows advance by 6 (mod 16), each takes two consecutive slots.
#ifndef YF_MRES_TILE2X8
#define YF_MRES_TILE2X8 1
#endif
template <int TH, int TW, int RW, int S>
__device__ __forceinline__ void mres_out_px(int mo, int r, int& oy, int& ox)
{
    if constexpr (YF_MRES_TILE2X8 && S == 1 && RW % 16 == 6 && TW % 2 == 0 && TH % 8 == 0) {
        const int rb = mo / (TW / 2), cp = mo - rb * (TW / 2);
        oy = rb * 8 + (r >> 1);
        ox = cp * 2 + (r & 1);
    } else if constexpr (YF_MRES_TILE2X8 && S == 1 && RW % 16 == 12 && TW == 10 && TH == 8) {
        // row pitch 12 (mod 16) records (the 8x10 tiles of stride 32): rows advance by -4, so 4-column x 4-row M-tiles are
        // conflict-free; the last two columns go as one 2 x 8 M-tile (two of its lanes share a slot)
        if (mo < 4) { oy = (mo >> 1) * 4 + (r >> 2); ox = (mo & 1) * 4 + (r & 3); }
        else { oy = r >> 1; ox = 8 + (r & 1); }
    } else {
        const int op = mo * 16 + r;
        oy = op / TW;
        ox = op - oy * TW;
    }
}

// S = 2: the stride-2 triples (pw-expand -> dw3x3 stride 2 -> pw-project, no residual): a.H / a.W are the INPUT dims, the tile
// is TH x TW OUTPUT pixels and the region (TH - 1) S + 3 rows.
template <int CIN, int CEXP, int COUT, bool RES, int S, int TH, int TW, int NWAVE, typename T>
__global__ void __launch_bounds__(NWAVE * 64) YF_MRES_WAVES_ATTR mres_kernel(MresArgs a)
{
    constexpr int RH = (TH - 1) * S + 3, RW = (TW - 1) * S + 3, NRP = RH * RW;
    constexpr int MTR = (NRP + 15) / 16, MTO = (TH * TW) / 16;
    constexpr int MTRW = (MTR + NWAVE - 1) / NWAVE, MTOW = (MTO + NWAVE - 1) / NWAVE;
    constexpr bool EVEN_R = MTR % NWAVE == 0, EVEN_O = MTO % NWAVE == 0;  // every wave owns the same number of tiles: no branches
    constexpr bool WEXP = mres_writes_expansion(CIN, CEXP, COUT, S);       // conv4_2 + conv4_3 + conv5_1: conv4_2 is a skip tensor
    constexpr int XP = CIN + 4;                           // X row pitch: conflict-free b128/b64 fragment reads
    constexpr int EPL = mres_epl(MTR, S);   // pixels per 4-channel plane
    constexpr bool X3 = is_x3<T>::value;  // fp32 storage, split-operand fp16 MFMAs (3 per k-group): yf_kernels.h DT_F16X3
    constexpr bool H16 = sizeof(T) == 2 || X3;  // the pointwise GEMMs run on v_mfma_f32_16x16x16_f16 (4 k-steps each)
    constexpr int WM = X3 ? 2 : 1;
    constexpr int KS1 = mres_ksteps(CIN), NB1 = CIN / 16, NT2 = (COUT + 15) / 16, NCH = (CEXP + 15) / 16;
    constexpr int NK1 = (KS1 + 3) / 4;    // f16 MFMAs per expansion tile
    constexpr int OFF_B1 = H16 ? WM * NK1 * 128 : KS1 * 64, OFF_WD = OFF_B1 + 16, OFF_BD = OFF_WD + 144, OFF_W2 = OFF_BD + 16;
    // Q4 (round 5): an 8-channel projection fills half of a 16x16x4 MFMA's N tile.  v_mfma_f32_4x4x1_16B_f32 multiplies, in 16 independent
    // 4x4 blocks, the 4 output channels of lanes 4b .. 4b+3's A registers with the 4 pixels of their B registers at ONE k: with the depthwise
    // result as B (lane (r, q): channel 4q + j of pixel r) block (q, r / 4) accumulates channels 4h .. 4h+3 of ITS four pixels over the k
    // values lane group q holds -- two instructions (h = 0, 1) of 8 cycles per k-step instead of one of 32 -- and the four lane groups'
    // partial sums are added once per tile in the epilogue (k19r_kernel's scheme for conv1_9's channels 16 .. 23).
    constexpr bool Q4 = YF_MRES_Q4 && !H16 && COUT == 8;
    constexpr int CHUNK = OFF_W2 + (H16 ? WM * NT2 * 128 : (Q4 ? 2 * 64 * 4 : 4 * NT2 * 64));
    static_assert((TH * TW) % 16 == 0 && CIN % 8 == 0 && COUT % 4 == 0, "shape");
    static_assert(!RES || (CIN == COUT && S == 1), "residual needs same shape");
    static_assert(CHUNK == mres_chunk_floats(CIN, COUT, wmode_of<T>()), "pack layout");
    static_assert(MTRW <= 32, "in-image mask bits");
    extern __shared__ __attribute__((aligned(16))) float mres_smem[];
    float* X = mres_smem;                 // [MTR*16][XP]
    float* E = mres_smem + MTR * 16 * XP; // [4][EPL][4]
    float* WL = E + 16 * EPL;             // the block's whole weight stream (NCH chunks + b2), staged once
    constexpr int WFLOATS = (NCH * CHUNK + COUT + 3) & ~3;

    const int b = xcd_tile(blockIdx.x, gridDim.x);
    const int tx = b % a.tiles_x, ty = (b / a.tiles_x) % a.tiles_y, n = b / (a.tiles_x * a.tiles_y);
    const int oy0 = ty * TH, ox0 = tx * TW;   // output coordinates
    const int Ho = a.H / S, Wo = a.W / S;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;

    MRES_STAMP_DECL
    mres_stage<CIN, S, RW, NRP, MTR, XP, WFLOATS, NWAVE * 64, T>(a, n, oy0, ox0, X, WL);
    __syncthreads();
    MRES_STAMP(0)   // staging + barrier

    // ---- expansion A fragments (constant over chunks) and the in-image mask of this lane's 4 C rows ----
    float a1[MTRW][KS1];
    unsigned inmask = 0;
    int eoff[MTRW];
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
        {   // this lane's pixel of the expansion result (see the expand step): region pixel mt*16 + r
            const int rp = mt * 16 + r;
            const int ry = rp / RW, rx = rp - ry * RW;
            const int iy = oy0 * S - 1 + ry, ix = ox0 * S - 1 + rx;
            if (mt < MTR && rp < NRP && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) inmask |= 1u << i;
            // expanded-tensor writeback (a.out_exp): each in-image pixel is written by ONE tile -- the first region row / column is
            // the previous tile's last
            eoff[i] = (inmask >> i & 1) && ry > 0 && rx > 0 ? (iy * a.W + ix) * CEXP + 4 * q : -1;  // within frame n
        }
    }
    // split-operand mode: the activation operand of the expansion is split ONCE per block (hi and lo halves, 4 k-steps per f16x4)
    f16x4 a1h[X3 ? MTRW : 1][NK1], a1l[X3 ? MTRW : 1][NK1];
    if constexpr (X3) {
#pragma unroll
        for (int i = 0; i < MTRW; ++i)
#pragma unroll
            for (int s = 0; s < NK1; ++s)
                split_f16x4(a1[i][4 * s], a1[i][4 * s + 1], 4 * s + 2 < KS1 ? a1[i][(4 * s + 2) % KS1] : 0.f,
                            4 * s + 3 < KS1 ? a1[i][(4 * s + 3) % KS1] : 0.f, a1h[i][s], a1l[i][s]);
    }
    // ---- projection accumulators and the E offsets of this lane's output pixel (as A-fragment row r) ----
    f32x4 acc[MTOW][NT2];
    f32x4 accq[Q4 ? MTOW : 1][2];   // Q4: channels 4h .. 4h+3 of this lane's pixel, partial over this lane group's k values
    int rp0[MTOW];
#pragma unroll
    for (int i = 0; i < MTOW; ++i) {
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) acc[i][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (Q4) { accq[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; accq[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        const int mo = wave + i * NWAVE;
        int oy, ox;
        mres_out_px<TH, TW, RW, S>(mo < MTO ? mo : 0, r, oy, ox);
        rp0[i] = (oy * S + 1) * RW + ox * S + 1;
    }

    MRES_STAMP(1)   // fragments, offsets
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
        const float* wc = WL + c * CHUNK;
        float w1f[KS1];
        f16x4 w1h[NK1], w1l[X3 ? NK1 : 1];
        if constexpr (H16) {
#pragma unroll
            for (int s = 0; s < NK1; ++s) w1h[s] = reinterpret_cast<const f16x4*>(wc)[s * 64 + lane];
            if constexpr (X3) {
#pragma unroll
                for (int s = 0; s < NK1; ++s) w1l[s] = reinterpret_cast<const f16x4*>(wc)[(NK1 + s) * 64 + lane];
            }
        } else {
#pragma unroll
            for (int s = 0; s < KS1; ++s) w1f[s] = wc[s * 64 + lane];
        }
        const float4 b1 = *reinterpret_cast<const float4*>(wc + OFF_B1 + 4 * q);
        float4 wd[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) wd[t] = *reinterpret_cast<const float4*>(wc + OFF_WD + t * 16 + 4 * q);
        const float4 bd = *reinterpret_cast<const float4*>(wc + OFF_BD + 4 * q);
        float w2f[4][NT2];
        f16x4 w2h[NT2], w2l[X3 ? NT2 : 1];
        if constexpr (H16) {
#pragma unroll
            for (int nt = 0; nt < NT2; ++nt) w2h[nt] = reinterpret_cast<const f16x4*>(wc + OFF_W2)[nt * 64 + lane];
            if constexpr (X3) {
#pragma unroll
                for (int nt = 0; nt < NT2; ++nt) w2l[nt] = reinterpret_cast<const f16x4*>(wc + OFF_W2)[(NT2 + nt) * 64 + lane];
            }
        } else if constexpr (!Q4) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int nt = 0; nt < NT2; ++nt) w2f[j][nt] = wc[OFF_W2 + (j * NT2 + nt) * 64 + lane];
        }
        float4 wq[2];   // Q4: W2[4h + (r & 3)][chunk channel 4q + j], j = x y z w
        if constexpr (Q4) {
            wq[0] = reinterpret_cast<const float4*>(wc + OFF_W2)[lane];
            wq[1] = reinterpret_cast<const float4*>(wc + OFF_W2)[64 + lane];
        }

        MRES_STAMP(2)   // chunk weights from LDS
        // ---- expand: E[channels 4q .. 4q+3][pixel mt*16 + r] ----
#pragma unroll
        for (int i = 0; i < MTRW; ++i) {
            const int mt = wave + i * NWAVE;
            if (EVEN_R || i < MTRW - 1 || mt < MTR) {   // only a wave's LAST tile can be missing: the others share one basic block
                f32x4 cf = f32x4{b1.x, b1.y, b1.z, b1.w};   // the bias rides in as the MFMA's C operand
                if constexpr (X3) {   // small terms first, then hi * hi
#pragma unroll
                    for (int s = 0; s < NK1; ++s) {
                        cf = __builtin_amdgcn_mfma_f32_16x16x16f16(w1l[s], a1h[i][s], cf, 0, 0, 0);
                        cf = __builtin_amdgcn_mfma_f32_16x16x16f16(w1h[s], a1l[i][s], cf, 0, 0, 0);
                    }
#pragma unroll
                    for (int s = 0; s < NK1; ++s) cf = __builtin_amdgcn_mfma_f32_16x16x16f16(w1h[s], a1h[i][s], cf, 0, 0, 0);
                } else if constexpr (H16) {
#pragma unroll
                    for (int s = 0; s < NK1; ++s) {
                        const f16x4 ah = f16x4{(half_t)a1[i][4 * s], (half_t)a1[i][4 * s + 1], 4 * s + 2 < KS1 ? (half_t)a1[i][(4 * s + 2) % KS1] : (half_t)0.f,
                                               4 * s + 3 < KS1 ? (half_t)a1[i][(4 * s + 3) % KS1] : (half_t)0.f};
                        cf = __builtin_amdgcn_mfma_f32_16x16x16f16(w1h[s], ah, cf, 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int s = 0; s < ((YF_MRES_DBG & 2) ? 1 : KS1); ++s) cf = __builtin_amdgcn_mfma_f32_16x16x4f32(w1f[s], a1[i][s], cf, 0, 0, 0);
                }
                // weights as the A operand: the lane holds channels 4q .. 4q+3 of region pixel mt*16 + r = one E record
                // ReLU and the zero outside the image in ONE instruction: median(x, 0, lim) with lim = +inf inside, 0 outside
                const float lim = (inmask >> i) & 1 ? __builtin_inff() : 0.f;
                const float4 ev = make_float4(__builtin_amdgcn_fmed3f(cf[0], 0.f, lim), __builtin_amdgcn_fmed3f(cf[1], 0.f, lim),
                                              __builtin_amdgcn_fmed3f(cf[2], 0.f, lim), __builtin_amdgcn_fmed3f(cf[3], 0.f, lim));
                *reinterpret_cast<float4*>(E + (q * EPL + mt * 16 + r) * 4) = ev;
                if constexpr (WEXP)
                    if (eoff[i] >= 0 && c * 16 + 4 * q < CEXP)
                        st4<T>(reinterpret_cast<T*>(a.out_exp) + (long)n * a.H * a.W * CEXP + eoff[i] + c * 16, ev);
            }
        }
        MRES_STAMP(3)   // expansion
        __syncthreads();
        MRES_STAMP(4)   // barrier 1
        // ---- depthwise 3x3 of channels 4q..4q+3 at output pixel r  ==  A fragment of the projection ----
#pragma unroll
        for (int i = 0; i < MTOW; ++i) {
            const int mo = wave + i * NWAVE;
            if (EVEN_O || i < MTOW - 1 || mo < MTO) {
                const float4* e = reinterpret_cast<const float4*>(E) + q * EPL + rp0[i];
                float d[4] = {bd.x, bd.y, bd.z, bd.w};
                // All nine window reads go out together (left to the scheduler they were split 3 + 1 + 5 with a wait each) ... except in
                // the smallest block (8/32 stride-2 triple), where three workgroups fit a CU's LDS and the +24 VGPRs of the batch take it
                // from 74 to 98 registers = from three resident workgroups to two (measured 35 -> 42 us).
                constexpr bool BATCH9 = CEXP > YF_MRES_BATCH9_MIN;
                float4 v9[9];
                if constexpr (BATCH9) {
#pragma unroll
                    for (int t = 0; t < 9; ++t) v9[t] = e[((YF_MRES_DBG & 1) ? 0 : (t / 3 - 1) * RW + (t % 3 - 1))];
                    __builtin_amdgcn_sched_barrier(0);
                }
#if YF_MRES_PK
                f32x2 dl = {d[0], d[1]}, dh2 = {d[2], d[3]};
#pragma unroll
                for (int t = 0; t < ((YF_MRES_DBG & 8) ? 1 : 9); ++t) {
                    float4 v;
                    if constexpr (BATCH9) v = v9[t]; else v = e[(t / 3 - 1) * RW + (t % 3 - 1)];
                    const float4 w = wd[t];
                    dl = __builtin_elementwise_fma(f32x2{v.x, v.y}, f32x2{w.x, w.y}, dl);
                    dh2 = __builtin_elementwise_fma(f32x2{v.z, v.w}, f32x2{w.z, w.w}, dh2);
                }
                d[0] = dl[0]; d[1] = dl[1]; d[2] = dh2[0]; d[3] = dh2[1];
#else
#pragma unroll
                for (int t = 0; t < ((YF_MRES_DBG & 8) ? 1 : 9); ++t) {
                    float4 v;
                    if constexpr (BATCH9) v = v9[t]; else v = e[(t / 3 - 1) * RW + (t % 3 - 1)];
                    const float4 w = wd[t];
                    d[0] = fmaf(v.x, w.x, d[0]);
                    d[1] = fmaf(v.y, w.y, d[1]);
                    d[2] = fmaf(v.z, w.z, d[2]);
                    d[3] = fmaf(v.w, w.w, d[3]);
                }
#endif
                if constexpr (X3) {
                    f16x4 dh, dl;
                    split_f16x4(relu_bits(d[0]), relu_bits(d[1]), relu_bits(d[2]), relu_bits(d[3]), dh, dl);
#pragma unroll
                    for (int nt = 0; nt < NT2; ++nt) acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x16f16(w2l[nt], dh, acc[i][nt], 0, 0, 0);
#pragma unroll
                    for (int nt = 0; nt < NT2; ++nt) acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x16f16(w2h[nt], dl, acc[i][nt], 0, 0, 0);
#pragma unroll
                    for (int nt = 0; nt < NT2; ++nt) acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x16f16(w2h[nt], dh, acc[i][nt], 0, 0, 0);
                } else if constexpr (H16) {
                    const f16x4 dh = f16x4{(half_t)fmaxf(d[0], 0.f), (half_t)fmaxf(d[1], 0.f), (half_t)fmaxf(d[2], 0.f), (half_t)fmaxf(d[3], 0.f)};
#pragma unroll
                    for (int nt = 0; nt < NT2; ++nt) acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x16f16(w2h[nt], dh, acc[i][nt], 0, 0, 0);
                } else if constexpr (Q4) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float dj = relu_bits(d[j]);
                        accq[i][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(((const float*)&wq[0])[j], dj, accq[i][0], 0, 0, 0);
                        accq[i][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(((const float*)&wq[1])[j], dj, accq[i][1], 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < ((YF_MRES_DBG & 4) ? 1 : 4); ++j) {
                        const float dj = relu_bits(d[j] + ((YF_MRES_DBG & 4) ? d[1] + d[2] + d[3] : 0.f));
#pragma unroll
                        for (int nt = 0; nt < NT2; ++nt)
                            acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2f[j][nt], dj, acc[i][nt], 0, 0, 0);
                    }
                }
            }
        }
        MRES_STAMP(5)   // depthwise + projection
        __syncthreads();
        MRES_STAMP(6)   // barrier 2
    }

    // ---- epilogue: + bias (+ residual from X), NHWC store.  The projection ran with the weights as the MFMA's A operand and the
    // depthwise result as B (same fragments, swapped): lane (r, q) holds output channels nt*16 + 4q .. +3 of pixel mo*16 + r
    // -- one 16-byte residual read and one 16-byte store per lane and n-tile ----
    const float* b2 = WL + NCH * CHUNK;
#pragma unroll
    for (int i = 0; i < MTOW; ++i) {
        const int mo = wave + i * NWAVE;
        if constexpr (Q4) {   // add the four lane groups' partial sums (every group then holds the totals of its pixel); group q stores channels 4q .. 4q+3
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float v = accq[i][h][k];
                    v += __shfl_xor(v, 16);
                    v += __shfl_xor(v, 32);
                    accq[i][h][k] = v;
                }
            acc[i][0] = q == 0 ? accq[i][0] : accq[i][1];
        }
        if (!EVEN_O && i == MTOW - 1 && mo >= MTO) continue;
        int oy, ox;
        mres_out_px<TH, TW, RW, S>(mo, r, oy, ox);
        const int gy = oy0 + oy, gx = ox0 + ox;
        if (gy >= Ho || gx >= Wo) continue;
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) {
            const int col = nt * 16 + 4 * q;
            if (col >= COUT) continue;
            const float4 bias = *reinterpret_cast<const float4*>(b2 + col);
            float4 v = make_float4(acc[i][nt][0] + bias.x, acc[i][nt][1] + bias.y, acc[i][nt][2] + bias.z, acc[i][nt][3] + bias.w);
            if constexpr (RES) {
                const float4 x = *reinterpret_cast<const float4*>(&X[((oy + 1) * RW + ox + 1) * XP + col]);
                v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
            }
            if constexpr (mres_relu_out(CIN, CEXP, COUT, S)) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
            st4<T>(reinterpret_cast<T*>(a.out) + (((long)n * Ho + gy) * Wo + gx) * COUT + col, v);
        }
    }
    MRES_STAMP(7)   // epilogue
    MRES_STAMP_FLUSH(NWAVE)
}


// ---- POSTN: a 1x1 conv + ReLU on the LAST block's result while it is still in LDS (conv5_2 after res5_5, yolo_fastest.py:197-198).
// The chain's last block leaves its result in X instead of HBM; after one barrier every wave owns one n-tile of the conv and walks
// over M-tiles of the tile's pixels: B operand = the pixel's channels from X (the same 16-byte fragment reads as the expansion),
// A operand = the conv's weight fragments in the pw GEMM's own packing (mfma_pack_weights: [k-step][n-tile][lane], then the bias),
// read straight from L2 (12 coalesced 256-byte loads per wave, requested BEFORE the barrier).  acc = 0, k-steps in order, + bias,
// ReLU: the arithmetic -- and the bits -- of pw_ws_kernel, which this replaces as a launch.  Always exact fp32 MFMAs on the fp32
// tile (also in the fp16 / split-operand engines: 360 MFMAs per frame).
template <int K, int NOUT>
struct MresPostFrag { float w[mres_ksteps(K)]; float4 bias; };

template <int K, int NOUT, int NWAVE>
__device__ __forceinline__ void mres_post_fetch(const float* __restrict__ pw, int wave, int lane, MresPostFrag<K, NOUT>& f)
{
    constexpr int KS = mres_ksteps(K), NT = NOUT / 16;
    static_assert(K % 16 == 0 && NOUT % 16 == 0 && NWAVE >= NT, "post conv shape");
    const int nt = wave % NT;
#pragma unroll
    for (int s = 0; s < KS; ++s) f.w[s] = pw[(s * NT + nt) * 64 + lane];
    f.bias = *reinterpret_cast<const float4*>(pw + KS * NT * 64 + nt * 16 + 4 * (lane >> 4));
}

template <int K, int NOUT, int TH, int TW, int RW, int XP, int NWAVE, typename T>
__device__ __forceinline__ void mres_post_conv(const float* X, const MresPostFrag<K, NOUT>& f, T* __restrict__ out, int wave, int lane,
                                               int n, int oy0, int ox0, int H, int W)
{
    constexpr int NB = K / 16, NT = NOUT / 16, MTO = (TH * TW) / 16, NG = NWAVE / NT;
    const int r = lane & 15, q = lane >> 4;
    const int nt = wave % NT, g = wave / NT;
    if (g >= NG) return;   // wave-uniform: the waves beyond NG full groups have no n-tile
#pragma unroll 1
    for (int mt = g; mt < MTO; mt += NG) {
        const int p = mt * 16 + r, oy = p / TW, ox = p - oy * TW;
        const float* xr = X + ((oy + 1) * RW + ox + 1) * XP + 4 * q;
        float4 av[NB];
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) av[kb] = *reinterpret_cast<const float4*>(xr + kb * 16);
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < NB; ++kb)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(f.w[kb * 4 + j], ((const float*)&av[kb])[j], acc, 0, 0, 0);
        const int gy = oy0 + oy, gx = ox0 + ox;
        if (gy >= H || gx >= W) continue;
        const float4 v = make_float4(fmaxf(acc[0] + f.bias.x, 0.f), fmaxf(acc[1] + f.bias.y, 0.f), fmaxf(acc[2] + f.bias.z, 0.f),
                                     fmaxf(acc[3] + f.bias.w, 0.f));
        st4<T>(out + (((long)n * H + gy) * W + gx) * NOUT + nt * 16 + 4 * q, v);
    }
}

// ------------------------------------------------------------------------------------------------
// mres_pc_kernel: the same block with the workgroup's waves split into NWP *producers* (expansion MFMAs of chunk c+1
// into one of two E buffers) and NWC *consumers* (depthwise + projection of chunk c from the other buffer): the matrix
// pipe and the LDS/VALU pipes work on different chunks at the same time and there is ONE barrier per chunk instead of
// two.  The two roles run in different branches of a wave-uniform condition (disjoint register live ranges); both execute
// exactly NCH + 1 barriers per block.
// CHAIN (a.nblk > 1): when the tile is the whole frame (strides 16 / 32 of the 320x256 net) consecutive residual blocks of the same
// shape need no halo exchange, so one launch runs a.nblk of them: a block's result replaces X in LDS (each lane overwrites exactly
// the residual it read), only the last block stores to HBM, and the next block's weight stream (a.wp + blk * a.wstride) is
// requested into registers before the epilogue and lands in LDS behind two barriers.  Saves, per chained block, a launch's
// fill/drain, the input staging and the output round trip.
// ------------------------------------------------------------------------------------------------
// FRAME: the tile is the whole frame (what the chains need anyway), so every halo pixel lies outside the image and its expansion is zero
// by definition: the producers expand the TH x TW interior pixels only (MTO M-tiles instead of MTR: 5 instead of 8 at stride 32, 20
// instead of 25 at stride 16 -- 37 % / 20 % of the expansion MFMAs were spent on zeros) and the halo ring of both E buffers is zeroed
// once.  The interior pixels' values, and the order they are summed in, do not change.
// CPP chunks per barrier phase (2 x CPP E buffers): the producers expand chunks t CPP .. t CPP + CPP - 1 while the consumers work on the
// previous CPP -- half the barriers per block for CPP = 2, where the LDS has room for the extra buffers (the stride-32 chain).
template <int CIN, int CEXP, int COUT, bool RES, int TH, int TW, int NWP, int NWC, typename T, int POSTN = 0, bool FRAME = false, int CPP = 1>
__global__ void __launch_bounds__((NWP + NWC) * 64) mres_pc_kernel(MresArgs a)
{
    constexpr int NWAVE = NWP + NWC;
    constexpr int RH = TH + 2, RW = TW + 2, NRP = RH * RW;
    constexpr int MTR = (NRP + 15) / 16, MTO = (TH * TW) / 16;
    constexpr int MTP = FRAME ? MTO : MTR;      // M-tiles of the expansion
    constexpr int MTRW = (MTP + NWP - 1) / NWP, MTOW = (MTO + NWC - 1) / NWC;
    constexpr int XP = CIN + 4;
    constexpr int EPL = mres_epl(MTR);
    constexpr bool X3 = is_x3<T>::value;  // fp32 storage, split-operand fp16 MFMAs (see mres_kernel)
    constexpr bool H16 = sizeof(T) == 2 || X3;  // the pointwise GEMMs run on v_mfma_f32_16x16x16_f16 (4 k-steps each)
    constexpr int WM = X3 ? 2 : 1;
    constexpr int KS1 = mres_ksteps(CIN), NB1 = CIN / 16, NT2 = (COUT + 15) / 16, NCH = (CEXP + 15) / 16;
    constexpr int NK1 = (KS1 + 3) / 4;    // f16 MFMAs per expansion tile
    constexpr int OFF_B1 = H16 ? WM * NK1 * 128 : KS1 * 64, OFF_WD = OFF_B1 + 16, OFF_BD = OFF_WD + 144, OFF_W2 = OFF_BD + 16;
    constexpr int CHUNK = OFF_W2 + (H16 ? WM * NT2 * 128 : 4 * NT2 * 64);
    static_assert((TH * TW) % 16 == 0 && CIN % 8 == 0 && COUT % 4 == 0, "shape");
    static_assert(CHUNK == mres_chunk_floats(CIN, COUT, wmode_of<T>()), "pack layout");
    static_assert(!RES || CIN == COUT, "residual needs same shape");
    static_assert(MTRW * 4 <= 64, "in-image mask bits");
    extern __shared__ __attribute__((aligned(16))) float mres_smem[];
    float* X = mres_smem;                  // [MTR*16][XP]
    float* E = mres_smem + MTR * 16 * XP;  // [2 * CPP][4][EPL][4]
    float* WL = E + 2 * CPP * 16 * EPL;    // weight stream
    constexpr int NPH = ((CEXP + 15) / 16) / CPP;   // barrier phases per block
    static_assert(((CEXP + 15) / 16) % CPP == 0, "chunks per phase must divide the chunk count");
    constexpr int WFLOATS = (NCH * CHUNK + COUT + 3) & ~3;
    // PSUM (round 6; fp32 stride-32 blocks): the projection's sum over the expanded channels is formed as (p0 + p1 + ... ) with one partial sum
    // p_c per 16-channel chunk -- each chunk's k-steps start from a zero accumulator and its result is ADDED to the running total -- instead of
    // one accumulator running through all chunks.  That is the association of mres_esplit_kernel (the few-frames form: chunk c on its own
    // workgroup, the partial sums added in chunk order at the launch boundary), so that form and this one give the SAME BITS and a frame's
    // logits do not depend on which of them its batch size selects.  Costs 12 v_add_f32 per chunk and consumer wave.
    constexpr bool PSUM = !H16 && CIN == 48 && CEXP == 224 && COUT == 48;

    const int b = xcd_tile(blockIdx.x, gridDim.x);
    const int tx = b % a.tiles_x, ty = (b / a.tiles_x) % a.tiles_y, n = b / (a.tiles_x * a.tiles_y);
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;

    mres_stage<CIN, 1, RW, NRP, MTR, XP, WFLOATS, NWAVE * 64, T>(a, n, oy0, ox0, X, WL);
    if constexpr (FRAME) {   // the producers never write the halo ring: zero both E buffers once
        for (int i = threadIdx.x; i < 2 * CPP * 16 * EPL / 4; i += NWAVE * 64) reinterpret_cast<float4*>(E)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();

    // next block's weight stream, moved by the PRODUCER waves (idle while the consumers finish the last chunk and the epilogue, and
    // their A fragments are dead by then): global -> registers after their last expansion, registers -> LDS between the two block
    // barriers
    constexpr int NW4 = WFLOATS / 4, NWI = (NW4 + NWP * 64 - 1) / (NWP * 64);
    const int nblk = a.nblk > 1 ? a.nblk : 1;

    if (wave < NWP) {
        // ================= producer: expansion of chunk s into E[s & 1] =================
        unsigned long long inmask = 0;
        int prow[MTRW];   // region pixel (= row of X, record of E) of this lane's pixel of the wave's i-th expansion tile
#pragma unroll
        for (int i = 0; i < MTRW; ++i) {
            const int mt = wave + i * NWP;
            int rp = (mt < MTP ? mt : 0) * 16 + r;
            if constexpr (FRAME) { const int py = rp / TW, px = rp - py * TW; rp = (py + 1) * RW + px + 1; }
            prow[i] = rp;
            const int ry = rp / RW, rx = rp - ry * RW;
            const int iy = oy0 - 1 + ry, ix = ox0 - 1 + rx;
            if (mt < MTP && rp < NRP && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) inmask |= 1ull << i;
        }
#pragma unroll 1
      for (int blk = 0; blk < nblk; ++blk) {
        float a1[MTRW][KS1];
#pragma unroll
        for (int i = 0; i < MTRW; ++i) {
            const int row = prow[i];
#pragma unroll
            for (int kb = 0; kb < NB1; ++kb) {
                const float4 t = *reinterpret_cast<const float4*>(&X[row * XP + kb * 16 + 4 * q]);
                a1[i][kb * 4 + 0] = t.x; a1[i][kb * 4 + 1] = t.y; a1[i][kb * 4 + 2] = t.z; a1[i][kb * 4 + 3] = t.w;
            }
            if constexpr (CIN % 16 != 0) {
                const float2 t = *reinterpret_cast<const float2*>(&X[row * XP + NB1 * 16 + 2 * q]);
                a1[i][NB1 * 4 + 0] = t.x; a1[i][NB1 * 4 + 1] = t.y;
            }
        }
        f16x4 a1h[X3 ? MTRW : 1][NK1], a1l[X3 ? MTRW : 1][NK1];   // split once per block
        if constexpr (X3) {
#pragma unroll
            for (int i = 0; i < MTRW; ++i)
#pragma unroll
                for (int k = 0; k < NK1; ++k)
                    split_f16x4(a1[i][4 * k], a1[i][4 * k + 1], 4 * k + 2 < KS1 ? a1[i][(4 * k + 2) % KS1] : 0.f,
                                4 * k + 3 < KS1 ? a1[i][(4 * k + 3) % KS1] : 0.f, a1h[i][k], a1l[i][k]);
        }
#pragma unroll 1
        for (int ph = 0; ph <= NPH; ++ph) {
#pragma unroll
            for (int u = 0; u < CPP; ++u)
            if (ph < NPH) {
                const int s = ph * CPP + u;
                const float* wc = WL + s * CHUNK;
                float* Eb = E + ((ph & 1) * CPP + u) * 16 * EPL;
                float w1f[KS1];
                f16x4 w1h[NK1], w1l[X3 ? NK1 : 1];
                if constexpr (H16) {
#pragma unroll
                    for (int k = 0; k < NK1; ++k) w1h[k] = reinterpret_cast<const f16x4*>(wc)[k * 64 + lane];
                    if constexpr (X3) {
#pragma unroll
                        for (int k = 0; k < NK1; ++k) w1l[k] = reinterpret_cast<const f16x4*>(wc)[(NK1 + k) * 64 + lane];
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < KS1; ++k) w1f[k] = wc[k * 64 + lane];
                }
                const float4 b1 = *reinterpret_cast<const float4*>(wc + OFF_B1 + 4 * q);
#pragma unroll
                for (int i = 0; i < MTRW; ++i) {
                    const int mt = wave + i * NWP;
                    if (i < MTRW - 1 || mt < MTP) {
                        f32x4 cf = f32x4{b1.x, b1.y, b1.z, b1.w};   // bias as the C operand
                        if constexpr (X3) {
#pragma unroll
                            for (int k = 0; k < NK1; ++k) {
                                cf = __builtin_amdgcn_mfma_f32_16x16x16f16(w1l[k], a1h[i][k], cf, 0, 0, 0);
                                cf = __builtin_amdgcn_mfma_f32_16x16x16f16(w1h[k], a1l[i][k], cf, 0, 0, 0);
                            }
#pragma unroll
                            for (int k = 0; k < NK1; ++k) cf = __builtin_amdgcn_mfma_f32_16x16x16f16(w1h[k], a1h[i][k], cf, 0, 0, 0);
                        } else if constexpr (H16) {
#pragma unroll
                            for (int k = 0; k < NK1; ++k) {
                                const f16x4 ah = f16x4{(half_t)a1[i][4 * k], (half_t)a1[i][4 * k + 1],
                                                       4 * k + 2 < KS1 ? (half_t)a1[i][(4 * k + 2) % KS1] : (half_t)0.f,
                                                       4 * k + 3 < KS1 ? (half_t)a1[i][(4 * k + 3) % KS1] : (half_t)0.f};
                                cf = __builtin_amdgcn_mfma_f32_16x16x16f16(w1h[k], ah, cf, 0, 0, 0);
                            }
                        } else {
#pragma unroll
                            for (int k = 0; k < KS1; ++k) cf = __builtin_amdgcn_mfma_f32_16x16x4f32(w1f[k], a1[i][k], cf, 0, 0, 0);
                        }
                        const float lim = (inmask >> i) & 1 ? __builtin_inff() : 0.f;   // median(x, 0, lim): ReLU + outside-the-image zero
                        *reinterpret_cast<float4*>(Eb + (q * EPL + prow[i]) * 4) =
                            make_float4(__builtin_amdgcn_fmed3f(cf[0], 0.f, lim), __builtin_amdgcn_fmed3f(cf[1], 0.f, lim),
                                        __builtin_amdgcn_fmed3f(cf[2], 0.f, lim), __builtin_amdgcn_fmed3f(cf[3], 0.f, lim));
                    }
                }
            }
            __syncthreads();
        }
        if (blk + 1 < nblk) {   // block boundary (the consumers run the same two barriers)
            f32x4 wv[NWI];   // (a native vector: as HIP's float4 struct the copies become memcpys through a scratch array)
            const float* wsrc = a.wp + (long)(blk + 1) * a.wstride;
#pragma unroll
            for (int i = 0; i < NWI; ++i) {
                const int idx = threadIdx.x + i * NWP * 64;
                wv[i] = *reinterpret_cast<const f32x4*>(wsrc + 4 * (idx < NW4 ? idx : NW4 - 1));
            }
            // epilogue done: X holds the block's result, nobody reads WL any more
            __syncthreads();
#pragma unroll
            for (int i = 0; i < NWI; ++i) {
                const int idx = threadIdx.x + i * NWP * 64;
                if (idx < NW4) *reinterpret_cast<f32x4*>(&WL[4 * idx]) = wv[i];
            }
            __syncthreads();
        }
      }
      if constexpr (POSTN > 0) {   // the last block's result is in X once the consumers pass this barrier
          MresPostFrag<COUT, POSTN> pf;
          mres_post_fetch<COUT, POSTN, NWAVE>(a.post_w, wave, lane, pf);
          __syncthreads();
          mres_post_conv<COUT, POSTN, TH, TW, RW, XP, NWAVE, T>(X, pf, reinterpret_cast<T*>(a.post_out), wave, lane, n, oy0, ox0, a.H, a.W);
      }
    } else {
        // ================= consumer: depthwise + projection of chunk s - 1 from E[(s - 1) & 1] =================
        const int cw = wave - NWP;
        int rp0[MTOW];
#pragma unroll
        for (int i = 0; i < MTOW; ++i) {
            const int mo = cw + i * NWC;
            int oy, ox;
            mres_out_px<TH, TW, RW, 1>(mo < MTO ? mo : 0, r, oy, ox);
            rp0[i] = (oy + 1) * RW + ox + 1;
        }
#pragma unroll 1
      for (int blk = 0; blk < nblk; ++blk) {
        f32x4 acc[MTOW][NT2];
#pragma unroll
        for (int i = 0; i < MTOW; ++i)
#pragma unroll
            for (int nt = 0; nt < NT2; ++nt) acc[i][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int ph = 0; ph <= NPH; ++ph) {
#pragma unroll
            for (int u = 0; u < CPP; ++u)
            if (ph >= 1) {
                const int s1 = (ph - 1) * CPP + u;
                const float* wc = WL + s1 * CHUNK;
                const float* Eb = E + (((ph - 1) & 1) * CPP + u) * 16 * EPL;
                float4 wd[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) wd[t] = *reinterpret_cast<const float4*>(wc + OFF_WD + t * 16 + 4 * q);
                const float4 bd = *reinterpret_cast<const float4*>(wc + OFF_BD + 4 * q);
                float w2f[4][NT2];
                f16x4 w2h[NT2], w2l[X3 ? NT2 : 1];
                if constexpr (H16) {
#pragma unroll
                    for (int nt = 0; nt < NT2; ++nt) w2h[nt] = reinterpret_cast<const f16x4*>(wc + OFF_W2)[nt * 64 + lane];
                    if constexpr (X3) {
#pragma unroll
                        for (int nt = 0; nt < NT2; ++nt) w2l[nt] = reinterpret_cast<const f16x4*>(wc + OFF_W2)[(NT2 + nt) * 64 + lane];
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int nt = 0; nt < NT2; ++nt) w2f[j][nt] = wc[OFF_W2 + (j * NT2 + nt) * 64 + lane];
                }
#pragma unroll
                for (int i = 0; i < MTOW; ++i) {
                    const int mo = cw + i * NWC;
                    if (i < MTOW - 1 || mo < MTO) {
                        const float4* e = reinterpret_cast<const float4*>(Eb) + q * EPL + rp0[i];
                        float d[4] = {bd.x, bd.y, bd.z, bd.w};
                        float4 v9[9];   // all nine window reads go out together
#pragma unroll
                        for (int t = 0; t < 9; ++t) v9[t] = e[(t / 3 - 1) * RW + (t % 3 - 1)];
                        __builtin_amdgcn_sched_barrier(0);
#if YF_MRES_PK
                        f32x2 dl = {d[0], d[1]}, dh2 = {d[2], d[3]};
#pragma unroll
                        for (int t = 0; t < 9; ++t) {
                            const float4 v = v9[t], w = wd[t];
                            dl = __builtin_elementwise_fma(f32x2{v.x, v.y}, f32x2{w.x, w.y}, dl);
                            dh2 = __builtin_elementwise_fma(f32x2{v.z, v.w}, f32x2{w.z, w.w}, dh2);
                        }
                        d[0] = dl[0]; d[1] = dl[1]; d[2] = dh2[0]; d[3] = dh2[1];
#else
#pragma unroll
                        for (int t = 0; t < 9; ++t) {
                            const float4 v = v9[t], w = wd[t];
                            d[0] = fmaf(v.x, w.x, d[0]); d[1] = fmaf(v.y, w.y, d[1]);
                            d[2] = fmaf(v.z, w.z, d[2]); d[3] = fmaf(v.w, w.w, d[3]);
                        }
#endif
                        if constexpr (X3) {
                            f16x4 dh, dl;
                            split_f16x4(relu_bits(d[0]), relu_bits(d[1]), relu_bits(d[2]), relu_bits(d[3]), dh, dl);
#pragma unroll
                            for (int nt = 0; nt < NT2; ++nt) acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x16f16(w2l[nt], dh, acc[i][nt], 0, 0, 0);
#pragma unroll
                            for (int nt = 0; nt < NT2; ++nt) acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x16f16(w2h[nt], dl, acc[i][nt], 0, 0, 0);
#pragma unroll
                            for (int nt = 0; nt < NT2; ++nt) acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x16f16(w2h[nt], dh, acc[i][nt], 0, 0, 0);
                        } else if constexpr (H16) {
                            const f16x4 dh = f16x4{(half_t)fmaxf(d[0], 0.f), (half_t)fmaxf(d[1], 0.f), (half_t)fmaxf(d[2], 0.f),
                                                   (half_t)fmaxf(d[3], 0.f)};
#pragma unroll
                            for (int nt = 0; nt < NT2; ++nt)
                                acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x16f16(w2h[nt], dh, acc[i][nt], 0, 0, 0);
                        } else if constexpr (PSUM) {
                            f32x4 part[NT2];
#pragma unroll
                            for (int nt = 0; nt < NT2; ++nt) part[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const float dj = relu_bits(d[j]);
#pragma unroll
                                for (int nt = 0; nt < NT2; ++nt) part[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2f[j][nt], dj, part[nt], 0, 0, 0);
                            }
#pragma unroll
                            for (int nt = 0; nt < NT2; ++nt) acc[i][nt] += part[nt];
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const float dj = relu_bits(d[j]);
#pragma unroll
                                for (int nt = 0; nt < NT2; ++nt)
                                    acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2f[j][nt], dj, acc[i][nt], 0, 0, 0);
                            }
                        }
                    }
                }
            }
            __syncthreads();
        }
        const bool last = blk + 1 == nblk;
        const float* b2 = WL + NCH * CHUNK;   // lane (r, q): output channels nt*16 + 4q .. +3 of pixel mo*16 + r (see mres_kernel)
#pragma unroll
        for (int i = 0; i < MTOW; ++i) {
            const int mo = cw + i * NWC;
            if (i == MTOW - 1 && mo >= MTO) continue;
            int oy, ox;
            mres_out_px<TH, TW, RW, 1>(mo, r, oy, ox);
            const int gy = oy0 + oy, gx = ox0 + ox;
            if (gy >= a.H || gx >= a.W) continue;
#pragma unroll
            for (int nt = 0; nt < NT2; ++nt) {
                const int col = nt * 16 + 4 * q;
                if (col >= COUT) continue;
                const float4 bias = *reinterpret_cast<const float4*>(b2 + col);
                float4 v = make_float4(acc[i][nt][0] + bias.x, acc[i][nt][1] + bias.y, acc[i][nt][2] + bias.z, acc[i][nt][3] + bias.w);
                float* xr = &X[((oy + 1) * RW + ox + 1) * XP + col];
                if constexpr (RES) {
                    const float4 x = *reinterpret_cast<const float4*>(xr);
                    v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
                }
                if (last && POSTN == 0) st4<T>(reinterpret_cast<T*>(a.out) + (((long)n * a.H + gy) * a.W + gx) * COUT + col, v);
                else *reinterpret_cast<float4*>(xr) = v;   // the next block's (or the post conv's) input (CIN == COUT), in place
            }
        }
        if (!last) {
            __syncthreads();
            __syncthreads();
        }
      }
      if constexpr (POSTN > 0) {
          MresPostFrag<COUT, POSTN> pf;
          mres_post_fetch<COUT, POSTN, NWAVE>(a.post_w, wave, lane, pf);
          __syncthreads();
          mres_post_conv<COUT, POSTN, TH, TW, RW, XP, NWAVE, T>(X, pf, reinterpret_cast<T*>(a.post_out), wave, lane, n, oy0, ox0, a.H, a.W);
      }
    }
}

template <int CIN, int CEXP, int COUT, bool RES, int TH, int TW, int NWP, int NWC, typename T, int POSTN = 0, bool FRAME = false>
static int launch_mres_pc_t(MresArgs a, int N, hipStream_t s)
{
    constexpr int CPP = (CIN == 48 && CEXP == 224) ? YF_RES5_CPP : 1;   // the stride-32 chain has the LDS for 2 x 2 E buffers
    a.tiles_y = (a.H + TH - 1) / TH;
    a.tiles_x = (a.W + TW - 1) / TW;
#if YF_MRES_FRAME
    if constexpr (!FRAME) {   // tile == frame: the interior-only expansion
        if (a.tiles_y == 1 && a.tiles_x == 1) return launch_mres_pc_t<CIN, CEXP, COUT, RES, TH, TW, NWP, NWC, T, POSTN, true>(a, N, s);
    }
#endif
    constexpr int MTR = ((TH + 2) * (TW + 2) + 15) / 16;
    constexpr size_t lds = ((size_t)MTR * 16 * (CIN + 4) + 2 * CPP * 16 * mres_epl(MTR) +
                            ((((CEXP + 15) / 16) * mres_chunk_floats(CIN, COUT, wmode_of<T>()) + COUT + 3) & ~3)) * sizeof(float);
    static_assert(lds <= 160 * 1024, "LDS");
    static bool attr_done[YF_MAX_DEVICES] = {};
    const int dev = current_device();
    if (dev < 0) return -2;
    if (lds > 64 * 1024 && !attr_done[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(mres_pc_kernel<CIN, CEXP, COUT, RES, TH, TW, NWP, NWC, T, POSTN, FRAME, CPP>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return -2;
        attr_done[dev] = true;
    }
    if (a.nblk > 1 && (a.tiles_y != 1 || a.tiles_x != 1 || !RES)) return -4;  // a chain needs tile == frame
    if ((POSTN > 0) != (a.post_w != nullptr) || (POSTN > 0 && !a.post_out)) return -5;
    hipLaunchKernelGGL((mres_pc_kernel<CIN, CEXP, COUT, RES, TH, TW, NWP, NWC, T, POSTN, FRAME, CPP>), dim3((unsigned)(N * a.tiles_y * a.tiles_x)),
                       dim3((NWP + NWC) * 64), lds, s, a);
    return 0;
}

template <int CIN, int CEXP, int COUT, bool RES, int S, int TH, int TW, int NWAVE, typename T>
static int launch_mres_t(MresArgs a, int N, hipStream_t s)
{
    a.tiles_y = (a.H / S + TH - 1) / TH;
    a.tiles_x = (a.W / S + TW - 1) / TW;
    if (mres_writes_expansion(CIN, CEXP, COUT, S) && !a.out_exp) return -3;
    if (a.nblk > 1) return -4;  // chains: producer/consumer kernel only
    constexpr int MTR = (((TH - 1) * S + 3) * ((TW - 1) * S + 3) + 15) / 16;
    constexpr size_t lds = ((size_t)MTR * 16 * (CIN + 4) + 16 * mres_epl(MTR, S) +
                            ((((CEXP + 15) / 16) * mres_chunk_floats(CIN, COUT, wmode_of<T>()) + COUT + 3) & ~3)) * sizeof(float);
    static_assert(lds <= 160 * 1024, "LDS");
    static bool attr_done[YF_MAX_DEVICES] = {};
    const int dev = current_device();
    if (dev < 0) return -2;
    if (lds > 64 * 1024 && !attr_done[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(mres_kernel<CIN, CEXP, COUT, RES, S, TH, TW, NWAVE, T>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return -2;
        attr_done[dev] = true;
    }
    hipLaunchKernelGGL((mres_kernel<CIN, CEXP, COUT, RES, S, TH, TW, NWAVE, T>), dim3((unsigned)(N * a.tiles_y * a.tiles_x)),
                       dim3(NWAVE * 64), lds, s, a);
    return 0;
}

// producer / consumer wave counts of the two chained shapes (A/B builds: -DYF_RES5_NWP=.. etc.).  Measured with the interior-only
// expansion (tools/ops_abn.sh, us per launch at batch 256): res5 chain (8,5) 82.5 -- the three idle producers still sit in every
// barrier --, (5,5) 71.9, (5,7) 71.3, (3,5) 69.3 [8 waves = 2 per SIMD], (5,3) / (5,4) 80, 9 or 13 waves with 3 producers > 100
// (a third wave on one SIMD caps the registers at 168: spills); res4 chain (8,8) 76.7, (4,8) 75.7, (6,8) 79.3, (5,10) 79.7.
#ifndef YF_RES4_NWP
#define YF_RES4_NWP 4
#endif
#ifndef YF_RES4_NWC
#define YF_RES4_NWC 8
#endif
#ifndef YF_RES5_NWP
#define YF_RES5_NWP 3
#endif
#ifndef YF_RES5_NWC
#define YF_RES5_NWC 5
#endif
//      (cin, cexp, cout, residual, stride, TH, TW, producer waves (0 = two-barrier kernel), waves / consumer waves)
// Producer/consumer pays where one workgroup owns the CU anyway (strides 16, 32); at stride 8 its second E buffer
// halves the workgroups per CU and it is slower (tools/kbench.hip mrespc).
// A/B builds (round 6): the stride-8 blocks on the producer/consumer kernel, e.g. -DYF_S8A_NWP=3 -DYF_S8A_NW=5
#ifndef YF_S8A_NWP
#define YF_S8A_NWP 0   // res3_1, res3_2, conv3_2 triple (8 -> 48 -> 8 / 16)
#define YF_S8A_NW 8
#endif
#ifndef YF_S8B_NWP
#define YF_S8B_NWP 0   // res3_3 .. res3_6 (16 -> 96 -> 16)
#define YF_S8B_NW 8
#endif
#define YF_MRES_SHAPES(MR)                                                            \
    MR(8, 48, 8, true, 1, 16, 20, YF_S8A_NWP, YF_S8A_NW)     /* res3_1, res3_2           @ H/8  */         \
    MR(8, 48, 16, false, 1, 16, 20, YF_S8A_NWP, YF_S8A_NW)   /* conv3_2/3_3/3_4          @ H/8  */         \
    MR(16, 96, 16, true, 1, 16, 20, YF_S8B_NWP, YF_S8B_NW)   /* res3_3 .. res3_6         @ H/8  */         \
    MR(16, 96, 24, false, 2, 8, 10, 0, 8)   /* conv3_5/3_6/4_1          H/8 -> H/16 */    \
    MR(8, 32, 8, false, 2, 8, 10, 0, 8)     /* conv2_2/2_3/3_1          H/4 -> H/8  */    \
    MR(8, 32, 8, true, 1, 16, 20, 0, 8)     /* res2_1, res2_2 @ H/4: planned for DT_F16X3 only (fp32: the VALU block is as fast) */ \
    MR(24, 136, 48, false, 2, 8, 10, 0, 8)  /* conv4_2/4_3/5_1 (+ conv4_2 written) H/16 -> H/32 */ \
    MR(24, 136, 24, true, 1, 16, 20, YF_RES4_NWP, YF_RES4_NWC)  /* res4_1 .. res4_4         @ H/16 */         \
    MR(48, 224, 48, true, 1, 8, 10, YF_RES5_NWP, YF_RES5_NWC)   /* res5_1 .. res5_5         @ H/32 */

template <int CIN, int CEXP, int COUT, bool RES, int S, int TH, int TW, int NWP, int NW, typename T>
static int launch_mres_any(const MresArgs& a, int N, hipStream_t s)
{
    if constexpr (NWP == 0) return launch_mres_t<CIN, CEXP, COUT, RES, S, TH, TW, NW, T>(a, N, s);
    else { static_assert(S == 1, "producer/consumer kernel: stride 1 only"); return launch_mres_pc_t<CIN, CEXP, COUT, RES, TH, TW, NWP, NW, T>(a, N, s); }
}

// the one block shape with a fused trailing 1x1 conv: res5_x + conv5_2 (48 -> 96, ReLU)
bool mres_has_post(int cin, int cexp, int cout, int postn) { return cin == 48 && cexp == 224 && cout == 48 && postn == 96; }
size_t mres_post_packed_floats(int cout, int postn) { return (size_t)mres_ksteps(cout) * (postn / 16) * 64 + postn; }

// ------------------------------------------------------------------------------------------------
// mres_esplit_kernel (round 5, batch-1 latency): the stride-32 residual chain res5_1 .. res5_5 (+ conv5_2) when only a few frames are in
// flight.  One workgroup per frame runs a block at ~50 % of ONE CU's fp32 peak (12 us); a block can only get faster on several CUs, and
// every block ends in a sum over ALL expanded channels.  Here the NCH = 14 chunks of 16 expanded channels of a block go to 14 workgroups
// (expansion -> depthwise -> the projection's PARTIAL sum over those 16 channels), the partial sums go to HBM, and the kernel LAUNCH
// BOUNDARY is the exchange: launch k first adds the 14 partial sums of block k - 1, in chunk order, + bias + the block's input (what
// its input X_k is), then computes its own chunk of block k.  nblk + 1 launches of ~5 us instead of one of 64; no grid barrier, nothing
// spins, every launch only reads what earlier launches wrote and writes the other half of two alternating buffers (repeatable).
// The last launch forms the chain's result and applies the trailing 1x1 conv (conv5_2: n-tile g on workgroup g) or stores the result.
// THE BITS of the chained launch since round 6: a block's 224-term sums are associated per chunk here ((c0 + c1 + ... + c13) + bias + x), and
// mres_pc_kernel forms the same per-chunk partial sums and adds them in the same order (PSUM); tests/test_gpu_parity.py holds the two forms bitwise.
// fp32 engines, frames of exactly TH x TW pixels.
// ------------------------------------------------------------------------------------------------
struct EsplitArgs {
    const float* in;      // chain input [N, TH*TW, CIN]
    const float* wp;      // block streams, wstride apart
    long wstride;
    float* out;           // chain result (no trailing conv) or nullptr
    const float* post_w;  // trailing conv: pw fragments + bias (or nullptr)
    float* post_out;      // [N, TH*TW, POSTN]
    float* xbuf;          // [2][N][TH*TW][CIN]
    float* slab;          // [2][N][NCH][TH*TW][CIN]
    int nblk, k, n_frames;
};

template <int CIN, int CEXP, int TH, int TW, int POSTN>
__global__ void __launch_bounds__(((TH * TW) / 16) * 64) mres_esplit_kernel(EsplitArgs a)
{
    constexpr int COUT = CIN, NPX = TH * TW, MT = NPX / 16, NWAVE = MT, RH = TH + 2, RW = TW + 2, XP = CIN + 4;
    constexpr int KS1 = mres_ksteps(CIN), NB1 = CIN / 16, NT2 = COUT / 16, NCH = (CEXP + 15) / 16;
    constexpr int OFF_B1 = KS1 * 64, OFF_WD = OFF_B1 + 16, OFF_BD = OFF_WD + 144, OFF_W2 = OFF_BD + 16, CHUNK = OFF_W2 + 4 * NT2 * 64;
    constexpr int EPL = ((RH * RW + 15) / 16) * 16 + YF_MRES_EPL_PAD;
    static_assert(NPX % 16 == 0 && CIN % 16 == 0 && CHUNK == mres_chunk_floats(CIN, COUT, WM_F32), "shape / pack layout");
    __shared__ __attribute__((aligned(16))) float X[NPX * XP];
    __shared__ __attribute__((aligned(16))) float E[4 * EPL * 4];
    const int n = blockIdx.x / NCH, g = blockIdx.x - n * NCH;
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int k = a.k;
    constexpr int C4 = CIN / 4, NX4 = NPX * C4, NTHR = NWAVE * 64;
    const long fr = (long)NPX * CIN;                       // floats per frame
    // ---- X_k: the chain's input (k = 0) or x + (sum of the partial sums of block k - 1, chunk order) + bias ----
    if (k == 0) {
        for (int i = threadIdx.x; i < NX4; i += NTHR) {
            const int px = i / C4, c4 = i - px * C4;
            *reinterpret_cast<float4*>(&X[px * XP + c4 * 4]) = *reinterpret_cast<const float4*>(a.in + n * fr + (long)px * CIN + c4 * 4);
        }
    } else {
        const float* xprev = (k == 1 ? a.in : a.xbuf + ((long)((k - 1) & 1) * a.n_frames) * fr) + n * fr;
        const float* sl = a.slab + (((long)((k - 1) & 1) * a.n_frames + n) * NCH) * fr;
        const float* b2 = a.wp + (long)(k - 1) * a.wstride + NCH * CHUNK;
        float* xnext = a.xbuf + ((long)(k & 1) * a.n_frames + n) * fr;
        for (int i = threadIdx.x; i < NX4; i += NTHR) {
            const int px = i / C4, c4 = i - px * C4;
            const long o = (long)px * CIN + c4 * 4;
            float4 v = *reinterpret_cast<const float4*>(sl + o);
#pragma unroll
            for (int c = 1; c < NCH; ++c) {
                const float4 t = *reinterpret_cast<const float4*>(sl + c * fr + o);
                v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
            }
            const float4 bb = *reinterpret_cast<const float4*>(b2 + c4 * 4);
            const float4 xx = *reinterpret_cast<const float4*>(xprev + o);
            v.x = (v.x + bb.x) + xx.x; v.y = (v.y + bb.y) + xx.y; v.z = (v.z + bb.z) + xx.z; v.w = (v.w + bb.w) + xx.w;
            *reinterpret_cast<float4*>(&X[px * XP + c4 * 4]) = v;
            if (g == 0 && k < a.nblk) *reinterpret_cast<float4*>(xnext + o) = v;                    // the next launch's x
            if (g == 0 && k == a.nblk && a.out) *reinterpret_cast<float4*>(a.out + n * fr + o) = v;   // the chain's result
        }
    }
    for (int i = threadIdx.x; i < 4 * EPL; i += NTHR) reinterpret_cast<float4*>(E)[i] = make_float4(0.f, 0.f, 0.f, 0.f);   // the halo ring stays zero
    __syncthreads();
    const int p = wave * 16 + r, py = p / TW, pxx = p - py * TW;          // this lane's pixel of the wave's M-tile (row-major)
    const int rp = (py + 1) * RW + pxx + 1;                              // its record in E
    if (k == a.nblk) {
        // ---- the trailing 1x1 conv + ReLU on the chain's result: n-tile g, M-tile = wave; mres_post_conv's arithmetic and order ----
        if constexpr (POSTN > 0) {
            constexpr int NTP = POSTN / 16;
            if (!a.post_w || g >= NTP) return;
            const float* pw = a.post_w;
            float4 av[NB1];
#pragma unroll
            for (int kb = 0; kb < NB1; ++kb) av[kb] = *reinterpret_cast<const float4*>(&X[p * XP + kb * 16 + 4 * q]);
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < NB1; ++kb)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pw[((kb * 4 + j) * NTP + g) * 64 + lane], ((const float*)&av[kb])[j], acc, 0, 0, 0);
            const float4 bias = *reinterpret_cast<const float4*>(pw + KS1 * NTP * 64 + g * 16 + 4 * q);
            *reinterpret_cast<float4*>(a.post_out + ((long)n * NPX + p) * POSTN + g * 16 + 4 * q) =
                make_float4(fmaxf(acc[0] + bias.x, 0.f), fmaxf(acc[1] + bias.y, 0.f), fmaxf(acc[2] + bias.z, 0.f), fmaxf(acc[3] + bias.w, 0.f));
        }
        return;
    }
    // ---- chunk g of block k: expansion of the wave's M-tile -> E -> depthwise -> partial projection ----
    const float* wc = a.wp + (long)k * a.wstride + g * CHUNK;
    float a1[KS1], w1f[KS1];
#pragma unroll
    for (int kb = 0; kb < NB1; ++kb) {
        const float4 t = *reinterpret_cast<const float4*>(&X[p * XP + kb * 16 + 4 * q]);
        a1[kb * 4 + 0] = t.x; a1[kb * 4 + 1] = t.y; a1[kb * 4 + 2] = t.z; a1[kb * 4 + 3] = t.w;
    }
#pragma unroll
    for (int s = 0; s < KS1; ++s) w1f[s] = wc[s * 64 + lane];
    const float4 b1 = *reinterpret_cast<const float4*>(wc + OFF_B1 + 4 * q);
    float4 wd[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wd[t] = *reinterpret_cast<const float4*>(wc + OFF_WD + t * 16 + 4 * q);
    const float4 bd = *reinterpret_cast<const float4*>(wc + OFF_BD + 4 * q);
    float w2f[4][NT2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) w2f[j][nt] = wc[OFF_W2 + (j * NT2 + nt) * 64 + lane];
    f32x4 cf = f32x4{b1.x, b1.y, b1.z, b1.w};
#pragma unroll
    for (int s = 0; s < KS1; ++s) cf = __builtin_amdgcn_mfma_f32_16x16x4f32(w1f[s], a1[s], cf, 0, 0, 0);
    *reinterpret_cast<float4*>(E + (q * EPL + rp) * 4) = make_float4(fmaxf(cf[0], 0.f), fmaxf(cf[1], 0.f), fmaxf(cf[2], 0.f), fmaxf(cf[3], 0.f));
    __syncthreads();
    const float4* e = reinterpret_cast<const float4*>(E) + q * EPL + rp;
    float d[4] = {bd.x, bd.y, bd.z, bd.w};
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const float4 v = e[(t / 3 - 1) * RW + (t % 3 - 1)];
        d[0] = fmaf(v.x, wd[t].x, d[0]); d[1] = fmaf(v.y, wd[t].y, d[1]); d[2] = fmaf(v.z, wd[t].z, d[2]); d[3] = fmaf(v.w, wd[t].w, d[3]);
    }
    f32x4 acc[NT2];
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float dj = relu_bits(d[j]);
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2f[j][nt], dj, acc[nt], 0, 0, 0);
    }
    float* so = a.slab + ((((long)(k & 1) * a.n_frames + n) * NCH + g) * NPX + p) * CIN + 4 * q;
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt) *reinterpret_cast<float4*>(so + nt * 16) = make_float4(acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]);
}

enum { ESPLIT_MAX_FRAMES = 9 };   // N x 14 workgroups stay a small share of the chip; larger batches keep the chained launch
size_t mres_esplit_scratch_floats() { return (size_t)ESPLIT_MAX_FRAMES * (2 + 2 * 14) * 80 * 48; }
static bool mres_esplit_ok(const MresArgs& a, int N, int dtype)
{
    static const bool off = getenv("YF_MRES_SMALL_OFF") != nullptr || getenv("YF_MRES_ESPLIT_OFF") != nullptr;
    return !off && dtype == DT_F32 && a.esplit && a.nblk > 1 && a.H == 8 && a.W == 10 && N <= ESPLIT_MAX_FRAMES;
}
static int launch_res5_esplit(const MresArgs& a, int N, hipStream_t s)
{
    EsplitArgs e{a.in, a.wp, a.wstride, a.post_w ? nullptr : a.out, a.post_w, a.post_out, a.esplit, a.esplit + (size_t)2 * N * 80 * 48, a.nblk, 0, N};
    for (int k = 0; k <= a.nblk; ++k) {
        e.k = k;
        hipLaunchKernelGGL((mres_esplit_kernel<48, 224, 8, 10, 96>), dim3((unsigned)(N * 14)), dim3(5 * 64), 0, s, e);
    }
    return 0;
}

// Small batches (round 5): when the whole-frame / 16x20 tiling of a launch would leave more than half of the CUs idle, the same arithmetic
// runs on 8x10 tiles -- four times the workgroups.  A residual CHAIN (tile == frame, one workgroup per frame for all its blocks) is then
// issued block by block on those tiles: each block reads the previous one's result WITH its halo, so consecutive blocks go through HBM and
// alternate between a scratch tensor of the same shape (MresArgs::out_exp, from the engine's plan) and the output tensor, the last block
// writing the output; the chain's input is only read (the launch stays repeatable).  fp32 storage only (a block's result is the same fp32 value in LDS and in HBM: bitwise the chained launch; fp16 storage would
// round between the blocks).  The launch count rises by nblk - 1; measured in DESIGN.md section 4 "Small batches".
static bool mres_small_batch(int N, int H, int W, int th, int tw)
{
    static const bool off = getenv("YF_MRES_SMALL_OFF") != nullptr;   // developer switch (A/B)
    const int n_cu = device_cu_count(current_device());
    const long big_tiles = (long)N * ((H + th - 1) / th) * ((W + tw - 1) / tw);
    return !off && n_cu > 0 && 2 * big_tiles <= n_cu;
}

template <int CIN, int CEXP, int TH, int TW, int NWP, int NWC, typename T>
static int launch_chain_unchained(const MresArgs& a, int N, hipStream_t s)
{
    // Block k writes the output tensor or the scratch tensor, alternating so that the LAST block writes the output; the chain's input is only
    // ever read.  (A first version also used the input tensor as a buffer: the launch was then not repeatable -- yf_set_profile_repeats issues
    // every op several times back to back -- and test_profile_with_repeated_launches_changes_nothing caught it.)
    if (a.post_w || !a.out_exp || !a.out) return -5;
    const float* src = a.in;
    for (int k = 0; k < a.nblk; ++k) {
        float* dst = ((a.nblk - 1 - k) & 1) ? a.out_exp : a.out;
        MresArgs b = a;
        b.in = src; b.out = dst; b.wp = a.wp + (size_t)k * a.wstride; b.nblk = 1; b.wstride = 0; b.out_exp = nullptr; b.post_w = nullptr; b.post_out = nullptr;
        if (int rc = launch_mres_pc_t<CIN, CEXP, CIN, true, TH, TW, NWP, NWC, T>(b, N, s)) return rc;
        src = dst;
    }
    return 0;
}

// kernel dispatches launch_mres() issues for this op at batch N (counter tools match dispatches to launches by order)
int mres_dispatches(int cin, int cexp, int cout, bool res, int stride, int nblk, bool has_scratch, bool has_post, int H, int W, int N, int dtype, bool esplit)
{
    if (dtype != DT_F16 && cin == 24 && cexp == 136 && cout == 24 && res && stride == 1 && nblk > 1 && has_scratch && !has_post && mres_small_batch(N, H, W, 16, 20))
        return nblk;
    if (esplit && cin == 48 && cexp == 224 && cout == 48 && res && stride == 1 && nblk > 1 && has_scratch && dtype == DT_F32 && H == 8 && W == 10 && N <= ESPLIT_MAX_FRAMES &&
        !getenv("YF_MRES_SMALL_OFF") && !getenv("YF_MRES_ESPLIT_OFF"))
        return nblk + 1;
    return 1;
}

int launch_mres(int cin, int cexp, int cout, bool res, int stride, const MresArgs& a, int N, hipStream_t s, int dtype)
{
    if (dtype != DT_F16 && cin == 24 && cexp == 136 && cout == 24 && res && stride == 1 && a.nblk > 1 && a.out_exp && !a.post_w &&
        mres_small_batch(N, a.H, a.W, 16, 20))
        return dtype == DT_F16X3 ? launch_chain_unchained<24, 136, 8, 10, 3, 5, x3_t>(a, N, s) : launch_chain_unchained<24, 136, 8, 10, 3, 5, float>(a, N, s);
    if (cin == 48 && cexp == 224 && cout == 48 && res && stride == 1 && mres_esplit_ok(a, N, dtype)) return launch_res5_esplit(a, N, s);
    // (the stride-32 chain the same way -- 8x4 tiles, three workgroups per 8x10 frame, conv5_2 as a launch of its own behind it -- measures 64 -> 58 us
    //  at batch 1 for six launches instead of one and no gain end to end: every workgroup stages the block's 94 KB weight stream; not kept)
    if (cin == 24 && cexp == 136 && cout == 48 && !res && stride == 2 && mres_small_batch(N, a.H / 2, a.W / 2, 8, 10))   // conv4_2 triple: 8x4 output tiles
        return dtype == DT_F16 ? launch_mres_t<24, 136, 48, false, 2, 8, 4, 8, half_t>(a, N, s)
             : dtype == DT_F16X3 ? launch_mres_t<24, 136, 48, false, 2, 8, 4, 8, x3_t>(a, N, s) : launch_mres_t<24, 136, 48, false, 2, 8, 4, 8, float>(a, N, s);
    if (cin == 16 && cexp == 96 && cout == 24 && !res && stride == 2 && mres_small_batch(N, a.H / 2, a.W / 2, 8, 10))   // conv3_5 triple: 8x4 output tiles
        return dtype == DT_F16 ? launch_mres_t<16, 96, 24, false, 2, 8, 4, 8, half_t>(a, N, s)
             : dtype == DT_F16X3 ? launch_mres_t<16, 96, 24, false, 2, 8, 4, 8, x3_t>(a, N, s) : launch_mres_t<16, 96, 24, false, 2, 8, 4, 8, float>(a, N, s);
    if (cin == 16 && cexp == 96 && cout == 16 && res && stride == 1 && a.nblk <= 1 && mres_small_batch(N, a.H, a.W, 16, 20))
        return dtype == DT_F16 ? launch_mres_t<16, 96, 16, true, 1, 8, 10, 8, half_t>(a, N, s)
             : dtype == DT_F16X3 ? launch_mres_t<16, 96, 16, true, 1, 8, 10, 8, x3_t>(a, N, s) : launch_mres_t<16, 96, 16, true, 1, 8, 10, 8, float>(a, N, s);
    if (a.post_w) {
        if (!(cin == 48 && cexp == 224 && cout == 48 && res && stride == 1)) return -5;
        return dtype == DT_F16 ? launch_mres_pc_t<48, 224, 48, true, 8, 10, YF_RES5_NWP, YF_RES5_NWC, half_t, 96>(a, N, s)
             : dtype == DT_F16X3 ? launch_mres_pc_t<48, 224, 48, true, 8, 10, YF_RES5_NWP, YF_RES5_NWC, x3_t, 96>(a, N, s)
                                 : launch_mres_pc_t<48, 224, 48, true, 8, 10, YF_RES5_NWP, YF_RES5_NWC, float, 96>(a, N, s);
    }
#define MR(ci, ce, co, rs, st, th, tw, np, nw)                                                                    \
    if (cin == ci && cexp == ce && cout == co && res == rs && stride == st)                                       \
        return dtype == DT_F16 ? launch_mres_any<ci, ce, co, rs, st, th, tw, np, nw, half_t>(a, N, s)              \
             : dtype == DT_F16X3 ? launch_mres_any<ci, ce, co, rs, st, th, tw, np, nw, x3_t>(a, N, s)              \
                               : launch_mres_any<ci, ce, co, rs, st, th, tw, np, nw, float>(a, N, s);
    YF_MRES_SHAPES(MR)
#undef MR
    return -1;
}

// Can consecutive residual blocks of this shape run as ONE launch on H x W frames?  (producer/consumer kernel whose tile is the frame)
bool mres_can_chain(int cin, int cexp, int cout, int H, int W)
{
#define MR(ci, ce, co, rs, st, th, tw, np, nw) \
    if (cin == ci && cexp == ce && cout == co && rs && st == 1) return np > 0 && H <= th && W <= tw;
    YF_MRES_SHAPES(MR)
#undef MR
    return false;
}

bool mres_has_kernel(int cin, int cexp, int cout, bool res, int stride, bool relu_out, int dtype)
{
    // the 8/32 residual blocks at stride 4: on fp32 MFMAs this kernel only ties the VALU block kernel (61.5 vs 64 us alone, nothing end
    // to end); with split-operand fp16 MFMAs, which run beside the depthwise VALU work, it wins
    if (cin == 8 && cexp == 32 && cout == 8 && res && stride == 1) {
        // The res2 pair (8 / 32 / 8 at stride 4) stays on the fp32 VALU block kernel in EVERY dtype since round 4: in an f16x3 engine the
        // split-operand MFMA form spends as much VALU time (depthwise + operand splits + E records: 0.66 of 80 us at 640x512) as the VALU
        // kernel needs in total (0.72 of 73 us) and is slower alone -- interleaved A/B (tools/res2_ab.sh, three rounds): 640x512 batch 128
        // 92.1 -> 92.7 k frames/s (one batch at a time 90.5 -> 92.2 k), 320x256 batch 256 363.9 -> 366.9 k (352.8 -> 359.6 k).
        // YF_RES2_X3=1 (developer switch) restores round 3's choice.
        static const bool res2_x3 = getenv("YF_RES2_X3") != nullptr;
        // (fp16 storage, single fp16 MFMAs, no operand splits: 72 us against the VALU kernel's 61 at 640x512 batch 128 -- round 5)
        return dtype == DT_F16X3 && !relu_out && res2_x3;
    }
#define MR(ci, ce, co, rs, st, th, tw, np, nw) \
    if (cin == ci && cexp == ce && cout == co && res == rs && stride == st) return relu_out == mres_relu_out(ci, ce, co, st);
    YF_MRES_SHAPES(MR)
#undef MR
    return false;
}

// Host-side weight stream of one block: NCH chunks of [W1 frags | b1 | wd 9x16 | bd | W2 frags], then b2.
// h16: the fragments are f16x4 per lane (one v_mfma_f32_16x16x16_f16 per 4 k-steps); b1 / wd / bd / b2 stay fp32.
size_t mres_packed_floats(int cin, int cexp, int cout, int wmode)
{
    return (((size_t)((cexp + 15) / 16) * mres_chunk_floats(cin, cout, wmode) + cout) + 3) & ~(size_t)3;
}

void mres_pack_weights(const float* w1 /*[cin][cexp]*/, const float* b1, const float* wd /*[9][cexp]*/, const float* bd,
                       const float* w2 /*[cexp][cout]*/, const float* b2, int cin, int cexp, int cout, float* out, int wmode)
{
    const bool h16 = wmode != WM_F32, x3 = wmode == WM_F16X3;
    const int KS1 = mres_ksteps(cin), NB1 = cin / 16, NK1 = (KS1 + 3) / 4, NT2 = (cout + 15) / 16, NCH = (cexp + 15) / 16;
    const int CH = mres_chunk_floats(cin, cout, wmode);
    for (int c = 0; c < NCH; ++c) {
        float* o = out + (size_t)c * CH;
        auto ch_ok = [&](int ch) { return c * 16 + ch < cexp; };
        auto w1_at = [&](int s, int lane) -> float {  // k-step s of the fp32 scheme, lane (q, n)
            const int q = lane >> 4, nn = lane & 15;
            const int kb = s / 4, j = s % 4;
            if (s >= KS1) return 0.f;
            const int k = kb < NB1 ? kb * 16 + 4 * q + j : NB1 * 16 + 2 * q + j;  // trailing 8-block: j in {0,1}
            return ch_ok(nn) ? w1[(size_t)k * cexp + c * 16 + nn] : 0.f;
        };
        if (h16) {
            uint16_t* o16 = reinterpret_cast<uint16_t*>(o);
            for (int m = 0; m < NK1; ++m)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 4; ++j) {
                        const float v = w1_at(4 * m + j, lane);
                        o16[(m * 64 + lane) * 4 + j] = f32_to_f16_bits(v);
                        if (x3) o16[((NK1 + m) * 64 + lane) * 4 + j] = f16_lo_bits(v);
                    }
            o += (x3 ? 2 : 1) * NK1 * 128;
        } else {
            for (int s = 0; s < KS1; ++s)
                for (int lane = 0; lane < 64; ++lane) o[s * 64 + lane] = w1_at(s, lane);
            o += KS1 * 64;
        }
        for (int ch = 0; ch < 16; ++ch) o[ch] = ch_ok(ch) ? b1[c * 16 + ch] : 0.f;
        o += 16;
        for (int t = 0; t < 9; ++t)
            for (int ch = 0; ch < 16; ++ch) o[t * 16 + ch] = ch_ok(ch) ? wd[(size_t)t * cexp + c * 16 + ch] : 0.f;
        o += 144;
        for (int ch = 0; ch < 16; ++ch) o[ch] = ch_ok(ch) ? bd[c * 16 + ch] : 0.f;
        o += 16;
        auto w2_at = [&](int j, int nt, int lane) -> float {
            const int q = lane >> 4, nn = nt * 16 + (lane & 15), ch = 4 * q + j;
            return (ch_ok(ch) && nn < cout) ? w2[(size_t)(c * 16 + ch) * cout + nn] : 0.f;
        };
        if (h16) {
            uint16_t* o16 = reinterpret_cast<uint16_t*>(o);
            for (int nt = 0; nt < NT2; ++nt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 4; ++j) {
                        const float v = w2_at(j, nt, lane);
                        o16[(nt * 64 + lane) * 4 + j] = f32_to_f16_bits(v);
                        if (x3) o16[((NT2 + nt) * 64 + lane) * 4 + j] = f16_lo_bits(v);
                    }
        } else if (YF_MRES_Q4 && cout == 8) {   // 4x4x1 form: [h][lane][j] = W2[chunk channel 4q + j][output channel 4h + (lane & 3)]
            for (int h = 0; h < 2; ++h)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 4; ++j) {
                        const int ch = 4 * (lane >> 4) + j;
                        o[(h * 64 + lane) * 4 + j] = ch_ok(ch) ? w2[(size_t)(c * 16 + ch) * cout + 4 * h + (lane & 3)] : 0.f;
                    }
        } else {
            for (int j = 0; j < 4; ++j)
                for (int nt = 0; nt < NT2; ++nt)
                    for (int lane = 0; lane < 64; ++lane) o[(j * NT2 + nt) * 64 + lane] = w2_at(j, nt, lane);
        }
    }
    for (int i = 0; i < cout; ++i) out[(size_t)NCH * CH + i] = b2[i];
}

}  // namespace yf
