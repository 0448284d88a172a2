// yf_k19_kernels.hip -- conv1_8 (pw 4->24, ReLU) -> conv1_9 (dense 3x3 stride 2 pad 1, 24->24, ReLU) -> conv2_1 (pw 24->8,
// linear) in one launch, the dense 3x3 on the matrix cores.            (reference: src/model_training/model/yolo_fastest.py:86-89)
//
// conv1_9 is 22 % of the network's MACs: an implicit GEMM  D[cout][pixel] = sum_k W[cout][k] X[k][pixel]  with k = (tap, cin),
// K = 9 x 24 = 216.  The VALU version (k19_kernel, yf_fused_kernels.hip) runs at ~30 TMAC/s (250 us for 256 frames); this one
// at 37 TMAC/s fp32 (200 us) and 80 us with fp16 storage:
//   * a persistent workgroup (512 threads: 8 waves, two per SIMD, one output row each) per CU walks over 8x16-pixel output
//     tiles; two region buffers in LDS (2 x 55 KB fp32), ONE barrier per tile;
//   * phase 1 (conv1_8 over the tile's 17x33 input region -> LDS, as even-column / odd-column planes of 24-channel records,
//     so that the 16 pixels of an output row, at a fixed tap, are 16 consecutive records) ALSO runs on the matrix cores (K = 4
//     is one v_mfma_f32_16x16x4_f32 k-step, the bias is the C operand, the result layout is one 16-byte LDS record write per
//     lane) and is spread over the k groups of phase 2 of the previous tile; its input elements are requested two tiles ahead;
//   * phase 2: K is walked in 14 groups of 16 k-values; lane (pixel p = l & 15, j = l >> 4) supplies the 4 consecutive
//     channels of flat chunk 4 g + j -- ONE ds_read_b128 (fp32: four v_mfma_f32_16x16x4_f32 k-steps, exact fp32) or ONE
//     ds_read_b64 (fp16: one v_mfma_f32_16x16x16_f16) -- as the B operand; the weights are the A operand and live in registers
//     for the lifetime of the workgroup (112 VGPRs fp32 / 56 fp16).  24 output channels = 1.5 M-tiles (25 % of the MFMA rows
//     are padding), 216 = 13.5 groups (the last half group has zero weights);
//   * epilogue: the accumulator layout (lane: pixel p, channels 16 mt + 4 j + r) IS the B operand of conv2_1's GEMM with the
//     k-order permuted on the host, so bias + ReLU + conv2_1 stay in registers; lanes j < 2 store 4 channels each.
// Measured on the way (tools/kbench.hip k19, SQ counters): the fp32 MFMA floor of this shape is 128 us (SQ_VALU_MFMA_BUSY =
// 32.3 cycles per MFMA, 68 % busy); fp32 MFMAs and VALU instructions do not overlap (each VALU instruction adds its 4 cycles:
// the fp32 matrix rate equals the packed fp32 vector rate), which is why phase 1 went from VALU (68 us) to MFMA and why its
// loads carry no address arithmetic; two workgroups per CU with one buffer each run in lockstep and never overlap their
// phases (offsets between them are neutrally stable), hence the in-wave interleave; LDS strides are padded so that both the
// reads and phase 1's writes are conflict-free (bank rules of MI355X_MICROARCH.md).
#include "yf_kernels.h"
#include <math.h>
#include <stdlib.h>
#include <type_traits>

namespace yf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {
constexpr int TH = 8, TW = 16;             // output tile
constexpr int RH = 2 * TH + 1;             // region rows
constexpr int RS = 24;                     // record stride in elements
// plane / row strides in elements: padded so that BOTH the phase-2 reads (b128 / b64 lane groups, 64 banks) and the phase-1
// writes (8- / 16-lane groups, 32 banks) are conflict-free -- unpadded, the writes are 2.3-way conflicted and the LDS write
// path, not the VALU, is what phase 1 costs (measured: 47 us of 210)
#ifndef YF_K19_HPAD
#define YF_K19_HPAD 8
#endif
// (fp16 storage: multiples of 8 halves, the 16-byte pair records of the K = 32 MFMA fragments must stay 16-byte aligned)
constexpr int plane_stride(bool h16) { return 17 * RS + (h16 ? YF_K19_HPAD : 4); }
constexpr int row_stride(bool h16) { return 2 * plane_stride(h16) + (h16 ? YF_K19_HPAD : 4); }
constexpr int NG = 14;                     // k groups
constexpr int NU = 5;                      // px-tiles of phase 1 per wave (36 over 8 waves)
constexpr int NCHUNK = 54;                 // 9 taps x 6 chunks of 4 channels
constexpr int W9_F32 = NG * 4 * 2 * 64, W21_F32 = 2 * 4 * 64;
// fp32 only: channels 16..23 of conv1_9 (the half-empty second M-tile) on v_mfma_f32_4x4x1_16B_f32 instead -- weights
// [g][cg][lane][s] (staged in LDS) and conv2_1's two k-steps for those channels [t][lane]
constexpr int WQ_F32 = NG * 2 * 64 * 4, W21Q_F32 = 2 * 64;
constexpr int W9_F16 = NG * 2 * 64 * 2, W21_F16 = 2 * 64 * 2;  // in floats (f16x4 = 2 floats per lane)
// split-operand mode (DT_F16X3): [W9 hi | W21 hi | W9 lo | W21 lo], each in the fp16 layout
constexpr int WX3_HALF = W9_F16 + W21_F16;
// X3 walks K in NG2 = 7 groups of 32 k-values on v_mfma_f32_16x16x32_f16 (the K = 32 form issues in the cycles of the K = 16 one:
// tools/mfma16_probe.hip): lane group j holds the PAIR of 4-channel chunks 2 (4 g + j), 2 (4 g + j) + 1 -- 27 pairs = 9 taps x 3
constexpr int NG2 = 7, NPAIR = 27;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
}  // namespace

#ifndef YF_K19_Q4
#define YF_K19_Q4 1
#endif
#ifndef YF_K19R_DBG
#define YF_K19R_DBG 0   // timing builds only (tools/build_variant.sh): 1 = no 4x4x1 MFMAs, 4 = no 16x16x4 k-steps of conv1_9
#endif
#ifndef YF_K19R_PIPE
#define YF_K19R_PIPE 2   // k19r_kernel: the next tap's conv1_8 is issued in front of the current tap's k-steps (0: behind them; 2: in front, no scheduling barrier per tap)
#endif
#ifndef YF_K19_PF
#define YF_K19_PF 1   // how many k groups ahead phase 2's LDS operands are requested
#endif
// Q4 (fp32): output channels 16..23 of conv1_9 do not ride in a second 16-row M-tile (half of whose rows are padding: 25 % of all
// MFMA cycles) but in 4x4 blocks: v_mfma_f32_4x4x1_16B_f32 is 16 independent 4x4 outer products, block b = lanes 4b..4b+3, and
// with lane = (pixel p, chunk j) block (j, p >> 2) multiplies 4 channels x the 4 pixels of its lanes at the k-value THAT lane
// group holds anyway (the same B register as the 16x16x4 k-step).  Each lane group accumulates its own k-values, so the four
// partial sums are added across lane groups once per tile (ds_bpermute), and conv2_1 takes channel 16 + 4 t + j from lane group
// j.  Measured (tools/mfma4_probe.hip): a wave issues one 4x4x1 per 12 cycles and two waves of a SIMD do not slow each other,
// against 32 cycles of pipe per 16x16x4.
// DBG (tools/kbench.hip only): 1 = skip phase 1, 2 = skip phase 2's MFMAs
template <typename TT, int DBG = 0>
__global__ void __launch_bounds__(512) k19m_kernel(K19Args a)
{
    // X3 (DT_F16X3): fp32 in HBM; conv1_8 (K = 4) stays an exact fp32 MFMA; its result is SPLIT ONCE, where phase 1 stores it: a
    // region record of 4 channels is 16 bytes either way -- four floats (fp32 mode) or [hi f16x4 | lo f16x4] -- so phase 2 reads
    // both halves of its B operand with plain ds_read_b128s and spends no VALU on splitting; conv1_9's weights are register-resident
    // as hi and lo fragments (2 x 56 VGPRs) and a k group (32 k-values, v_mfma_f32_16x16x32_f16) issues w_lo*x_hi + w_hi*x_lo + w_hi*x_hi.
    constexpr bool X3 = is_x3<TT>::value;
    constexpr bool H16 = sizeof(TT) == 2;
    constexpr bool M16 = H16 || X3;                // conv1_9 / conv2_1 on the fp16 matrix pipe
    constexpr int PS = plane_stride(H16), RWS = row_stride(H16);
    constexpr int BUF = RH * RWS;  // elements per region buffer
    extern __shared__ __attribute__((aligned(16))) unsigned char k19_smem[];
    TT* const R = reinterpret_cast<TT*>(k19_smem);  // [2][BUF]
    constexpr bool Q4 = !M16 && YF_K19_Q4;
    float* const WQ = reinterpret_cast<float*>(k19_smem + (size_t)2 * BUF * sizeof(TT));  // Q4: [NG][2][64][4]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // = the tile's output row this wave owns
    const int p = lane & 15, j = lane >> 4;

    // ---- weights: registers for the lifetime of the workgroup ----
    float wf[M16 ? 1 : NG][4][2];
    f16x8 wh8[M16 ? NG2 : 1][2], wl8[X3 ? NG2 : 1][2];
    float w21f[2][4];
    f16x4 w21h[2], w21l[2];
    if constexpr (X3) {
        const f16x8* w = reinterpret_cast<const f16x8*>(a.wp);
        const f16x8* wlo = reinterpret_cast<const f16x8*>(a.wp + WX3_HALF);
#pragma unroll
        for (int g = 0; g < NG2; ++g)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) { wh8[g][mt] = w[(g * 2 + mt) * 64 + lane]; wl8[g][mt] = wlo[(g * 2 + mt) * 64 + lane]; }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            w21h[mt] = reinterpret_cast<const f16x4*>(a.wp + W9_F16)[mt * 64 + lane];
            w21l[mt] = reinterpret_cast<const f16x4*>(a.wp + WX3_HALF + W9_F16)[mt * 64 + lane];
        }
    } else if constexpr (H16) {
        const f16x8* w = reinterpret_cast<const f16x8*>(a.wp);
#pragma unroll
        for (int g = 0; g < NG2; ++g)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) wh8[g][mt] = w[(g * 2 + mt) * 64 + lane];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) w21h[mt] = reinterpret_cast<const f16x4*>(a.wp + W9_F16)[mt * 64 + lane];
    } else {
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int mt = 0; mt < (Q4 ? 1 : 2); ++mt) wf[g][s][mt] = a.wp[((g * 4 + s) * 2 + mt) * 64 + lane];
#pragma unroll
        for (int mt = 0; mt < (Q4 ? 1 : 2); ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) w21f[mt][r] = a.wp[W9_F32 + (mt * 4 + r) * 64 + lane];
    }
    float w21q[2] = {0.f, 0.f}, biasq[2] = {0.f, 0.f};
    if constexpr (Q4) {
        stage_to_lds<WQ_F32, 512>(WQ, a.wp + W9_F32 + W21_F32);   // all loads in flight at once; visible after the prologue's barrier
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            w21q[t] = a.wp[W9_F32 + W21_F32 + WQ_F32 + t * 64 + lane];
            biasq[t] = a.b9[16 + 4 * t + j];
        }
    }
    float bias9[2][4], bias21[4], w8a[2], bias8[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int c = 16 * mt + p;
        w8a[mt] = c < 24 ? a.w8[j * 24 + c] : 0.f;  // conv1_8's A operand: row = cout, k = cin
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int cc = 16 * mt + 4 * j + r;
            bias9[mt][r] = cc < 24 ? a.b9[cc] : 0.f;
            bias8[mt][r] = cc < 24 ? a.b8[cc] : 0.f;
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) bias21[r] = j < 2 ? a.b21[4 * j + r] : 0.f;

    // ---- per-lane LDS element offsets (relative to a region buffer) of the 14 B-operand reads of phase 2 ----
    int adr[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        int fc = 4 * g + j;
        if (fc >= NCHUNK) fc = 0;  // zero weights there; any finite data will do
        const int tap = fc / 6, c4 = fc - 6 * tap, ky = tap / 3, kx = tap - 3 * ky;
        adr[g] = (2 * wave + ky) * RWS + (kx == 1 ? PS : 0) + (p + (kx >> 1)) * RS + c4 * 4;
    }

    int adr2[M16 ? NG2 : 1];   // the pair record lane group j reads in group g -- X3: 32 bytes [hi8 | lo8]; fp16 storage: 8 halves
    if constexpr (M16) {
#pragma unroll
        for (int g = 0; g < NG2; ++g) {
            int fp = 4 * g + j;
            if (fp >= NPAIR) fp = 0;   // zero weights there
            const int tap = fp / 3, pr = fp - 3 * tap, ky = tap / 3, kx = tap - 3 * ky;
            adr2[g] = (2 * wave + ky) * RWS + (kx == 1 ? PS : 0) + (p + (kx >> 1)) * RS + pr * 8;
        }
    }

    // ---- phase 1 bookkeeping: the wave's px-tiles are wave, wave + 8, .. (NU of them, the last only for waves 0..3) ----
    // rr: region row | column << 8 of the lane's pixel; vo: element offset of the lane's input element from the region origin;
    // wo0 / wo1: where the lane's two 4-channel results go.  Lanes without a pixel, and the M-tile-1 lanes whose channels
    // 24..31 do not exist, write into the unused 17th record of an odd-column plane (one slot per lane: no same-address
    // serialisation) and load the region origin's element, so phase 1 is branch-free.
    int rr[NU], wo0[NU], wo1[NU];
    unsigned vo[NU];
    {
        const int dummy = (lane / 6) * RWS + PS + 16 * RS + 4 * (lane % 6);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int pt = wave + 8 * u, q = 16 * pt + p;
            const bool v = pt < 36 && q < RH * 33;
            const int ry = v ? q / 33 : 0, rx = v ? q - 33 * ry : 0;
            rr[u] = v ? (ry | (rx << 8)) : 0xffff;  // 0xffff: never inside any image window
            vo[u] = (unsigned)((ry * a.W + rx) * 4 + j);
            const int o = ry * RWS + (rx & 1) * PS + (rx >> 1) * RS + (X3 ? 0 : 4 * j);
            if constexpr (X3) {   // chunk c = j (M-tile 0) / 4 + j (M-tile 1) -> hi4 slot of its pair record
                wo0[u] = v ? o + 8 * (j >> 1) + 2 * (j & 1) : dummy;
                wo1[u] = (v && j < 2) ? o + 16 + 2 * j : dummy;
            } else {
                wo0[u] = v ? o : dummy;
                wo1[u] = (v && j < 2) ? o + 16 : dummy;
            }
        }
    }

    const int tiles = a.tiles_y * a.tiles_x;
    const int total = a.n_frames * tiles;
    const TT* const in = reinterpret_cast<const TT*>(a.in);
    float xin[NU];
    struct Tile { int n, oy0, ox0; };
    auto decode = [&](int t) {
        t = min(t, total - 1);  // beyond the end: a harmless repeat of the last tile (its results are never used)
        const int n = t / tiles, tt = t - n * tiles;
        const int ty = tt / a.tiles_x, tx = tt - ty * a.tiles_x;
        return Tile{n, ty * TH, tx * TW};
    };
    // The region origin (row 2 oy0 - 1, column 2 ox0 - 1) may lie one row / column outside the image, and a partial tile's
    // region reaches beyond its far edges: the loads are NOT bounds-checked (wave-uniform base + per-lane constant offset, no
    // address arithmetic on the VALU) -- the engine keeps a guard band around the workspace slots (k19m_guard_elems) and
    // whatever is read there is replaced by zeros in p1_store.
    auto origin = [&](const Tile& c) { return in + (((long)c.n * a.H + (2 * c.oy0 - 1)) * a.W + (2 * c.ox0 - 1)) * 4; };
    auto p1_mfma = [&](int u, f32x4* d) {  // C operand = bias
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
            d[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w8a[mt], xin[u], f32x4{bias8[mt][0], bias8[mt][1], bias8[mt][2], bias8[mt][3]}, 0, 0, 0);
    };
    // window of region rows / columns that are inside the image, as one packed compare per axis: inside <=> (unsigned)(r - lo) < n
    struct Win { bool border; int ylo, yn, xlo, xn; };
    auto window = [&](const Tile& c) {
        const int iy0 = 2 * c.oy0 - 1, ix0 = 2 * c.ox0 - 1;
        const int ylo = max(0, -iy0), yhi = min(RH, a.H - iy0), xlo = max(0, -ix0), xhi = min(33, a.W - ix0);
        return Win{ylo > 0 || yhi < RH || xlo > 0 || xhi < 33, ylo, yhi - ylo, xlo, xhi - xlo};
    };
    auto p1_store = [&](int u, const f32x4* d, TT* buf, const Win& w) {
        float o[2][4];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[mt][r] = __int_as_float(max(__float_as_int(d[mt][r]), 0));  // ReLU as ONE v_max_i32 (fmaxf costs a canonicalising second op)
        if (w.border) {  // wave-uniform; zeros outside the image: conv1_9 pads conv1_8's OUTPUT
            const bool inimg = (unsigned)((rr[u] & 255) - w.ylo) < (unsigned)w.yn && (unsigned)((rr[u] >> 8) - w.xlo) < (unsigned)w.xn;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[mt][r] = inimg ? o[mt][r] : 0.f;
        }
        if constexpr (X3) {
            // a 24-channel record is three PAIR records of 32 bytes, [hi4(chunk 2m) hi4(chunk 2m+1) | lo4(chunk 2m) lo4(chunk 2m+1)]:
            // phase 2 reads hi8 and lo8 of a pair as two 16-byte pieces that ARE the K = 32 MFMA's B fragments.  wo0 / wo1 point at
            // the lane's chunk (4 floats per chunk): chunk c -> pair c >> 1, half c & 1; hi4 at 8 (c >> 1) + 2 (c & 1), lo4 4 floats on
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                f16x4 hi, lo;
                split_f16x4(o[mt][0], o[mt][1], o[mt][2], o[mt][3], hi, lo);
                TT* dst = buf + (mt ? wo1[u] : wo0[u]);
                *reinterpret_cast<f16x4*>(dst) = hi;
                *reinterpret_cast<f16x4*>(dst + 4) = lo;
            }
        } else {
            st4<TT>(buf + wo0[u], make_float4(o[0][0], o[0][1], o[0][2], o[0][3]));
            st4<TT>(buf + wo1[u], make_float4(o[1][0], o[1][1], o[1][2], o[1][3]));
        }
    };

    int t = xcd_tile(blockIdx.x, gridDim.x);  // grid <= total; XCD x walks the contiguous runs [x * grid / 8, (x + 1) * grid / 8) + k * grid
    const int step = gridDim.x;
    // ---- prologue: region of the first tile -> buffer 0; input elements of the second tile on their way ----
    {
        const Tile c = decode(t);
        const TT* const o0 = origin(c);
        const Win w0 = window(c);
#pragma unroll
        for (int u = 0; u < NU; ++u) xin[u] = ld1<TT>(o0 + vo[u]);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            f32x4 d[2];
            p1_mfma(u, d);
            p1_store(u, d, R, w0);
        }
        const TT* const o1 = origin(decode(t + step));
#pragma unroll
        for (int u = 0; u < NU; ++u) xin[u] = ld1<TT>(o1 + vo[u]);
        __syncthreads();
    }

    int cur = 0;  // element offset of the buffer phase 2 reads
    // The scalar bookkeeping of the NEXT iteration (one tile decode with two integer divisions, the image window, the load
    // origin) is computed inside this iteration's k groups 10..12, in the shadow of MFMA issue, and carried over: at the top of
    // an iteration, where both waves of a SIMD arrive together from the barrier and the matrix pipe is idle, nothing is left to do.
    Tile c = decode(t);                                    // the tile of phase 2 (and of the output store)
    Win w1 = window(decode(t + step));                     // the tile phase 1 prepares during the iteration
    Tile tl = decode(t + 2 * step);                        // the tile whose input elements are requested during the iteration
    long off2 = origin(tl) - in;
    Tile c_n = c; Win w1_n = w1; Tile tl_n = tl; long off2_n = off2;
    for (; t < total; t += step) {
        asm volatile("" : "+s"(off2));  // a plain SGPR value from here on (otherwise the origin arithmetic is re-done before every load)
        const TT* const o2 = in + off2;
        const TT* const Rc = R + cur;
        TT* const Rn = R + (BUF - cur);

        // ---- phase 2 of tile t (conv1_9 on the matrix cores; acc[mt]: channels 16 mt + 4 j + r of the wave's row), with phase 1
        // of tile t + step spread over its k groups: px-tile u's two MFMAs go out in group 2 u, their results are finished and
        // stored in group 2 u + 1, and the element of tile t + 2 step that replaces xin[u] is requested right after.  fp32 MFMAs
        // and VALU instructions do NOT overlap on this part (measured: the two phases' times add up exactly; the fp32 matrix
        // rate equals the fp32 vector rate), so phase 1 is written for the fewest VALU instructions, not for overlap ----
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        f32x4 accq[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};  // Q4: channels 16 + 4 cg + i, this lane group's k-values only
        f32x4 d[2];
        if constexpr (M16) {
            // fp16 matrix pipe: 7 groups of 32 k-values on v_mfma_f32_16x16x32_f16 -- 2 M-tiles x 3 MFMAs with split operands (X3),
            // 2 x 1 with fp16 storage; phase 1 of the next tile rides along: px-tile u's two fp32 MFMAs in group u, its (split +)
            // store in group u + 1
            f16x8 xh = *reinterpret_cast<const f16x8*>(Rc + adr2[0]), xl = xh;
            if constexpr (X3) xl = *reinterpret_cast<const f16x8*>(Rc + adr2[0] + 4);
#pragma unroll
            for (int g = 0; g < NG2; ++g) {
                f16x8 xhn = xh, xln = xl;
                if (g + 1 < NG2) {
                    xhn = *reinterpret_cast<const f16x8*>(Rc + adr2[g + 1]);
                    if constexpr (X3) xln = *reinterpret_cast<const f16x8*>(Rc + adr2[g + 1] + 4);
                }
                if (!(DBG & 1) && g >= 1 && g - 1 < NU) {
                    p1_store(g - 1, d, Rn, w1);
                    xin[g - 1] = ld1<TT>(o2 + vo[g - 1]);
                }
                if (!(DBG & 1) && g < NU) p1_mfma(g, d);
                if constexpr (!(DBG & 2)) {
                    if constexpr (X3) {
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl8[g][mt], xh, acc[mt], 0, 0, 0);
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh8[g][mt], xl, acc[mt], 0, 0, 0);
                    }
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh8[g][mt], xh, acc[mt], 0, 0, 0);
                }
                if (g == 3) { c_n = decode(t + step); w1_n = window(tl); }
                if (g == 4) { tl_n = decode(t + 3 * step); }
                if (g == 5) { off2_n = origin(tl_n) - in; }
                __builtin_amdgcn_sched_barrier(0);
                xh = xhn; xl = xln;
            }
        }
        using xfrag = f32x4;
        // LDS operands of group g are requested YF_K19_PF groups ahead (1: 188-190 us; 2: A/B in DESIGN.md)
        constexpr int PF = YF_K19_PF;
        // The 4x4x1 weight table lies behind the two region buffers, > 64 KiB into the LDS allocation: past the 16-bit offset field of a
        // DS instruction, so the compiler paid one v_add_u32 per read (28 per tile).  An opaque per-lane base keeps the per-group
        // part (<= 27 KiB) in the immediate.
        int wq_lane = (int)((size_t)2 * BUF * sizeof(TT) / sizeof(float)) + lane * 4;
        asm volatile("" : "+v"(wq_lane));
        const float* const wq_base = reinterpret_cast<const float*>(k19_smem) + wq_lane;
        xfrag xq[PF + 1];
        f32x4 wqq[PF + 1][2];
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            xq[d] = *reinterpret_cast<const xfrag*>(Rc + (M16 ? 0 : adr[d < NG ? d : 0]));
#pragma unroll
            for (int cg = 0; cg < 2; ++cg)
                wqq[d][cg] = Q4 ? *reinterpret_cast<const f32x4*>(wq_base + (d * 2 + cg) * 256) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        xq[PF] = xq[0]; wqq[PF][0] = wqq[0][0]; wqq[PF][1] = wqq[0][1];
        xfrag xc = xq[0];
        f32x4 wq[2] = {wqq[0][0], wqq[0][1]};
#pragma unroll
        for (int g = 0; g < (M16 ? 0 : NG); ++g) {
            if (g + PF < NG) {
                xq[PF] = *reinterpret_cast<const xfrag*>(Rc + adr[g + PF]);
                if constexpr (Q4) {
#pragma unroll
                    for (int cg = 0; cg < 2; ++cg) wqq[PF][cg] = *reinterpret_cast<const f32x4*>(wq_base + ((g + PF) * 2 + cg) * 256);
                }
            }
            if (!(DBG & 1) && (g & 1) == 0 && g / 2 < NU) p1_mfma(g / 2, d);
            if constexpr (!(DBG & 2)) {
                if constexpr (Q4) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[g][s][0], xc[s], acc[0], 0, 0, 0);
#pragma unroll
                        for (int cg = 0; cg < 2; ++cg) accq[cg] = __builtin_amdgcn_mfma_f32_4x4x1f32(wq[cg][s], xc[s], accq[cg], 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt)
                            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[g][s][mt], xc[s], acc[mt], 0, 0, 0);
                }
            }
            if (!(DBG & 1) && (g & 1) == 1 && g / 2 < NU) {
                p1_store(g / 2, d, Rn, w1);
                xin[g / 2] = ld1<TT>(o2 + vo[g / 2]);
            }
            if (g == 10) { c_n = decode(t + step); w1_n = window(tl); }
            if (g == 11) { tl_n = decode(t + 3 * step); }
            if (g == 12) { off2_n = origin(tl_n) - in; }
            __builtin_amdgcn_sched_barrier(0);  // keeps the PF-groups-ahead LDS reads where they are (hoisting all 14 costs 56 VGPRs)
#pragma unroll
            for (int d = 0; d < PF; ++d) { xq[d] = xq[d + 1]; wqq[d][0] = wqq[d + 1][0]; wqq[d][1] = wqq[d + 1][1]; }
            xc = xq[0];
            if constexpr (Q4) { wq[0] = wqq[0][0]; wq[1] = wqq[0][1]; }
        }

        // ---- epilogue: bias + ReLU, conv2_1 (24 -> 8) chained in registers, store ----
        {
            f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (Q4) {
                // channels 16..23: add the four lane groups' partial sums (every lane group then holds all eight totals of its
                // pixel), bias + ReLU, and hand channel 16 + 4 t + j to conv2_1's k-step t from lane group j
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    float tot[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v = accq[t][i];
                        v += __shfl_xor(v, 16);
                        v += __shfl_xor(v, 32);
                        tot[i] = v;
                    }
                    const float mine = j == 0 ? tot[0] : j == 1 ? tot[1] : j == 2 ? tot[2] : tot[3];
                    o = __builtin_amdgcn_mfma_f32_16x16x4f32(w21q[t], fmaxf(mine + biasq[t], 0.f), o, 0, 0, 0);
                }
            }
#pragma unroll
            for (int mt = 0; mt < (Q4 ? 1 : 2); ++mt) {
                f32x4 h;
#pragma unroll
                for (int r = 0; r < 4; ++r) h[r] = fmaxf(acc[mt][r] + bias9[mt][r], 0.f);
                if constexpr (X3) {
                    f16x4 hh, hl;
                    split_f16x4(h[0], h[1], h[2], h[3], hh, hl);
                    o = __builtin_amdgcn_mfma_f32_16x16x16f16(w21l[mt], hh, o, 0, 0, 0);
                    o = __builtin_amdgcn_mfma_f32_16x16x16f16(w21h[mt], hl, o, 0, 0, 0);
                    o = __builtin_amdgcn_mfma_f32_16x16x16f16(w21h[mt], hh, o, 0, 0, 0);
                } else if constexpr (H16) {
                    const f16x4 hh = f16x4{(half_t)h[0], (half_t)h[1], (half_t)h[2], (half_t)h[3]};
                    o = __builtin_amdgcn_mfma_f32_16x16x16f16(w21h[mt], hh, o, 0, 0, 0);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) o = __builtin_amdgcn_mfma_f32_16x16x4f32(w21f[mt][r], h[r], o, 0, 0, 0);
                }
            }
            const int oy = c.oy0 + wave, ox = c.ox0 + p;
            if (j < 2 && oy < a.Ho && ox < a.Wo)
                st4<TT>(reinterpret_cast<TT*>(a.out) + (((long)c.n * a.Ho + oy) * a.Wo + ox) * 8 + 4 * j,
                        make_float4(o[0] + bias21[0], o[1] + bias21[1], o[2] + bias21[2], o[3] + bias21[3]));
        }
        __syncthreads();  // buffer `cur` is free for the tile after next; the other one is complete
        cur = BUF - cur;
        c = c_n; w1 = w1_n; tl = tl_n; off2 = off2_n;
    }
}


// ------------------------------------------------------------------------------------------------
// k19r_kernel (fp32): the same three layers WITHOUT the region buffers.  conv1_8 is a K = 4 GEMM, one v_mfma_f32_16x16x4_f32 per 16
// pixels and 16 channels, and the MFMA's result layout (lane (pixel p, group j): channels 4j .. 4j+3 of pixel p) IS the B operand
// of conv1_9's k-steps for those channels -- the chain yf_dcat_kernels.hip uses for deconv5_1 -> conv4_1_1.  So conv1_8 is evaluated
// PER TAP, in registers, right in front of that tap's conv1_9 k-steps (channels 16..23: eight FMAs per lane on the VALU, two channels per
// lane group), and its 24-channel tensor never exists -- not in HBM and not in LDS.  What that buys against k19m_kernel: no 2 x 55 KB
// region buffers (a wave needs its 3 x 33 four-channel input pixels: 2.4 KB), hence no workgroup barrier per tile and no phase-1 store
// traffic, waves that are independent of each other, and twelve of them per CU (three per SIMD) instead of eight.  The matrix-pipe work
// is unchanged (per 16 output pixels 9 + 54 + 8 v_mfma_f32_16x16x4_f32 and 108 v_mfma_f32_4x4x1_16B_f32: conv1_8 costs one MFMA per
// tap here, one per region pixel tile there); the k order inside an output element differs (per tap: channels 0..15, then 16..23).
//   item = 16 consecutive output pixels of one output row; a persistent wave walks items w, w + S, ..; the 99 input pixels of the NEXT
//   item are in flight (two 16-byte loads per lane) while the current one is computed, and go to the wave's own LDS slice
//   (even-column / odd-column planes per row, so that the 16 pixels of a tap are consecutive records) when it is done.
//   Padding: conv1_9 pads conv1_8's OUTPUT.  With even H and W only input row -1 (output row 0: the three ky = 0 taps are skipped,
//   wave-uniform) and input column -1 (first segment of a row: lane p = 0 of the kx = 0 taps is zeroed) are outside; the loads
//   themselves are not bounds-checked (guard band: k19m_guard_elems).
// ------------------------------------------------------------------------------------------------
namespace {
constexpr int R_PS = 100, R_RS = 2 * R_PS;   // floats: plane (17 even-column records + pad) and row stride of a wave's slice
constexpr int R_SLICE = 3 * R_RS;            // floats per wave
constexpr int R_WA = 9 * 4 * 64, R_WB = 9 * 2 * 64, R_WQ = 9 * 3 * 64 * 4;   // floats: conv1_9 A fragments (channels 0..15 / 16..23), 4x4x1 table
constexpr int R_OFF = W9_F32 + W21_F32 + WQ_F32 + W21Q_F32;                     // the k19r stream follows k19m's in the packed blob
}  // namespace

// WLDS: conv1_9's A fragments (54 per lane) are read from LDS per tap instead of living in registers -- 16 waves per CU fit then
template <int R_NW, bool WLDS = false>   // waves per workgroup, one workgroup per CU
__global__ void __launch_bounds__(R_NW * 64) k19r_kernel(K19Args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char k19_smem[];
    float* const WQ = reinterpret_cast<float*>(k19_smem);          // [9][3][64][4]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = lane & 15, j = lane >> 4;
    float* const WAL = WQ + R_WQ;                                   // WLDS: [9][64][4] + [9][64][2] (lane-major: one b128 + one b64 per tap)
    float* const SL = WQ + R_WQ + (WLDS ? R_WA + R_WB : 0) + wave * R_SLICE;

    // the 4x4x1 table and, WLDS, the lane-major copies of the fragments ([t][lane][4] | [t][lane][2]) that follow it in the blob: one staging
    // pass with every 16-byte load in flight (transposing [t][s][lane] here cost three dependent rounds of 4-byte loads per workgroup)
    stage_to_lds<R_WQ + (WLDS ? R_WA + R_WB : 0), R_NW * 64>(WQ, a.wp + R_OFF + R_WA + R_WB);
    // ---- weights in registers for the lifetime of the wave ----
    float wA[WLDS ? 1 : 9][4], wB[WLDS ? 1 : 9][2];
    if constexpr (!WLDS) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
#pragma unroll
            for (int s = 0; s < 4; ++s) wA[t][s] = a.wp[R_OFF + (t * 4 + s) * 64 + lane];
#pragma unroll
            for (int s = 0; s < 2; ++s) wB[t][s] = a.wp[R_OFF + R_WA + (t * 2 + s) * 64 + lane];
        }
    }
    float w21f[4], w21q[2], biasq[2], bias9[4], bias21[4], bias8[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        w21f[r] = a.wp[W9_F32 + r * 64 + lane];            // M-tile 0 of k19m's conv2_1 fragments: k-step r <-> channel 4j + r
        bias9[r] = a.b9[4 * j + r];
        bias8[r] = a.b8[4 * j + r];
        bias21[r] = j < 2 ? a.b21[4 * j + r] : 0.f;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        w21q[t] = a.wp[W9_F32 + W21_F32 + WQ_F32 + t * 64 + lane];
        biasq[t] = a.b9[16 + 4 * t + j];
    }
    const float w8a = a.w8[j * 24 + p];                     // conv1_8's A operand, channels 0..15: row = cout p, k = cin j
    f32x2 w8h[4], b8h;                                      // channels 16 + 2j, 17 + 2j on the VALU
#pragma unroll
    for (int c = 0; c < 4; ++c) w8h[c] = f32x2{a.w8[c * 24 + 16 + 2 * j], a.w8[c * 24 + 17 + 2 * j]};
    b8h = f32x2{a.b8[16 + 2 * j], a.b8[17 + 2 * j]};

    // ---- the lane's two staging records (region pixel idx = lane, lane + 64 < 99): global offset from the item origin, LDS offset ----
    const int row0 = lane / 33, c0 = lane - 33 * row0;
    const int idx1 = lane + 64 < 99 ? lane + 64 : 0, row1 = idx1 / 33, c1 = idx1 - 33 * row1;
    const unsigned vo0 = (unsigned)((row0 * a.W + c0) * 4), vo1 = (unsigned)((row1 * a.W + c1) * 4);
    const int so0 = row0 * R_RS + (c0 & 1) * R_PS + (c0 >> 1) * 4, so1 = row1 * R_RS + (c1 & 1) * R_PS + (c1 >> 1) * 4;
    const int segs = (a.Wo + 15) >> 4;
    const int nwaves = gridDim.x * R_NW;
    // item = (frame n, output row oy, segment sx), walked incrementally: the stride's (dn, doy, dsx) decomposition is added with carries
    const int per_frame = a.Ho * segs;
    const int d_n = nwaves / per_frame, d_r = nwaves - d_n * per_frame, d_oy = d_r / segs, d_sx = d_r - d_oy * segs;
    const int w0 = blockIdx.x * R_NW + wave;
    int n = w0 / per_frame, oy = (w0 - n * per_frame) / segs, sx = w0 - n * per_frame - oy * segs;
    auto advance = [&](int& n_, int& oy_, int& sx_) {
        sx_ += d_sx; oy_ += d_oy; n_ += d_n;
        if (sx_ >= segs) { sx_ -= segs; ++oy_; }
        if (oy_ >= a.Ho) { oy_ -= a.Ho; ++n_; }
    };
    auto origin = [&](int n_, int oy_, int sx_) {   // region pixel (row 0, column 0) = input (2 oy - 1, 32 sx - 1); beyond the last frame: frame 0 (unused)
        const int nn = n_ < a.n_frames ? n_ : 0;
        return a.in + (((long)nn * a.H + (2 * oy_ - 1)) * a.W + (32 * sx_ - 1)) * 4;
    };
    f32x4 xin0, xin1;   // (scalars, not an array: the array form lived in scratch)
    {
        const float* o0 = origin(n, oy, sx);
        xin0 = *reinterpret_cast<const f32x4*>(o0 + vo0);
        xin1 = *reinterpret_cast<const f32x4*>(o0 + vo1);
    }
    __syncthreads();   // the 4x4x1 table is staged (the only workgroup-wide step)
    int n2 = n, oy2 = oy, sx2 = sx;
    advance(n2, oy2, sx2);
    // per-lane LDS bases: the tap reads are immediates on top of them
    const float* const xb = SL + p * 4;        // the 16-byte record of pixel p
    const float* const xjb = SL + p * 4 + j;   // its channel j (conv1_8's B operand)
    const float* const wqb = WQ + lane * 4;

    for (; n < a.n_frames;) {
        // the item's region -> the wave's slice (its loads were requested one iteration ago); the next item's loads go out
        *reinterpret_cast<f32x4*>(SL + so0) = xin0;
        if (lane < 35) *reinterpret_cast<f32x4*>(SL + so1) = xin1;
        {
            const float* o2 = origin(n2, oy2, sx2);
            xin0 = *reinterpret_cast<const f32x4*>(o2 + vo0);
            xin1 = *reinterpret_cast<const f32x4*>(o2 + vo1);
        }
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 accq0 = f32x4{0.f, 0.f, 0.f, 0.f}, accq1 = f32x4{0.f, 0.f, 0.f, 0.f};   // channels 16 + 4 cg + i of pixel p, this lane group's k-values
        // Two copies of the nine taps: items that touch the top row or the left column (a fifth of them) select zeros into the padded
        // taps' operands; the others run without those selects.  (One copy with per-tap wave-uniform branches: accumulators in scratch.)
        auto taps = [&](auto border) {
            constexpr bool BORDER = decltype(border)::value;
            const bool top = oy == 0, left = sx == 0;
            // conv1_8 of tap t's pixel: channels 4j .. 4j+3 on the matrix pipe (bias = C operand), 16 + 2j, 17 + 2j on the VALU
            auto c8 = [&](int t, f32x4& d, f32x2& h) {
                const int ky = t / 3, kx = t - 3 * ky;
                const int off = ky * R_RS + (kx == 1 ? R_PS : 0) + (kx == 2 ? 4 : 0);
                const float xj = xjb[off];
                const f32x4 x4 = *reinterpret_cast<const f32x4*>(xb + off);
                d = __builtin_amdgcn_mfma_f32_16x16x4f32(w8a, xj, f32x4{bias8[0], bias8[1], bias8[2], bias8[3]}, 0, 0, 0);
                h = b8h;
                h = __builtin_elementwise_fma(f32x2{x4[0], x4[0]}, w8h[0], h);
                h = __builtin_elementwise_fma(f32x2{x4[1], x4[1]}, w8h[1], h);
                h = __builtin_elementwise_fma(f32x2{x4[2], x4[2]}, w8h[2], h);
                h = __builtin_elementwise_fma(f32x2{x4[3], x4[3]}, w8h[3], h);
            };
            f32x4 d;
            f32x2 h;
            c8(0, d, h);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int ky = t / 3, kx = t - 3 * ky;
                const f32x4 wq0 = *reinterpret_cast<const f32x4*>(wqb + (t * 3 + 0) * 256);
                const f32x4 wq1 = *reinterpret_cast<const f32x4*>(wqb + (t * 3 + 1) * 256);
                const f32x4 wq2 = *reinterpret_cast<const f32x4*>(wqb + (t * 3 + 2) * 256);
                f32x4 wa;
                f32x2 wb;
                if constexpr (WLDS) {
                    wa = *reinterpret_cast<const f32x4*>(WAL + lane * 4 + t * 256);
                    wb = *reinterpret_cast<const f32x2*>(WAL + R_WA + lane * 2 + t * 128);
                } else {
                    wa = f32x4{wA[t][0], wA[t][1], wA[t][2], wA[t][3]};
                    wb = f32x2{wB[t][0], wB[t][1]};
                }
                float b[6];
#pragma unroll
                for (int r = 0; r < 4; ++r) b[r] = __int_as_float(max(__float_as_int(d[r]), 0));   // ReLU as one v_max_i32
                b[4] = __int_as_float(max(__float_as_int(h[0]), 0));
                b[5] = __int_as_float(max(__float_as_int(h[1]), 0));
                if constexpr (BORDER) {
                    if (ky == 0 || kx == 0) {   // input row -1 (all lanes of a top item) / column -1 (lane p = 0 of a left item): conv1_9's zero padding
                        const bool z = (ky == 0 && top) || (kx == 0 && left && p == 0);
#pragma unroll
                        for (int r = 0; r < 6; ++r) b[r] = z ? 0.f : b[r];
                    }
                }
#if YF_K19R_PIPE
                if (t + 1 < 9) c8(t + 1, d, h);   // the next tap's conv1_8 goes out in front of this tap's k-steps: its result is ready when they are done
#endif
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if (!(YF_K19R_DBG & 4)) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s], b[s], acc, 0, 0, 0);
                    if (!(YF_K19R_DBG & 1)) {
                        accq0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wq0[s], b[s], accq0, 0, 0, 0);
                        accq1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wq1[s], b[s], accq1, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    if (!(YF_K19R_DBG & 4)) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[s], b[4 + s], acc, 0, 0, 0);
                    if (!(YF_K19R_DBG & 1)) {
                        accq0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wq2[s], b[4 + s], accq0, 0, 0, 0);
                        accq1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wq2[2 + s], b[4 + s], accq1, 0, 0, 0);
                    }
                }
                if (YF_K19R_DBG & 5) {   // (timing builds: keep the operands alive)
                    acc[0] += b[0] + b[1] + b[2] + b[3] + b[4] + b[5];
                    if (YF_K19R_DBG & 1) accq0[0] += wq0[0] + wq1[1] + wq2[2];
                }
#if YF_K19R_PIPE == 1
                __builtin_amdgcn_sched_barrier(0);
#elif YF_K19R_PIPE == 2
#else
                if (t + 1 < 9) c8(t + 1, d, h);
#endif
            }
        };
        if (oy == 0 || sx == 0) taps(std::true_type{});
        else taps(std::false_type{});
        // ---- epilogue: bias + ReLU, conv2_1 (24 -> 8) chained in registers, store (k19m_kernel's) ----
        {
            f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float tot[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v = t == 0 ? accq0[i] : accq1[i];
                    v += __shfl_xor(v, 16);
                    v += __shfl_xor(v, 32);
                    tot[i] = v;
                }
                const float mine = j == 0 ? tot[0] : j == 1 ? tot[1] : j == 2 ? tot[2] : tot[3];
                o = __builtin_amdgcn_mfma_f32_16x16x4f32(w21q[t], fmaxf(mine + biasq[t], 0.f), o, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) o = __builtin_amdgcn_mfma_f32_16x16x4f32(w21f[r], fmaxf(acc[r] + bias9[r], 0.f), o, 0, 0, 0);
            const int ox = 16 * sx + p;
            if (j < 2 && ox < a.Wo)
                *reinterpret_cast<float4*>(a.out + (((long)n * a.Ho + oy) * a.Wo + ox) * 8 + 4 * j) =
                    make_float4(o[0] + bias21[0], o[1] + bias21[1], o[2] + bias21[2], o[3] + bias21[3]);
        }
        n = n2; oy = oy2; sx = sx2;
        advance(n2, oy2, sx2);
    }
}


// ------------------------------------------------------------------------------------------------
// k19h_kernel (fp16 storage): k19r_kernel's scheme on the fp16 matrix pipe.  Per tap and 16 output pixels
//   * conv1_8 (K = 4) is two v_mfma_f32_4x4x4_16B_f16 -- 16 independent 4x4 blocks with K = 4 EXACTLY: block (j, p >> 2) = lanes
//     4b .. 4b+3 multiplies channels 4j .. 4j+3 (second instruction: 16 + 2j, 17 + 2j) by the 4 pixels of its lanes, so lane (p, j) ends up
//     with channels 4j + i (and 16 + 2j + i, i < 2) of pixel p in its result registers; A = the lane's own weight row, B = the pixel's
//     four input halves (ONE ds_read_b64 per tap), C = the bias;
//   * ReLU + RNE to fp16 (3 v_cvt_pk_f16_f32 + 3 v_pk_max_f16) makes those values the lane's f16x8 B operand of ONE K = 32 k-step per
//     M-tile of conv1_9: k = 8j + e <-> channel 4j + e (e < 4), 16 + 2j + e - 4 (e = 4, 5; e = 6, 7 are the K padding) -- six live channels in
//     every lane group, so the second result block costs one conversion and one ReLU, not two; the channel permutation lives in the
//     host-side packing of A (k19_pack_weights), for conv1_9's own channels 16..23 (rows of its second M-tile) and conv2_1's k order likewise;
//   * the epilogue (bias + ReLU -> the same k <-> channel map -> ONE K = 32 k-step of conv2_1) stays in registers.
// Against k19m_kernel<half_t> (138 us at 640x512 batch 128): no region buffers, no workgroup barrier per tile, no phase-1 LDS store traffic;
// per 16 output pixels 18 v_mfma_f32_16x16x32_f16 + 18 v_mfma_f32_4x4x4_16B_f16 instead of 14 + 9 v_mfma_f32_16x16x4_f32, and the VALU
// work of a wave (ReLU, conversions, padding selects) runs under the fp16 MFMAs of the SIMD's other waves (tools/coissue_probe.hip:
// the fp16 matrix pipe co-issues with VALU, the fp32 one does not).  conv1_8's weights are rounded to fp16 here (k19m keeps them fp32).
// Items, staging, padding and the unchecked loads: k19r_kernel's, in 8-byte pixels.
// ------------------------------------------------------------------------------------------------
namespace {
constexpr int H_PS = 72, H_RS = 2 * H_PS, H_SLICE = 3 * H_RS;   // halves: plane (17 records of 4 + pad), row and slice of a wave
constexpr int H_BT = 48;                                        // floats: the wave's bias table behind its slice ([j][bias9 x 8 | bias21 x 4])
constexpr int H_WAVE_BYTES = H_SLICE * 2 + H_BT * 4 + 64 * 16; // + conv2_1's fragment, 16 bytes per lane
constexpr int H_W9 = 9 * 2 * 64 * 4, H_W21 = 64 * 4;            // floats (an f16x8 fragment = 4 floats per lane)
constexpr int H_OFF = W9_F16 + W21_F16;                         // the k19h stream follows k19m's in the fp16 blob
}  // namespace
#ifndef YF_K19H_DBG
#define YF_K19H_DBG 0   // timing builds only: 1 = no K = 32 k-steps, 2 = no 4x4x4 MFMAs, 4 = no ReLU / conversion, 8 = no global loads in the loop
#endif
#ifndef YF_K19H_C8
#define YF_K19H_C8 1   // conv1_8 on 1: v_mfma_f32_4x4x4_16B_f16 | 0: v_mfma_f32_16x16x16_f16 with K padded from 4 (A/B builds)
#endif

// WPS: waves per SIMD the register budget is set for (HIP's second launch bound): 3 -> 128 VGPRs without spills; 4 -> two weight fragments
// are reloaded from scratch per item.  Measured at 640x512 batch 128 (tools/scratch/k19h_forms.sh): 4 waves per workgroup x 3 per SIMD
// 69.6 us | 4 x 4 70.5 | 8 x 4 70.5 | 4 x 2 78.6 | k19m_kernel<half_t> 138 (with four + four conversions per tap: 74 | 80 | 80 | 87, and conv1_8
// on K-padded 16x16x16 MFMAs, YF_K19H_C8=0, 80).
// Where the time goes (timing builds, YF_K19H_DBG, at 74 us): the 18 K = 32 k-steps of an item are 288 of its ~1100 SIMD cycles; the 4x4x4 MFMAs, the
// conversions and the global loads are worth 4 / 10 / 4 us.  The fp16 matrix pipe co-issues with another wave's VALU instruction only
// every 8 cycles while it is saturated (tools/coissue_probe.hip --f16: v_cvt_pk_f16_f32 / v_pk_max_f16 at 124 per 1000 cycles beside
// back-to-back 16x16x32 or 4x4x4 MFMAs, against 245 alone), and a tap has 6 such instructions per 48 MFMA cycles.
template <int R_NW, int WPS>
__global__ void __launch_bounds__(R_NW * 64, WPS) k19h_kernel(K19Args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char k19_smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = lane & 15, j = lane >> 4;
    half_t* const SL = reinterpret_cast<half_t*>(k19_smem + wave * H_WAVE_BYTES);
    float* const BT = reinterpret_cast<float*>(k19_smem + wave * H_WAVE_BYTES + H_SLICE * 2);
    const half_t* const in = reinterpret_cast<const half_t*>(a.in);
    half_t* const out = reinterpret_cast<half_t*>(a.out);

    // ---- weights in registers for the lifetime of the wave: 18 x 4 + 4 + 4 VGPRs ----
    f16x8 wA[9][2];
    {
        const f16x8* w = reinterpret_cast<const f16x8*>(a.wp + H_OFF);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) wA[t][mt] = w[(t * 2 + mt) * 64 + lane];
    }
    f32x4* const W21L = reinterpret_cast<f32x4*>(k19_smem + wave * H_WAVE_BYTES + H_SLICE * 2 + H_BT * 4) + lane;   // used once per item: LDS, not 4 VGPRs
    *W21L = reinterpret_cast<const f32x4*>(a.wp + H_OFF + H_W9)[lane];
    f16x4 w8A[2];
    f32x4 bias8[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        // result row i of lane group jj: channel 4 jj + i (first instruction) / 16 + 2 jj + i, i < 2 (second: rows 2, 3 carry no channel, their
        // weights and bias are zero and their results are not even converted) -- all four lane groups hold six live channels
        auto chan = [&](int jj, int i) { return mt == 0 ? 4 * jj + i : i < 2 ? 16 + 2 * jj + i : -1; };
#if YF_K19H_C8
        const int cout = chan(j, p & 3);               // the row this lane supplies to its 4x4 block
#pragma unroll
        for (int k = 0; k < 4; ++k) w8A[mt][k] = cout >= 0 ? (half_t)a.w8[k * 24 + cout] : (half_t)0.f;
#else
        const int cout = chan(p >> 2, p & 3);          // row p of the 16-row tile; k = 4j + i: only lane group 0 holds real k-values
#pragma unroll
        for (int k = 0; k < 4; ++k) w8A[mt][k] = cout >= 0 && j == 0 ? (half_t)a.w8[k * 24 + cout] : (half_t)0.f;
#endif
#pragma unroll
        for (int r = 0; r < 4; ++r) bias8[mt][r] = chan(j, r) >= 0 ? a.b8[chan(j, r)] : 0.f;
    }
    // conv1_9's and conv2_1's biases are needed once per item: a per-wave LDS table (12 floats per lane group) instead of 12 VGPRs
    if (lane < 48) {
        const int jj = lane / 12, e = lane - 12 * jj;
        BT[lane] = e < 4 ? a.b9[4 * jj + e] : e < 6 ? a.b9[16 + 2 * jj + e - 4] : e < 8 ? 0.f : (jj < 2 ? a.b21[4 * jj + e - 8] : 0.f);
    }
    const float* const bt = BT + 12 * j;

    // ---- the lane's two staging records (region pixel idx = lane, lane + 64 < 99), in halves ----
    const int row0 = lane / 33, c0 = lane - 33 * row0;
    const int idx1 = lane + 64 < 99 ? lane + 64 : 0, row1 = idx1 / 33, c1 = idx1 - 33 * row1;
    const unsigned vo0 = (unsigned)((row0 * a.W + c0) * 4), vo1 = (unsigned)((row1 * a.W + c1) * 4);
    const int so0 = row0 * H_RS + (c0 & 1) * H_PS + (c0 >> 1) * 4, so1 = row1 * H_RS + (c1 & 1) * H_PS + (c1 >> 1) * 4;
    const int segs = (a.Wo + 15) >> 4;
    const int nwaves = gridDim.x * R_NW;
    const int per_frame = a.Ho * segs;
    const int d_n = nwaves / per_frame, d_r = nwaves - d_n * per_frame, d_oy = d_r / segs, d_sx = d_r - d_oy * segs;
    const int w0 = blockIdx.x * R_NW + wave;
    int n = w0 / per_frame, oy = (w0 - n * per_frame) / segs, sx = w0 - n * per_frame - oy * segs;
    auto advance = [&](int& n_, int& oy_, int& sx_) {
        sx_ += d_sx; oy_ += d_oy; n_ += d_n;
        if (sx_ >= segs) { sx_ -= segs; ++oy_; }
        if (oy_ >= a.Ho) { oy_ -= a.Ho; ++n_; }
    };
    auto origin = [&](int n_, int oy_, int sx_) {   // region pixel (0, 0) = input (2 oy - 1, 32 sx - 1); beyond the last frame: frame 0 (unused)
        const int nn = n_ < a.n_frames ? n_ : 0;
        return in + (((long)nn * a.H + (2 * oy_ - 1)) * a.W + (32 * sx_ - 1)) * 4;
    };
    f16x4 xin0, xin1;
    {
        const half_t* o0 = origin(n, oy, sx);
        xin0 = *reinterpret_cast<const f16x4*>(o0 + vo0);
        xin1 = *reinterpret_cast<const f16x4*>(o0 + vo1);
    }
    int n2 = n, oy2 = oy, sx2 = sx;
    advance(n2, oy2, sx2);
    const half_t* const xb = SL + p * 4;   // the 8-byte record of pixel p: the tap reads are immediates on top of it
    const f16x4 zero4 = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
    const f16x2 zero2 = {(half_t)0.f, (half_t)0.f};

    for (; n < a.n_frames;) {
        *reinterpret_cast<f16x4*>(SL + so0) = xin0;
        if (lane < 35) *reinterpret_cast<f16x4*>(SL + so1) = xin1;
        if (!(YF_K19H_DBG & 8)) {
            const half_t* o2 = origin(n2, oy2, sx2);
            xin0 = *reinterpret_cast<const f16x4*>(o2 + vo0);
            xin1 = *reinterpret_cast<const f16x4*>(o2 + vo1);
        }
        f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
        auto taps = [&](auto border) {
            constexpr bool BORDER = decltype(border)::value;
            const bool top = oy == 0, left = sx == 0;
            auto c8 = [&](int t, f32x4& d0, f32x4& d1) {
                const int ky = t / 3, kx = t - 3 * ky;
                const f16x4 x = *reinterpret_cast<const f16x4*>(xb + ky * H_RS + (kx == 1 ? H_PS : 0) + (kx == 2 ? 4 : 0));
                const f32x4 c0 = bias8[0], c1 = bias8[1];
#if YF_K19H_DBG & 2
                d0 = c0 + __builtin_convertvector(x, f32x4); d1 = c1;
#elif YF_K19H_C8
                d0 = __builtin_amdgcn_mfma_f32_4x4x4f16(w8A[0], x, c0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_4x4x4f16(w8A[1], x, c1, 0, 0, 0);
#else
                d0 = __builtin_amdgcn_mfma_f32_16x16x16f16(w8A[0], x, c0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x16f16(w8A[1], x, c1, 0, 0, 0);
#endif
            };
            f32x4 d0, d1;
            c8(0, d0, d1);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int ky = t / 3, kx = t - 3 * ky;
#if YF_K19H_DBG & 4
                f16x4 b0 = *reinterpret_cast<const f16x4*>(&d0);
                f16x2 b1 = *reinterpret_cast<const f16x2*>(&d1);
#else
                f16x4 b0 = __builtin_elementwise_max(__builtin_convertvector(d0, f16x4), zero4);
                f16x2 b1 = __builtin_elementwise_max(__builtin_convertvector(f32x2{d1[0], d1[1]}, f16x2), zero2);
#endif
                if constexpr (BORDER) {
                    if (ky == 0 || kx == 0) {   // input row -1 / column -1: conv1_9's zero padding of conv1_8's output
                        const bool z = (ky == 0 && top) || (kx == 0 && left && p == 0);
                        b0 = z ? zero4 : b0;
                        b1 = z ? zero2 : b1;
                    }
                }
                if (t + 1 < 9) c8(t + 1, d0, d1);   // the next tap's conv1_8 goes out in front of this tap's k-steps
                const f16x8 b = f16x8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], (half_t)0.f, (half_t)0.f};
#if YF_K19H_DBG & 1
                acc0 += __builtin_convertvector(b0, f32x4) * wA[t][0][0]; acc1[0] += (float)b1[0] * wA[t][1][0]; acc1[1] += (float)b1[1] * wA[t][1][1];
#else
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wA[t][0], b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wA[t][1], b, acc1, 0, 0, 0);
#endif
            }
        };
        if (oy == 0 || sx == 0) taps(std::true_type{});
        else taps(std::false_type{});
        {
            const f32x4 bias9a = *reinterpret_cast<const f32x4*>(bt), bias9b = *reinterpret_cast<const f32x4*>(bt + 4), bias21 = *reinterpret_cast<const f32x4*>(bt + 8);
            const f16x4 h0 = __builtin_elementwise_max(__builtin_convertvector(acc0 + bias9a, f16x4), zero4);
            const f16x2 h1 = __builtin_elementwise_max(__builtin_convertvector(f32x2{acc1[0] + bias9b[0], acc1[1] + bias9b[1]}, f16x2), zero2);
            const f16x8 h = f16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], (half_t)0.f, (half_t)0.f};
            const f32x4 w21v = *W21L;
            const f32x4 o = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const f16x8*>(&w21v), h, bias21, 0, 0, 0);
            const int ox = 16 * sx + p;
            if (j < 2 && ox < a.Wo)
                *reinterpret_cast<f16x4*>(out + (((long)n * a.Ho + oy) * a.Wo + ox) * 8 + 4 * j) = __builtin_convertvector(o, f16x4);
        }
        n = n2; oy = oy2; sx = sx2;
        advance(n2, oy2, sx2);
    }
}

// ------------------------------------------------------------------------------------------------
// k19x_kernel (round 6; DT_F16X3: fp32 storage, split-operand fp16 MFMAs): the buffer-free scheme of k19r_kernel / k19h_kernel for the engine
// that meets BASELINE configs[2]'s tolerance.  BUILT, MEASURED, A TIE with k19m_kernel<x3_t> (launch_k19m below has the numbers): not the
// default; YF_K19X=43 selects it, tests/test_gpu_parity.py::test_f16x3_both_forms_of_the_stride2_block runs both.  Per tap and 16 output pixels
//   * conv1_8 stays EXACT fp32 exactly as in k19r_kernel: one v_mfma_f32_16x16x4_f32 (K = 4, bias as the C operand) leaves channels
//     4j .. 4j+3 of pixel p in lane (p, j), four packed FMAs give channels 16 + 2j, 17 + 2j -- six live channels per lane group, the k <-> channel
//     map of k19h_kernel's K = 32 fragments;
//   * ReLU, then ONE split of the six values into fp16 hi = rne(v) and lo = rne(v - hi) (3 + 3 conversions, 6 subtractions) -- per tap, in
//     registers: k19m_kernel<x3_t> split once per REGION pixel but paid for it with two 55 KB region buffers, a workgroup barrier per tile and
//     the phase-1 store traffic (203 us at 640x512 batch 128, VALU 0.44 beside matrix pipe 0.37);
//   * conv1_9: per M-tile w_lo x_hi + w_hi x_lo + w_hi x_hi on v_mfma_f32_16x16x32_f16 (six per tap) into fp32 accumulators; the hi fragments
//     live in registers (72 VGPRs), the lo fragments come from LDS (18 KB per workgroup, two ds_read_b128 per tap);
//   * epilogue: bias + ReLU + split -> conv2_1 as three K = 32 MFMAs, fp32 store.
// Items, staging, padding and the unchecked loads (guard band) are k19r_kernel's.  Weight stream (k19_pack_weights, behind k19m's at X_OFF):
// [W9 hi | W9 lo | W21 hi | W21 lo] in k19h_kernel's fragment layout.
// ------------------------------------------------------------------------------------------------
namespace {
constexpr int X_OFF = 2 * WX3_HALF;
constexpr int X_LDSW = H_W9 + 2 * H_W21;           // floats staged in LDS: W9 lo | W21 hi | W21 lo
constexpr int X_BT = 48;                           // floats: [j][bias9 x 6, 2 unused | bias21 x 4]
}  // namespace
template <int R_NW, int WPS>
__global__ void __launch_bounds__(R_NW * 64, WPS) k19x_kernel(K19Args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char k19_smem[];
    float* const WL = reinterpret_cast<float*>(k19_smem);           // [9][2][64] f16x8 lo fragments | [64] W21 hi | [64] W21 lo
    float* const BT = WL + X_LDSW;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = lane & 15, j = lane >> 4;
    float* const SL = BT + X_BT + wave * R_SLICE;

    stage_to_lds<X_LDSW, R_NW * 64>(WL, a.wp + X_OFF + H_W9);
    if (threadIdx.x < 48) {
        const int jj = threadIdx.x / 12, e = threadIdx.x - 12 * jj;
        BT[threadIdx.x] = e < 4 ? a.b9[4 * jj + e] : e < 6 ? a.b9[16 + 2 * jj + e - 4] : e < 8 ? 0.f : (jj < 2 ? a.b21[4 * jj + e - 8] : 0.f);
    }
    // ---- conv1_9's hi fragments in registers for the lifetime of the wave ----
    f16x8 wA[9][2];
    {
        const f16x8* w = reinterpret_cast<const f16x8*>(a.wp + X_OFF);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) wA[t][mt] = w[(t * 2 + mt) * 64 + lane];
    }
    float bias8[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bias8[r] = a.b8[4 * j + r];
    const float w8a = a.w8[j * 24 + p];                     // conv1_8's A operand, channels 0..15: row = cout p, k = cin j
    f32x2 w8h[4], b8h;                                      // channels 16 + 2j, 17 + 2j on the VALU
#pragma unroll
    for (int c = 0; c < 4; ++c) w8h[c] = f32x2{a.w8[c * 24 + 16 + 2 * j], a.w8[c * 24 + 17 + 2 * j]};
    b8h = f32x2{a.b8[16 + 2 * j], a.b8[17 + 2 * j]};

    // ---- the lane's two staging records, item walk: k19r_kernel's ----
    const int row0 = lane / 33, c0 = lane - 33 * row0;
    const int idx1 = lane + 64 < 99 ? lane + 64 : 0, row1 = idx1 / 33, c1 = idx1 - 33 * row1;
    const unsigned vo0 = (unsigned)((row0 * a.W + c0) * 4), vo1 = (unsigned)((row1 * a.W + c1) * 4);
    const int so0 = row0 * R_RS + (c0 & 1) * R_PS + (c0 >> 1) * 4, so1 = row1 * R_RS + (c1 & 1) * R_PS + (c1 >> 1) * 4;
    const int segs = (a.Wo + 15) >> 4;
    const int nwaves = gridDim.x * R_NW;
    const int per_frame = a.Ho * segs;
    const int d_n = nwaves / per_frame, d_r = nwaves - d_n * per_frame, d_oy = d_r / segs, d_sx = d_r - d_oy * segs;
    const int w0 = blockIdx.x * R_NW + wave;
    int n = w0 / per_frame, oy = (w0 - n * per_frame) / segs, sx = w0 - n * per_frame - oy * segs;
    auto advance = [&](int& n_, int& oy_, int& sx_) {
        sx_ += d_sx; oy_ += d_oy; n_ += d_n;
        if (sx_ >= segs) { sx_ -= segs; ++oy_; }
        if (oy_ >= a.Ho) { oy_ -= a.Ho; ++n_; }
    };
    auto origin = [&](int n_, int oy_, int sx_) {
        const int nn = n_ < a.n_frames ? n_ : 0;
        return a.in + (((long)nn * a.H + (2 * oy_ - 1)) * a.W + (32 * sx_ - 1)) * 4;
    };
    f32x4 xin0, xin1;
    {
        const float* o0 = origin(n, oy, sx);
        xin0 = *reinterpret_cast<const f32x4*>(o0 + vo0);
        xin1 = *reinterpret_cast<const f32x4*>(o0 + vo1);
    }
    __syncthreads();   // the lo fragments and the bias table are staged (the only workgroup-wide step)
    int n2 = n, oy2 = oy, sx2 = sx;
    advance(n2, oy2, sx2);
    const float* const xb = SL + p * 4;        // the 16-byte record of pixel p
    const float* const xjb = SL + p * 4 + j;   // its channel j (conv1_8's B operand)
    const f16x8* const wlo = reinterpret_cast<const f16x8*>(WL) + lane;
    const float* const bt = BT + 12 * j;
    const half_t hz = (half_t)0.f;

    for (; n < a.n_frames;) {
        *reinterpret_cast<f32x4*>(SL + so0) = xin0;
        if (lane < 35) *reinterpret_cast<f32x4*>(SL + so1) = xin1;
        {
            const float* o2 = origin(n2, oy2, sx2);
            xin0 = *reinterpret_cast<const f32x4*>(o2 + vo0);
            xin1 = *reinterpret_cast<const f32x4*>(o2 + vo1);
        }
        f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
        auto taps = [&](auto border) {
            constexpr bool BORDER = decltype(border)::value;
            const bool top = oy == 0, left = sx == 0;
            auto c8 = [&](int t, f32x4& d, f32x2& h) {
                const int ky = t / 3, kx = t - 3 * ky;
                const int off = ky * R_RS + (kx == 1 ? R_PS : 0) + (kx == 2 ? 4 : 0);
                const float xj = xjb[off];
                const f32x4 x4 = *reinterpret_cast<const f32x4*>(xb + off);
                d = __builtin_amdgcn_mfma_f32_16x16x4f32(w8a, xj, f32x4{bias8[0], bias8[1], bias8[2], bias8[3]}, 0, 0, 0);
                h = b8h;
                h = __builtin_elementwise_fma(f32x2{x4[0], x4[0]}, w8h[0], h);
                h = __builtin_elementwise_fma(f32x2{x4[1], x4[1]}, w8h[1], h);
                h = __builtin_elementwise_fma(f32x2{x4[2], x4[2]}, w8h[2], h);
                h = __builtin_elementwise_fma(f32x2{x4[3], x4[3]}, w8h[3], h);
            };
            f32x4 d;
            f32x2 h;
            c8(0, d, h);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int ky = t / 3, kx = t - 3 * ky;
                float b[6];
#pragma unroll
                for (int r = 0; r < 4; ++r) b[r] = __int_as_float(max(__float_as_int(d[r]), 0));   // ReLU as one v_max_i32
                b[4] = __int_as_float(max(__float_as_int(h[0]), 0));
                b[5] = __int_as_float(max(__float_as_int(h[1]), 0));
                if constexpr (BORDER) {
                    if (ky == 0 || kx == 0) {   // input row -1 / column -1: conv1_9's zero padding of conv1_8's output
                        const bool z = (ky == 0 && top) || (kx == 0 && left && p == 0);
#pragma unroll
                        for (int r = 0; r < 6; ++r) b[r] = z ? 0.f : b[r];
                    }
                }
                f16x4 h4, l4;
                f16x2 h2, l2;
                split_f16x4(b[0], b[1], b[2], b[3], h4, l4);
                split_f16x2(b[4], b[5], h2, l2);
                const f16x8 bh = f16x8{h4[0], h4[1], h4[2], h4[3], h2[0], h2[1], hz, hz};
                const f16x8 bl = f16x8{l4[0], l4[1], l4[2], l4[3], l2[0], l2[1], hz, hz};
                const f16x8 wl0 = wlo[(t * 2 + 0) * 64], wl1 = wlo[(t * 2 + 1) * 64];
                if (t + 1 < 9) c8(t + 1, d, h);   // the next tap's conv1_8 goes out in front of this tap's k-steps
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl0, bh, acc0, 0, 0, 0);       // small terms first, then hi * hi
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl1, bh, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wA[t][0], bl, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wA[t][1], bl, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wA[t][0], bh, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wA[t][1], bh, acc1, 0, 0, 0);
            }
        };
        if (oy == 0 || sx == 0) taps(std::true_type{});
        else taps(std::false_type{});
        {   // bias + ReLU + split -> conv2_1 (24 -> 8) on three K = 32 MFMAs, fp32 store
            const f32x4 bias9a = *reinterpret_cast<const f32x4*>(bt), bias9b = *reinterpret_cast<const f32x4*>(bt + 4), bias21 = *reinterpret_cast<const f32x4*>(bt + 8);
            f16x4 h4, l4;
            f16x2 h2, l2;
            split_f16x4(fmaxf(acc0[0] + bias9a[0], 0.f), fmaxf(acc0[1] + bias9a[1], 0.f), fmaxf(acc0[2] + bias9a[2], 0.f), fmaxf(acc0[3] + bias9a[3], 0.f), h4, l4);
            split_f16x2(fmaxf(acc1[0] + bias9b[0], 0.f), fmaxf(acc1[1] + bias9b[1], 0.f), h2, l2);
            const f16x8 hh = f16x8{h4[0], h4[1], h4[2], h4[3], h2[0], h2[1], hz, hz};
            const f16x8 hl = f16x8{l4[0], l4[1], l4[2], l4[3], l2[0], l2[1], hz, hz};
            const f16x8 w21h = wlo[18 * 64], w21l = wlo[19 * 64];
            f32x4 o = bias21;
            o = __builtin_amdgcn_mfma_f32_16x16x32_f16(w21l, hh, o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_16x16x32_f16(w21h, hl, o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_16x16x32_f16(w21h, hh, o, 0, 0, 0);
            const int ox = 16 * sx + p;
            if (j < 2 && ox < a.Wo)
                *reinterpret_cast<float4*>(a.out + (((long)n * a.Ho + oy) * a.Wo + ox) * 8 + 4 * j) = make_float4(o[0], o[1], o[2], o[3]);
        }
        n = n2; oy = oy2; sx = sx2;
        advance(n2, oy2, sx2);
    }
}

size_t k19_packed_floats(int wmode)
{
    return wmode == WM_F16X3 ? (size_t)(X_OFF + 2 * (H_W9 + H_W21)) : wmode == WM_F16 ? (size_t)(H_OFF + H_W9 + H_W21) : (size_t)(R_OFF + 2 * (R_WA + R_WB) + R_WQ);
}

// w9: [tap][cin][cout] (blob layout of the dense 3x3), w21: [cin][cout]
void k19_pack_weights(const float* w9, const float* w21, float* out, int wmode)
{
    const bool h16 = wmode != WM_F32, x3 = wmode == WM_F16X3;
    uint16_t* oh = reinterpret_cast<uint16_t*>(out);
    uint16_t* ol = reinterpret_cast<uint16_t*>(out + WX3_HALF);   // x3: the lo halves, same layout
    if (h16) {   // K = 32 fragments: lane (cout, jj) holds k = 8 jj + e = chunk 2 (4 g + jj) + (e >> 2), channel e & 3
        for (int g = 0; g < NG2; ++g)
            for (int mt = 0; mt < 2; ++mt)
                for (int l = 0; l < 64; ++l)
                    for (int e = 0; e < 8; ++e) {
                        const int cout = 16 * mt + (l & 15), jj = l >> 4, fp = 4 * g + jj, fc = 2 * fp + (e >> 2);
                        const int tap = fc / 6, c = (fc % 6) * 4 + (e & 3);
                        const float v = (fp < NPAIR && cout < 24) ? w9[((size_t)tap * 24 + c) * 24 + cout] : 0.f;
                        oh[((size_t)(g * 2 + mt) * 64 + l) * 8 + e] = f32_to_f16_bits(v);
                        if (x3) ol[((size_t)(g * 2 + mt) * 64 + l) * 8 + e] = f16_lo_bits(v);
                    }
    }
    if (h16) {   // k19h_kernel / k19x_kernel: per tap ONE K = 32 fragment per M-tile (x3: [W9 hi | W9 lo | W21 hi | W21 lo] behind k19m's stream)
        uint16_t* o9 = reinterpret_cast<uint16_t*>(out + (x3 ? X_OFF : H_OFF));
        uint16_t* o21 = reinterpret_cast<uint16_t*>(out + (x3 ? X_OFF + 2 * H_W9 : H_OFF + H_W9));
        uint16_t* o9l = reinterpret_cast<uint16_t*>(out + X_OFF + H_W9);
        uint16_t* o21l = reinterpret_cast<uint16_t*>(out + X_OFF + 2 * H_W9 + H_W21);
        for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 8; ++e) {
                // k-value e of lane group jj = row i of conv1_8's two result blocks: channels 4 jj + e (e < 4), 16 + 2 jj + e - 4 (e = 4, 5), none (6, 7);
                // the second M-tile's rows carry conv1_9's channels 16..23 the same way (row 4 g + i: channel 16 + 2 g + i, i < 2)
                const int m = l & 15, jj = l >> 4, ch = e < 4 ? 4 * jj + e : e < 6 ? 16 + 2 * jj + e - 4 : -1;
                for (int tap = 0; tap < 9; ++tap)
                    for (int mt = 0; mt < 2; ++mt) {
                        const int cout = mt == 0 ? m : (m & 3) < 2 ? 16 + 2 * (m >> 2) + (m & 3) : -1;
                        const float v = ch >= 0 && cout >= 0 ? w9[((size_t)tap * 24 + ch) * 24 + cout] : 0.f;
                        o9[((size_t)(tap * 2 + mt) * 64 + l) * 8 + e] = f32_to_f16_bits(v);
                        if (x3) o9l[((size_t)(tap * 2 + mt) * 64 + l) * 8 + e] = f16_lo_bits(v);
                    }
                const float v21 = ch >= 0 && m < 8 ? w21[ch * 8 + m] : 0.f;
                o21[(size_t)l * 8 + e] = f32_to_f16_bits(v21);
                if (x3) o21l[(size_t)l * 8 + e] = f16_lo_bits(v21);
            }
    }
    for (int g = 0; g < (h16 ? 0 : NG); ++g)
        for (int s = 0; s < 4; ++s)
            for (int mt = 0; mt < 2; ++mt)
                for (int l = 0; l < 64; ++l) {
                    const int cout = 16 * mt + (l & 15), jj = l >> 4, fc = 4 * g + jj;
                    const int tap = fc / 6, c = (fc % 6) * 4 + s;
                    const float v = (fc < NCHUNK && cout < 24) ? w9[((size_t)tap * 24 + c) * 24 + cout] : 0.f;
                    out[((g * 4 + s) * 2 + mt) * 64 + l] = v;
                }
    for (int mt = 0; mt < 2; ++mt)
        for (int r = 0; r < 4; ++r)
            for (int l = 0; l < 64; ++l) {
                const int c2 = l & 15, jj = l >> 4, c1 = 16 * mt + 4 * jj + r;
                const float v = (c2 < 8 && c1 < 24) ? w21[c1 * 8 + c2] : 0.f;
                if (h16) oh[(size_t)W9_F16 * 2 + ((size_t)mt * 64 + l) * 4 + r] = f32_to_f16_bits(v);
                else out[W9_F32 + (mt * 4 + r) * 64 + l] = v;
                if (x3) ol[(size_t)W9_F16 * 2 + ((size_t)mt * 64 + l) * 4 + r] = f16_lo_bits(v);
            }
    if (!h16) {   // the 4x4-block form of channels 16..23 (k19m_kernel, Q4)
        float* wq = out + W9_F32 + W21_F32;
        for (int g = 0; g < NG; ++g)
            for (int cg = 0; cg < 2; ++cg)
                for (int l = 0; l < 64; ++l)
                    for (int s = 0; s < 4; ++s) {
                        const int jj = l >> 4, fc = 4 * g + jj, tap = fc / 6, c = (fc % 6) * 4 + s, cout = 16 + 4 * cg + (l & 3);
                        wq[((g * 2 + cg) * 64 + l) * 4 + s] = fc < NCHUNK ? w9[((size_t)tap * 24 + c) * 24 + cout] : 0.f;
                    }
        float* w21q = wq + WQ_F32;
        for (int t = 0; t < 2; ++t)
            for (int l = 0; l < 64; ++l) {
                const int c2 = l & 15, c1 = 16 + 4 * t + (l >> 4);
                w21q[t * 64 + l] = c2 < 8 ? w21[c1 * 8 + c2] : 0.f;
            }
        // k19r_kernel: per tap, k-step s of lane group jj is input channel 4 jj + s (channels 0..15) resp. 16 + 2 jj + s (16..23)
        float* wa = out + R_OFF;
        float* wb = wa + R_WA;
        float* wr = wb + R_WB;
        for (int tap = 0; tap < 9; ++tap)
            for (int l = 0; l < 64; ++l) {
                const int m = l & 15, jj = l >> 4;
                for (int s = 0; s < 4; ++s) wa[(tap * 4 + s) * 64 + l] = w9[((size_t)tap * 24 + 4 * jj + s) * 24 + m];
                for (int s = 0; s < 2; ++s) wb[(tap * 2 + s) * 64 + l] = w9[((size_t)tap * 24 + 16 + 2 * jj + s) * 24 + m];
                // the 4x4x1 table: [tap][0: cg 0, s 0..3 | 1: cg 1, s 0..3 | 2: (cg 0, s 4..5), (cg 1, s 4..5)][lane][4]
                for (int cg = 0; cg < 2; ++cg) {
                    const int cout = 16 + 4 * cg + (l & 3);
                    for (int s = 0; s < 4; ++s) wr[((tap * 3 + cg) * 64 + l) * 4 + s] = w9[((size_t)tap * 24 + 4 * jj + s) * 24 + cout];
                    for (int s = 0; s < 2; ++s) wr[((tap * 3 + 2) * 64 + l) * 4 + 2 * cg + s] = w9[((size_t)tap * 24 + 16 + 2 * jj + s) * 24 + cout];
                }
                // lane-major copies of the A fragments (the 16-wave form reads them from LDS: one b128 + one b64 per tap), behind the 4x4x1 table
                for (int s = 0; s < 4; ++s) wr[R_WQ + (tap * 64 + l) * 4 + s] = wa[(tap * 4 + s) * 64 + l];
                for (int s = 0; s < 2; ++s) wr[R_WQ + R_WA + (tap * 64 + l) * 2 + s] = wb[(tap * 2 + s) * 64 + l];
            }
    }
}

// Pixels of the 4-channel input that the two kernels' UNCHECKED 16-byte loads may touch before / after the tensor (W = its width in pixels).
// launch_k19m() must only be handed an input with k19m_guard_elems(W) readable elements on either side (the engine's workspace guard band,
// yf_engine.hip plan_workspace; tools/kbench.hip allocates the same): nothing else bounds these loads.
//   k19m_kernel: the 17x33 region of an 8x16 output tile starts one row and one column outside and ends up to 33 columns past a row end;
//   k19r_kernel: a wave's 3x33 region starts at input (2 oy - 1, 32 sx - 1) -> W + 1 pixels before the tensor, and its last segment
//                reaches column 32 segs - 1 + 32 <= W + 30 of the last row -> at most 31 pixels past the end.
constexpr size_t k19m_over_pixels(int W) { return (size_t)W + 34; }
constexpr size_t k19r_over_pixels(int W) { return (size_t)W + 1 > 31 ? (size_t)W + 1 : 31; }
static_assert(k19r_over_pixels(2) <= k19m_over_pixels(2) && k19r_over_pixels(4096) <= k19m_over_pixels(4096), "the guard band must cover both kernels");
size_t k19m_guard_elems(int W)
{
    const size_t px = k19m_over_pixels(W) > k19r_over_pixels(W) ? k19m_over_pixels(W) : k19r_over_pixels(W);
    return (px * 4 + 63) & ~(size_t)63;
}

size_t k19m_lds_bytes(int dtype)
{
    return (size_t)2 * RH * row_stride(dtype == DT_F16) * (dtype == DT_F16 ? 2 : 4) + (dtype != DT_F32 || !YF_K19_Q4 ? 0 : (size_t)WQ_F32 * 4);
}

template <int R_NW, bool WLDS>
static int launch_k19r_t(const K19Args& a, long items, int n_cu, int dev, hipStream_t s)
{
    static bool attr_r[YF_MAX_DEVICES] = {};
    constexpr size_t lds = (size_t)(R_WQ + (WLDS ? R_WA + R_WB : 0) + R_NW * R_SLICE) * sizeof(float);
    if (!attr_r[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k19r_kernel<R_NW, WLDS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return -1;
        attr_r[dev] = true;
    }
    const long wgs = (items + R_NW - 1) / R_NW;
    hipLaunchKernelGGL((k19r_kernel<R_NW, WLDS>), dim3((unsigned)(wgs < n_cu ? wgs : n_cu)), dim3(R_NW * 64), lds, s, a);
    return 0;
}

template <int R_NW, int WPS>
static int launch_k19h_t(const K19Args& a, long items, int n_cu, hipStream_t s)
{
    constexpr size_t lds = (size_t)R_NW * H_WAVE_BYTES;
    const long wgs = (items + R_NW - 1) / R_NW, cap = (long)n_cu * (4 * WPS / R_NW);
    hipLaunchKernelGGL((k19h_kernel<R_NW, WPS>), dim3((unsigned)(wgs < cap ? wgs : cap)), dim3(R_NW * 64), lds, s, a);
    return 0;
}

template <int R_NW, int WPS>
static int launch_k19x_t(const K19Args& a, long items, int n_cu, int dev, hipStream_t s)
{
    static bool attr_x[YF_MAX_DEVICES] = {};
    constexpr size_t lds = (size_t)(X_LDSW + X_BT + R_NW * R_SLICE) * sizeof(float);
    if (lds > 64 * 1024 && !attr_x[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k19x_kernel<R_NW, WPS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return -1;
        attr_x[dev] = true;
    }
    const long wgs = (items + R_NW - 1) / R_NW, cap = (long)n_cu * (4 * WPS / R_NW > 0 ? 4 * WPS / R_NW : 1);
    hipLaunchKernelGGL((k19x_kernel<R_NW, WPS>), dim3((unsigned)(wgs < cap ? wgs : cap)), dim3(R_NW * 64), lds, s, a);
    return 0;
}

// fp32: k19r_kernel unless YF_K19R=0 (A/B: the region-buffer kernel of rounds 1-3)
static bool k19r_enabled()
{
    static const bool on = [] { const char* v = getenv("YF_K19R"); return !(v && v[0] == '0'); }();
    return on;
}

int launch_k19m(K19Args a, int N, hipStream_t s, int dtype)
{
    const int dev = current_device(), n_cu = device_cu_count(dev);
    if (n_cu <= 0) return -1;
    static bool attr_done[YF_MAX_DEVICES] = {};
    if (!attr_done[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k19m_kernel<float, 0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)k19m_lds_bytes(DT_F32)) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k19m_kernel<half_t, 0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)k19m_lds_bytes(DT_F16)) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k19m_kernel<x3_t, 0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)k19m_lds_bytes(DT_F16X3)) != hipSuccess)
            return -1;
        attr_done[dev] = true;
    }
    a.n_frames = N;
    if (dtype == DT_F32 && k19r_enabled() && a.H % 2 == 0 && a.W % 2 == 0) {
        // developer switch: 16 = weights from LDS (default: 156 us); A/B: 8 (166 us) | 12 (160 us) | 1012 = 12 waves, weights from LDS (162 us);
        // anything else is refused (it used to select the 12-wave form silently)
        static const int nw = [] {
            const char* v = getenv("YF_K19R_NW");
            if (!v || !*v) return 16;
            const int k = atoi(v);
            if (k != 8 && k != 12 && k != 16 && k != 1012) { fprintf(stderr, "yolo_fastest_hip: YF_K19R_NW=%s is not one of 8, 12, 16, 1012\n", v); return -1; }
            return k;
        }();
        if (nw < 0) return -1;
        const long items = (long)N * a.Ho * ((a.Wo + 15) / 16);
        // small batches: the 8-wave form (conv1_9's fragments in registers, a third of the LDS prologue) -- 11.6 -> 8.3 us at batch 1, 12.4 -> 9.0 at
        // 4, 20.8 -> 18.4 at 16 (tools/small_batch_ops.py); unless the developer switch asks for a form
        if (!getenv("YF_K19R_NW") && items <= 64L * n_cu) return launch_k19r_t<8, false>(a, items, n_cu, dev, s);
        return nw == 8 ? launch_k19r_t<8, false>(a, items, n_cu, dev, s) : nw == 16 ? launch_k19r_t<16, true>(a, items, n_cu, dev, s)
                       : nw == 1012 ? launch_k19r_t<12, true>(a, items, n_cu, dev, s) : launch_k19r_t<12, false>(a, items, n_cu, dev, s);
    }
    if (dtype == DT_F16 && k19r_enabled() && a.H % 2 == 0 && a.W % 2 == 0) {   // fp16 storage: k19h_kernel (YF_K19R=0: the region-buffer kernel)
        // developer switch YF_K19H_FORM = waves per workgroup * 10 + waves per SIMD (A/B; default 43)
        static const int form = [] { const char* v = getenv("YF_K19H_FORM"); return v && *v ? atoi(v) : 43; }();
        const long items = (long)N * a.Ho * ((a.Wo + 15) / 16);
        switch (form) {
        case 43: return launch_k19h_t<4, 3>(a, items, n_cu, s);
        case 44: return launch_k19h_t<4, 4>(a, items, n_cu, s);
        case 42: return launch_k19h_t<4, 2>(a, items, n_cu, s);
        case 84: return launch_k19h_t<8, 4>(a, items, n_cu, s);
        default: fprintf(stderr, "yolo_fastest_hip: YF_K19H_FORM=%d is not one of 42, 43, 44, 84\n", form); return -1;
        }
    }
    if (dtype == DT_F16X3 && k19r_enabled() && a.H % 2 == 0 && a.W % 2 == 0) {   // split operands: k19x_kernel (YF_K19X=0 / YF_K19R=0: the region-buffer kernel)
        // Measured (tools/x3_ab.sh, 640x512 batch 128, two interleaved rounds, us): k19m_kernel<x3_t> 198-201 | k19x 4 waves x 3 per SIMD 197-198 |
        // 12 x 3 201 (with the v_fma_mix split, yf_kernels.h YF_X3_MIX: 190-192 | 190-191 | 194) -- a TIE: per 16 output pixels k19x issues ~340 VALU
        // instructions (six conversions + six subtractions + six ReLUs per tap: the per-tap split) + 9 fp32 and 57 fp16 MFMAs = ~2600 issue cycles,
        // and they do not overlap (2700 measured); the region-buffer kernel splits once per region pixel and pays for its barrier instead.
        // Default: k19m_kernel (YF_K19X unset or 0); YF_K19X=43 / 42 / 83 / 123 select a k19x form (waves per workgroup * 10 + waves per SIMD).
        static const int form = [] { const char* v = getenv("YF_K19X"); return v && *v ? atoi(v) : 0; }();
        const long items = (long)N * a.Ho * ((a.Wo + 15) / 16);
        switch (form) {
        case 0: break;
        case 43: return launch_k19x_t<4, 3>(a, items, n_cu, dev, s);
        case 42: return launch_k19x_t<4, 2>(a, items, n_cu, dev, s);
        case 83: return launch_k19x_t<8, 3>(a, items, n_cu, dev, s);
        case 123: return launch_k19x_t<12, 3>(a, items, n_cu, dev, s);
        default: fprintf(stderr, "yolo_fastest_hip: YF_K19X=%d is not one of 0, 42, 43, 83, 123\n", form); return -1;
        }
    }
    a.tiles_y = (a.Ho + TH - 1) / TH;
    a.tiles_x = (a.Wo + TW - 1) / TW;
    const long total = (long)N * a.tiles_y * a.tiles_x;
    // persistent: one workgroup (8 waves, two per SIMD) per CU
    const long want = (long)n_cu;
    const unsigned grid = (unsigned)(total < want ? total : want);
    if (dtype == DT_F16) hipLaunchKernelGGL((k19m_kernel<half_t, 0>), dim3(grid), dim3(512), k19m_lds_bytes(DT_F16), s, a);
    else if (dtype == DT_F16X3) hipLaunchKernelGGL((k19m_kernel<x3_t, 0>), dim3(grid), dim3(512), k19m_lds_bytes(DT_F16X3), s, a);
    else hipLaunchKernelGGL((k19m_kernel<float, 0>), dim3(grid), dim3(512), k19m_lds_bytes(DT_F32), s, a);
    return 0;
}

}  // namespace yf
