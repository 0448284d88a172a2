// yf_cv_kernels.hip -- the two OpenCV calls of Detect_YOLO.__pre_process (src/detect.py:107-116) on the device:
//     img = cv2.cvtColor(ori_img, cv2.COLOR_BGR2GRAY)             (:110-111)
//     img = cv2.resize(img, (input_shape[1], input_shape[0]))     (:115-116, INTER_LINEAR; exactly 1/2 -> INTER_AREA's 2x2 mean)
// for ANY source size, as ONE pass over the source frames: u8 [N, sh, sw(, 3)] -> u8 [N, H, W(, C)] (what the stem kernel's fused
// `(v - 128) / 255` entry then reads: yf_forward_u8).  OpenCV's 8-bit fixed-point arithmetic is restated (oracle/cv_oracle.py holds the
// same restatement on the CPU and this kernel is held to it bit for bit; parity with a real OpenCV build is UNPINNED: no cv2 here):
//   gray   = (B * BY + G * GY + R * RY + (1 << (shift - 1))) >> shift            (RY, GY, BY, shift) = (4899, 9617, 1868, 14) | (9798, 19235, 3735, 15)
//   linear : per destination column {sx, sx + 1 (clamped), a0, a1}, per row {y0, y1, b0, b1} -- 11-bit coefficients from cv::resize's own
//            float arithmetic, built on the HOST exactly as OpenCV builds its xofs / ialpha / yofs / ibeta tables (yf_engine.hip cv_tables) --
//            rows r = S[sx] * a0 + S[sx + 1] * a1 (int32), dst = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2
//   area2  : (a + b + c + d + 2) >> 2 over the 2x2 block (cv::resize turns INTER_LINEAR into INTER_AREA at exactly 1/2)
// HBM-bound byte work (983 KB in, 82 KB out per 640x512 BGR frame): no LDS staging -- a wave covers 64 consecutive destination pixels of
// a row, its taps are two short runs of consecutive source bytes per source row (hit in L1/L2 after the first touch), every source byte
// crosses HBM once, and four destination pixels leave as one 4-byte store where the row is 4-aligned.
#include "yf_kernels.h"
#include <stdlib.h>

namespace yf {

namespace {

template <int GRAY>   // 0: plain channel c; 14 / 15: BGR -> gray on the fly
__device__ __forceinline__ int cv_px(const uint8_t* __restrict__ row, int x, int sc, int c)
{
    if constexpr (GRAY == 0) {
        return row[(long)x * sc + c];
    } else {
        const uint8_t* p = row + (long)x * 3;
        constexpr int RY = GRAY == 14 ? 4899 : 9798, GY = GRAY == 14 ? 9617 : 19235, BY = GRAY == 14 ? 1868 : 3735;
        return (p[0] * BY + p[1] * GY + p[2] * RY + (1 << (GRAY - 1))) >> GRAY;
    }
}

// one thread = one destination pixel position x 4 consecutive columns (all channels)
template <int GRAY, int MODE, int DC>   // MODE 0: same size (cvtColor only), 1: exact 1/2 (area), 2: linear; DC destination channels
__global__ void __launch_bounds__(256) cv_pre_kernel(CvArgs a)
{
    const int quads = (a.dw + 3) >> 2;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const long per_frame = (long)a.dh * quads;
    if (t >= (long)a.n * per_frame) return;
    const int n = (int)(t / per_frame);
    const int r = (int)(t - (long)n * per_frame);
    const int dy = r / quads, dx0 = (r - dy * quads) * 4;
    const uint8_t* const src = a.src + (long)n * a.sh * a.sw * a.sc;
    uint8_t* const dst = a.dst + ((long)n * a.dh + dy) * a.dw * DC;
    const long rs = (long)a.sw * a.sc;   // source row stride in bytes
    int y0 = dy, y1 = dy, b0 = 0, b1 = 0;
    if constexpr (MODE == 1) { y0 = 2 * dy; y1 = 2 * dy + 1; }
    if constexpr (MODE == 2) { const int4 ty = a.ytab[dy]; y0 = ty.x; y1 = ty.y; b0 = ty.z; b1 = ty.w; }
    const uint8_t* const row0 = src + y0 * rs;
    const uint8_t* const row1 = src + y1 * rs;
    uint32_t packed[DC];   // DC channels x 4 pixels
#pragma unroll
    for (int w = 0; w < DC; ++w) packed[w] = 0u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int dx = dx0 + i;
        if (dx >= a.dw) break;
        int x0 = dx, x1 = dx, a0 = 0, a1 = 0;
        if constexpr (MODE == 1) { x0 = 2 * dx; x1 = 2 * dx + 1; }
        if constexpr (MODE == 2) { const int4 tx = a.xtab[dx]; x0 = tx.x; x1 = tx.y; a0 = tx.z; a1 = tx.w; }
#pragma unroll
        for (int c = 0; c < DC; ++c) {
            int v;
            if constexpr (MODE == 0) {
                v = cv_px<GRAY>(row0, x0, a.sc, c);
            } else if constexpr (MODE == 1) {
                v = (cv_px<GRAY>(row0, x0, a.sc, c) + cv_px<GRAY>(row0, x1, a.sc, c) + cv_px<GRAY>(row1, x0, a.sc, c) + cv_px<GRAY>(row1, x1, a.sc, c) + 2) >> 2;
            } else {
                const int r0 = cv_px<GRAY>(row0, x0, a.sc, c) * a0 + cv_px<GRAY>(row0, x1, a.sc, c) * a1;
                const int r1 = cv_px<GRAY>(row1, x0, a.sc, c) * a0 + cv_px<GRAY>(row1, x1, a.sc, c) * a1;
                v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
            }
            const int k = i * DC + c;   // byte index inside the thread's 4 * DC output bytes
            packed[k >> 2] |= (uint32_t)(v & 255) << (8 * (k & 3));
        }
    }
    const int nb = (a.dw - dx0 < 4 ? a.dw - dx0 : 4) * DC;
    uint8_t* o = dst + (long)dx0 * DC;
    if (nb == 4 * DC && ((reinterpret_cast<uintptr_t>(o) & 3) == 0)) {
#pragma unroll
        for (int w = 0; w < DC; ++w) reinterpret_cast<uint32_t*>(o)[w] = packed[w];
    } else {
#pragma unroll
        for (int k = 0; k < 4 * DC; ++k)
            if (k < nb) o[k] = (uint8_t)(packed[k >> 2] >> (8 * (k & 3)));
    }
}

// The same arithmetic with the source rows staged through LDS (the direct kernel above gathers single bytes from global memory: 12 byte
// loads per destination pixel, 2.4 TB/s on 480x640 BGR frames).  A workgroup owns CV_TR destination rows of one frame: the 2 CV_TR source
// rows they read are fetched ONCE with coalesced 4-byte loads -- BGR pixels converted to gray on the way, so that LDS holds the image the
// resize works on (detect.py:110-116: cvtColor first, then resize) -- and the taps come from LDS.  Needs 4-byte aligned source rows
// ((sw * sc) % 4 == 0) and rows that fit the staging buffer; the launcher falls back to the direct kernel otherwise.
// A/B builds (tools/cv_ab.sh, round 6, us per 256 BGR frames of 480x640 / 128 of 720x1280): TR 4 (default) 73-76 / 70; TR 8 71 / 82; TR 2 90 / 75; loads in
// flight 3 / 12: 71.5 / 85.  A persistent, software-pipelined form (the next row group's loads requested before the current one is interpolated): 88 / 95 --
// slower (more live registers, a second barrier per group); the one-group-per-workgroup form with seven resident workgroups per CU stays.
#ifndef YF_CV_TR
#define YF_CV_TR 4
#endif
#ifndef YF_CV_UNR
#define YF_CV_UNR 6
#endif
constexpr int CV_TR = YF_CV_TR;          // destination rows per workgroup

template <int GRAY, int MODE, int DC>
__global__ void __launch_bounds__(256) cv_pre_lds_kernel(CvArgs a, int pitch)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t cv_smem[];   // [2 * CV_TR][pitch]
    const int groups = (a.dh + CV_TR - 1) / CV_TR;
    const int n = blockIdx.x / groups, dy0 = (blockIdx.x - n * groups) * CV_TR;
    const uint8_t* const src = a.src + (long)n * a.sh * a.sw * a.sc;
    const long rs = (long)a.sw * a.sc;
    // ---- stage: row slot 2 k + j = source row y_j of destination row dy0 + k; CV_UNR units per thread in flight together (one unit per
    // trip waited for every load in turn: 108 -> 71 us with the staging alone, the batched loads and the LDS copy of the column table take
    // the rest) ----
    const int upr = a.sw >> 2;                       // 4-pixel units per row (a tail of sw % 4 pixels: byte path below)
    const int total = 2 * CV_TR * upr;
    int4* const xl = reinterpret_cast<int4*>(cv_smem + 2 * CV_TR * pitch);   // the column table, staged for the CV_TR rows that share it
    if constexpr (MODE == 2)
        for (int i = threadIdx.x; i < a.dw; i += 256) xl[i] = a.xtab[i];
    constexpr int CV_UNR = YF_CV_UNR, NW = GRAY != 0 ? 3 : DC;
#pragma unroll 1
    for (int base = threadIdx.x; base < total; base += 256 * CV_UNR) {
        uint32_t w[CV_UNR][NW];
        int dsto[CV_UNR];
#pragma unroll
        for (int j = 0; j < CV_UNR; ++j) {
            const int u = base + j * 256;
            dsto[j] = -1;
            if (u < total) {
                const int slot = u / upr, x4 = (u - slot * upr) * 4;
                const int dy = dy0 + (slot >> 1);
                if (dy < a.dh) {
                    int y;
                    if constexpr (MODE == 0) y = dy;
                    else if constexpr (MODE == 1) y = 2 * dy + (slot & 1);
                    else { const int4 ty = a.ytab[dy]; y = (slot & 1) ? ty.y : ty.x; }
                    const uint32_t* p = reinterpret_cast<const uint32_t*>(src + y * rs + (long)x4 * (GRAY != 0 ? 3 : DC));
#pragma unroll
                    for (int k = 0; k < NW; ++k) w[j][k] = p[k];
                    dsto[j] = slot * pitch + x4 * DC;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < CV_UNR; ++j) {
            if (dsto[j] < 0) continue;
            uint32_t* o = reinterpret_cast<uint32_t*>(cv_smem + dsto[j]);
            if constexpr (GRAY != 0) {
                const uint32_t w0 = w[j][0], w1 = w[j][1], w2 = w[j][2];     // B G R B | G R B G | R B G R
                constexpr int RY = GRAY == 14 ? 4899 : 9798, GY = GRAY == 14 ? 9617 : 19235, BY = GRAY == 14 ? 1868 : 3735, HALF = 1 << (GRAY - 1);
                const int g0 = (int)((w0 & 255) * BY + ((w0 >> 8) & 255) * GY + ((w0 >> 16) & 255) * RY + HALF) >> GRAY;
                const int g1 = (int)((w0 >> 24) * BY + (w1 & 255) * GY + ((w1 >> 8) & 255) * RY + HALF) >> GRAY;
                const int g2 = (int)(((w1 >> 16) & 255) * BY + (w1 >> 24) * GY + (w2 & 255) * RY + HALF) >> GRAY;
                const int g3 = (int)(((w2 >> 8) & 255) * BY + ((w2 >> 16) & 255) * GY + (w2 >> 24) * RY + HALF) >> GRAY;
                o[0] = (uint32_t)g0 | ((uint32_t)g1 << 8) | ((uint32_t)g2 << 16) | ((uint32_t)g3 << 24);
            } else {
#pragma unroll
                for (int k = 0; k < DC; ++k) o[k] = w[j][k];
            }
        }
    }
    if (a.sw & 3) {   // the last sw % 4 pixels of every row, byte by byte
        const int x_lo = a.sw & ~3, tail = a.sw - x_lo;
        for (int u = threadIdx.x; u < 2 * CV_TR * tail; u += 256) {
            const int slot = u / tail, x = x_lo + (u - slot * tail);
            const int dy = dy0 + (slot >> 1);
            if (dy >= a.dh) continue;
            int y;
            if constexpr (MODE == 0) y = dy;
            else if constexpr (MODE == 1) y = 2 * dy + (slot & 1);
            else { const int4 ty = a.ytab[dy]; y = (slot & 1) ? ty.y : ty.x; }
            for (int c = 0; c < DC; ++c) cv_smem[slot * pitch + x * DC + c] = (uint8_t)cv_px<GRAY>(src + y * rs, x, a.sc, c);
        }
    }
    __syncthreads();
    // ---- interpolate from LDS: one task = 4 consecutive destination pixels of one of the CV_TR rows ----
    const int quads = (a.dw + 3) >> 2;
#pragma unroll 1
    for (int t = threadIdx.x; t < CV_TR * quads; t += 256) {
        const int k = t / quads, dx0 = (t - k * quads) * 4, dy = dy0 + k;
        if (dy >= a.dh) continue;
        const uint8_t* const row0 = cv_smem + (2 * k) * pitch;
        const uint8_t* const row1 = cv_smem + (2 * k + 1) * pitch;
        int b0 = 0, b1 = 0;
        if constexpr (MODE == 2) { const int4 ty = a.ytab[dy]; b0 = ty.z; b1 = ty.w; }
        uint32_t packed[DC];
#pragma unroll
        for (int w = 0; w < DC; ++w) packed[w] = 0u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int dx = dx0 + i;
            if (dx >= a.dw) break;
            int x0 = dx, x1 = dx, a0 = 0, a1 = 0;
            if constexpr (MODE == 1) { x0 = 2 * dx; x1 = 2 * dx + 1; }
            if constexpr (MODE == 2) { const int4 tx = xl[dx]; x0 = tx.x; x1 = tx.y; a0 = tx.z; a1 = tx.w; }
#pragma unroll
            for (int c = 0; c < DC; ++c) {
                int v;
                if constexpr (MODE == 0) {
                    v = row0[x0 * DC + c];
                } else if constexpr (MODE == 1) {
                    v = (row0[x0 * DC + c] + row0[x1 * DC + c] + row1[x0 * DC + c] + row1[x1 * DC + c] + 2) >> 2;
                } else {
                    const int r0 = row0[x0 * DC + c] * a0 + row0[x1 * DC + c] * a1;
                    const int r1 = row1[x0 * DC + c] * a0 + row1[x1 * DC + c] * a1;
                    v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
                }
                const int kk = i * DC + c;
                packed[kk >> 2] |= (uint32_t)(v & 255) << (8 * (kk & 3));
            }
        }
        uint8_t* o = a.dst + (((long)n * a.dh + dy) * a.dw + dx0) * DC;
        const int nb = (a.dw - dx0 < 4 ? a.dw - dx0 : 4) * DC;
        if (nb == 4 * DC && ((reinterpret_cast<uintptr_t>(o) & 3) == 0)) {
#pragma unroll
            for (int w = 0; w < DC; ++w) reinterpret_cast<uint32_t*>(o)[w] = packed[w];
        } else {
#pragma unroll
            for (int kk = 0; kk < 4 * DC; ++kk)
                if (kk < nb) o[kk] = (uint8_t)(packed[kk >> 2] >> (8 * (kk & 3)));
        }
    }
}

template <int GRAY, int DC>
void launch_mode(const CvArgs& a, unsigned grid, hipStream_t s)
{
    // rows staged through LDS when they are 4-byte aligned and fit (every common frame size); the byte-gather kernel otherwise
    const bool aligned = ((long)a.sw * a.sc) % 4 == 0 && (reinterpret_cast<uintptr_t>(a.src) & 3) == 0 && ((long)a.sh * a.sw * a.sc) % 4 == 0;
    static const bool lds_off = getenv("YF_CV_DIRECT") != nullptr;   // developer switch: A/B against the direct kernel
    const int pitch = ((a.sw * DC + 15) & ~15) + 16;
    const size_t lds = (size_t)2 * CV_TR * pitch + (a.mode == 2 ? (size_t)a.dw * sizeof(int4) : 0);
    if (aligned && lds <= 64 * 1024 && !lds_off) {
        const unsigned g2 = (unsigned)((long)a.n * ((a.dh + CV_TR - 1) / CV_TR));
        if (a.mode == 0) hipLaunchKernelGGL((cv_pre_lds_kernel<GRAY, 0, DC>), dim3(g2), dim3(256), lds, s, a, pitch);
        else if (a.mode == 1) hipLaunchKernelGGL((cv_pre_lds_kernel<GRAY, 1, DC>), dim3(g2), dim3(256), lds, s, a, pitch);
        else hipLaunchKernelGGL((cv_pre_lds_kernel<GRAY, 2, DC>), dim3(g2), dim3(256), lds, s, a, pitch);
        return;
    }
    if (a.mode == 0) hipLaunchKernelGGL((cv_pre_kernel<GRAY, 0, DC>), dim3(grid), dim3(256), 0, s, a);
    else if (a.mode == 1) hipLaunchKernelGGL((cv_pre_kernel<GRAY, 1, DC>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((cv_pre_kernel<GRAY, 2, DC>), dim3(grid), dim3(256), 0, s, a);
}

}  // namespace

// cv::resize(INTER_LINEAR, 8-bit): the tables OpenCV's resize() builds on the host -- xofs / ialpha with the edge resets of its horizontal
// pass, yofs / ibeta with clamped row indices -- from ITS float arithmetic (modules/imgproc/src/resize.cpp; restated in oracle/cv_oracle.py):
// `scale = 1. / inv_scale` in double, `fx = (float)((dx + 0.5) * scale - 0.5)`, cvFloor, `cvRound(f * 2048)` (half to even).  IEEE double /
// float operations, built with -ffp-contract=off: the device evaluates exactly what the host expression does.
__global__ void cv_tables_kernel(int src_h, int src_w, int dst_h, int dst_w, int4* __restrict__ xtab, int4* __restrict__ ytab)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= dst_w + dst_h) return;
    const bool isx = i < dst_w;
    const int d = isx ? i : i - dst_w, src = isx ? src_w : src_h, dst = isx ? dst_w : dst_h;
    const double inv_scale = (double)dst / (double)src, scale = 1.0 / inv_scale;
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int sidx = (int)floorf(f);
    f -= (float)sidx;
    if (isx) {   // the horizontal pass resets at the edges (xofs), the vertical pass clamps its row indices
        if (sidx < 0) { f = 0.f; sidx = 0; }
        if (sidx >= src - 1) { f = 0.f; sidx = src - 1; }
    }
    const int c1 = (int)nearbyintf(f * 2048.f), c0 = (int)nearbyintf((1.f - f) * 2048.f);
    const int i0 = sidx < 0 ? 0 : (sidx > src - 1 ? src - 1 : sidx), i1 = sidx + 1 < 0 ? 0 : (sidx + 1 > src - 1 ? src - 1 : sidx + 1);
    (isx ? xtab : ytab)[d] = make_int4(i0, i1, c0, c1);
}

void launch_cv_tables(int src_h, int src_w, int dst_h, int dst_w, int4* xtab, int4* ytab, hipStream_t s)
{
    hipLaunchKernelGGL(cv_tables_kernel, dim3((unsigned)((dst_w + dst_h + 255) / 256)), dim3(256), 0, s, src_h, src_w, dst_h, dst_w, xtab, ytab);
}

int launch_cv_pre(const CvArgs& a, hipStream_t s)
{
    if (a.n <= 0 || a.sh <= 0 || a.sw <= 0 || a.dh <= 0 || a.dw <= 0 || (a.dc != 1 && a.dc != 3) || (a.sc != 1 && a.sc != 3)) return -1;
    if (a.gray != 0 && (a.sc != 3 || a.dc != 1)) return -1;
    if (a.gray == 0 && a.sc != a.dc) return -1;
    if (a.mode == 2 && (!a.xtab || !a.ytab)) return -1;
    const long threads = (long)a.n * a.dh * ((a.dw + 3) >> 2);
    const unsigned grid = (unsigned)((threads + 255) / 256);
    if (a.mode == 1 && (a.sh != 2 * a.dh || a.sw != 2 * a.dw)) return -1;
    if (a.mode == 0 && (a.sh != a.dh || a.sw != a.dw)) return -1;
    if (a.gray == 14) launch_mode<14, 1>(a, grid, s);
    else if (a.gray == 15) launch_mode<15, 1>(a, grid, s);
    else if (a.gray == 0 && a.dc == 1) launch_mode<0, 1>(a, grid, s);
    else if (a.gray == 0) launch_mode<0, 3>(a, grid, s);
    else return -1;
    return 0;
}

}  // namespace yf
