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

template <int GRAY, int DC>
void launch_mode(const CvArgs& a, unsigned grid, hipStream_t s)
{
    if (a.mode == 0) hipLaunchKernelGGL((cv_pre_kernel<GRAY, 0, DC>), dim3(grid), dim3(256), 0, s, a);
    else if (a.mode == 1) hipLaunchKernelGGL((cv_pre_kernel<GRAY, 1, DC>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((cv_pre_kernel<GRAY, 2, DC>), dim3(grid), dim3(256), 0, s, a);
}

}  // namespace

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
