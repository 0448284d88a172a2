// yf_train_kernels.hip -- operators of the reference's TRAINING step (second slice): forward in train mode
// and backward of every layer type of YoloFastest (src/model_training/model/yolo_fastest.py:16-66), and the optimizer update of
// src/model_training/train.py:84 (Adam), on NCHW fp32 tensors like the reference's.
//
//   conv_norm_relu / conv_norm :16-38   Conv2d(bias=False) -> BatchNorm2d(train: batch statistics) -> [ReLU]
//   deconv_norm_relu           :42-48   ConvTranspose2d(k=2, s=2) -> BN -> ReLU
//   BasicResBlock              :52-66   three units + residual add
//   heads                      :136,146 Conv2d 1x1 with bias
//
// Pointwise / dense convs are GEMMs on the fp32 matrix pipe read straight from global memory; depthwise convs keep their weights
// wave-uniform; every reduction that spans workgroups (BatchNorm sums, weight gradients) stores partials into a scratch and a second
// pass adds them in a fixed order -- no float atomics, no in-kernel fences (both measured slow on this multi-XCD part; DESIGN.md
// section 4).  The generic one-thread-per-output kernels remain as the fallback for shapes the fast paths do not take (odd widths).
// They are NOT the tuned inference kernels (those fold BN, which training cannot: batch statistics).  Everything is stream-ordered;
// no allocation, no synchronisation.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <vector>

#include "yf_kernels.h"

namespace yf {

#include "yf_train_common.h"
#include "yf_train_pw_kernels.h"
#include "yf_train_dense_kernels.h"
#include "yf_train_dw_kernels.h"
#include "yf_train_bn_kernels.h"
#include "yf_train_opt_kernels.h"

// ---- launchers: which kernel of which family a layer geometry takes ----

static void launch_tpw_gemm(const float* x, const float* a, const float* bias, const float* addend, float* y, long Q, long HW, int M, int K, long sm,
                            long sk, hipStream_t s, TStatPart* st = nullptr)
{
    if (st) st->count = 0;
    const int wm = tpw_waves_m(M), wq = 4 / wm;
    const unsigned my = (unsigned)((M + 16 * wm - 1) / (16 * wm));
    const long b4 = (Q + 64L * wq - 1) / (64L * wq), b1 = (Q + 16L * wq - 1) / (16L * wq);     // workgroups along the pixels, NT = 4 / 1
    static const bool old_only = getenv("YF_TPW_OLD") != nullptr;
    const int tiles = (M + 15) / 16;
    int mgroups = (tiles + 3) / 4, mt = (tiles + mgroups - 1) / mgroups;
    while (mt > 1 && (Q + 63) / 64 * mgroups < 512) {                  // few pixels: channel tiles on separate waves (the B re-reads hit L2)
        mt = (mt + 1) / 2;
        mgroups = (tiles + mt - 1) / mt;
    }
    const long wg = (Q + 255) / 256 * mgroups;
    // big A operand and enough pixel blocks per workgroup to pay for staging it: the weight-stationary form
    static const bool lds_off = getenv("YF_TPW_LDS_OFF") != nullptr;
    const bool want_stat = st && !bias && !addend && mt <= 2 && HW % 4 == 0 && K % 4 == 0 && Q % 4 == 0 && !old_only;
    if (!want_stat && !old_only && !lds_off && HW % 4 == 0 && K % 4 == 0 && Q % 4 == 0 && (long)M * K >= 8192 && Q >= 32768) {
        int rc = -1;
        if (mt == 1) rc = launch_tpw4_lds<1, false>(x, a, bias, addend, y, Q, HW, M, K, sm, sk, mgroups, 0, s);
        else if (mt == 2) rc = launch_tpw4_lds<2, false>(x, a, bias, addend, y, Q, HW, M, K, sm, sk, mgroups, 0, s);
        else if (mt == 3) rc = launch_tpw4_lds<3, false>(x, a, bias, addend, y, Q, HW, M, K, sm, sk, mgroups, 0, s);
        else rc = launch_tpw4_lds<4, false>(x, a, bias, addend, y, Q, HW, M, K, sm, sk, mgroups, 0, s);
        if (rc == 0) return;
    }
    if (!old_only && HW % 4 == 0 && K % 4 == 0 && Q % 4 == 0) {
        float2* sp = (want_stat && tstat_room(st, (Q + 255) / 256 * 4, M)) ? st->part : nullptr;
#define YF_PW4(MT_) hipLaunchKernelGGL(tpw4_mfma_kernel<MT_>, dim3((unsigned)wg), dim3(256), 0, s, x, a, bias, addend, y, Q, HW, M, K, sm, sk, mgroups, 0, sp)
        if (mt == 1) YF_PW4(1); else if (mt == 2) YF_PW4(2); else if (mt == 3) YF_PW4(3); else YF_PW4(4);
#undef YF_PW4
        return;
    }
    if (b4 * my >= 512)
        hipLaunchKernelGGL(tpw_mfma_kernel<4>, dim3((unsigned)(b4 * my)), dim3(256), 0, s, x, a, bias, addend, y, Q, HW, M, K, sm, sk);
    else
        hipLaunchKernelGGL(tpw_mfma_kernel<1>, dim3((unsigned)(b1 * my)), dim3(256), 0, s, x, a, bias, addend, y, Q, HW, M, K, sm, sk);
}
void launch_tconv_fwd(const float* x, const float* w, const float* bias, float* y, int N, int Cin, int H, int W, int Cout, int k, int stride,
                      int depthwise, hipStream_t s, TStatPart* st)
{
    const int pad = (k - 1) / 2, Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    if (st) st->count = 0;
    if (st && (long)N * Ho * Wo <= 4096 * 8) st = nullptr;              // BatchNorm's one-launch kernels read z once anyway
    if (!depthwise && k == 1 && stride == 1) {
        launch_tpw_gemm(x, w, bias, nullptr, y, (long)N * H * W, (long)H * W, Cout, Cin, (long)Cin, 1L, s, st);
        return;
    }
    static const bool s2_off = getenv("YF_TCONV3S2_OFF") != nullptr;
    if (!depthwise && k == 3 && stride == 2 && !s2_off && H % 2 == 0 && W % 8 == 0) {
        const long G = (long)N * Ho * (Wo / 4);
        if (Cin == 1 && Cout <= 8) {
            hipLaunchKernelGGL(tconv3s2_c1_kernel<8>, dim3(nblk(G)), dim3(256), 0, s, x, w, bias, y, N, H, W, Cout);
            return;
        }
        if (Cin % 4 == 0 && Cout <= 32) {
            const dim3 grid((unsigned)((G + 63) / 64));
            if (Cout <= 16) hipLaunchKernelGGL(tconv3s2_mfma_kernel<1>, grid, dim3(256), 0, s, x, w, bias, y, N, Cin, H, W, Cout);
            else hipLaunchKernelGGL(tconv3s2_mfma_kernel<2>, grid, dim3(256), 0, s, x, w, bias, y, N, Cin, H, W, Cout);
            return;
        }
    }
    if (!depthwise && k == 3) {
        hipLaunchKernelGGL(tconv_im2col_mfma_kernel<3>, dim3((unsigned)(((long)N * Ho * Wo + 63) / 64), (Cout + 63) / 64), dim3(256), 0, s, x, w, bias, y,
                           N, Cin, H, W, Ho, Wo, Cout, stride);
        return;
    }
    if (!depthwise) {
        const long Q = (long)N * Ho * Wo;
        if (Cout > 8)
            hipLaunchKernelGGL(tconv_mc_kernel<16>, dim3(nblk(Q), (Cout + 15) / 16), dim3(256), 0, s, x, w, bias, y, N, Cin, H, W, Cout, Ho, Wo, k,
                               stride, (long)Cin * k * k, (long)k * k);
        else
            hipLaunchKernelGGL(tconv_mc_kernel<8>, dim3(nblk(Q), (Cout + 7) / 8), dim3(256), 0, s, x, w, bias, y, N, Cin, H, W, Cout, Ho, Wo, k, stride,
                               (long)Cin * k * k, (long)k * k);
        return;
    }
    if (depthwise && !bias && stride == 1 && W % 4 != 0 && W <= 16 && (k == 3 || k == 5)) {
        const long nrows = (long)N * Cout * H;
        if (k == 3) hipLaunchKernelGGL((tdw_plane_kernel<3, false>), dim3(nblk(nrows)), dim3(256), 0, s, x, w, y, Cout, H, W, nrows);
        else hipLaunchKernelGGL((tdw_plane_kernel<5, false>), dim3(nblk(nrows)), dim3(256), 0, s, x, w, y, Cout, H, W, nrows);
        return;
    }
    if (depthwise && !bias && Wo % 4 == 0) {
        if (k == 3 && stride == 1) return launch_tdw_conv<3, 1, false>(x, w, y, N, Cout, H, W, Ho, Wo, s, st);
        if (k == 3 && stride == 2) return launch_tdw_conv<3, 2, false>(x, w, y, N, Cout, H, W, Ho, Wo, s);
        if (k == 5 && stride == 1) return launch_tdw_conv<5, 1, false>(x, w, y, N, Cout, H, W, Ho, Wo, s, st);
    }
    hipLaunchKernelGGL(tconv_fwd_kernel, dim3(nblk((long)N * Cout * Ho * Wo)), dim3(256), 0, s, x, w, bias, y, N, Cin, H, W, Cout, Ho, Wo, k, stride, depthwise);
}
// addend (optional, like dx): added to the result -- fused for the pointwise GEMM, a separate pass otherwise
void launch_tconv_bwd_data(const float* dy, const float* w, float* dx, int N, int Cin, int H, int W, int Cout, int k, int stride, int depthwise,
                           hipStream_t s, const float* addend, TBnRed* red)
{
    if (red) red->count = 0;
    if (red && (long)N * H * W <= 4096 * 8) red = nullptr;              // BatchNorm's one-launch backward reads dy and z once anyway
    if (addend && !(!depthwise && k == 1 && stride == 1)) {
        launch_tconv_bwd_data(dy, w, dx, N, Cin, H, W, Cout, k, stride, depthwise, s, nullptr, nullptr);
        launch_tadd(dx, addend, dx, (long)N * Cin * H * W, s);
        return;
    }
    const int pad = (k - 1) / 2, Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    if (!depthwise && k == 1 && stride == 1) {       // pointwise: dx[ci] = sum_co dy[co] w[co][ci] -- the same GEMM with the weight transposed
        // (the same epilogue in the pointwise GEMM: a z load, 4 constants and ~45 instructions for each of a lane's 4 MT rows cost what
        //  the reduction pass saved -- +60 / -58 us on the 8-channel layers at batch 256; only the depthwise kernel carries it)
        launch_tpw_gemm(dy, w, nullptr, addend, dx, (long)N * H * W, (long)H * W, Cin, Cout, 1L, (long)Cin, s);
        return;
    }
    static const bool s2m_off = getenv("YF_TCONV3S2_OFF") != nullptr;
    if (!depthwise && k == 3 && stride == 2 && H == 2 * Ho && W == 2 * Wo && !s2m_off && Cout % 4 == 0 && Wo % 4 == 0 && Cin <= 32 &&
        (long)N * Ho * (Wo / 4) >= 2048) {
        const dim3 grid((unsigned)(((long)N * Ho * (Wo / 4) + 63) / 64));
        if (Cin <= 16) hipLaunchKernelGGL(tconv3s2_bwd_mfma_kernel<1>, grid, dim3(256), 0, s, dy, w, dx, N, Cin, Cout, Ho, Wo);
        else hipLaunchKernelGGL(tconv3s2_bwd_mfma_kernel<2>, grid, dim3(256), 0, s, dy, w, dx, N, Cin, Cout, Ho, Wo);
        return;
    }
    if (!depthwise && k == 3 && stride == 2 && H == 2 * Ho && W == 2 * Wo) {
        hipLaunchKernelGGL(tconv3s2_bwd_data_kernel<8>, dim3(nblk((long)N * Ho * Wo), (Cin + 7) / 8), dim3(256), 0, s, dy, w, dx, N, Cin, Cout, Ho, Wo);
        return;
    }
    if (depthwise) {
        if (stride == 1 && W % 4 != 0 && W <= 16 && (k == 3 || k == 5)) {
            const long nrows = (long)N * Cin * H;
            if (k == 3) hipLaunchKernelGGL((tdw_plane_kernel<3, true>), dim3(nblk(nrows)), dim3(256), 0, s, dy, w, dx, Cin, H, W, nrows);
            else hipLaunchKernelGGL((tdw_plane_kernel<5, true>), dim3(nblk(nrows)), dim3(256), 0, s, dy, w, dx, Cin, H, W, nrows);
            return;
        }
        if (stride == 1 && W % 4 == 0 && k == 3) return launch_tdw_conv<3, 1, true>(dy, w, dx, N, Cin, H, W, H, W, s, nullptr, red);
        if (stride == 1 && W % 4 == 0 && k == 5) return launch_tdw_conv<5, 1, true>(dy, w, dx, N, Cin, H, W, H, W, s, nullptr, red);
        if (stride == 2 && k == 3 && H == 2 * Ho && W == 2 * Wo) {
            const int threads = Ho * Wo, bs = threads <= 64 ? 64 : 256;
            hipLaunchKernelGGL(tdw3s2_bwd_data_kernel, dim3(N * Cin, (threads + bs - 1) / bs), dim3(bs), 0, s, dy, w, dx, Cin, Ho, Wo);
            return;
        }
    }
    hipLaunchKernelGGL(tconv_bwd_data_kernel, dim3(nblk((long)N * Cin * H * W)), dim3(256), 0, s, dy, w, dx, N, Cin, H, W, Cout, Ho, Wo, k, stride, depthwise);
}
// waves per workgroup of the weight-gradient GEMM: fill the chip (>= 16 waves per CU) when the slab limit keeps the grid small and a
// slice is long enough to deal out
static inline int twgrad_waves(long workgroups, long q_per)
{
    static const int forced = getenv("YF_WGRAD_NW") ? atoi(getenv("YF_WGRAD_NW")) : 0;
    if (forced == 1 || forced == 4 || forced == 8) return forced;
    if (workgroups * 8 <= 8192 && q_per >= 8 * 64) return 8;
    if (workgroups * 4 <= 8192 && q_per >= 4 * 64) return 4;
    return 1;
}
// the slabs of the split reductions: nsplit <= what fits into the scratch
static inline void tsum_partials(const float* part, long nsplit, long nw, float* dw, hipStream_t s)
{
    int spl = 1;
    while (spl < 64 && spl * 8 <= nsplit) spl *= 2;                 // >= 8 slabs per lane
    hipLaunchKernelGGL(tsum_partials_kernel, dim3(nblk(nw * spl)), dim3(256), 0, s, part, (int)nsplit, nw, nw, dw, spl);
}
// the multi-tensor form: every layer's slabs stay alive in the defer region until ONE launch at the end of the pass adds them all
// (tsum_multi_kernel: the same lanes, order and arithmetic per output as tsum_partials_kernel -- bitwise the same sums)
__global__ void __launch_bounds__(256) tsum_multi_kernel(const TSumEntry* __restrict__ tab, int n, const float* __restrict__ slab, float* __restrict__ dst)
{
    // this block's entry = the last one that starts at or before it: one parallel probe of the (<= 256) entries, not a dependent scan
    const int e = __syncthreads_count((int)threadIdx.x < n && tab[threadIdx.x].blk0 <= (long)blockIdx.x) - 1;
    const TSumEntry E = tab[e];
    const float* part = slab + E.part_off;
    const long t = ((long)blockIdx.x - E.blk0) * 256 + threadIdx.x;
    const long i = t / E.spl;
    const int j = (int)(t - i * E.spl);
    float v = 0.f;
    if (i < E.nw) {
        int sidx = j;
        for (; sidx + 3 * E.spl < E.nsplit; sidx += 4 * E.spl) {          // four slabs requested at once, added in slab order
            const float p0 = part[(long)sidx * E.nw + i], p1 = part[(long)(sidx + E.spl) * E.nw + i];
            const float p2 = part[(long)(sidx + 2 * E.spl) * E.nw + i], p3 = part[(long)(sidx + 3 * E.spl) * E.nw + i];
            v += p0; v += p1; v += p2; v += p3;
        }
        for (; sidx < E.nsplit; sidx += E.spl) v += part[(long)sidx * E.nw + i];
    }
    for (int o = E.spl >> 1; o > 0; o >>= 1) v += __shfl_down(v, o);
    if (i < E.nw && j == 0) dst[E.dst_off + i] = v;
}
float* TSumDefer::take(long floats)
{
    const size_t need = ((size_t)floats + 63) & ~(size_t)63;
    if (!slab || used + need > cap_floats) return nullptr;
    float* p = slab + used;
    used += need;
    return p;
}
void TSumDefer::push(const float* part, long nsplit, long nw, float* dw)
{
    TSumEntry E;
    E.part_off = part - slab; E.dst_off = dw - dst_base; E.nw = nw; E.nsplit = (int)nsplit;
    int spl = 1;
    while (spl < 64 && spl * 8 <= nsplit) spl *= 2;
    E.spl = spl;
    E.blk0 = nblocks;
    nblocks += (nw * spl + 255) / 256;
    entries.push_back(E);
}
void launch_tsum_multi(const TSumEntry* d_tab, int n, long nblocks, const float* slab, float* dst, hipStream_t s)
{
    if (n > 0) hipLaunchKernelGGL(tsum_multi_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, d_tab, n, slab, dst);
}
// where a split reduction writes: the gradient itself without a split, a slab of the defer region, or the shared scratch
static inline float* tsum_out(void* scratch, long nsplit, long nw, float* dw, TSumDefer* d)
{
    if (nsplit <= 1) return dw;
    if (d) {
        float* p = d->take(nsplit * nw);
        if (p) return p;
    }
    return (float*)scratch;
}
static inline void tsum_finish(const float* out, void* scratch, long nsplit, long nw, float* dw, hipStream_t s, TSumDefer* d)
{
    if (nsplit <= 1) return;
    if (d && out != (const float*)scratch) d->push(out, nsplit, nw, dw);
    else tsum_partials(out, nsplit, nw, dw, s);
}
void launch_tconv_bwd_weight(const float* x, const float* dy, float* dw, int N, int Cin, int H, int W, int Cout, int k, int stride, int depthwise,
                             void* scratch, size_t scratch_bytes, hipStream_t s, TSumDefer* defer)
{
    const int pad = (k - 1) / 2, Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    const long nw = (long)Cout * (depthwise ? 1 : Cin) * k * k, P = (long)N * Ho * Wo;
    const long fit = scratch ? (long)(scratch_bytes / ((size_t)nw * sizeof(float))) : 0;      // slabs that fit
    static const bool s2_off = getenv("YF_TCONV3S2_OFF") != nullptr;
    if (!depthwise && k == 3 && stride == 2 && !s2_off && H % 2 == 0 && W % 8 == 0 && fit >= 1) {
        const long G = (long)N * Ho * (Wo / 4);
        if (Cin == 1 && Cout <= 8 && G >= 4096) {
            long chunks = G / 2048;                                     // >= 8 groups per thread
            if (chunks > 1024) chunks = 1024;
            if (chunks > fit) chunks = fit;
            float* out = tsum_out(scratch, chunks, nw, dw, defer);
            hipLaunchKernelGGL(tconv3s2_c1_wgrad_kernel<8>, dim3((unsigned)chunks), dim3(256), 0, s, x, dy, out, N, H, W, Cout, nw);
            tsum_finish(out, scratch, chunks, nw, dw, s, defer);
            return;
        }
        if (Cin % 4 == 0 && Cin <= 32 && Cout <= 32 && G >= 2048) {
            long nsplit = G / 32;                                       // >= 8 steps per wave
            if (nsplit > 1024) nsplit = 1024;
            if (nsplit > fit) nsplit = fit;
            const int n_cu = device_cu_count(current_device());        // equal slices, one workgroup each: a whole number per CU
            if (n_cu > 0 && nsplit > n_cu) nsplit -= nsplit % n_cu;
            long g_per = (G + nsplit - 1) / nsplit;
            g_per = (g_per + 3) / 4 * 4;
            nsplit = (G + g_per - 1) / g_per;
            float* out = tsum_out(scratch, nsplit, nw, dw, defer);
            hipLaunchKernelGGL(tconv3s2_wgrad_mfma_kernel, dim3((unsigned)nsplit), dim3(256), 0, s, x, dy, out, N, Cin, H, W, Cout, g_per, nw);
            tsum_finish(out, scratch, nsplit, nw, dw, s, defer);
            return;
        }
    }
    // 4 consecutive output pixels per lane, in one frame
    if (!depthwise && (k == 3 || (k == 1 && stride == 1)) && ((long)Ho * Wo) % 4 == 0) {
        const int R = Cin * k * k, tiles = ((Cout + 15) / 16) * ((R + 63) / 64);
        // many pixels: 4 waves per workgroup share a slice (added through LDS) rather than 4 slices -- the same waves in flight, a
        // quarter of the slabs to write and to add up afterwards
        static const bool forced_nw = getenv("YF_WGRAD_NW") != nullptr;
        const long pw = (!forced_nw && P >= 65536) ? 4 : 1;
        long nsplit = (P + 128 * pw - 1) / (128 * pw);                  // >= 8 MFMA steps per wave ...
        while (nsplit * tiles * pw > 8192 && nsplit > 1) nsplit = (nsplit + 1) / 2;   // ... and a bounded grid
        if (nsplit > 1024) nsplit = 1024;
        if (nsplit > fit) nsplit = fit < 1 ? 1 : fit;
        long q_per = (P + nsplit - 1) / nsplit;
        q_per = (q_per + 15) / 16 * 16;
        nsplit = (P + q_per - 1) / q_per;
        float* out = tsum_out(scratch, nsplit, nw, dw, defer);
        const dim3 grid((unsigned)(nsplit * tiles));
        int nwv = twgrad_waves(nsplit * tiles, q_per);
        if (pw == 4 && nwv < 4 && q_per >= 4 * 64) nwv = 4;
#define YF_WG(KS_, NW_) hipLaunchKernelGGL((tconv_wgrad_mfma_kernel<KS_, NW_>), grid, dim3(64 * NW_), 0, s, x, dy, out, N, Cin, H, W, Cout, Ho, Wo, stride, q_per, nw)
        if (k == 1) { if (nwv == 8) YF_WG(1, 8); else if (nwv == 4) YF_WG(1, 4); else YF_WG(1, 1); }
        else { if (nwv == 8) YF_WG(3, 8); else if (nwv == 4) YF_WG(3, 4); else YF_WG(3, 1); }
#undef YF_WG
        tsum_finish(out, scratch, nsplit, nw, dw, s, defer);
        return;
    }
    if (depthwise && (k == 3 || k == 5)) {
        long chunks = (P + 2047) / 2048;                                // ~8 pixels per thread
        while (chunks * Cout > 4096 && chunks > 1) chunks = (chunks + 1) / 2;
        const bool rows = !tdw_rows_off && stride == 1 && W % 4 == 0 && H % 4 == 0 && (long)N * (H / 4) * (W / 4) >= 2048;
        if (rows && (long)N * (H / 4) * (W / 4) / 512 < chunks) chunks = (long)N * (H / 4) * (W / 4) / 512;   // 16 outputs per thread and trip
        if (chunks > fit) chunks = fit < 1 ? 1 : fit;
        float* out = tsum_out(scratch, chunks, nw, dw, defer);
        const dim3 grid((unsigned)chunks, Cout);
        if (rows) {
            if (k == 3) hipLaunchKernelGGL((tdw_wgrad_rows_kernel<3, 4>), grid, dim3(256), 0, s, x, dy, out, N, Cout, H, W, nw);
            else hipLaunchKernelGGL((tdw_wgrad_rows_kernel<5, 4>), grid, dim3(256), 0, s, x, dy, out, N, Cout, H, W, nw);
            tsum_finish(out, scratch, chunks, nw, dw, s, defer);
            return;
        }
        if (stride == 1 && W % 4 != 0 && W <= 16) {
            if (k == 3) hipLaunchKernelGGL(tdw_plane_wgrad_kernel<3>, grid, dim3(256), 0, s, x, dy, out, N, Cout, H, W, nw);
            else hipLaunchKernelGGL(tdw_plane_wgrad_kernel<5>, grid, dim3(256), 0, s, x, dy, out, N, Cout, H, W, nw);
            tsum_finish(out, scratch, chunks, nw, dw, s, defer);
            return;
        }
        if (k == 3)
            hipLaunchKernelGGL(tdw_wgrad_kernel<3>, grid, dim3(256), 0, s, x, dy, out, N, Cout, H, W, Ho, Wo, stride, nw);
        else
            hipLaunchKernelGGL(tdw_wgrad_kernel<5>, grid, dim3(256), 0, s, x, dy, out, N, Cout, H, W, Ho, Wo, stride, nw);
        tsum_finish(out, scratch, chunks, nw, dw, s, defer);
        return;
    }
    (void)hipMemsetAsync(dw, 0, (size_t)nw * sizeof(float), s);        // the fallbacks below accumulate with atomics
    if (!depthwise) {
        const int R = Cin * k * k, tiles = ((R + 63) / 64) * ((Cout + 63) / 64);
        long nsplit = (P + 255) / 256;                                  // >= 8 staged tiles per workgroup ...
        while (nsplit * tiles > 2048 && nsplit > 1) nsplit = (nsplit + 1) / 2;   // ... and a bounded grid
        long p_per = (P + nsplit - 1) / nsplit;
        p_per = (p_per + 31) / 32 * 32;
        nsplit = (P + p_per - 1) / p_per;
        hipLaunchKernelGGL(tconv_bwd_weight_gemm_kernel, dim3((unsigned)nsplit, (R + 63) / 64, (Cout + 63) / 64), dim3(256), 0, s, x, dy, dw, N, Cin, H,
                           W, Cout, Ho, Wo, k, stride, p_per);
        return;
    }
    int nchunk = (int)((P + 4095) / 4096);                       // ~16 reduction elements per thread
    while ((long)nchunk * nw > 262144 && nchunk > 1) nchunk /= 2;   // bound the grid
    hipLaunchKernelGGL(tconv_bwd_weight_kernel, dim3((unsigned)(nw * nchunk)), dim3(256), 0, s, x, dy, dw, N, Cin, H, W, Cout, Ho, Wo, k, stride,
                       depthwise, nchunk);
}
// backward-data + weight gradient of a pointwise layer in one launch (tpw_bwd_dual_kernel); false: not applicable, launch them separately
bool launch_tpw_bwd_dual(const float* x, const float* dz, const float* w, float* dw, float* dx, const float* addend, int N, int Cin, int H, int W,
                         int Cout, void* scratch, size_t scratch_bytes, hipStream_t s, TSumDefer* defer)
{
    static const bool off = getenv("YF_TPW_DUAL_OFF") != nullptr;
    const long HW = (long)H * W, Q = (long)N * HW;
    // up to 2 M pixels per batch (batch 256: strides 4 and up; measured 18.11 -> 17.94 -> 17.80 ms for limits of 128 k / 400 k / 2 M, no
    // further gain without a limit): below that the two kernels mostly wait, and side by side they wait once
    static const long qmax = getenv("YF_TPW_DUAL_QMAX") ? atol(getenv("YF_TPW_DUAL_QMAX")) : 2000000;
    if (off || HW % 4 || Cout % 4 || Q > qmax || Q < 256) return false;
    const long nw = (long)Cout * Cin;
    const long fit = scratch ? (long)(scratch_bytes / ((size_t)nw * sizeof(float))) : 0;
    // the data gradient's grid (launch_tpw_gemm's choice without the weight-stationary variant)
    const int mtiles = (Cin + 15) / 16;
    int mgroups = (mtiles + 3) / 4, mt = (mtiles + mgroups - 1) / mgroups;
    while (mt > 1 && (Q + 63) / 64 * mgroups < 512) { mt = (mt + 1) / 2; mgroups = (mtiles + mt - 1) / mt; }
    const long nA = (Q + 255) / 256 * mgroups;
    // the weight gradient's: 4 waves per slice
    const int tiles = ((Cout + 15) / 16) * ((Cin + 63) / 64);
    long nsplit = (Q + 511) / 512;
    while (nsplit * tiles * 4 > 8192 && nsplit > 1) nsplit = (nsplit + 1) / 2;
    if (nsplit > 1024) nsplit = 1024;
    if (nsplit > fit) nsplit = fit < 1 ? 1 : fit;
    long q_per = (Q + nsplit - 1) / nsplit;
    q_per = (q_per + 15) / 16 * 16;
    nsplit = (Q + q_per - 1) / q_per;
    float* out = tsum_out(scratch, nsplit, nw, dw, defer);
    TPwBwdArgs a{dz, w, addend, dx, x, out, Q, HW, N, Cin, Cout, H, W, mgroups, (unsigned)nA, q_per, nw};
    const dim3 grid((unsigned)(nA + nsplit * tiles));
    if (mt == 1) hipLaunchKernelGGL(tpw_bwd_dual_kernel<1>, grid, dim3(256), 0, s, a);
    else if (mt == 2) hipLaunchKernelGGL(tpw_bwd_dual_kernel<2>, grid, dim3(256), 0, s, a);
    else if (mt == 3) hipLaunchKernelGGL(tpw_bwd_dual_kernel<3>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(tpw_bwd_dual_kernel<4>, grid, dim3(256), 0, s, a);
    tsum_finish(out, scratch, nsplit, nw, dw, s, defer);
    return true;
}
void launch_tdeconv_fwd(const float* x, const float* w, float* y, int N, int Cin, int H, int W, int Cout, hipStream_t s)
{
    static const bool off = getenv("YF_TDECONV_OLD") != nullptr;
    const long Q = (long)N * H * W, HW = (long)H * W;
    if (!off && Q >= 256 && HW % 4 == 0 && Cin % 4 == 0) {             // the GEMM form (tpw4_mfma_kernel<.., DECONV>): M = 4 Cout rows
        const int M = 4 * Cout, tiles = (M + 15) / 16;
        int mgroups = (tiles + 3) / 4, mt = (tiles + mgroups - 1) / mgroups;
        while (mt > 1 && (Q + 63) / 64 * mgroups < 512) { mt = (mt + 1) / 2; mgroups = (tiles + mt - 1) / mt; }
        const dim3 grid((unsigned)((Q + 255) / 256 * mgroups));
#define YF_DC4(MT_) hipLaunchKernelGGL((tpw4_mfma_kernel<MT_, true>), grid, dim3(256), 0, s, x, w, (const float*)nullptr, (const float*)nullptr, y, Q, HW, M, Cin, 1L, (long)M, mgroups, W)
        if (mt == 1) YF_DC4(1); else if (mt == 2) YF_DC4(2); else if (mt == 3) YF_DC4(3); else YF_DC4(4);
#undef YF_DC4
        return;
    }
    hipLaunchKernelGGL(tdeconv_fwd_kernel, dim3(nblk((long)N * Cout * 4 * H * W)), dim3(256), 0, s, x, w, y, N, Cin, H, W, Cout);
}
void launch_tdeconv_bwd_data(const float* dy, const float* w, float* dx, int N, int Cin, int H, int W, int Cout, hipStream_t s)
{
    static const bool off = getenv("YF_TDECONV_OLD") != nullptr;
    if ((long)N * H * W < (off ? 8192 : 256)) {      // (the old gather GEMM at 16 x 8 x 10 pixels: 134 us against 25 us for this one)
        hipLaunchKernelGGL(tdeconv_bwd_data_kernel, dim3(nblk((long)N * Cin * H * W)), dim3(256), 0, s, dy, w, dx, N, Cin, H, W, Cout);
        return;
    }
    if (!off) {
        const long Q = (long)N * H * W;
        const int tiles = (Cin + 15) / 16;
        int mgroups = (tiles + 2) / 3, mt = (tiles + mgroups - 1) / mgroups;
        while (mt > 1 && (Q + 63) / 64 * mgroups < 512) { mt = (mt + 1) / 2; mgroups = (tiles + mt - 1) / mt; }
        const dim3 grid((unsigned)((Q + 255) / 256 * mgroups));
#define YF_DB(MT_) hipLaunchKernelGGL(tdeconv_bwd_mfma_kernel<MT_>, grid, dim3(256), 0, s, dy, w, dx, Q, H, W, Cin, Cout, mgroups)
        if (mt == 1) YF_DB(1); else if (mt == 2) YF_DB(2); else YF_DB(3);
#undef YF_DB
        return;
    }
    // dx[ci][p] = sum over (co, a, b) of dY[co][2 iy + a][2 ix + b] w[ci][co][a][b]: a 2x2 stride-2 pad-0 convolution of dY with the weight
    // read as [Cin][(co, a, b)] -- the im2col GEMM
    hipLaunchKernelGGL(tconv_im2col_mfma_kernel<2>, dim3((unsigned)(((long)N * H * W + 63) / 64), (Cin + 63) / 64), dim3(256), 0, s, dy, w,
                       (const float*)nullptr, dx, N, Cout, 2 * H, 2 * W, H, W, Cin, 2);
}
void launch_tdeconv_bwd_weight(const float* x, const float* dy, float* dw, int N, int Cin, int H, int W, int Cout, void* scratch, size_t scratch_bytes,
                               hipStream_t s, TSumDefer* defer)
{
    const long nw = (long)Cin * Cout * 4, P = (long)N * H * W;
    if (((long)H * W) % 4 == 0) {                 // the conv weight-gradient GEMM with the operands swapped (see tconv_wgrad_mfma_kernel)
        const long fit = scratch ? (long)(scratch_bytes / ((size_t)nw * sizeof(float))) : 0;
        const int R = Cout * 4, tiles = ((Cin + 15) / 16) * ((R + 63) / 64);
        long nsplit = (P + 127) / 128;
        while (nsplit * tiles > 8192 && nsplit > 1) nsplit = (nsplit + 1) / 2;
        if (nsplit > 1024) nsplit = 1024;
        if (nsplit > fit) nsplit = fit < 1 ? 1 : fit;
        long q_per = (P + nsplit - 1) / nsplit;
        q_per = (q_per + 15) / 16 * 16;
        nsplit = (P + q_per - 1) / q_per;
        float* out = tsum_out(scratch, nsplit, nw, dw, defer);
        const int nwv = twgrad_waves(nsplit * tiles, q_per);
        const dim3 grid((unsigned)(nsplit * tiles));
        if (nwv == 8) hipLaunchKernelGGL((tconv_wgrad_mfma_kernel<2, 8>), grid, dim3(512), 0, s, dy, x, out, N, Cout, 2 * H, 2 * W, Cin, H, W, 2, q_per, nw);
        else if (nwv == 4) hipLaunchKernelGGL((tconv_wgrad_mfma_kernel<2, 4>), grid, dim3(256), 0, s, dy, x, out, N, Cout, 2 * H, 2 * W, Cin, H, W, 2, q_per, nw);
        else hipLaunchKernelGGL((tconv_wgrad_mfma_kernel<2, 1>), grid, dim3(64), 0, s, dy, x, out, N, Cout, 2 * H, 2 * W, Cin, H, W, 2, q_per, nw);
        tsum_finish(out, scratch, nsplit, nw, dw, s, defer);
        return;
    }
    int nchunk = (int)((P + 4095) / 4096);
    while ((long)nchunk * nw > 262144 && nchunk > 1) nchunk /= 2;
    (void)hipMemsetAsync(dw, 0, (size_t)nw * sizeof(float), s);
    hipLaunchKernelGGL(tdeconv_bwd_weight_kernel, dim3((unsigned)(nw * nchunk)), dim3(256), 0, s, x, dy, dw, N, Cin, H, W, Cout, nchunk);
}
// one scratch for the split reductions of a stream: BatchNorm partial pairs (256 KB) or weight-gradient slabs (all of it)
size_t train_scratch_bytes() { return (size_t)32 << 20; }
static inline unsigned tbn_apply_blocks(int N, int C, long HW, int V, bool flat)
{
    const long U = tbn_units(N, HW, V, flat);
    long n = 8192 / C;
    if (n > U) n = U;
    return n < 1 ? 1u : (unsigned)n;
}
static inline int tbn_chunks(int N, int C, long HW, int V, bool flat)
{
    const long U = tbn_units(N, HW, V, flat);
    long n = 8192 / C;
    if (n > TBN_MAXCHUNK) n = TBN_MAXCHUNK;
    if (n > U) n = U;
    return n < 1 ? 1 : (int)n;
}
// scratch: >= 1 MB of device memory (partial sums; needs no initialisation); C <= 256
// residual (optional, like y): y = bn(x) [relu] + residual
void launch_tbn_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var, float* stats, float* y, int N,
                    int C, long HW, int relu, void* scratch, hipStream_t s, const float* residual, const TStatPart* st)
{
    static const bool small4_off = getenv("YF_TBN_SMALL4_OFF") != nullptr;
    if (!small4_off && HW % 4 == 0 && (long)N * HW <= 4096 * 8) {
        const long P4 = (long)N * HW / 4;
#define YF_BNF(U_) hipLaunchKernelGGL(tbn_fwd_small4_kernel<U_>, dim3(C), dim3(1024), 0, s, x, gamma, beta, y, N, C, (int)HW, relu, 1e-5f, 0.1f, stats, running_mean, running_var, residual)
        if (P4 <= 1024) YF_BNF(1); else if (P4 <= 2048) YF_BNF(2); else if (P4 <= 4096) YF_BNF(4); else YF_BNF(8);
#undef YF_BNF
        return;
    }
    if ((long)N * HW <= TBN_SMALL) {
        hipLaunchKernelGGL(tbn_fwd_small_kernel, dim3(C), dim3(1024), 0, s, x, gamma, beta, y, N, C, (int)HW, relu, 1e-5f, 0.1f, stats, running_mean,
                           running_var, residual);
        return;
    }
    static const bool flat_off = getenv("YF_TBN_FLAT_OFF") != nullptr;
    const bool flat = !flat_off && HW % 4 == 0 && HW % 1024 != 0;      // float4 numbered across the frames (small / ragged planes)
    const bool parts = st && st->count > 0;                             // the conv left the partial sums: no pass over z for them
    int pchunks = parts ? (int)((st->count + 1023) / 1024) : 0;        // >= 4 pairs per thread
    if (pchunks > 64) pchunks = 64;
    const int V = (flat || (HW % 4 == 0 && HW >= 1024)) ? 4 : 1, nchunk = parts ? pchunks : tbn_chunks(N, C, HW, V, flat);
    const dim3 g1(nchunk, C), g2(tbn_apply_blocks(N, C, HW, V, flat), C);
    if (parts) hipLaunchKernelGGL(tbn_stats_from_parts_kernel, dim3(pchunks, C), dim3(256), 0, s, (const float2*)st->part, st->count, (double*)scratch);
    if (flat) {
        if (!parts) hipLaunchKernelGGL((tbn_stats_kernel<4, true>), g1, dim3(256), 0, s, x, N, C, HW, (double*)scratch);
        hipLaunchKernelGGL((tbn_apply_kernel<4, true>), g2, dim3(256), 0, s, x, (const double*)scratch, nchunk, gamma, beta, y, N, C, HW, relu, 1e-5f, 0.1f,
                           stats, running_mean, running_var, residual);
    } else if (V == 4) {
        if (!parts) hipLaunchKernelGGL(tbn_stats_kernel<4>, g1, dim3(256), 0, s, x, N, C, HW, (double*)scratch);
        hipLaunchKernelGGL(tbn_apply_kernel<4>, g2, dim3(256), 0, s, x, (const double*)scratch, nchunk, gamma, beta, y, N, C, HW, relu, 1e-5f, 0.1f,
                           stats, running_mean, running_var, residual);
    } else {
        hipLaunchKernelGGL(tbn_stats_kernel<1>, g1, dim3(256), 0, s, x, N, C, HW, (double*)scratch);
        hipLaunchKernelGGL(tbn_apply_kernel<1>, g2, dim3(256), 0, s, x, (const double*)scratch, nchunk, gamma, beta, y, N, C, HW, relu, 1e-5f, 0.1f,
                           stats, running_mean, running_var, residual);
    }
}
void launch_tbn_bwd(const float* x, const float* dy, const float* stats, const float* gamma, const float* beta, float* dgamma, float* dbeta, float* dx,
                    int N, int C, long HW, int relu, void* scratch, hipStream_t s, const TBnRed* red)
{
    static const bool small4_off = getenv("YF_TBN_SMALL4_OFF") != nullptr;
    if (!small4_off && HW % 4 == 0 && (long)N * HW <= 4096 * 8) {
        const long P4 = (long)N * HW / 4;
#define YF_BNB(U_) hipLaunchKernelGGL(tbn_bwd_small4_kernel<U_>, dim3(C), dim3(1024), 0, s, x, dy, stats, gamma, beta, dgamma, dbeta, dx, N, C, (int)HW, relu)
        if (P4 <= 1024) YF_BNB(1); else if (P4 <= 2048) YF_BNB(2); else if (P4 <= 4096) YF_BNB(4); else YF_BNB(8);
#undef YF_BNB
        return;
    }
    if ((long)N * HW <= TBN_SMALL) {
        hipLaunchKernelGGL(tbn_bwd_small_kernel, dim3(C), dim3(1024), 0, s, x, dy, stats, gamma, beta, dgamma, dbeta, dx, N, C, (int)HW, relu);
        return;
    }
    static const bool flat_off = getenv("YF_TBN_FLAT_OFF") != nullptr;
    const bool flat = !flat_off && HW % 4 == 0 && HW % 1024 != 0;
    const bool parts = red && red->count > 0;                          // the data-gradient kernel above left the pairs: no reduction pass
    int pchunks = parts ? (int)((red->count + 1023) / 1024) : 0;
    if (pchunks > 64) pchunks = 64;
    const int V = (flat || (HW % 4 == 0 && HW >= 1024)) ? 4 : 1, nchunk = parts ? pchunks : tbn_chunks(N, C, HW, V, flat);
    const dim3 g1(nchunk, C), g2(tbn_apply_blocks(N, C, HW, V, flat), C);
    if (parts) {
        hipLaunchKernelGGL(tbn_stats_from_parts_kernel, dim3(pchunks, C), dim3(256), 0, s, (const float2*)red->part, red->count, (double*)scratch);
        if (flat)
            hipLaunchKernelGGL((tbn_bwd_apply_kernel<4, true>), g2, dim3(256), 0, s, x, dy, stats, gamma, beta, (const double*)scratch, nchunk, dgamma, dbeta,
                               dx, N, C, HW, relu);
        else if (V == 4)
            hipLaunchKernelGGL(tbn_bwd_apply_kernel<4>, g2, dim3(256), 0, s, x, dy, stats, gamma, beta, (const double*)scratch, nchunk, dgamma, dbeta, dx, N,
                               C, HW, relu);
        else
            hipLaunchKernelGGL(tbn_bwd_apply_kernel<1>, g2, dim3(256), 0, s, x, dy, stats, gamma, beta, (const double*)scratch, nchunk, dgamma, dbeta, dx, N,
                               C, HW, relu);
        return;
    }
    if (flat) {
        hipLaunchKernelGGL((tbn_bwd_reduce_kernel<4, true>), g1, dim3(256), 0, s, x, dy, stats, gamma, beta, N, C, HW, relu, (double*)scratch);
        hipLaunchKernelGGL((tbn_bwd_apply_kernel<4, true>), g2, dim3(256), 0, s, x, dy, stats, gamma, beta, (const double*)scratch, nchunk, dgamma, dbeta,
                           dx, N, C, HW, relu);
    } else if (V == 4) {
        hipLaunchKernelGGL(tbn_bwd_reduce_kernel<4>, g1, dim3(256), 0, s, x, dy, stats, gamma, beta, N, C, HW, relu, (double*)scratch);
        hipLaunchKernelGGL(tbn_bwd_apply_kernel<4>, g2, dim3(256), 0, s, x, dy, stats, gamma, beta, (const double*)scratch, nchunk, dgamma, dbeta, dx, N,
                           C, HW, relu);
    } else {
        hipLaunchKernelGGL(tbn_bwd_reduce_kernel<1>, g1, dim3(256), 0, s, x, dy, stats, gamma, beta, N, C, HW, relu, (double*)scratch);
        hipLaunchKernelGGL(tbn_bwd_apply_kernel<1>, g2, dim3(256), 0, s, x, dy, stats, gamma, beta, (const double*)scratch, nchunk, dgamma, dbeta, dx, N,
                           C, HW, relu);
    }
}
void launch_tchan_sum(const float* dy, float* out, int N, int C, long HW, hipStream_t s, void* scratch)
{
    const long U = (long)N * HW / 4;
    if (scratch && HW % 4 == 0 && U >= 16384) {                        // enough to spread over the chip: <= 64 chunks of >= 8 trips
        int nchunk = (int)(U / 2048);
        if (nchunk > 64) nchunk = 64;
        hipLaunchKernelGGL(tchan_sum_part_kernel, dim3(nchunk, C), dim3(256), 0, s, dy, N, C, HW, (double*)scratch);
        hipLaunchKernelGGL(tchan_sum_final_kernel, dim3((C + 63) / 64), dim3(64), 0, s, (const double*)scratch, nchunk, C, out);
        return;
    }
    hipLaunchKernelGGL(tchan_sum_kernel, dim3(C), dim3(256), 0, s, dy, N, C, HW, out);
}
void launch_tadd(const float* a, const float* b, float* out, long total, hipStream_t s)
{
    hipLaunchKernelGGL(tadd_kernel, dim3(nblk(total)), dim3(256), 0, s, a, b, out, total);
}
void launch_tslice(const float* src, float* dst, int N, int C, long HW, int Cs, int sc0, int Cd, int dc0, hipStream_t s)
{
    hipLaunchKernelGGL(tslice_kernel, dim3(nblk((long)N * C * HW)), dim3(256), 0, s, src, dst, N, C, HW, Cs, sc0, Cd, dc0);
}
void launch_tadam(float* p, const float* g, float* m, float* v, long total, double lr, double b1, double b2, double eps, int step, hipStream_t s)
{
    const double bc1 = 1.0 - pow(b1, (double)step), bc2 = 1.0 - pow(b2, (double)step);
    hipLaunchKernelGGL(tadam_kernel, dim3(nblk(total)), dim3(256), 0, s, p, g, m, v, total, (float)(1.0 - b1), (float)b2, (float)(1.0 - b2), (float)eps,
                       (float)(lr / bc1), (float)sqrt(bc2));
}

// host arrays of device pointers -> (one table upload +) one launch.  d_table: at least nt * 48 bytes of device memory.
// h_table: nt * 48 bytes of PINNED host memory owned by the caller (one per optimizer), or null.  With it the table is written there and
// uploaded with a truly asynchronous copy, and only when `upload` is set -- the caller sets it when the pointer set changed (p, m, v are
// stable; g is too when the flat gradient buffer is reused) and must not call again with upload = 1 before the previous upload has
// executed (training.Adam waits on an event).  Without it (null) the table is staged from pageable memory on every call, which the
// runtime completes before returning -- correct, but it can hold the host until the stream has drained.
int launch_tadam_multi(int nt, float* const* p, const float* const* g, float* const* m, float* const* v, const long* sizes, double lr, double b1,
                       double b2, double eps, int step, void* d_table, void* h_table, int upload, hipStream_t s)
{
    static thread_local std::vector<TAdamEntry> tab;
    TAdamEntry* host = static_cast<TAdamEntry*>(h_table);
    if (!host) { tab.resize(nt); host = tab.data(); upload = 1; }
    long blocks = 0;
    for (int t = 0; t < nt; ++t) {
        if (upload) host[t] = TAdamEntry{p[t], g[t], m[t], v[t], blocks, sizes[t]};
        blocks += (sizes[t] + 255) / 256;
    }
    if (upload && hipMemcpyAsync(d_table, host, (size_t)nt * sizeof(TAdamEntry), hipMemcpyHostToDevice, s) != hipSuccess) return -1;
    const double bc1 = 1.0 - pow(b1, (double)step), bc2 = 1.0 - pow(b2, (double)step);
    hipLaunchKernelGGL(tadam_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const TAdamEntry*)d_table, nt, (float)(1.0 - b1), (float)b2,
                       (float)(1.0 - b2), (float)eps, (float)(lr / bc1), (float)sqrt(bc2));
    return 0;
}

}  // namespace yf
