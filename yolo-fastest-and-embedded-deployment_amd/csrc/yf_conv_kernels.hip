// yf_conv_kernels.hip -- per-layer convolution kernels of the YOLO-Fastest forward pass for gfx950.
//
// Replaces what torch.nn does under the reference's YoloFastest.forward
// (src/model_training/model/yolo_fastest.py:150-218): Conv2d(+folded BN)(+ReLU), depthwise conv,
// ConvTranspose2d 2x2 s2, the residual add of BasicResBlock (:65) and the channel concat (:209).
//
// Layout: activations are NHWC fp32 in HBM (channel innermost) so that a lane owning one pixel reads
// its channels with 16-byte loads and a wave covers 64 consecutive pixels.  Weights are BN-folded and
// stored [cin][cout]: every lane of a wave needs the SAME weight at the same time, so they are read
// through the scalar data path (s_load_dwordx{4,8,16} into SGPRs) and cost no VGPRs, no LDS and no
// vector-memory bandwidth; each v_fma_f32 takes its weight as an SGPR operand.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "yf_kernels.h"

namespace yf {

// ------------------------------------------------------------------------------------------------
// Pointwise (1x1) convolution, optionally reading a two-tensor channel concat, adding a residual,
// and writing NHWC / NCHW / the 2x2-stride-2 deconvolution's interleaved pixels.
//   thread: P pixels (pixel p*256 apart so that lanes stay on consecutive pixels) x CT output channels
//   grid  : x = pixel groups, y = COUT/CT channel tiles, z = 4 deconv quadrants (OMODE 2) else 1
// ------------------------------------------------------------------------------------------------
template <int CIN1, int CIN2, int COUT, int CT, int P, bool RELU, bool RES, int OMODE>
__global__ void __launch_bounds__(256) pw_kernel(PwArgs a)
{
    static_assert(CIN1 % 4 == 0 && CIN2 % 4 == 0 && CT % 4 == 0 && COUT % CT == 0, "shape");
    constexpr int CIN = CIN1 + CIN2;
    const int co0 = blockIdx.y * CT;
    const float* __restrict__ w = a.w + (OMODE == 2 ? (size_t)blockIdx.z * CIN * COUT : 0) + co0;
    const float* __restrict__ bias = a.b + co0;

    long pix[P];
    bool ok[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        pix[p] = (long)blockIdx.x * (256 * P) + p * 256 + threadIdx.x;
        ok[p] = pix[p] < a.npix;
        if (!ok[p]) pix[p] = 0;
    }
    float acc[P][CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        float bv = bias[c];
#pragma unroll
        for (int p = 0; p < P; ++p) acc[p][c] = bv;
    }
    {
        const float* __restrict__ in = a.in1;
        for (int ci = 0; ci < CIN1; ci += 4) {
            float4 x[P];
#pragma unroll
            for (int p = 0; p < P; ++p) x[p] = *reinterpret_cast<const float4*>(in + pix[p] * CIN1 + ci);
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    float wv = w[(size_t)(ci + cc) * COUT + c];
#pragma unroll
                    for (int p = 0; p < P; ++p) acc[p][c] = fmaf(((const float*)&x[p])[cc], wv, acc[p][c]);
                }
            }
        }
    }
    if constexpr (CIN2 > 0) {
        const float* __restrict__ in = a.in2;
        for (int ci = 0; ci < CIN2; ci += 4) {
            float4 x[P];
#pragma unroll
            for (int p = 0; p < P; ++p) x[p] = *reinterpret_cast<const float4*>(in + pix[p] * CIN2 + ci);
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    float wv = w[(size_t)(CIN1 + ci + cc) * COUT + c];
#pragma unroll
                    for (int p = 0; p < P; ++p) acc[p][c] = fmaf(((const float*)&x[p])[cc], wv, acc[p][c]);
                }
            }
        }
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
        if (!ok[p]) continue;
        if constexpr (RES) {
            const float* r = a.res + pix[p] * COUT + co0;
#pragma unroll
            for (int c = 0; c < CT; c += 4) {
                float4 rv = *reinterpret_cast<const float4*>(r + c);
                acc[p][c] += rv.x; acc[p][c + 1] += rv.y; acc[p][c + 2] += rv.z; acc[p][c + 3] += rv.w;
            }
        }
        if constexpr (RELU) {
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[p][c] = fmaxf(acc[p][c], 0.f);
        }
        if constexpr (OMODE == 1) {  // NCHW (the heads): out[(n*COUT+co)*HW + hw]
            long n = pix[p] / a.HW, hw = pix[p] - n * a.HW;
            float* o = a.out + (n * COUT + co0) * a.HW + hw;
#pragma unroll
            for (int c = 0; c < CT; ++c) o[(long)c * a.HW] = acc[p][c];
        } else {
            long opix = pix[p];
            if constexpr (OMODE == 2) {  // ConvTranspose2d k=2 s=2: quadrant (dy,dx) -> pixel (2y+dy, 2x+dx)
                long n = pix[p] / a.HW, hw = pix[p] - n * a.HW;
                int y = (int)(hw / a.W), x = (int)(hw - (long)y * a.W);
                int dy = blockIdx.z >> 1, dx = blockIdx.z & 1;
                opix = (n * (2 * (a.HW / a.W)) + 2 * y + dy) * (2 * a.W) + 2 * x + dx;
            }
            float* o = a.out + opix * COUT + co0;
#pragma unroll
            for (int c = 0; c < CT; c += 4)
                *reinterpret_cast<float4*>(o + c) = make_float4(acc[p][c], acc[p][c + 1], acc[p][c + 2], acc[p][c + 3]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Depthwise KxK convolution (+ReLU, every depthwise layer of the net has one), stride 1 or 2, pad (K-1)/2.
//   thread: one output pixel x 4 channels; lanes run over channel groups first (16-byte coalesced).
// ------------------------------------------------------------------------------------------------
template <int K, int S, typename T>
__global__ void __launch_bounds__(256) dw_kernel(DwArgs a)
{
    const int C4 = a.C >> 2;
    long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.total) return;
    int c4 = (int)(idx % C4);
    long opix = idx / C4;
    int ox = (int)(opix % a.Wo);
    long t = opix / a.Wo;
    int oy = (int)(t % a.Ho);
    long n = t / a.Ho;
    const T* __restrict__ in = reinterpret_cast<const T*>(a.in) + (n * a.H * a.W) * a.C + c4 * 4;
    const float* __restrict__ w = a.w + c4 * 4;
    float4 acc = *reinterpret_cast<const float4*>(a.b + c4 * 4);
    constexpr int PAD = (K - 1) / 2;
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
        int iy = oy * S - PAD + ky;
        if (iy < 0 || iy >= a.H) continue;
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            int ix = ox * S - PAD + kx;
            if (ix < 0 || ix >= a.W) continue;
            float4 x = ld4<T>(in + ((long)iy * a.W + ix) * a.C);
            float4 wv = *reinterpret_cast<const float4*>(w + (ky * K + kx) * a.C);
            acc.x = fmaf(x.x, wv.x, acc.x); acc.y = fmaf(x.y, wv.y, acc.y);
            acc.z = fmaf(x.z, wv.z, acc.z); acc.w = fmaf(x.w, wv.w, acc.w);
        }
    }
    acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
    st4<T>(reinterpret_cast<T*>(a.out) + opix * a.C + c4 * 4, acc);
}

// ------------------------------------------------------------------------------------------------
// Dense 3x3 stride-2 pad-1 convolution + ReLU (conv0: 1->8, conv1_9: 24->24).
//   thread: one output pixel x all COUT channels; weights [ky][kx][cin][cout] through the scalar path.
// ------------------------------------------------------------------------------------------------
template <int CIN, int COUT>
__global__ void __launch_bounds__(256) dense3x3s2_kernel(DenseArgs a)
{
    long opix = (long)blockIdx.x * 256 + threadIdx.x;
    bool ok = opix < a.total;
    if (!ok) opix = 0;
    int ox = (int)(opix % a.Wo);
    long t = opix / a.Wo;
    int oy = (int)(t % a.Ho);
    long n = t / a.Ho;
    const float* __restrict__ in = a.in + n * a.H * a.W * CIN;
    const float* __restrict__ w = a.w;
    float acc[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) acc[c] = a.b[c];
    for (int ky = 0; ky < 3; ++ky) {
        int iy = oy * 2 - 1 + ky;
        bool yok = iy >= 0 && iy < a.H;
        for (int kx = 0; kx < 3; ++kx) {
            int ix = ox * 2 - 1 + kx;
            bool v = yok && ix >= 0 && ix < a.W;
            const float* src = in + ((long)(v ? iy : 0) * a.W + (v ? ix : 0)) * CIN;
            const float* wt = w + (ky * 3 + kx) * CIN * COUT;
            if constexpr (CIN == 1) {
                float x = v ? src[0] : 0.f;
#pragma unroll
                for (int c = 0; c < COUT; ++c) acc[c] = fmaf(x, wt[c], acc[c]);
            } else {
                for (int ci = 0; ci < CIN; ci += 4) {
                    float4 x = *reinterpret_cast<const float4*>(src + ci);
                    if (!v) x = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
#pragma unroll
                        for (int c = 0; c < COUT; ++c)
                            acc[c] = fmaf(((const float*)&x)[cc], wt[(ci + cc) * COUT + c], acc[c]);
                    }
                }
            }
        }
    }
    if (!ok) return;
    float* o = a.out + opix * COUT;
#pragma unroll
    for (int c = 0; c < COUT; c += 4)
        *reinterpret_cast<float4*>(o + c) = make_float4(fmaxf(acc[c], 0.f), fmaxf(acc[c + 1], 0.f),
                                                         fmaxf(acc[c + 2], 0.f), fmaxf(acc[c + 3], 0.f));
}

// conv0 of an RGB model (io_params input_channel = 3, yolo_fastest.py:78): the net input is NCHW, i.e. three PLANES per frame, not
// NHWC.  thread: one output pixel x 8 channels; k order (ky, kx, ci) -- the order of the fused stem's conv0 (yf_fused_kernels.hip).
template <int C0>
__global__ void __launch_bounds__(256) conv0_planar_kernel(DenseArgs a)
{
    long opix = (long)blockIdx.x * 256 + threadIdx.x;
    if (opix >= a.total) return;
    const int ox = (int)(opix % a.Wo);
    long t = opix / a.Wo;
    const int oy = (int)(t % a.Ho);
    const long n = t / a.Ho;
    const long plane = (long)a.H * a.W;
    const float* __restrict__ in = a.in + n * C0 * plane;
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = a.b[c];
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * 2 - 1 + ky;
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox * 2 - 1 + kx;
            const bool v = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
#pragma unroll
            for (int ci = 0; ci < C0; ++ci) {
                const float x = v ? in[ci * plane + (long)iy * a.W + ix] : 0.f;
                const float* wt = a.w + ((ky * 3 + kx) * C0 + ci) * 8;
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[c] = fmaf(x, wt[c], acc[c]);
            }
        }
    }
    float* o = a.out + opix * 8;
    *reinterpret_cast<float4*>(o) = make_float4(fmaxf(acc[0], 0.f), fmaxf(acc[1], 0.f), fmaxf(acc[2], 0.f), fmaxf(acc[3], 0.f));
    *reinterpret_cast<float4*>(o + 4) = make_float4(fmaxf(acc[4], 0.f), fmaxf(acc[5], 0.f), fmaxf(acc[6], 0.f), fmaxf(acc[7], 0.f));
}

// conv0 for ANY number of input channels (io_params input_channel > 4: yolo_fastest.py:78 takes any; the fused stem kernel is instantiated
// for 1 .. 4): the same thread mapping and k order (ky, kx, ci) with a run-time channel loop; the 8-channel result goes out NHWC in the
// engine's storage type, so the fused plans can continue with a block kernel.
template <typename T>
__global__ void __launch_bounds__(256) conv0_any_kernel(DenseArgs a, int cin)
{
    long opix = (long)blockIdx.x * 256 + threadIdx.x;
    if (opix >= a.total) return;
    const int ox = (int)(opix % a.Wo);
    long t = opix / a.Wo;
    const int oy = (int)(t % a.Ho);
    const long n = t / a.Ho;
    const long plane = (long)a.H * a.W;
    const float* __restrict__ in = a.in + n * cin * plane;
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = a.b[c];
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * 2 - 1 + ky;
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox * 2 - 1 + kx;
            const bool v = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            for (int ci = 0; ci < cin; ++ci) {
                const float x = v ? in[ci * plane + (long)iy * a.W + ix] : 0.f;
                const float* wt = a.w + ((ky * 3 + kx) * cin + ci) * 8;
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[c] = fmaf(x, wt[c], acc[c]);
            }
        }
    }
    T* o = reinterpret_cast<T*>(a.out) + opix * 8;
    st4<T>(o, make_float4(fmaxf(acc[0], 0.f), fmaxf(acc[1], 0.f), fmaxf(acc[2], 0.f), fmaxf(acc[3], 0.f)));
    st4<T>(o + 4, make_float4(fmaxf(acc[4], 0.f), fmaxf(acc[5], 0.f), fmaxf(acc[6], 0.f), fmaxf(acc[7], 0.f)));
}

// Head conv of the per-layer plan for any Cout = num_anchors * (5 + num_cls) (yolo_fastest.py:138,148; the shipped 24 keeps its
// pw_kernel instantiation).  thread: one pixel x CT consecutive output channels (blockIdx.y = channel tile, wave-uniform -> the
// weights come through the scalar path); input NHWC in the engine's storage type, logits NCHW float32.
template <typename T, int CT>
__global__ void __launch_bounds__(256) head_conv_kernel(const T* __restrict__ in, const float* __restrict__ w, const float* __restrict__ b,
                                                        float* __restrict__ out, int cin, int cout, long HW, long total)
{
    const long pix = (long)blockIdx.x * 256 + threadIdx.x;   // over N * HW
    if (pix >= total) return;
    const int c0 = blockIdx.y * CT;
    float acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[c] = c0 + c < cout ? b[c0 + c] : 0.f;
    const T* __restrict__ x = in + pix * cin;
    for (int k = 0; k < cin; k += 4) {      // every 1x1 input of this net has a multiple of 4 channels
        const float4 v = ld4<T>(x + k);
        const float xv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int c = 0; c < CT; ++c)
                acc[c] = fmaf(xv[kk], c0 + c < cout ? w[(long)(k + kk) * cout + c0 + c] : 0.f, acc[c]);
    }
    const long n = pix / HW, hw = pix - n * HW;
#pragma unroll
    for (int c = 0; c < CT; ++c)
        if (c0 + c < cout) out[(n * cout + c0 + c) * HW + hw] = acc[c];
}

// NHWC -> NCHW copy for yf_forward_probe (test hook only).
template <typename T>
__global__ void __launch_bounds__(256) nhwc_to_nchw_kernel(const T* __restrict__ in, float* __restrict__ out,
                                                            long total, int C, long HW)
{
    long idx = (long)blockIdx.x * 256 + threadIdx.x;  // index into NCHW output
    if (idx >= total) return;
    long hw = idx % HW;
    long t = idx / HW;
    int c = (int)(t % C);
    long n = t / C;
    out[idx] = (float)in[(n * HW + hw) * C + c];
}

// Detect_YOLO.__pre_process arithmetic (src/detect.py:115-124): optional 2x2 box mean, then (v-128)/255.
__global__ void __launch_bounds__(256) preprocess_kernel(const uint8_t* __restrict__ in, float* __restrict__ out,
                                                          long total, int H, int W, int down2)
{
    long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    float v;
    if (down2) {
        int x = (int)(idx % W);
        long t = idx / W;
        int y = (int)(t % H);
        long n = t / H;
        const uint8_t* p = in + (n * 2 * H + 2 * y) * (2L * W) + 2 * x;
        v = (float)((p[0] + p[1] + p[2 * W] + p[2 * W + 1] + 2) >> 2);
    } else {
        v = (float)in[idx];
    }
    out[idx] = (v - 128.0f) / 255.0f;
}

// The same for multi-channel frames: `in` is what cv2.imread returns, HWC u8 [N,h,w,C] (C = 3: BGR); `out` NCHW float [N,C,H,W] with the
// channel order reversed (detect.py:119 `img[:, :, ::-1].transpose(2, 0, 1)`); the exact-2x resize is the 2x2 box mean per channel.
__global__ void __launch_bounds__(256) preprocess3_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, long total, int H, int W,
                                                           int down2, int C)
{
    long idx = (long)blockIdx.x * 256 + threadIdx.x;   // over the NCHW output
    if (idx >= total) return;
    const int x = (int)(idx % W);
    long t = idx / W;
    const int y = (int)(t % H);
    t /= H;
    const int c = (int)(t % C);
    const long n = t / C;
    const int sc = C - 1 - c;
    float v;
    if (down2) {
        const long sw = 2L * W * C;
        const uint8_t* p = in + (n * 2 * H + 2 * y) * sw + 2 * x * C + sc;
        v = (float)((p[0] + p[C] + p[sw] + p[sw + C] + 2) >> 2);
    } else {
        v = (float)in[((n * H + y) * W + x) * C + sc];
    }
    out[idx] = (v - 128.0f) / 255.0f;
}

// ------------------------------------------------------------------------------------------------
// Launchers
// ------------------------------------------------------------------------------------------------
template <int CIN1, int CIN2, int COUT, int CT, int P, bool RELU, bool RES, int OMODE>
static void launch_pw_t(const PwArgs& a, hipStream_t s)
{
    dim3 grid((unsigned)((a.npix + 256 * P - 1) / (256 * P)), COUT / CT, OMODE == 2 ? 4 : 1);
    hipLaunchKernelGGL((pw_kernel<CIN1, CIN2, COUT, CT, P, RELU, RES, OMODE>), grid, dim3(256), 0, s, a);
}

#define PW_CASE(ci1, ci2, co, ct, p, relu, res, om)                                              \
    if (cin1 == ci1 && cin2 == ci2 && cout == co && relu_ == relu && res_ == res && omode == om) { \
        launch_pw_t<ci1, ci2, co, ct, p, relu, res, om>(a, s);                                     \
        return 0;                                                                                  \
    }

int launch_pw(int cin1, int cin2, int cout, bool relu_, bool res_, int omode, const PwArgs& a, hipStream_t s)
{
    // (cin1, cin2, cout, CT, P, relu, residual, omode) -- one line per distinct layer shape of the net
    PW_CASE(8, 0, 8, 8, 4, true, false, 0)       // conv1_2
    PW_CASE(8, 0, 4, 4, 4, false, false, 0)      // conv1_4
    PW_CASE(4, 0, 8, 8, 4, true, false, 0)       // res1_1.conv1
    PW_CASE(8, 0, 4, 4, 4, false, true, 0)       // res1_1.conv3
    PW_CASE(4, 0, 24, 24, 4, true, false, 0)     // conv1_8
    PW_CASE(24, 0, 8, 8, 4, false, false, 0)     // conv2_1
    PW_CASE(8, 0, 32, 32, 2, true, false, 0)     // res2_x.conv1, conv2_2
    PW_CASE(32, 0, 8, 8, 4, false, true, 0)      // res2_x.conv3
    PW_CASE(32, 0, 8, 8, 4, false, false, 0)     // conv3_1
    PW_CASE(8, 0, 48, 48, 2, true, false, 0)     // res3_{1,2}.conv1, conv3_2
    PW_CASE(48, 0, 8, 8, 4, false, true, 0)      // res3_{1,2}.conv3
    PW_CASE(48, 0, 16, 16, 4, false, false, 0)   // conv3_4
    PW_CASE(16, 0, 96, 48, 2, true, false, 0)    // res3_{3..6}.conv1, conv3_5
    PW_CASE(96, 0, 16, 16, 4, false, true, 0)    // res3_{3..6}.conv3
    PW_CASE(96, 0, 24, 24, 2, false, false, 0)   // conv4_1
    PW_CASE(24, 0, 136, 68, 1, true, false, 0)   // res4_x.conv1, conv4_2
    PW_CASE(136, 0, 24, 24, 2, false, true, 0)   // res4_x.conv3
    PW_CASE(136, 0, 48, 24, 1, true, false, 0)   // conv5_1
    PW_CASE(48, 0, 224, 32, 1, true, false, 0)   // res5_x.conv1
    PW_CASE(224, 0, 48, 16, 1, false, true, 0)   // res5_x.conv3
    PW_CASE(48, 0, 96, 32, 1, true, false, 0)    // conv5_2
    PW_CASE(96, 0, 128, 32, 1, false, false, 0)  // conv5_4
    PW_CASE(128, 0, 128, 32, 1, false, false, 0) // conv5_6
    PW_CASE(128, 0, 24, 8, 1, false, false, 1)   // head_5 (NCHW out)
    PW_CASE(96, 0, 96, 32, 1, true, false, 2)    // deconv5_1 (4 quadrants)
    PW_CASE(136, 96, 96, 48, 2, true, false, 0)  // conv4_1_1 over cat(conv4_2, deconv5_1)
    PW_CASE(96, 0, 96, 48, 2, false, false, 0)   // conv4_1_3, conv4_1_5
    PW_CASE(96, 0, 24, 24, 2, false, false, 1)   // head_4 (NCHW out)
    return -1;
}

template <typename T>
static int launch_dw_t(int k, int stride, const DwArgs& a, hipStream_t s)
{
    dim3 grid((unsigned)((a.total + 255) / 256));
    if (k == 3 && stride == 1) hipLaunchKernelGGL((dw_kernel<3, 1, T>), grid, dim3(256), 0, s, a);
    else if (k == 3 && stride == 2) hipLaunchKernelGGL((dw_kernel<3, 2, T>), grid, dim3(256), 0, s, a);
    else if (k == 5 && stride == 1) hipLaunchKernelGGL((dw_kernel<5, 1, T>), grid, dim3(256), 0, s, a);
    else return -1;
    return 0;
}

int launch_dw(int k, int stride, const DwArgs& a, hipStream_t s, int dtype)
{
    return dtype == DT_F16 ? launch_dw_t<half_t>(k, stride, a, s) : launch_dw_t<float>(k, stride, a, s);
}

int launch_dense3x3s2(int cin, int cout, const DenseArgs& a, hipStream_t s, int out_dtype)
{
    dim3 grid((unsigned)((a.total + 255) / 256));
    if (cout == 8 && (cin > 4 || out_dtype == DT_F16)) {   // conv0 of a model with more than 4 input channels (any plan), fp32 or fp16 storage out
        if (cin < 1) return -1;
        if (out_dtype == DT_F16) hipLaunchKernelGGL(conv0_any_kernel<half_t>, grid, dim3(256), 0, s, a, cin);
        else hipLaunchKernelGGL(conv0_any_kernel<float>, grid, dim3(256), 0, s, a, cin);
        return 0;
    }
    if (cin == 1 && cout == 8) hipLaunchKernelGGL((dense3x3s2_kernel<1, 8>), grid, dim3(256), 0, s, a);
    else if (cin == 24 && cout == 24) hipLaunchKernelGGL((dense3x3s2_kernel<24, 24>), grid, dim3(256), 0, s, a);
    else if (cin == 2 && cout == 8) hipLaunchKernelGGL(conv0_planar_kernel<2>, grid, dim3(256), 0, s, a);
    else if (cin == 3 && cout == 8) hipLaunchKernelGGL(conv0_planar_kernel<3>, grid, dim3(256), 0, s, a);
    else if (cin == 4 && cout == 8) hipLaunchKernelGGL(conv0_planar_kernel<4>, grid, dim3(256), 0, s, a);
    else return -1;
    return 0;
}

int launch_head_conv(const float* in, const float* w, const float* b, float* out, int cin, int cout, long HW, int N, hipStream_t s, int dtype)
{
    if (cin % 4 || cout <= 0) return -1;
    constexpr int CT = 8;
    const long total = (long)N * HW;
    const dim3 grid((unsigned)((total + 255) / 256), (unsigned)((cout + CT - 1) / CT));
    if (dtype == DT_F16)
        hipLaunchKernelGGL((head_conv_kernel<half_t, CT>), grid, dim3(256), 0, s, reinterpret_cast<const half_t*>(in), w, b, out, cin, cout, HW, total);
    else
        hipLaunchKernelGGL((head_conv_kernel<float, CT>), grid, dim3(256), 0, s, in, w, b, out, cin, cout, HW, total);
    return 0;
}

void launch_nhwc_to_nchw(const float* in, float* out, long N, int C, long HW, hipStream_t s, int dtype)
{
    long total = N * C * HW;
    if (dtype == DT_F16)
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<half_t>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                           reinterpret_cast<const half_t*>(in), out, total, C, HW);
    else
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, out, total, C, HW);
}

void launch_preprocess(const uint8_t* in, float* out, long N, int H, int W, int down2, hipStream_t s, int channels)
{
    long total = N * H * W * channels;
    if (channels > 1)
        hipLaunchKernelGGL(preprocess3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, out, total, H, W, down2, channels);
    else
        hipLaunchKernelGGL(preprocess_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, out, total, H, W, down2);
}

}  // namespace yf
