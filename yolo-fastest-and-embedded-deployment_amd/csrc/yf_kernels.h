// yf_kernels.h -- internal launch interface between the engine (yf_engine.hip) and the kernel files.
#pragma once
#include <vector>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace yf {

// ---- activation storage types: fp32 (BASELINE configs[1]) or fp16 (configs[2]); all arithmetic accumulates in fp32 ----
//   DT_F32   fp32 storage, fp32 arithmetic (v_mfma_f32_16x16x4_f32 / VALU): BASELINE configs[1]
//   DT_F16   fp16 storage in HBM, single fp16 MFMA operands: the fastest, but 11-bit operands cost up to 8e-2 on the logits
//   DT_F16X3 fp32 storage (HBM and LDS), the pointwise GEMMs on the fp16 matrix pipe with SPLIT operands: a = hi + lo (two fp16
//            halves, lo = fp16(a - hi) computed exactly in fp32), products w_lo*a_hi + w_hi*a_lo + w_hi*a_hi accumulated in fp32 --
//            22-bit operands, as accurate as the fp32 path (tools/mfma16_probe.hip, tools/fp16_sim.py), at 3 fp16 MFMAs (17 cycles
//            each for K = 16) instead of 4 fp32 MFMAs (32 cycles each for K = 4)
enum { DT_F32 = 0, DT_F16 = 1, DT_F16X3 = 2 };
enum { WM_F32 = 0, WM_F16 = 1, WM_F16X3 = 2 };   // host-side weight-fragment packing of the MFMA kernels
typedef _Float16 half_t;
struct x3_t { float v; };   // storage tag of DT_F16X3: a float in memory; kernels instantiated on it use the split-fp16 MFMAs
template <typename T> struct is_x3 { static constexpr bool value = false; };
template <> struct is_x3<x3_t> { static constexpr bool value = true; };
template <typename T> constexpr int wmode_of() { return is_x3<T>::value ? WM_F16X3 : sizeof(T) == 2 ? WM_F16 : WM_F32; }
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
#ifdef __HIPCC__
template <typename T> __device__ __forceinline__ float4 ld4(const T* p);
template <> __device__ __forceinline__ float4 ld4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <> __device__ __forceinline__ float4 ld4<x3_t>(const x3_t* p) { return *reinterpret_cast<const float4*>(p); }
template <> __device__ __forceinline__ float4 ld4<half_t>(const half_t* p)
{
    const f16x4 v = *reinterpret_cast<const f16x4*>(p);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
template <typename T> __device__ __forceinline__ float2 ld2(const T* p);
template <> __device__ __forceinline__ float2 ld2<float>(const float* p) { return *reinterpret_cast<const float2*>(p); }
template <> __device__ __forceinline__ float2 ld2<x3_t>(const x3_t* p) { return *reinterpret_cast<const float2*>(p); }
template <> __device__ __forceinline__ float2 ld2<half_t>(const half_t* p)
{
    const f16x2 v = *reinterpret_cast<const f16x2*>(p);
    return make_float2((float)v[0], (float)v[1]);
}
template <typename T> __device__ __forceinline__ void st4(T* p, float4 v);
template <> __device__ __forceinline__ void st4<float>(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
template <> __device__ __forceinline__ void st4<x3_t>(x3_t* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
template <> __device__ __forceinline__ void st4<half_t>(half_t* p, float4 v)
{
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    *reinterpret_cast<f16x4*>(p) = __builtin_convertvector((f32x4v{v.x, v.y, v.z, v.w}), f16x4);  // 2 x v_cvt_pk_f16_f32 (RNE)
}
template <typename T> __device__ __forceinline__ float ld1(const T* p) { return (float)*p; }
template <> __device__ __forceinline__ float ld1<x3_t>(const x3_t* p) { return p->v; }
template <typename T> __device__ __forceinline__ void st1(T* p, float v) { *p = (T)v; }
template <> __device__ __forceinline__ void st1<x3_t>(x3_t* p, float v) { p->v = v; }
// split fp32 values into fp16 hi and lo halves: hi = rne(a), lo = rne(a - hi); a - hi is exact in fp32.
// YF_X3_MIX (round 6): lo through v_fma_mixlo_f16 / v_fma_mixhi_f16 -- fma(hi as f16, -1.0, a) rounded once to f16 and written straight into a
// half of the packed register: one instruction per value where the C form costs 2.5 (v_cvt_f32_f16, a packed subtraction per pair, a
// v_cvt_pk_f16_f32 per pair).  The same bits: a - hi is exact, both forms round the same real number once (tools/mix_probe.hip: 2 M pairs,
// normal and denormal results, bit-identical).  No builtin exists, so this is inline asm -- and the compiler's hazard recogniser does not see
// inside it: an MFMA that reads the result in the next cycle got the STALE register (VALU write -> MFMA read needs wait states the compiler
// inserts only for instructions it knows to be VALU; test_deep_stage_fusion_is_bitwise_neutral caught it as two fusion levels disagreeing).
// Hence the trailing s_nop in the asm block: YF_X3_MIX_NOP wait states (2 = what the compiler places for its own VALU writes).
// Measured with the s_nop in place (tools/x3_ab.sh, 640x512 batch 128 f16x3, two interleaved rounds): conv1_8+conv1_9+conv2_1 198-201 -> 190-192 us,
// deconv5_1+conv4_1_1 46.0 -> 48.4 (its compiler-scheduled form was the tighter one), every other launch +-1 %: launch sum 1410 -> 1411-1415 us.
// No gain for the pass: OFF.
#ifndef YF_X3_MIX
#define YF_X3_MIX 0
#endif
#ifndef YF_X3_MIX_NOP
#define YF_X3_MIX_NOP 1    // s_nop operand: N + 1 wait states
#endif
#define YF_STR2(x) #x
#define YF_STR(x) YF_STR2(x)
__device__ __forceinline__ void split_f16x2(float a0, float a1, f16x2& hi, f16x2& lo)
{
    typedef float f32x2v __attribute__((ext_vector_type(2)));
    const f32x2v a = {a0, a1};
    hi = __builtin_convertvector(a, f16x2);
#if YF_X3_MIX
    unsigned r, h = __builtin_bit_cast(unsigned, hi);
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "s_nop " YF_STR(YF_X3_MIX_NOP)
        : "=&v"(r) : "v"(h), "v"(a0), "v"(a1));
    lo = __builtin_bit_cast(f16x2, r);
#else
    lo = __builtin_convertvector(a - __builtin_convertvector(hi, f32x2v), f16x2);
#endif
}
__device__ __forceinline__ void split_f16x4(float a0, float a1, float a2, float a3, f16x4& hi, f16x4& lo)
{
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    const f32x4v a = {a0, a1, a2, a3};
    hi = __builtin_convertvector(a, f16x4);
#if YF_X3_MIX
    typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
    const u32x2v h = __builtin_bit_cast(u32x2v, hi);
    unsigned r0, r1;
    asm("v_fma_mixlo_f16 %0, %2, -1.0, %4 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %1, %3, -1.0, %6 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %2, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %1, %3, -1.0, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "s_nop " YF_STR(YF_X3_MIX_NOP)
        : "=&v"(r0), "=&v"(r1) : "v"(h[0]), "v"(h[1]), "v"(a0), "v"(a1), "v"(a2), "v"(a3));
    lo = __builtin_bit_cast(f16x4, u32x2v{r0, r1});
#else
    lo = __builtin_convertvector(a - __builtin_convertvector(hi, f32x4v), f16x4);
#endif
}
#endif

#ifdef __HIPCC__
// Copy NFLOATS (a multiple of 4) floats from global memory to LDS with ALL of a thread's 16-byte loads in flight before its first LDS
// write.  Written as the obvious rolled loop (load, store, next) the compiler waits for every load before its store: one L2 round
// trip per iteration, up to 22 in a row at the top of a kernel (measured: the "10-13 us floor" of the per-frame head kernels).
template <int NFLOATS, int NTHR>
struct LdsStage {   // issue() the loads early, commit() the LDS writes when convenient (other loads may be requested in between)
    typedef float stage_f32x4 __attribute__((ext_vector_type(4)));
    static_assert(NFLOATS % 4 == 0, "16-byte pieces");
    static constexpr int N4 = NFLOATS / 4, NI = (N4 + NTHR - 1) / NTHR;
    stage_f32x4 v[NI];
    __device__ __forceinline__ void issue(const float* __restrict__ src)
    {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int idx = threadIdx.x + i * NTHR;
            v[i] = *reinterpret_cast<const stage_f32x4*>(src + 4 * (idx < N4 ? idx : N4 - 1));
        }
    }
    __device__ __forceinline__ void commit(float* __restrict__ dst) const
    {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int idx = threadIdx.x + i * NTHR;
            if (idx < N4) *reinterpret_cast<stage_f32x4*>(dst + 4 * idx) = v[i];
        }
    }
};
template <int NFLOATS, int NTHR>
__device__ __forceinline__ void stage_to_lds(float* __restrict__ dst, const float* __restrict__ src)
{
    LdsStage<NFLOATS, NTHR> st;
    st.issue(src);
    st.commit(dst);
}
#endif

// Workgroups are dealt round-robin to the 8 XCDs of the chip (block i -> XCD i % 8), each with its own L2.  xcd_tile() gives XCD x the
// CONTIGUOUS range [x * per, (x + 1) * per) of logical tiles, so that neighbouring tiles of a frame -- which share halo rows and
// columns -- are fetched through one L2 instead of up to eight.  The ragged tail (grid % 8) keeps its index.  -DYF_XCD_SWIZZLE=0: off.
#ifndef YF_XCD_SWIZZLE
#define YF_XCD_SWIZZLE 1
#endif
#ifdef __HIPCC__
__device__ __forceinline__ int xcd_tile(unsigned b, unsigned nb)
{
#if YF_XCD_SWIZZLE
    const unsigned per = nb >> 3, main = per << 3;
    return b < main ? (int)((b & 7) * per + (b >> 3)) : (int)b;
#else
    (void)nb;
    return (int)b;
#endif
}
#endif

// hipFuncSetAttribute (dynamic LDS above 64 KiB) and the CU count belong to (kernel, DEVICE), not to the process: launchers keep
// their "done once" state per device so that a second GPU in the same process gets the attribute too.
enum { YF_MAX_DEVICES = 64 };
inline int current_device()
{
    int d = -1;
    return (hipGetDevice(&d) == hipSuccess && d >= 0 && d < YF_MAX_DEVICES) ? d : -1;
}
inline int device_cu_count(int dev)   // cached per device; <= 0 on failure
{
    static int n[YF_MAX_DEVICES] = {};
    if (dev < 0) return -1;
    if (n[dev] == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return -1;
        n[dev] = v;
    }
    return n[dev];
}

struct PwArgs {
    const float* in1;  // NHWC [npix, CIN1]
    const float* in2;  // NHWC [npix, CIN2] (second half of a channel concat) or null
    const float* w;    // [cin][cout] (deconv: [4][cin][cout])
    const float* b;    // [cout]
    const float* res;  // NHWC [npix, COUT] residual or null
    float* out;
    long npix;         // N*H*W input pixels
    long HW;           // pixels per frame (input resolution of this layer)
    int W;             // row length (input resolution of this layer)
};

struct DwArgs {
    const float* in;  // NHWC [N,H,W,C]
    const float* w;   // [K*K][C]
    const float* b;   // [C]
    float* out;       // NHWC [N,Ho,Wo,C]
    long total;       // N*Ho*Wo*C/4 threads
    int C, H, W, Ho, Wo;
};

struct DenseArgs {
    const float* in;  // NHWC [N,H,W,CIN]
    const float* w;   // [3][3][cin][cout]
    const float* b;
    float* out;       // NHWC [N,Ho,Wo,COUT]
    long total;       // N*Ho*Wo
    int H, W, Ho, Wo;
};

int launch_pw(int cin1, int cin2, int cout, bool relu, bool res, int omode, const PwArgs& a, hipStream_t s);
int launch_dw(int k, int stride, const DwArgs& a, hipStream_t s, int dtype = DT_F32);
// matrix-core pointwise GEMM (yf_mfma_kernels.hip); a.w points at the layer's PACKED B fragments
int launch_pw_mfma(int cin1, int cin2, int cout, bool relu, bool res, int omode, const PwArgs& a, hipStream_t s, int dtype = DT_F32);
uint16_t f32_to_f16_bits(float f);
// WM_F16X3 packing: the lo half of a weight, rne(w - (float)rne_f16(w)), as fp16 bits (the hi half is f32_to_f16_bits(w))
inline uint16_t f16_lo_bits(float w)
{
    const uint16_t h = f32_to_f16_bits(w);
    const int e = (h >> 10) & 31, m = h & 1023;
    float hf = e == 0 ? ldexpf((float)m, -24) : e == 31 ? (m ? NAN : INFINITY) : ldexpf((float)(m | 1024), e - 25);
    if (h & 0x8000) hf = -hf;
    return f32_to_f16_bits(w - hf);
}
size_t mfma_packed_floats_f16(int k1, int k2, int n);
void mfma_pack_weights_f16(const float* w, int k1, int k2, int n, float* out);
bool mfma_has_kernel(int cin1, int cin2, int cout, bool relu, bool res, int omode);
bool mfma_has_x3_kernel(int cin1, int cin2, int cout, bool relu, bool res, int omode);   // DT_F16X3 instantiation exists
size_t mfma_packed_floats_x3(int k1, int k2, int n);
void mfma_pack_weights_x3(const float* w, int k1, int k2, int n, float* out);
size_t mfma_packed_floats(int k1, int k2, int n);
void mfma_pack_weights(const float* w, int k1, int k2, int n, float* out);
int launch_dense3x3s2(int cin, int cout, const DenseArgs& a, hipStream_t s, int out_dtype = DT_F32);   // cin 3 (conv0 on RGB): `in` is NCHW planes [N,3,H,W]
// head conv of the per-layer plan for any Cout = num_anchors * (5 + num_cls) (yolo_fastest.py:138,148): NHWC [N,HW,cin] (storage
// dtype) x w[cin][cout] + b -> NCHW float32 [N,cout,HW]; fp32 arithmetic in every engine dtype
int launch_head_conv(const float* in, const float* w, const float* b, float* out, int cin, int cout, long HW, int N, hipStream_t s, int dtype = DT_F32);
void launch_nhwc_to_nchw(const float* in, float* out, long N, int C, long HW, hipStream_t s, int dtype = DT_F32);
// cv2.cvtColor(BGR2GRAY) + cv2.resize of detect.py:110-116 for any source size (yf_cv_kernels.hip).  u8 in, u8 out.
struct CvArgs {
    const uint8_t* src; uint8_t* dst;
    int n, sh, sw, sc;          // frames, source rows / columns / channels (1, or 3 = cv2.imread's BGR)
    int dh, dw, dc;             // destination (the net's input) rows / columns / channels
    int mode;                   // 0: same size, 1: exactly 1/2 (2x2 mean), 2: INTER_LINEAR through the tables
    int gray;                   // 0: channels kept (sc == dc); 14 | 15: BGR -> gray with OpenCV's 14- / 15-bit coefficients (sc 3, dc 1)
    const int4* xtab;           // [dw] {sx, sx + 1 clamped, a0, a1}   (mode 2)
    const int4* ytab;           // [dh] {y0, y1, b0, b1}
};
int launch_cv_pre(const CvArgs& a, hipStream_t s);
// cv::resize's xofs / ialpha and yofs / ibeta tables for one source size, built ON THE DEVICE (stream-ordered, no allocation, capturable)
void launch_cv_tables(int src_h, int src_w, int dst_h, int dst_w, int4* xtab, int4* ytab, hipStream_t s);
void launch_preprocess(const uint8_t* in, float* out, long N, int H, int W, int down2, hipStream_t s, int channels = 1);   // channels 3: HWC BGR in, NCHW RGB planes out

struct FbArgs {
    const float* in;              // NHWC [N,H,W,CIN]; PRE: the net input, NCHW planes [N,C0,2H,2W] (C0 = io_params input_channel: 1 .. 4)
    const float *w0, *b0;         // PRE only: conv0 [9][C0][8], [8]
    const float* wp;              // chunk-major weight stream of expand / depthwise / project (fb_pack_weights)
    float* out;                   // NHWC [N,Ho,Wo,COUT]
    int H, W, Ho, Wo;             // expansion-resolution and output-resolution frame sizes
    int tiles_y, tiles_x;         // filled by the launcher
    const uint8_t* in_u8;         // PRE only, optional: u8 frames instead of `in` (pre-process fused into the load): gray [N,h,w], or --
                                  // C0 = 3 -- cv2's HWC BGR [N,h,w,3], channel-flipped on the fly (detect.py:119)
    int u8_down2;                 //   1: the u8 frame is exactly 2x the net input (2x2 box mean first)
    unsigned long long* dbg;      // diagnostic builds only (-DYF_STAMP): per-phase cycle sums; null in the product
};
// pre: 0 = no conv0 in front; 1 / 3 = conv0 on that many input channels evaluated on the fly
int launch_fused_block(int cin, int cexp, int cout, int stride, bool res, bool relu_out, int pre, const FbArgs& a, int N,
                       hipStream_t s, int dtype = DT_F32);
int fb_chunk_channels(int cin, int cexp, int cout, int stride, bool res, bool relu_out, int pre);  // EC of that shape, -1 if none
size_t fb_packed_floats(int cin, int cexp, int cout, int ec);
void fb_pack_weights(const float* w1, const float* b1, const float* wd, const float* bd, const float* w2, const float* b2, int cin,
                     int cexp, int cout, int ec, float* out);

struct MresArgs {
    const float* in;   // NHWC [N,H,W,CIN]
    const float* wp;   // host-packed weight stream (mres_pack_weights)
    float* out;        // NHWC [N,H,W,COUT]
    int H, W;
    int tiles_y, tiles_x;  // filled by the launcher
    float* out_exp;    // optional: the EXPANDED tensor (after ReLU) is also written, NHWC [N,H,W,CEXP] -- conv4_2 is a skip tensor
    int nblk;          // > 1: a chain of that many residual blocks of this shape in one launch (tile == frame only)
    long wstride;      // floats between the packed weight streams of consecutive chained blocks
    const float* post_w;  // optional (mres_has_post shapes): a 1x1 conv + ReLU applied to the last block's result on chip --
                          // mfma_pack_weights fragments followed by the bias; the block's own result is then NOT stored
    float* post_out;      // NHWC [N,H,W,POSTN]
    float* esplit;        // optional: mres_esplit_scratch_floats() floats of engine-owned scratch (the stride-32 chain at a handful of frames)
};
size_t mres_esplit_scratch_floats();
int launch_mres(int cin, int cexp, int cout, bool res, int stride, const MresArgs& a, int N, hipStream_t s, int dtype = DT_F32);
int mres_dispatches(int cin, int cexp, int cout, bool res, int stride, int nblk, bool has_scratch, bool has_post, int H, int W, int N, int dtype, bool esplit);
bool mres_has_kernel(int cin, int cexp, int cout, bool res, int stride = 1, bool relu_out = false, int dtype = DT_F32);
bool mres_can_chain(int cin, int cexp, int cout, int H, int W);  // relu_out: ReLU after the projection
bool mres_has_post(int cin, int cexp, int cout, int postn);      // a trailing 1x1 conv (+ReLU) of postn channels can ride in the launch
size_t mres_post_packed_floats(int cout, int postn);
size_t mres_packed_floats(int cin, int cexp, int cout, int wmode = WM_F32);
void mres_pack_weights(const float* w1, const float* b1, const float* wd, const float* bd, const float* w2, const float* b2,
                       int cin, int cexp, int cout, float* out, int wmode = WM_F32);

struct MdwArgs {
    const float* in;   // NHWC [N,H,W,C]
    const float* wp;   // host-packed weight stream (mdw_pack_weights)
    float* out;        // NHWC [N,H,W,N_out], or NCHW [N,headn,H,W] when the head conv is fused
    int H, W;
    int tiles_y, tiles_x;
    int headn;         // fused head conv: its output channels = num_anchors * (5 + num_cls) (filled by the launcher)
};
int launch_mdw(int c, int n, int headn, const MdwArgs& a, int Nf, hipStream_t s, int dtype = DT_F32);
bool mdw_has_kernel(int c, int n, int headn);
// the small head's two pairs as ONE launch (frames that fit one tile): stage 1 (c1 -> n1, no head) feeds stage 2 (n1 -> n2 -> head) in LDS
bool mdw2_can_chain(int c1, int n1, int n2, int headn, int H, int W);
int launch_mdw2(const float* in, const float* wp1, const float* wp2, float* out_nchw, int H, int W, int headn, int Nf, hipStream_t s, int dtype = DT_F32,
                float* esplit_scratch = nullptr);   // scratch: mdw2_esplit_scratch_floats() floats (a handful of frames: three launches over 6 / 8 / 1 workgroups per frame)
size_t mdw2_esplit_scratch_floats();
bool mdw2_esplit_ok(int H, int W, int headn, int Nf, int dtype, const float* scratch);
size_t mdw_packed_floats(int c, int n, int headn, int wmode = WM_F32);
void mdw_pack_weights(const float* wd, const float* bd, const float* w, const float* b, const float* hw, const float* hb, int c,
                      int n, int headn, float* out, int wmode = WM_F32);

// deconv5_1 + conv4_1_1 over cat(conv4_2, deconv5_1) as one launch (yf_dcat_kernels.hip)
bool dcat_has_kernel(int cin, int cskip, int cout);
size_t dcat_packed_floats();
void dcat_pack_weights(const float* w_cat /*[136 + 96][96]*/, const float* b_deconv, const float* b_conv, float* out);
int launch_dcat(const float* x /*conv5_2*/, const float* skip /*conv4_2*/, const float* w_deconv /*4 x mfma_pack_weights[_x3]*/, const float* w_conv,
                float* out, int h, int w, int Nf, hipStream_t s, int dtype = DT_F32);
size_t dcat_packed_floats_f16();  // DT_F16: fp16 fragments (the deconv fragments: 4 x mfma_pack_weights_f16)
void dcat_pack_weights_f16(const float* w_cat, const float* b_deconv, const float* b_conv, float* out);
size_t dcat_packed_floats_x3();   // DT_F16X3: the split-operand stream
void dcat_pack_weights_x3(const float* w_cat, const float* b_deconv, const float* b_conv, float* out);

struct K19Args {
    const float* in;              // NHWC [N,H,W,4] (res1_1 output, stride-2 resolution)
    const float *w8, *b8;         // conv1_8 [4][24]
    const float *w9, *b9;         // conv1_9 [3][3][24][24]
    const float *w21, *b21;       // conv2_1 [24][8]
    float* out;                   // NHWC [N,Ho,Wo,8]
    int H, W, Ho, Wo;
    int tiles_y, tiles_x;
    const float* wp;              // k19m: MFMA A fragments of conv1_9 and conv2_1 (k19_pack_weights)
    int n_frames;
};
int launch_k19(K19Args a, int N, hipStream_t s, int dtype = DT_F32);    // VALU version (kept for tools/kbench.hip)
int launch_k19m(K19Args a, int N, hipStream_t s, int dtype = DT_F32);   // matrix-core version (the plan's)
size_t k19_packed_floats(int wmode = WM_F32);
size_t k19m_lds_bytes(int dtype);
size_t k19m_guard_elems(int W);   // the engine keeps this many elements free before and after the workspace slots
void k19_pack_weights(const float* w9, const float* w21, float* out, int wmode = WM_F32);

enum { POST_MAX_ANCHORS = 8 };   // anchors per head the post-process kernels take (the reference ships 3)
struct PostArgs {
    const float* head_large;  // [N,na*(5+nc),hl,wl]
    const float* head_small;  // [N,na*(5+nc),hs,ws]
    int na, nc;               // io_params num_anchors, num_cls (detect.py:15-21)
    int hl, wl, hs, ws;
    int in_h, in_w;           // net-input rows/cols
    float logit_min;          // smallest fp32 conf logit the reference's `sigmoid(t) > conf_thres` accepts
    double nms_thres;
    double anchors[2 * POST_MAX_ANCHORS * 2];   // [2][na][2]
    double adj_h, adj_w;      // __adjust_coord scales (0 = no adjustment)
    int kmax;
    int32_t* boxes;           // [N,kmax,4]
    float* scores;            // [N,kmax,2]
    int32_t* cls;             // [N,kmax]
    int32_t* src;             // [N,kmax]
    int32_t* counts;          // [N]
    // optional, instead of the five arrays above: ONE packed record row per frame, int32 [N, 1 + 8 kmax] =
    // count | boxes (4 kmax) | conf, cls_score as float bits (2 kmax) | cls (kmax) | src (kmax) -- the layout of the multi-GPU exchange
    // (dist.pack_records), so that the all-gather sends the kernel's own buffer
    int32_t* records;
    // optional: scratch of post_split_kernel + post_assemble_kernel -- one workgroup per (frame, class) instead of one per frame: dense frames.
    // split_tmp int32 [N, nc, 2 + 5 kmax] (post_split_tmp_ints)
    int32_t* split_tmp;
};
int launch_post(const PostArgs& a, int N, hipStream_t s);
size_t post_split_tmp_ints(int N, int nc, int kmax);
size_t post_lds_bytes(int ncell);
// anc: na (w, h) pairs in feature-map units; rows of `out` / `pred` are 5 + nc floats
void launch_val_decode(const float* in, float* out, int N, int h, int w, int M_total, int m_off, const float* anc, int na, int nc, float stride_w,
                       float stride_h, hipStream_t s);
int launch_val_nms(const float* pred, int N, int M, int nc, float conf_thres, float nms_thres, int kmax, float* det, int32_t* counts, hipStream_t s);
// training-time loss of one head and its gradient with respect to the head tensor (yf_loss_kernels.hip)
size_t train_loss_workspace_bytes(int N, int fh, int fw, int na = 3, int nc = 3);
void launch_train_loss(const float* head, int N, int fh, int fw, const float* anc, int na, int nc, const float* targets, int T, float ignore_thres,
                       void* work, float* losses, float* grad, hipStream_t s);
// training-step operators (yf_train_kernels.hip): NCHW fp32, correctness-first
// BatchNorm's partial sums out of the conv's epilogue (the large maps): the caller sets part / cap_bytes (a region no concurrent kernel
// uses: behind BatchNorm's own megabyte of the scratch), the conv launcher sets count (> 0: it left count pairs per channel in
// part[channel][block]), launch_tbn_fwd then adds those instead of reading z for its statistics
struct TStatPart { float2* part; size_t cap_bytes; long count; };
void launch_tconv_fwd(const float* x, const float* w, const float* bias, float* y, int N, int Cin, int H, int W, int Cout, int k, int stride,
                      int depthwise, hipStream_t s, TStatPart* st = nullptr);
// The backward's BatchNorm sums of the layer BELOW out of the data-gradient kernel that produces its dy (the large maps): dx of this
// layer IS dy of the layer whose output it consumed, so the epilogue reads that layer's z (the one extra pass) and leaves (sum dy_eff,
// sum dy_eff xhat) pairs per channel; launch_tbn_bwd adds them instead of reading dy and z for its reduction.  The caller fills
// z / stats / gamma / beta / relu of the layer below and part / cap_bytes; the launcher sets count (> 0: pairs left).
struct TBnRed { const float* z; const float* stats; const float* gamma; const float* beta; int relu; float2* part; size_t cap_bytes; long count; };
void launch_tconv_bwd_data(const float* dy, const float* w, float* dx, int N, int Cin, int H, int W, int Cout, int k, int stride, int depthwise,
                           hipStream_t s, const float* addend = nullptr, TBnRed* red = nullptr);
// split weight-gradient reductions, summed once per pass: a layer's launcher puts its slabs into `slab` and records an entry instead of
// launching its own sum; launch_tsum_multi adds them all (offsets relative to slab / dst_base, so the table is the same every iteration)
struct TSumEntry { long part_off, dst_off, nw, blk0; int nsplit, spl; };
struct TSumDefer {
    float* slab; size_t cap_floats, used; float* dst_base; long nblocks;
    std::vector<TSumEntry> entries;
    float* take(long floats);
    void push(const float* part, long nsplit, long nw, float* dw);
};
void launch_tsum_multi(const TSumEntry* d_tab, int n, long nblocks, const float* slab, float* dst, hipStream_t s);
void launch_tconv_bwd_weight(const float* x, const float* dy, float* dw, int N, int Cin, int H, int W, int Cout, int k, int stride, int depthwise,
                             void* scratch, size_t scratch_bytes, hipStream_t s, TSumDefer* defer = nullptr);
bool launch_tpw_bwd_dual(const float* x, const float* dz, const float* w, float* dw, float* dx, const float* addend, int N, int Cin, int H, int W,
                         int Cout, void* scratch, size_t scratch_bytes, hipStream_t s, TSumDefer* defer);
void launch_tdeconv_fwd(const float* x, const float* w, float* y, int N, int Cin, int H, int W, int Cout, hipStream_t s);
void launch_tdeconv_bwd_data(const float* dy, const float* w, float* dx, int N, int Cin, int H, int W, int Cout, hipStream_t s);
void launch_tdeconv_bwd_weight(const float* x, const float* dy, float* dw, int N, int Cin, int H, int W, int Cout, void* scratch, size_t scratch_bytes,
                               hipStream_t s, TSumDefer* defer = nullptr);
size_t train_scratch_bytes();
void launch_tbn_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var, float* stats, float* y, int N,
                    int C, long HW, int relu, void* scratch, hipStream_t s, const float* residual = nullptr, const TStatPart* st = nullptr);
void launch_tbn_bwd(const float* x, const float* dy, const float* stats, const float* gamma, const float* beta, float* dgamma, float* dbeta, float* dx,
                    int N, int C, long HW, int relu, void* scratch, hipStream_t s, const TBnRed* red = nullptr);
int launch_tadam_multi(int nt, float* const* p, const float* const* g, float* const* m, float* const* v, const long* sizes, double lr, double b1,
                       double b2, double eps, int step, void* d_table, void* h_table, int upload, hipStream_t s);
void launch_tchan_sum(const float* dy, float* out, int N, int C, long HW, hipStream_t s, void* scratch = nullptr /* >= 64 C doubles */);
void launch_tadd(const float* a, const float* b, float* out, long total, hipStream_t s);
void launch_tslice(const float* src, float* dst, int N, int C, long HW, int Cs, int sc0, int Cd, int dc0, hipStream_t s);
void launch_tadam(float* p, const float* g, float* m, float* v, long total, double lr, double b1, double b2, double eps, int step, hipStream_t s);
void launch_nms_sorted(const int32_t* boxes, int n, double nms_thres, int32_t* suppressor, hipStream_t s);

}  // namespace yf
