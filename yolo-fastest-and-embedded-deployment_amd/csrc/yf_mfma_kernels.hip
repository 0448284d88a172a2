// yf_mfma_kernels.hip -- pointwise (1x1) convolutions with GEMM-worthy channel counts on the matrix cores.
//
// Replaces torch.nn.Conv2d(k=1)+BN(+ReLU) of the stride-16/32 stages and the two detection heads
// (src/model_training/model/yolo_fastest.py:108-146, forward :182-216), the 2x2 stride-2 ConvTranspose2d
// (deconv5_1 = four 96x96 GEMMs writing interleaved pixels, :138) and the 1x1 conv over the channel concat
// (conv4_1_1 reads its two sources through two pointers, :209-211).
//
// Out[M, N] = act(A[M, K] . W[K, N] + b) (+ residual), M = pixels of the whole batch (NHWC rows), fp32 in, fp32
// accumulate: v_mfma_f32_16x16x4_f32 is exact fp32 (a k-ordered fmaf chain) at the fp32 vector peak rate, but one
// A value per lane feeds 16 outputs and one B value 16 rows, so there is no per-FMA weight traffic at all.
//   * one wave = 16*MT rows x all N columns, no LDS, no barriers (waves are independent);
//   * A fragment: lane (r = l&15, q = l>>4) loads the 16 B  A[row r][k0 + 4q .. 4q+3]  -> four MFMA k-steps whose
//     k index is permuted (k = k0 + 4q + j); the permutation is folded into B, which the host pre-packs per
//     (k-step, n-tile) in lane order, so a B fragment is one coalesced 256-B load (L1/L2 resident, ~40 KB/layer);
//   * C fragment: col = l&15, row = 4q + reg  ->  64-B row segments on store.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "yf_kernels.h"

namespace yf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// number of MFMA k-steps for a source of K channels: 4 per full 16-block, 2 for a trailing 8-block
__host__ __device__ constexpr int ksteps(int K) { return (K / 16) * 4 + ((K % 16) ? 2 : 0); }

// fp16 variant (storage type half_t): one v_mfma_f32_16x16x16_f16 covers the four k-steps of a 16-wide k block -- its A/B
// lane layout (row/col = l & 15, k = 4*(l >> 4) + j) is exactly the 16-B fragment this kernel already loads; B fragments are
// host-packed f16x4 (8 B per lane); a trailing 8-block is one MFMA with k-slots 2,3 zero.
template <int KS, int NT, int MT>
__device__ __forceinline__ void mfma_source_f16(const half_t* __restrict__ in, const long (&rows)[MT], int q,
                                                const float* __restrict__ bp, int lane, f32x4 (&acc)[MT][NT])
{
    constexpr int NB = KS / 16;
    const f16x4* __restrict__ b4 = reinterpret_cast<const f16x4*>(bp);
#pragma unroll 2
    for (int kb = 0; kb < NB; ++kb) {
        f16x4 av[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) av[mt] = *reinterpret_cast<const f16x4*>(in + rows[mt] * KS + kb * 16 + 4 * q);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const f16x4 bv = b4[(kb * NT + nt) * 64 + lane];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x16f16(bv, av[mt], acc[mt][nt], 0, 0, 0);
        }
    }
    if constexpr (KS % 16 != 0) {
        f16x4 av[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const f16x2 t = *reinterpret_cast<const f16x2*>(in + rows[mt] * KS + NB * 16 + 2 * q);
            av[mt] = f16x4{t[0], t[1], (half_t)0.f, (half_t)0.f};
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const f16x4 bv = b4[(NB * NT + nt) * 64 + lane];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x16f16(bv, av[mt], acc[mt][nt], 0, 0, 0);
        }
    }
}

template <int KS, int NT, int MT>
__device__ __forceinline__ void mfma_source(const float* __restrict__ in, const long (&rows)[MT], int q,
                                            const float* __restrict__ bp, int lane, f32x4 (&acc)[MT][NT])
{
    // full 16-wide k blocks
    constexpr int NB = KS / 16;
#pragma unroll 2
    for (int kb = 0; kb < NB; ++kb) {
        float4 av[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) av[mt] = *reinterpret_cast<const float4*>(in + rows[mt] * KS + kb * 16 + 4 * q);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float bv = bp[((kb * 4 + j) * NT + nt) * 64 + lane];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv, ((const float*)&av[mt])[j], acc[mt][nt], 0, 0, 0);
            }
        }
    }
    if constexpr (KS % 16 != 0) {  // trailing 8 channels: 8 B per lane, two k-steps
        static_assert(KS % 16 == 8, "channel counts are multiples of 8");
        float2 av[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) av[mt] = *reinterpret_cast<const float2*>(in + rows[mt] * KS + NB * 16 + 2 * q);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float bv = bp[((NB * 4 + j) * NT + nt) * 64 + lane];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv, ((const float*)&av[mt])[j], acc[mt][nt], 0, 0, 0);
            }
        }
    }
}

__host__ __device__ constexpr int kmfmas16(int K) { return K / 16 + ((K % 16) ? 1 : 0); }  // f16 MFMAs per n-tile for K channels

template <int K1, int K2, int N, int MT, bool RELU, bool RES, int OMODE, typename T>
__global__ void __launch_bounds__(256) pw_mfma_kernel(PwArgs a)
{
    constexpr bool H16 = sizeof(T) == 2;
    constexpr int NT = (N + 15) / 16;
    // packed-B floats per source: fp32 = one float per lane per k-step; fp16 = 2 floats (f16x4) per lane per 16-wide block
    constexpr int S1 = H16 ? 2 * kmfmas16(K1) : ksteps(K1), S2 = K2 ? (H16 ? 2 * kmfmas16(K2) : ksteps(K2)) : 0;
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long row0 = wave * (16 * MT);
    if (row0 >= a.npix) return;  // wave-uniform
    const float* __restrict__ bp = a.w + (OMODE == 2 ? (size_t)blockIdx.z * (S1 + S2) * NT * 64 : 0);

    long rows[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        long rr = row0 + mt * 16 + r;
        rows[mt] = rr < a.npix ? rr : a.npix - 1;  // clamp loads, guard stores
    }
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    if constexpr (H16) {
        mfma_source_f16<K1, NT, MT>(reinterpret_cast<const half_t*>(a.in1), rows, q, bp, lane, acc);
        if constexpr (K2 > 0) mfma_source_f16<K2, NT, MT>(reinterpret_cast<const half_t*>(a.in2), rows, q, bp + (size_t)S1 * NT * 64, lane, acc);
    } else {
        mfma_source<K1, NT, MT>(a.in1, rows, q, bp, lane, acc);
        if constexpr (K2 > 0) mfma_source<K2, NT, MT>(a.in2, rows, q, bp + (size_t)S1 * NT * 64, lane, acc);
    }

    // epilogue.  The weights are the MFMA's A operand and the activations its B operand (the two fragment layouts index
    // (l & 15, l >> 4) the same way: the same registers, swapped), so lane (r, q) holds output channels nt*16 + 4q .. +3 of row
    // row0 + mt*16 + r: one 16-byte (fp16: 8-byte) residual read and store per lane and n-tile
    static_assert(N % 4 == 0, "channel counts are multiples of 4");
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const long row = row0 + mt * 16 + r;
        if (row >= a.npix) continue;
        long opix = row;
        if constexpr (OMODE == 2) {  // ConvTranspose2d k=2 s=2: quadrant (dy,dx) -> pixel (2y+dy, 2x+dx)
            const long n = row / a.HW, hw = row - n * a.HW;
            const int y = (int)(hw / a.W), x = (int)(hw - (long)y * a.W);
            const int dy = blockIdx.z >> 1, dx = blockIdx.z & 1;
            opix = (n * (2 * (a.HW / a.W)) + 2 * y + dy) * (2 * a.W) + 2 * x + dx;
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int c = nt * 16 + 4 * q;
            if (c >= N) continue;
            float4 v = make_float4(acc[mt][nt][0] + a.b[c], acc[mt][nt][1] + a.b[c + 1], acc[mt][nt][2] + a.b[c + 2], acc[mt][nt][3] + a.b[c + 3]);
            if constexpr (RES) {
                const float4 x = ld4<T>(reinterpret_cast<const T*>(a.res) + row * N + c);
                v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
            }
            if constexpr (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if constexpr (OMODE == 1) {  // NCHW (the heads): always fp32, the reference's output
                const long n = row / a.HW, hw = row - n * a.HW;
                a.out[(n * N + c) * a.HW + hw] = v.x; a.out[(n * N + c + 1) * a.HW + hw] = v.y;
                a.out[(n * N + c + 2) * a.HW + hw] = v.z; a.out[(n * N + c + 3) * a.HW + hw] = v.w;
            } else {
                st4<T>(reinterpret_cast<T*>(a.out) + opix * N + c, v);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// pw_ws_kernel: the same GEMM, weight-stationary (fp32).  pw_mfma_kernel re-reads every B fragment per wave: on conv4_1_1
// (K = 232, N = 96) that is 89 KB through the L2 -> L1 fill path per 32 rows, the matrix pipe sits at 33 % and the waves
// are issue-stalled 67 % of the time (SQ counters, profiles/).  Here a wave owns UPW n-tiles ("units"; a deconv quadrant x
// n-tile is a unit too) and keeps ALL their B fragments in registers (58 k-steps x 2 units = 116 VGPRs on conv4_1_1); the
// waves of a workgroup own different units and walk over the same row tiles (persistent: tile t, t + grid, ..), so A comes
// through L1 once per workgroup and B is read once per wave for the whole launch.
// The A fragments of a row tile (K/4 registers) form a ring: each 16-byte piece is re-requested for the NEXT row tile
// right after the MFMAs that consumed it -- a full tile of prefetch distance with no extra registers.  No LDS, no barriers.
// ------------------------------------------------------------------------------------------------
// A workgroup is RS "row streams" x NUG unit groups of waves, sized to a multiple of 4 waves: the dispatcher deals a workgroup's
// waves to the SIMDs in a fixed pattern, so a 6-wave workgroup loads SIMDs 2,2,1,1 and a second one does not fit beside it
// (measured: 1 workgroup per CU resident, two sequential rounds).
__device__ __forceinline__ bool v0guard(float v) { return v != 123456.f; }  // DBG: keeps the value alive, never stores

// DBG (tools/kbench.hip only): 1 = no A refill, 2 = no stores
// pw_ws_x3_kernel: the same weight-stationary GEMM for DT_F16X3 (fp32 in HBM, split-operand fp16 MFMAs).  A wave keeps the hi AND
// lo f16x4 fragments of its units' weights in registers (the same register count as the fp32 fragments); every 16-byte piece of
// the A ring is split into its fp16 halves once per row tile (ten VALU instructions that run beside the matrix pipe) and feeds
// 3 x UPW MFMAs: w_lo*a_hi + w_hi*a_lo + w_hi*a_hi.
template <int K1, int K2, int N, int UPW, int RS, bool RELU, int OMODE>
__global__ void __launch_bounds__(64 * RS * ((((N + 15) / 16) * (OMODE == 2 ? 4 : 1) + UPW - 1) / UPW)) pw_ws_x3_kernel(PwArgs a)
{
    constexpr int NT = (N + 15) / 16, NQ = OMODE == 2 ? 4 : 1, NU = NT * NQ;
    constexpr int NB1 = K1 / 16, NB2 = K2 / 16;
    constexpr bool T1 = (K1 % 16) != 0, T2 = (K2 % 16) != 0;
    constexpr int NK1 = kmfmas16(K1), NK2 = K2 ? kmfmas16(K2) : 0, NK = NK1 + NK2;   // f16 MFMA k-blocks
    static_assert(OMODE == 0 || OMODE == 2, "NHWC outputs only");
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    constexpr int NUG = (NU + UPW - 1) / UPW;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int ug = wv % NUG, rs = wv / NUG;

    f16x4 bh[UPW][NK], bl[UPW][NK];
    float bias[UPW][4];
    int un[UPW], uq[UPW];
#pragma unroll
    for (int i = 0; i < UPW; ++i) {
        const int u = ug * UPW + i;
        const bool v = u < NU;
        const int uc = v ? u : NU - 1;
        uq[i] = uc / NT;
        un[i] = uc - uq[i] * NT;
        // per quadrant: [hi fragments NK x NT x 64 | lo fragments, same layout] (mfma_pack_weights_x3)
        const f16x4* w = reinterpret_cast<const f16x4*>(a.w) + ((size_t)uq[i] * 2 * NK * NT + un[i]) * 64 + lane;
#pragma unroll
        for (int kb = 0; kb < NK; ++kb) {
            bh[i][kb] = w[(size_t)kb * NT * 64];
            bl[i][kb] = w[(size_t)(NK + kb) * NT * 64];
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) bias[i][reg] = un[i] * 16 + 4 * q + reg < N ? a.b[un[i] * 16 + 4 * q + reg] : 0.f;
        if (!v) un[i] = NT;
    }

    const long ntiles = (a.npix + 15) / 16;
    const long t0 = (long)blockIdx.x * RS + rs, tstep = (long)gridDim.x * RS;
    auto rowptr1 = [&](long t) { long rr = t * 16 + r; rr = rr < a.npix ? rr : a.npix - 1; return a.in1 + rr * K1 + 4 * q; };
    auto rowptr2 = [&](long t) { long rr = t * 16 + r; rr = rr < a.npix ? rr : a.npix - 1; return a.in2 + rr * K2 + 4 * q; };

    float4 a1[NB1 > 0 ? NB1 : 1], a2[NB2 > 0 ? NB2 : 1];
    float2 a1t = make_float2(0.f, 0.f), a2t = make_float2(0.f, 0.f);
    {
        const float* p1 = rowptr1(t0 < ntiles ? t0 : ntiles - 1);
#pragma unroll
        for (int kb = 0; kb < NB1; ++kb) a1[kb] = *reinterpret_cast<const float4*>(p1 + kb * 16);
        if constexpr (T1) a1t = *reinterpret_cast<const float2*>(p1 - 4 * q + NB1 * 16 + 2 * q);
        if constexpr (K2 > 0) {
            const float* p2 = rowptr2(t0 < ntiles ? t0 : ntiles - 1);
#pragma unroll
            for (int kb = 0; kb < NB2; ++kb) a2[kb] = *reinterpret_cast<const float4*>(p2 + kb * 16);
            if constexpr (T2) a2t = *reinterpret_cast<const float2*>(p2 - 4 * q + NB2 * 16 + 2 * q);
        }
    }
    auto mac = [&](f32x4 (&acc)[UPW], int kb, float x0, float x1, float x2, float x3) {
        f16x4 ah, al;
        split_f16x4(x0, x1, x2, x3, ah, al);
#pragma unroll
        for (int i = 0; i < UPW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(bl[i][kb], ah, acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < UPW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(bh[i][kb], al, acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < UPW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(bh[i][kb], ah, acc[i], 0, 0, 0);
    };

#pragma unroll 1
    for (long t = t0; t < ntiles; t += tstep) {
        const long tn = (t + tstep < ntiles) ? t + tstep : t;
        const float* p1 = rowptr1(tn);
        f32x4 acc[UPW];
#pragma unroll
        for (int i = 0; i < UPW; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < NB1; ++kb) {
            mac(acc, kb, a1[kb].x, a1[kb].y, a1[kb].z, a1[kb].w);
            a1[kb] = *reinterpret_cast<const float4*>(p1 + kb * 16);
        }
        if constexpr (T1) {
            mac(acc, NB1, a1t.x, a1t.y, 0.f, 0.f);
            a1t = *reinterpret_cast<const float2*>(p1 - 4 * q + NB1 * 16 + 2 * q);
        }
        if constexpr (K2 > 0) {
            const float* p2 = rowptr2(tn);
#pragma unroll
            for (int kb = 0; kb < NB2; ++kb) {
                mac(acc, NK1 + kb, a2[kb].x, a2[kb].y, a2[kb].z, a2[kb].w);
                a2[kb] = *reinterpret_cast<const float4*>(p2 + kb * 16);
            }
            if constexpr (T2) {
                mac(acc, NK1 + NB2, a2t.x, a2t.y, 0.f, 0.f);
                a2t = *reinterpret_cast<const float2*>(p2 - 4 * q + NB2 * 16 + 2 * q);
            }
        }
        const long row = t * 16 + r;
        long obase;
        if constexpr (OMODE == 0) {
            obase = row * N;
        } else {
            const long n = row / a.HW, hw = row - n * a.HW;
            const int y = (int)(hw / a.W), x = (int)(hw - (long)y * a.W);
            obase = ((n * (2 * (a.HW / a.W)) + 2 * y) * (2 * a.W) + 2 * x) * N;
        }
#pragma unroll
        for (int i = 0; i < UPW; ++i) {
            const int c = un[i] * 16 + 4 * q;
            if (c >= N || row >= a.npix) continue;
            const long qoff = OMODE == 2 ? ((long)(uq[i] >> 1) * (2 * a.W) + (uq[i] & 1)) * N : 0;
            float4 v = make_float4(acc[i][0] + bias[i][0], acc[i][1] + bias[i][1], acc[i][2] + bias[i][2], acc[i][3] + bias[i][3]);
            if constexpr (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *reinterpret_cast<float4*>(a.out + obase + qoff + c) = v;
        }
    }
}

template <int K1, int K2, int N, int UPW, int RS, bool RELU, int OMODE, int DBG = 0>
__global__ void __launch_bounds__(64 * RS * ((((N + 15) / 16) * (OMODE == 2 ? 4 : 1) + UPW - 1) / UPW)) pw_ws_kernel(PwArgs a)
{
    constexpr int NT = (N + 15) / 16, NQ = OMODE == 2 ? 4 : 1, NU = NT * NQ;
    constexpr int NB1 = K1 / 16, NB2 = K2 / 16;
    constexpr bool T1 = (K1 % 16) != 0, T2 = (K2 % 16) != 0;
    constexpr int S1 = ksteps(K1), S2 = K2 ? ksteps(K2) : 0, SS = S1 + S2;
    static_assert(OMODE == 0 || OMODE == 2, "NHWC outputs only");
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    constexpr int NUG = (NU + UPW - 1) / UPW;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int ug = wv % NUG, rs = wv / NUG;  // unit group and row stream of this wave

    // ---- this wave's B fragments and biases ----
    float bw[UPW][SS], bias[UPW][4];
    int un[UPW], uq[UPW];
#pragma unroll
    for (int i = 0; i < UPW; ++i) {
        const int u = ug * UPW + i;
        const bool v = u < NU;             // a workgroup's last wave may have fewer units: it recomputes the last one, stores nothing
        const int uc = v ? u : NU - 1;
        uq[i] = uc / NT;
        un[i] = uc - uq[i] * NT;
        const float* w = a.w + ((size_t)uq[i] * SS * NT + un[i]) * 64 + lane;
#pragma unroll
        for (int s = 0; s < SS; ++s) bw[i][s] = w[(size_t)s * NT * 64];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) bias[i][reg] = un[i] * 16 + 4 * q + reg < N ? a.b[un[i] * 16 + 4 * q + reg] : 0.f;
        if (!v) un[i] = NT;  // marks "no stores"
    }

    const long ntiles = (a.npix + 15) / 16;
    const long t0 = (long)blockIdx.x * RS + rs, tstep = (long)gridDim.x * RS;
    auto rowptr1 = [&](long t) { long rr = t * 16 + r; rr = rr < a.npix ? rr : a.npix - 1; return a.in1 + rr * K1 + 4 * q; };
    auto rowptr2 = [&](long t) { long rr = t * 16 + r; rr = rr < a.npix ? rr : a.npix - 1; return a.in2 + rr * K2 + 4 * q; };

    // ---- the A ring, filled for the first tile ----
    float4 a1[NB1 > 0 ? NB1 : 1], a2[NB2 > 0 ? NB2 : 1];
    float2 a1t = make_float2(0.f, 0.f), a2t = make_float2(0.f, 0.f);
    {
        const float* p1 = rowptr1(t0 < ntiles ? t0 : ntiles - 1);
#pragma unroll
        for (int kb = 0; kb < NB1; ++kb) a1[kb] = *reinterpret_cast<const float4*>(p1 + kb * 16);
        if constexpr (T1) a1t = *reinterpret_cast<const float2*>(p1 - 4 * q + NB1 * 16 + 2 * q);
        if constexpr (K2 > 0) {
            const float* p2 = rowptr2(t0 < ntiles ? t0 : ntiles - 1);
#pragma unroll
            for (int kb = 0; kb < NB2; ++kb) a2[kb] = *reinterpret_cast<const float4*>(p2 + kb * 16);
            if constexpr (T2) a2t = *reinterpret_cast<const float2*>(p2 - 4 * q + NB2 * 16 + 2 * q);
        }
    }

#pragma unroll 1
    for (long t = t0; t < ntiles; t += tstep) {
        const long tn = (t + tstep < ntiles) ? t + tstep : t;  // the tile whose fragments are requested while this one is computed
        const float* p1 = rowptr1(tn);
        f32x4 acc[UPW];
#pragma unroll
        for (int i = 0; i < UPW; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < NB1; ++kb) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < UPW; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[i][kb * 4 + j], ((const float*)&a1[kb])[j], acc[i], 0, 0, 0);
            if constexpr (!(DBG & 1)) a1[kb] = *reinterpret_cast<const float4*>(p1 + kb * 16);
        }
        if constexpr (T1) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < UPW; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[i][NB1 * 4 + j], ((const float*)&a1t)[j], acc[i], 0, 0, 0);
            if constexpr (!(DBG & 1)) a1t = *reinterpret_cast<const float2*>(p1 - 4 * q + NB1 * 16 + 2 * q);
        }
        if constexpr (K2 > 0) {
            const float* p2 = rowptr2(tn);
#pragma unroll
            for (int kb = 0; kb < NB2; ++kb) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < UPW; ++i)
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[i][S1 + kb * 4 + j], ((const float*)&a2[kb])[j], acc[i], 0, 0, 0);
                if constexpr (!(DBG & 1)) a2[kb] = *reinterpret_cast<const float4*>(p2 + kb * 16);
            }
            if constexpr (T2) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < UPW; ++i)
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[i][S1 + NB2 * 4 + j], ((const float*)&a2t)[j], acc[i], 0, 0, 0);
                a2t = *reinterpret_cast<const float2*>(p2 - 4 * q + NB2 * 16 + 2 * q);
            }
        }
        // ---- epilogue.  The weights are the MFMA's A operand and the activations its B operand (both fragment layouts index
        // (l & 15, l >> 4) the same way, so this is the same registers with the operands swapped): D = (X W)^T, lane (r, q) holds
        // output channels un*16 + 4q .. +3 of row t*16 + r -- one 16-byte store per lane and unit instead of four 4-byte ones ----
        const long row = t * 16 + r;
        long obase;  // element offset of the row's output pixel (deconv: of its 2x2 block's top-left pixel)
        if constexpr (OMODE == 0) {
            obase = row * N;
        } else {  // ConvTranspose2d k=2 s=2: quadrant (dy,dx) of input pixel (y,x) -> output pixel (2y+dy, 2x+dx)
            const long n = row / a.HW, hw = row - n * a.HW;
            const int y = (int)(hw / a.W), x = (int)(hw - (long)y * a.W);
            obase = ((n * (2 * (a.HW / a.W)) + 2 * y) * (2 * a.W) + 2 * x) * N;
        }
#pragma unroll
        for (int i = 0; i < UPW; ++i) {
            const int c = un[i] * 16 + 4 * q;
            if (c >= N || row >= a.npix) continue;  // c >= N: also the "no unit" marker (N is a multiple of 4)
            const long qoff = OMODE == 2 ? ((long)(uq[i] >> 1) * (2 * a.W) + (uq[i] & 1)) * N : 0;
            float4 v = make_float4(acc[i][0] + bias[i][0], acc[i][1] + bias[i][1], acc[i][2] + bias[i][2], acc[i][3] + bias[i][3]);
            if constexpr (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if ((DBG & 2) && v0guard(v.x)) continue;
            *reinterpret_cast<float4*>(a.out + obase + qoff + c) = v;
        }
    }
}

//      (k1, k2, n, units per wave, row streams, relu, omode)  -- fp32 storage only.  UPW as large as the registers allow: every
//      wave of a row stream re-reads the same A tile through the texture path (16 B per lane, 16 rows per instruction), and that
//      path, not the matrix pipe, limits this kernel: conv4_1_1 with 1 / 2 / 3 units per wave runs 67 / 58 / 37 us (tools/kbench ws)
//      last column: workgroups that fit a CU (registers), for the persistent grid
#define YF_WS_SHAPES(WS)                                                              \
    WS(24, 0, 136, 9, 4, true, 0, 3)    /* conv4_2: all 9 n-tiles per wave, 4 streams  (stride 16) */  \
    WS(136, 0, 48, 3, 4, true, 0, 3)    /* conv5_1: 3 n-tiles per wave, 4 streams      (stride 32) */  \
    WS(48, 0, 96, 6, 4, true, 0, 3)     /* conv5_2: 6 n-tiles per wave, 4 streams */                   \
    WS(96, 0, 96, 6, 2, true, 2, 1)     /* deconv5_1: 4 quadrants x 6 n-tiles = 24 units: 4 waves x 2 streams */ \
    WS(136, 96, 96, 3, 2, true, 0, 1)   /* conv4_1_1 over cat(conv4_2, deconv5_1): 2 waves x 2 streams */

template <int K1, int K2, int N, int UPW, int RS, bool RELU, int OMODE, int WPC>
static int launch_ws(const PwArgs& a, hipStream_t s, bool x3 = false)
{
    constexpr int NU = ((N + 15) / 16) * (OMODE == 2 ? 4 : 1), NUG = (NU + UPW - 1) / UPW;
    static_assert((NUG * RS) % 4 == 0 && NUG * RS <= 16, "whole waves per SIMD");
    const int n_cu = device_cu_count(current_device());
    if (n_cu <= 0) return -2;
    const long ntiles = (a.npix + 15) / 16, streams = (ntiles + RS - 1) / RS;
    // persistent grid: every workgroup gets the same number of row tiles (no ragged last round)
    const long cap = (long)n_cu * WPC, rounds = (streams + cap - 1) / cap, grid = (streams + rounds - 1) / rounds;
    if (x3) hipLaunchKernelGGL((pw_ws_x3_kernel<K1, K2, N, UPW, RS, RELU, OMODE>), dim3((unsigned)grid), dim3(64 * NUG * RS), 0, s, a);
    else hipLaunchKernelGGL((pw_ws_kernel<K1, K2, N, UPW, RS, RELU, OMODE>), dim3((unsigned)grid), dim3(64 * NUG * RS), 0, s, a);
    return 0;
}

template <int K1, int K2, int N, int MT, bool RELU, bool RES, int OMODE>
static int launch_t(const PwArgs& a, hipStream_t s, int dtype)
{
    const long waves = (a.npix + 16 * MT - 1) / (16 * MT);
    dim3 grid((unsigned)((waves + 3) / 4), 1, OMODE == 2 ? 4 : 1);
    if (dtype == DT_F16) hipLaunchKernelGGL((pw_mfma_kernel<K1, K2, N, MT, RELU, RES, OMODE, half_t>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((pw_mfma_kernel<K1, K2, N, MT, RELU, RES, OMODE, float>), grid, dim3(256), 0, s, a);
    return 0;
}

// rows per wave: 16*MT. MT is picked so that even the smallest batch-256 stage (20 480 rows) fills 256 CUs.
//      (k1, k2, n, MT, relu, residual, omode)
#define YF_MFMA_SHAPES(MF)                                                          \
    MF(24, 0, 136, 2, true, false, 0)    /* res4_x.conv1, conv4_2       (stride 16) */ \
    MF(136, 0, 24, 2, false, true, 0)    /* res4_x.conv3 */                            \
    MF(136, 0, 48, 1, true, false, 0)    /* conv5_1                     (stride 32) */ \
    MF(48, 0, 224, 1, true, false, 0)    /* res5_x.conv1 */                            \
    MF(224, 0, 48, 1, false, true, 0)    /* res5_x.conv3 */                            \
    MF(48, 0, 96, 1, true, false, 0)     /* conv5_2 */                                 \
    MF(96, 0, 128, 1, false, false, 0)   /* conv5_4 */                                 \
    MF(128, 0, 128, 1, false, false, 0)  /* conv5_6 */                                 \
    MF(128, 0, 24, 1, false, false, 1)   /* head_5  (NCHW) */                          \
    MF(96, 0, 96, 1, true, false, 2)     /* deconv5_1 (4 quadrants) */                 \
    MF(136, 96, 96, 2, true, false, 0)   /* conv4_1_1 over cat(conv4_2, deconv5_1) */  \
    MF(96, 0, 96, 2, false, false, 0)    /* conv4_1_3, conv4_1_5 */                    \
    MF(96, 0, 24, 2, false, false, 1)    /* head_4  (NCHW) */

int launch_pw_mfma(int cin1, int cin2, int cout, bool relu_, bool res_, int omode, const PwArgs& a, hipStream_t s, int dtype)
{
#define WS(k1, k2, n, upw, rs, relu, om, wpc)                                                                  \
    if (dtype != DT_F16 && cin1 == k1 && cin2 == k2 && cout == n && relu_ == relu && !res_ && omode == om)      \
        return launch_ws<k1, k2, n, upw, rs, relu, om, wpc>(a, s, dtype == DT_F16X3);
    YF_WS_SHAPES(WS)
#undef WS
    if (dtype == DT_F16X3) return -1;   // split-operand mode: weight-stationary shapes only (the engine falls back to fp32 elsewhere)
#define MF(k1, k2, n, mt, relu, res, om)                                                        \
    if (cin1 == k1 && cin2 == k2 && cout == n && relu_ == relu && res_ == res && omode == om)    \
        return launch_t<k1, k2, n, mt, relu, res, om>(a, s, dtype);
    YF_MFMA_SHAPES(MF)
#undef MF
    return -1;
}

bool mfma_has_x3_kernel(int cin1, int cin2, int cout, bool relu_, bool res_, int omode)
{
#define WS(k1, k2, n, upw, rs, relu, om, wpc) \
    if (cin1 == k1 && cin2 == k2 && cout == n && relu_ == relu && !res_ && omode == om) return true;
    YF_WS_SHAPES(WS)
#undef WS
    return false;
}

bool mfma_has_kernel(int cin1, int cin2, int cout, bool relu_, bool res_, int omode)
{
#define MF(k1, k2, n, mt, relu, res, om) \
    if (cin1 == k1 && cin2 == k2 && cout == n && relu_ == relu && res_ == res && omode == om) return true;
    YF_MFMA_SHAPES(MF)
#undef MF
    return false;
}

// Host-side packing of W[K][N] (row-major, as in the blob) into MFMA B fragments, see the file header.
//   out[(step*NT + nt)*64 + lane] = W[kidx(step, lane>>4)][nt*16 + (lane&15)]   (0 beyond N)
// Sources of a concat are packed one after the other (K = K1 + K2 rows of W).
// IEEE fp32 -> fp16 bits, round to nearest even (what v_cvt_f16_f32 does), on the host
uint16_t f32_to_f16_bits(float f)
{
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | (x > 0x7f800000u ? 0x7e00u : 0x7c00u));   // nan / inf
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                                    // overflows to inf after rounding
    if (x < 0x33000001u) return (uint16_t)sign;                                                 // underflows to zero
    if (x < 0x38800000u) {                                                                      // subnormal half
        const int shift = 126 - (int)(x >> 23);   // 14 .. 24
        uint32_t m = (x & 0x7fffffu) | 0x800000u;
        uint32_t r = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (r & 1u))) ++r;
        return (uint16_t)(sign | r);
    }
    uint32_t r = ((x - 0x38000000u) >> 13), rem = x & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) ++r;
    return (uint16_t)(sign | r);
}

size_t mfma_packed_floats_f16(int k1, int k2, int n) { return (size_t)2 * (kmfmas16(k1) + (k2 ? kmfmas16(k2) : 0)) * ((n + 15) / 16) * 64; }

// fp16 B fragments: out16[((blk*NT + nt)*64 + lane)*4 + j] = half(W[kbase + blk*16 + (blk < NB ? 4 : 2)*q + j][nt*16 + (lane&15)]), j beyond the block = 0
// WM_F16X3: [hi fragments | lo fragments], each in the fp16 layout
size_t mfma_packed_floats_x3(int k1, int k2, int n) { return 2 * mfma_packed_floats_f16(k1, k2, n); }
static void mfma_pack_weights_f16_part(const float* w, int k1, int k2, int n, float* out, bool lo);
void mfma_pack_weights_x3(const float* w, int k1, int k2, int n, float* out)
{
    mfma_pack_weights_f16_part(w, k1, k2, n, out, false);
    mfma_pack_weights_f16_part(w, k1, k2, n, out + mfma_packed_floats_f16(k1, k2, n), true);
}
void mfma_pack_weights_f16(const float* w, int k1, int k2, int n, float* out) { mfma_pack_weights_f16_part(w, k1, k2, n, out, false); }

static void mfma_pack_weights_f16_part(const float* w, int k1, int k2, int n, float* out, bool lo)
{
    uint16_t* o16 = reinterpret_cast<uint16_t*>(out);
    const int NT = (n + 15) / 16;
    size_t blk = 0;
    int kbase = 0;
    for (int src = 0; src < 2; ++src) {
        const int K = src == 0 ? k1 : k2;
        if (K == 0) continue;
        const int NB = K / 16, NK = kmfmas16(K);
        for (int kb = 0; kb < NK; ++kb, ++blk) {
            const int per = kb < NB ? 4 : 2;
            for (int nt = 0; nt < NT; ++nt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 4; ++j) {
                        const int q = lane >> 4, c = nt * 16 + (lane & 15);
                        const int k = kbase + kb * 16 + per * q + j;
                        const float v = (j < per && c < n) ? w[(size_t)k * n + c] : 0.f;
                        o16[((blk * NT + nt) * 64 + lane) * 4 + j] = lo ? f16_lo_bits(v) : f32_to_f16_bits(v);
                    }
        }
        kbase += K;
    }
}

size_t mfma_packed_floats(int k1, int k2, int n) { return (size_t)(ksteps(k1) + (k2 ? ksteps(k2) : 0)) * ((n + 15) / 16) * 64; }

void mfma_pack_weights(const float* w, int k1, int k2, int n, float* out)
{
    const int NT = (n + 15) / 16;
    size_t step = 0;
    int kbase = 0;
    for (int src = 0; src < 2; ++src) {
        const int K = src == 0 ? k1 : k2;
        if (K == 0) continue;
        const int NB = K / 16;
        for (int kb = 0; kb <= NB; ++kb) {
            const int per = kb < NB ? 4 : ((K % 16) ? 2 : 0);  // k values per lane-quad in this block
            for (int j = 0; j < per; ++j, ++step)
                for (int nt = 0; nt < NT; ++nt)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int q = lane >> 4, c = nt * 16 + (lane & 15);
                        const int k = kbase + kb * 16 + per * q + j;
                        out[(step * NT + nt) * 64 + lane] = c < n ? w[(size_t)k * n + c] : 0.f;
                    }
        }
        kbase += K;
    }
}

}  // namespace yf
