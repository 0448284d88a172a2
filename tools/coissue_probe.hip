// coissue_probe.hip -- developer probe: does vector-ALU work issue under a v_mfma_f32_16x16x4_f32?
//
// The fp32 roof bench.py prices the block kernels against assumes that fp32 MFMAs and VALU instructions share one issue rate on a SIMD
// and therefore ADD.  That was inferred from one wave's in-order stream; this probe isolates it:
//   (1) two wave sets on the SAME SIMD -- set A streams independent MFMAs, set B streams independent v_fma_f32 / v_pk_fma_f32 -- each
//       counted over a fixed window of shader cycles, alone and together (1+1, 2+2 waves per SIMD);
//   (2) ONE wave's stream with k VALU fillers behind every MFMA (what a block kernel's in-order stream looks like).
// MFMA kinds: 16x16x4_f32, 4x4x1_16B_f32, 32x32x2_f32 and, as the control, 16x16x16_f16.
//   hipcc -O3 --offload-arch=gfx950 tools/coissue_probe.hip -o tools/kb_coissue && tools/kb_coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

enum { K_IDLE = 0, K_M16 = 1, K_M4 = 2, K_M32 = 3, K_H16 = 4, K_FMA = 5, K_PK = 6, K_MOV = 7, K_LDS = 8,
       K_H32 = 9, K_H4 = 10, K_PKMAX = 11, K_CVT = 12, K_MAXI = 13, K_CND = 14 };   // --f16: what an fp16 block kernel (k19h_kernel) issues
static const char* kname[] = {"idle", "mfma_f32_16x16x4_f32", "mfma_f32_4x4x1_16B_f32", "mfma_f32_32x32x2_f32", "mfma_f32_16x16x16_f16",
                              "v_fma_f32", "v_pk_fma_f32", "v_mov_b32", "ds_read_b128",
                              "mfma_f32_16x16x32_f16", "mfma_f32_4x4x4_16B_f16", "v_pk_max_f16", "v_cvt_pk_f16_f32", "v_max_i32", "v_cndmask_b32"};

#define FMA1(c) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))
#define PK1(c) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a2), "v"(b2))

// One "unit" = one instruction of the kind; a block = UNITS of them, straight-line, on independent registers.
template <int KIND> struct Stream;
template <> struct Stream<K_M16> {
    static constexpr int UNITS = 128;
    f32x4 c[8]; float a, b;
    __device__ void init(int l) { a = l * 0.001f; b = 1.f + l * 0.002f; for (int i = 0; i < 8; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    __device__ void block() {
#pragma unroll
        for (int r = 0; r < UNITS / 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[i], 0, 0, 0);
    }
    __device__ float fini() { float s = 0.f; for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][3]; return s; }
};
template <> struct Stream<K_M4> {
    static constexpr int UNITS = 256;
    f32x4 c[8]; float a, b;
    __device__ void init(int l) { a = l * 0.001f; b = 1.f + l * 0.002f; for (int i = 0; i < 8; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    __device__ void block() {
#pragma unroll
        for (int r = 0; r < UNITS / 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c[i], 0, 0, 0);
    }
    __device__ float fini() { float s = 0.f; for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][3]; return s; }
};
template <> struct Stream<K_M32> {
    static constexpr int UNITS = 64;
    f32x16 c[4]; float a, b;
    __device__ void init(int l) { a = l * 0.001f; b = 1.f + l * 0.002f; for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) c[i][j] = 0.f; }
    __device__ void block() {
#pragma unroll
        for (int r = 0; r < UNITS / 4; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[i], 0, 0, 0);
    }
    __device__ float fini() { float s = 0.f; for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][15]; return s; }
};
template <> struct Stream<K_H16> {
    static constexpr int UNITS = 256;
    f32x4 c[8]; f16x4 a, b;
    __device__ void init(int l) { for (int j = 0; j < 4; ++j) { a[j] = (_Float16)(l * 0.001f + j); b[j] = (_Float16)(1.f + l * 0.002f); } for (int i = 0; i < 8; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    __device__ void block() {
#pragma unroll
        for (int r = 0; r < UNITS / 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c[i], 0, 0, 0);
    }
    __device__ float fini() { float s = 0.f; for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][3]; return s; }
};
template <> struct Stream<K_FMA> {
    static constexpr int UNITS = 512;
    float c[16]; float a, b;
    __device__ void init(int l) { a = 1.f - l * 1e-6f; b = l * 1e-3f; for (int i = 0; i < 16; ++i) c[i] = (float)i; }
    __device__ void block() {
#pragma unroll
        for (int r = 0; r < UNITS / 16; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) FMA1(c[i]);
    }
    __device__ float fini() { float s = 0.f; for (int i = 0; i < 16; ++i) s += c[i]; return s; }
};
template <> struct Stream<K_PK> {
    static constexpr int UNITS = 512;
    f32x2 c[16]; f32x2 a2, b2;
    __device__ void init(int l) { a2 = f32x2{1.f - l * 1e-6f, 1.f - l * 2e-6f}; b2 = f32x2{l * 1e-3f, l * 2e-3f}; for (int i = 0; i < 16; ++i) c[i] = f32x2{(float)i, 1.f}; }
    __device__ void block() {
#pragma unroll
        for (int r = 0; r < UNITS / 16; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) PK1(c[i]);
    }
    __device__ float fini() { float s = 0.f; for (int i = 0; i < 16; ++i) s += c[i][0] + c[i][1]; return s; }
};
template <> struct Stream<K_MOV> {
    static constexpr int UNITS = 512;
    float c[16]; float a;
    __device__ void init(int l) { a = (float)l; for (int i = 0; i < 16; ++i) c[i] = (float)i; }
    __device__ void block() {
#pragma unroll
        for (int r = 0; r < UNITS / 16; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(c[i]) : "v"(a));
    }
    __device__ float fini() { float s = 0.f; for (int i = 0; i < 16; ++i) s += c[i]; return s; }
};
template <> struct Stream<K_LDS> {
    static constexpr int UNITS = 128;
    f32x4 c[8]; const f32x4* p;
    __device__ void init(int l) { extern __shared__ f32x4 lds4[]; p = lds4 + (l & 63); for (int i = 0; i < 8; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    __device__ void block() {
#pragma unroll
        for (int r = 0; r < UNITS / 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(c[i]) : "v"((unsigned)(size_t)p), "i"(i * 1024));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    __device__ float fini() { float s = 0.f; for (int i = 0; i < 8; ++i) s += c[i][0]; return s; }
};

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <> struct Stream<K_H32> {
    static constexpr int UNITS = 256;
    f32x4 c[8]; f16x8 a, b;
    __device__ void init(int l) { for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(l * 0.001f + j); b[j] = (_Float16)(1.f + l * 0.002f); } for (int i = 0; i < 8; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    __device__ void block() {
#pragma unroll
        for (int r = 0; r < UNITS / 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[i], 0, 0, 0);
    }
    __device__ float fini() { float s = 0.f; for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][3]; return s; }
};
template <> struct Stream<K_H4> {
    static constexpr int UNITS = 256;
    f32x4 c[8]; f16x4 a, b;
    __device__ void init(int l) { for (int j = 0; j < 4; ++j) { a[j] = (_Float16)(l * 0.001f + j); b[j] = (_Float16)(1.f + l * 0.002f); } for (int i = 0; i < 8; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    __device__ void block() {
#pragma unroll
        for (int r = 0; r < UNITS / 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, c[i], 0, 0, 0);
    }
    __device__ float fini() { float s = 0.f; for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][3]; return s; }
};
// the three VALU instructions of k19h_kernel's tap: independent destinations, one asm statement each
#define VALU_STREAM(KIND, ASM)                                                                                                   \
    template <> struct Stream<KIND> {                                                                                            \
        static constexpr int UNITS = 512;                                                                                        \
        unsigned c[16]; unsigned a, b;                                                                                           \
        __device__ void init(int l) { a = 0x3c003c00u + l; b = 0x38003800u; for (int i = 0; i < 16; ++i) c[i] = i; }             \
        __device__ void block() {                                                                                                \
            _Pragma("unroll") for (int r = 0; r < UNITS / 16; ++r)                                                               \
                _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "=v"(c[i]) : "v"(a), "v"(b));                   \
        }                                                                                                                        \
        __device__ float fini() { unsigned s = 0; for (int i = 0; i < 16; ++i) s += c[i]; return (float)s; }                     \
    };
VALU_STREAM(K_PKMAX, "v_pk_max_f16 %0, %1, %2")
VALU_STREAM(K_CVT, "v_cvt_pk_f16_f32 %0, %1, %2")
VALU_STREAM(K_MAXI, "v_max_i32 %0, %1, %2")
VALU_STREAM(K_CND, "v_cndmask_b32 %0, %1, %2, vcc")

template <int KIND>
__device__ unsigned long long run_window(unsigned long long t0, unsigned dur, int lane, float* sink, unsigned long long* t_end)
{
    Stream<KIND> st; st.init(lane);
    unsigned long long n = 0, t;
    do {
        t = __builtin_amdgcn_s_memtime();      // requested in front of the block, consumed behind it: its latency hides
        st.block();
        ++n;
    } while (t - t0 < dur);
    *t_end = __builtin_amdgcn_s_memtime();
    *sink = st.fini();
    return n * Stream<KIND>::UNITS;
}

struct Rec { unsigned long long units, cycles, t0, t1; unsigned hwid, kind; };

// waves [0, nA) of a workgroup run kind A, [nA, nA + nB) kind B, the rest leave.  Waves of a workgroup go round-robin to the 4 SIMDs
// (checked from HW_ID), so nA = nB = 4 puts one A and one B wave on every SIMD.
__global__ void __launch_bounds__(1024) pair_kernel(int kindA, int nA, int kindB, int nB, unsigned dur, Rec* rec, float* sink)
{
    extern __shared__ f32x4 lds4[];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x < 512) lds4[threadIdx.x] = f32x4{1.f, 2.f, 3.f, 4.f};
    __syncthreads();
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    // role by the SIMD the wave really sits on (the dispatcher's placement is NOT wave index % 4): the first nA / 4 arrivals on a SIMD run
    // kind A, the next nB / 4 kind B, the rest leave
    __shared__ int arrivals[4];
    __shared__ int slot_of[16];
    if (threadIdx.x < 4) arrivals[threadIdx.x] = 0;
    __syncthreads();
    if (lane == 0) slot_of[w] = atomicAdd(&arrivals[(hwid >> 4) & 3], 1);
    __syncthreads();
    const int slot = slot_of[w];
    const int kind = slot < nA / 4 ? kindA : (slot < (nA + nB) / 4 ? kindB : K_IDLE);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long units = 0, t1 = t0;
    float s = 0.f;
    switch (kind) {
        case K_M16: units = run_window<K_M16>(t0, dur, lane, &s, &t1); break;
        case K_M4:  units = run_window<K_M4>(t0, dur, lane, &s, &t1); break;
        case K_M32: units = run_window<K_M32>(t0, dur, lane, &s, &t1); break;
        case K_H16: units = run_window<K_H16>(t0, dur, lane, &s, &t1); break;
        case K_FMA: units = run_window<K_FMA>(t0, dur, lane, &s, &t1); break;
        case K_PK:  units = run_window<K_PK>(t0, dur, lane, &s, &t1); break;
        case K_MOV: units = run_window<K_MOV>(t0, dur, lane, &s, &t1); break;
        case K_LDS: units = run_window<K_LDS>(t0, dur, lane, &s, &t1); break;
        case K_H32: units = run_window<K_H32>(t0, dur, lane, &s, &t1); break;
        case K_H4:  units = run_window<K_H4>(t0, dur, lane, &s, &t1); break;
        case K_PKMAX: units = run_window<K_PKMAX>(t0, dur, lane, &s, &t1); break;
        case K_CVT: units = run_window<K_CVT>(t0, dur, lane, &s, &t1); break;
        case K_MAXI: units = run_window<K_MAXI>(t0, dur, lane, &s, &t1); break;
        case K_CND: units = run_window<K_CND>(t0, dur, lane, &s, &t1); break;
        default: break;
    }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) rec[blockIdx.x * (blockDim.x >> 6) + w] = Rec{units, t1 - t0, t0, t1, hwid, (unsigned)kind};
}

// One wave's in-order stream: NF fillers behind every MFMA.  FK = K_FMA / K_PK.  A scheduling barrier pins the order.
template <int MK, int FK, int NF>
__global__ void __launch_bounds__(1024) mix_kernel(int iters, Rec* rec, float* sink)
{
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float a = 1.f - lane * 1e-6f, b = lane * 1e-3f;
    f32x2 a2 = f32x2{a, a}, b2 = f32x2{b, b};
    f16x4 ah, bh;
    for (int j = 0; j < 4; ++j) { ah[j] = (_Float16)(lane * 0.001f + j); bh[j] = (_Float16)(1.f + lane * 0.002f); }
    f32x4 acc[8];
    float c[12]; f32x2 c2[12];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 12; ++i) { c[i] = (float)i; c2[i] = f32x2{(float)i, 1.f}; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MK == K_M16) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            if (MK == K_M4) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
            if (MK == K_H16) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bh, acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                if (FK == K_FMA) FMA1(c[f % 12]); else PK1(c2[f % 12]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    for (int i = 0; i < 12; ++i) s += c[i] + c2[i][0] + c2[i][1];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) rec[blockIdx.x * (blockDim.x >> 6) + w] = Rec{(unsigned long long)iters * 8, t1 - t0, t0, t1, hwid, (unsigned)MK};
}

static int g_verbose = 0;
static Rec* d_rec; static float* d_sink; static Rec h_rec[256 * 16];

// AGGREGATE issue rate of a wave set on one SIMD, instructions per 1000 shader cycles (the rates of a set's waves add; per-wave averages
// mislead because the arbiter serves the oldest ready wave first and a younger wave of the same pipe can starve for the whole window)
struct PairOut { double rateA, rateB; int simd_ok; };

static PairOut run_pair(int kA, int nA, int kB, int nB, int blocks, unsigned dur)
{
    const int waves = 16;   // more than needed: every SIMD can fill its set whatever the placement, the surplus leaves at once
    CK(hipMemset(d_rec, 0, sizeof(h_rec)));
    hipLaunchKernelGGL(pair_kernel, dim3(blocks), dim3(64 * waves), 160 * 1024 - 256, 0, kA, nA, kB, nB, dur, d_rec, d_sink);   // all the LDS: one workgroup per CU
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h_rec, d_rec, sizeof(Rec) * blocks * waves, hipMemcpyDeviceToHost));
    PairOut o = {0, 0, 1};
    for (int blk = 0; blk < blocks; ++blk) {
        int na[4] = {0, 0, 0, 0}, nb[4] = {0, 0, 0, 0};
        for (int w = 0; w < waves; ++w) {
            const Rec& r = h_rec[blk * waves + w];
            const int simd = (r.hwid >> 4) & 3;
            if (g_verbose && blk == 0 && r.kind != K_IDLE) printf("      wave %2d: simd %d slot %2d cu %2d kind %-22s %9llu in %9llu cycles = %7.2f cycles each\n", w, simd, r.hwid & 15, (r.hwid >> 8) & 15, kname[r.kind], r.units, r.cycles, (double)r.cycles / (double)r.units);
            if (r.kind == K_IDLE) continue;
            const double rate = 1000.0 * (double)r.units / (double)r.cycles;
            // A and B of the same kind (the controls) are told apart by nothing: count them all as A
            if ((int)r.kind == kA) { o.rateA += rate; ++na[simd]; } else { o.rateB += rate; ++nb[simd]; }
        }
        const int wantA = kA == kB ? (nA + nB) / 4 : nA / 4, wantB = kA == kB ? 0 : nB / 4;
        for (int q = 0; q < 4; ++q) if (na[q] != wantA || nb[q] != wantB) o.simd_ok = 0;
    }
    o.rateA /= 4.0 * blocks; o.rateB /= 4.0 * blocks;
    return o;
}

// NW waves per SIMD all running the same mixed stream: cycles of SIMD time per (MFMA + k fillers) = (last end - first start) / all units of the SIMD
template <int MK, int FK, int NF>
static double run_mix(int blocks, int iters, int per_simd)
{
    const int waves = 4 * per_simd;
    hipLaunchKernelGGL((mix_kernel<MK, FK, NF>), dim3(blocks), dim3(64 * waves), 0, 0, iters, d_rec, d_sink);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h_rec, d_rec, sizeof(Rec) * blocks * waves, hipMemcpyDeviceToHost));
    double s = 0;
    for (int blk = 0; blk < blocks; ++blk)
        for (int q = 0; q < 4; ++q) {
            unsigned long long lo = ~0ull, hi = 0, units = 0;
            for (int w = 0; w < waves; ++w) {
                const Rec& r = h_rec[blk * waves + w];
                if ((int)((r.hwid >> 4) & 3) != q) continue;
                if (r.t0 < lo) lo = r.t0;
                if (r.t1 > hi) hi = r.t1;
                units += r.units;
            }
            s += units ? (double)(hi - lo) / (double)units : 0.0;
        }
    return s / (4.0 * blocks);
}

template <int MK, int FK>
static void mix_row(int blocks, int per_simd)
{
    printf("  %-24s + k x %-12s %d wave(s)/SIMD, k = 0 1 2 3 4 6 8 12:", kname[MK], kname[FK], per_simd);
    const int it = 4000;
    printf(" %6.1f", run_mix<MK, FK, 0>(blocks, it, per_simd)); printf(" %6.1f", run_mix<MK, FK, 1>(blocks, it, per_simd));
    printf(" %6.1f", run_mix<MK, FK, 2>(blocks, it, per_simd)); printf(" %6.1f", run_mix<MK, FK, 3>(blocks, it, per_simd));
    printf(" %6.1f", run_mix<MK, FK, 4>(blocks, it, per_simd)); printf(" %6.1f", run_mix<MK, FK, 6>(blocks, it, per_simd));
    printf(" %6.1f", run_mix<MK, FK, 8>(blocks, it, per_simd)); printf(" %6.1f\n", run_mix<MK, FK, 12>(blocks, it, per_simd));
}

int main(int argc, char** argv)
{
    int blocks = 1, f16 = 0;
    for (int i = 1; i < argc; ++i) { if (!strcmp(argv[i], "--all-cus")) blocks = 256; if (!strcmp(argv[i], "-v")) g_verbose = 1; if (!strcmp(argv[i], "--f16")) f16 = 1; }
    const unsigned dur = 2000000;   // window, shader cycles (~1 ms)
    CK(hipMalloc(&d_rec, sizeof(h_rec))); CK(hipMalloc(&d_sink, 256 * 1024 * 4));
    CK(hipFuncSetAttribute((const void*)pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("# coissue_probe on %s, %d workgroup(s), one per CU, window %u shader cycles (s_memtime ticks)\n", p.gcnArchName, blocks, dur);
    printf("# rate = instructions per 1000 cycles issued by ALL waves of a set on one SIMD (average over the SIMDs); 'together' = both sets resident on the same SIMDs.\n");
    printf("# overlap = rateA_together/rateA_alone + rateB_together/rateB_alone: 2.0 = neither set slows the other (separate issue), 1.0 = ONE shared issue rate.\n");
    if (f16) {   // the fp16 matrix instructions against the VALU instructions an fp16 block kernel issues between them
        const int hm[] = {K_H16, K_H32, K_H4}, hv[] = {K_FMA, K_PKMAX, K_CVT, K_MAXI, K_CND, K_LDS};
        for (int per = 1; per <= 2; ++per) {
            printf("\n## %d + %d waves per SIMD\n", per, per);
            double aM[3], aV[6];
            for (int i = 0; i < 3; ++i) { PairOut o = run_pair(hm[i], 4 * per, K_IDLE, 0, blocks, dur); aM[i] = o.rateA; printf("  alone  %-24s %7.1f /kcycle = %6.2f cycles each\n", kname[hm[i]], o.rateA, 1000.0 / o.rateA); }
            for (int j = 0; j < 6; ++j) { PairOut o = run_pair(hv[j], 4 * per, K_IDLE, 0, blocks, dur); aV[j] = o.rateA; printf("  alone  %-24s %7.1f /kcycle = %6.2f cycles each\n", kname[hv[j]], o.rateA, 1000.0 / o.rateA); }
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 6; ++j) {
                    PairOut o = run_pair(hm[i], 4 * per, hv[j], 4 * per, blocks, dur);
                    printf("  together %-24s %7.1f (x%.2f)   %-18s %7.1f (x%.2f)   overlap %.2f\n", kname[hm[i]], o.rateA, o.rateA / aM[i], kname[hv[j]], o.rateB, o.rateB / aV[j],
                           o.rateA / aM[i] + o.rateB / aV[j]);
                }
            { PairOut o = run_pair(K_H32, 4 * per, K_H4, 4 * per, blocks, dur); printf("  together %-24s %7.1f (x%.2f)   %-18s %7.1f (x%.2f)\n", kname[K_H32], o.rateA, o.rateA / aM[1], kname[K_H4], o.rateB, o.rateB / aM[2]); }
        }
        return 0;
    }
    const int mk[] = {K_M16, K_M4, K_M32, K_H16};
    const int vk[] = {K_FMA, K_PK, K_MOV, K_LDS};
    for (int per = 1; per <= 2; ++per) {
        printf("\n## %d + %d waves per SIMD (set A: %d waves of a workgroup, set B: %d)\n", per, per, 4 * per, 4 * per);
        double aloneM[4], aloneV[4];
        for (int i = 0; i < 4; ++i) { PairOut o = run_pair(mk[i], 4 * per, K_IDLE, 0, blocks, dur); aloneM[i] = o.rateA; printf("  alone  %-24s %7.1f /kcycle = %6.2f cycles each%s\n", kname[mk[i]], o.rateA, 1000.0 / o.rateA, o.simd_ok ? "" : "  (a SIMD did NOT get its full wave set!)"); }
        for (int i = 0; i < 4; ++i) { PairOut o = run_pair(vk[i], 4 * per, K_IDLE, 0, blocks, dur); aloneV[i] = o.rateA; printf("  alone  %-24s %7.1f /kcycle = %6.2f cycles each\n", kname[vk[i]], o.rateA, 1000.0 / o.rateA); }
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                PairOut o = run_pair(mk[i], 4 * per, vk[j], 4 * per, blocks, dur);
                printf("  together %-24s %7.1f (x%.2f)   %-14s %7.1f (x%.2f)   overlap %.2f%s\n", kname[mk[i]], o.rateA, o.rateA / aloneM[i], kname[vk[j]], o.rateB,
                       o.rateB / aloneV[j], o.rateA / aloneM[i] + o.rateB / aloneV[j], o.simd_ok ? "" : "  (placement?)");
            }
        // age swapped: the VALU set takes the older wave slots (the arbiter serves the oldest ready wave first)
        { const int pi[3] = {0, 1, 3}, pj[3] = {0, 0, 1};
          for (int q = 0; q < 3; ++q) {
            PairOut o = run_pair(vk[pj[q]], 4 * per, mk[pi[q]], 4 * per, blocks, dur);
            printf("  together, VALU set older: %-14s %7.1f (x%.2f)   %-24s %7.1f (x%.2f)   overlap %.2f\n", kname[vk[pj[q]]], o.rateA, o.rateA / aloneV[pj[q]], kname[mk[pi[q]]], o.rateB,
                   o.rateB / aloneM[pi[q]], o.rateA / aloneV[pj[q]] + o.rateB / aloneM[pi[q]]);
          } }
        // controls: the same kind on both sets (2x the waves of 'alone'): does a second wave set add issue rate?
        for (int i = 0; i < 4; ++i) { PairOut o = run_pair(mk[i], 4 * per, mk[i], 4 * per, blocks, dur); printf("  control  %-24s twice the waves: %7.1f /kcycle (x%.2f of alone)\n", kname[mk[i]], o.rateA, o.rateA / aloneM[i]); }
        for (int j = 0; j < 4; ++j) { PairOut o = run_pair(vk[j], 4 * per, vk[j], 4 * per, blocks, dur); printf("  control  %-24s twice the waves: %7.1f /kcycle (x%.2f of alone)\n", kname[vk[j]], o.rateA, o.rateA / aloneV[j]); }
    }
    printf("\n## every wave runs ONE in-order stream, k fillers behind each MFMA: SIMD cycles per (MFMA + k fillers)\n");
    for (int per = 1; per <= 4; per *= 2) {
        mix_row<K_M16, K_FMA>(blocks, per); mix_row<K_M16, K_PK>(blocks, per);
        mix_row<K_M4, K_FMA>(blocks, per);  mix_row<K_M4, K_PK>(blocks, per);
        mix_row<K_H16, K_FMA>(blocks, per); mix_row<K_H16, K_PK>(blocks, per);
    }
    return 0;
}
