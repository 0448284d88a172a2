// yf_train_common.h -- what the kernel families of the training step share
// Part of the training-step operators (yf_train_kernels.hip includes the five family headers into ONE translation unit, INSIDE namespace yf,
// so the kernels keep their internal linkage and the launchers in that file see all of them).  Device code: include from there only.
#pragma once

typedef float f32x4_t __attribute__((ext_vector_type(4)));
__host__ __device__ inline int tpw_waves_m(int M) { return M > 32 ? 4 : M > 16 ? 2 : 1; }
static inline unsigned nblk(long total) { return (unsigned)((total + 255) / 256); }

// y before the ReLU, in ONE fixed operation order: the backward recomputes it from z to get the ReLU mask (y > 0) without reading y
__device__ __forceinline__ float tbn_affine(float x, float mean, float invstd, float gamma, float beta)
{
    return __fmaf_rn(__fmul_rn(__fsub_rn(x, mean), invstd), gamma, beta);
}

// ---- BatchNorm statistics out of the conv's epilogue (the large maps: one pass over z less).  A workgroup leaves one (sum, sum of
// squares) pair per output channel and pixel block in `stat` ([channel][block], float2 of values summed in double over the block);
// tbn_stats_from_parts_kernel adds a channel's pairs in double, in block order.  Deterministic: fixed rotation / wave order.
// what a data-gradient kernel needs of the layer below to leave that layer's backward BatchNorm sums (yf_kernels.h: TBnRed)
struct TRedArgs { const float* z; const float* stats; const float* gamma; const float* beta; float2* part; int relu; };
template <int N_> __device__ __forceinline__ float row16_rotate(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N_, 0xf, 0xf, false));   // row_ror:N_
}
__device__ __forceinline__ float row16_sum(float v)         // the sum over the 16 lanes of a DPP row (= the lanes lr of one lk), in every lane
{
    v += row16_rotate<8>(v);
    v += row16_rotate<4>(v);
    v += row16_rotate<2>(v);
    v += row16_rotate<1>(v);
    return v;
}
