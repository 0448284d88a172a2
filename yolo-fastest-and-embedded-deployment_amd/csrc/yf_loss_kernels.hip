// yf_loss_kernels.hip -- the loss end of the reference's training step (first slice):
//   YOLOLossV3.forward(input, targets)   src/model_training/loss/yolo_loss.py:48-97   (loss terms)
//   YOLOLossV3.get_target                src/model_training/loss/yolo_loss.py:144-196 (targets -> dense masks)
// plus d(total loss)/d(input), what `loss.backward()` (src/model_training/train.py:131) hands to the head tensors.
//
// Three small launches per head (HBM-bound, a few MB): init of the dense target arrays, one thread per image walking its
// targets IN ORDER (the reference's loop has sequential semantics: a later target overwrites an earlier one in the same cell),
// and one pass over all cells that accumulates the seven sums in double (block reduction + one atomicAdd per block) and
// writes the gradient.  The backward of the layers themselves is not part of this slice.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "yf_kernels.h"

namespace yf {

// dense target arrays, per (n, anchor, row, col): mask, noobj, tx, ty, tw, th, tcls0..nc-1   (6 + num_classes planes)
enum { LT_MASK = 0, LT_NOOBJ, LT_TX, LT_TY, LT_TW, LT_TH, LT_C0 };
enum { LOSS_MAX_ANCHORS = 8 };
struct LossAnchors { float w[LOSS_MAX_ANCHORS], h[LOSS_MAX_ANCHORS]; };

__global__ void loss_init_kernel(float* __restrict__ dense, long E, int planes, double* __restrict__ acc)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < 16 && blockIdx.x == 0) acc[e] = 0.0;
    if (e >= E) return;
    for (int p = 0; p < planes; ++p) dense[p * E + e] = p == LT_NOOBJ ? 1.f : 0.f;
}

// yolo_loss.py:156-194 for image n.  anc: the three (w, h) anchors in feature-map units as float32 (torch turns the Python-double
// scaled anchors into float32 for both the IoU and the division).  acc[8] counts targets whose cell is outside the map (the
// reference raises IndexError there).
__global__ void loss_targets_kernel(const float* __restrict__ targets, int T, float* __restrict__ dense, long E, int N, int fh, int fw,
                                    LossAnchors anc, int na, int nc, float ignore_thres, double* __restrict__ acc)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float* aw = anc.w;
    const float* ah = anc.h;
    for (int t = 0; t < T; ++t) {
        const float* g = targets + ((long)n * T + t) * 6;
        if (g[5] < 1.f) break;                                         // :158 the marker ends the list
        const float gx = g[0] * (float)fw, gy = g[1] * (float)fh, gw = g[2] * (float)fw, gh = g[3] * (float)fh;   // :161-164
        if (gw <= 0.f || gh <= 0.f) continue;                          // :166
        const int gi = (int)gx, gj = (int)gy;                          // :170-171 int() truncates
        if (gi < 0 || gi >= fw || gj < 0 || gj >= fh) { atomicAdd(&acc[8], 1.0); continue; }
        float iou[LOSS_MAX_ANCHORS];
        int best = 0;
        for (int a = 0; a < na; ++a) {                                 // bbox_iou([0,0,gw,gh], [0,0,aw,ah]), +1 convention (general.py:29-52)
            const float iw = fmaxf(fminf(gw, aw[a]) - 0.f + 1.f, 0.f), ih = fmaxf(fminf(gh, ah[a]) - 0.f + 1.f, 0.f);
            const float inter = iw * ih;
            const float b1 = (gw - 0.f + 1.f) * (gh - 0.f + 1.f), b2 = (aw[a] - 0.f + 1.f) * (ah[a] - 0.f + 1.f);
            iou[a] = inter / (b1 + b2 - inter + 1e-16f);
            if (iou[a] > iou[best]) best = a;                          // np.argmax: first maximum
        }
        const long cell = (long)gj * fw + gi;
        for (int a = 0; a < na; ++a)
            if (iou[a] > ignore_thres) dense[LT_NOOBJ * E + ((long)n * na + a) * fh * fw + cell] = 0.f;   // :181
        const long e = ((long)n * na + best) * fh * fw + cell;
        dense[LT_MASK * E + e] = 1.f;                                  // :184
        dense[LT_TX * E + e] = gx - (float)gi;                         // :187-188
        dense[LT_TY * E + e] = gy - (float)gj;
        dense[LT_TW * E + e] = (float)log((double)(gw / aw[best] + 1e-16f));   // :190-191 math.log of the float32 quotient, in double
        dense[LT_TH * E + e] = (float)log((double)(gh / ah[best] + 1e-16f));
        const int c = (int)g[4];
        if (c >= 0 && c < nc) dense[(LT_C0 + c) * E + e] = 1.f;         // :195 one-hot (accumulates over targets sharing the cell)
    }
}

// torch.nn.BCELoss element: -(t * max(log p, -100) + (1 - t) * max(log(1 - p), -100))
__device__ __forceinline__ float bce(float p, float t) { return -(t * fmaxf(logf(p), -100.f) + (1.f - t) * fmaxf(logf(1.f - p), -100.f)); }
// its derivative with respect to p: (p - t) / max((1 - p) p, 1e-12)
__device__ __forceinline__ float dbce(float p, float t) { return (p - t) / fmaxf((1.f - p) * p, 1e-12f); }
__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

// acc: [0] sum bce x, [1] y, [2] sum (w)^2, [3] h, [4] conf obj, [5] conf noobj, [6] cls, [7] n_pos
template <bool GRAD>
__global__ void __launch_bounds__(256) loss_cells_kernel(const float* __restrict__ head, const float* __restrict__ dense, long E, int fh, int fw,
                                                         int nc, double* __restrict__ acc, float* __restrict__ grad)
{
    const int attrs = 5 + nc;
    __shared__ double red[8][4];
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (e < E) {
        const long hw = (long)fh * fw;
        const long na = e / hw, cell = e - na * hw;          // na = n * num_anchors + anchor
        const float* t = head + na * attrs * hw + cell;      // [N, A*(5+C), fh, fw]: channel c of this anchor at t[c * hw]
        const float m = dense[LT_MASK * E + e], nm = dense[LT_NOOBJ * E + e];
        const float px = sigm(t[0]), py = sigm(t[hw]), w = t[2 * hw], h = t[3 * hw], pc = sigm(t[4 * hw]);
        const float tx = dense[LT_TX * E + e], ty = dense[LT_TY * E + e], tw = dense[LT_TW * E + e], th = dense[LT_TH * E + e];
        s[0] = bce(px * m, tx * m);                          // yolo_loss.py:78-79
        s[1] = bce(py * m, ty * m);
        const float dw = w * m - tw * m, dh = h * m - th * m;
        s[2] = dw * dw;                                      // :80-81
        s[3] = dh * dh;
        s[4] = bce(pc * m, m);                               // :84
        s[5] = bce(pc * nm, nm * 0.f);
        if (m == 1.f) {                                      // :87 pred_cls[mask == 1]
            for (int k = 0; k < nc; ++k) s[6] += bce(sigm(t[(5 + k) * hw]), dense[(LT_C0 + k) * E + e]);
            s[7] = 1.0;
        }
        if constexpr (GRAD) {
            // acc[7] already holds n_pos (the forward pass ran first); d(total)/d(logit), total = 2.5 (x + y + w + h) + conf + cls
            const float inv = 1.f / (float)E, npos = (float)acc[7];
            float* g = grad + na * attrs * hw + cell;
            g[0] = 2.5f * inv * dbce(px * m, tx * m) * m * (1.f - px) * px;
            g[hw] = 2.5f * inv * dbce(py * m, ty * m) * m * (1.f - py) * py;
            g[2 * hw] = 2.5f * inv * 2.f * dw * m;
            g[3 * hw] = 2.5f * inv * 2.f * dh * m;
            g[4 * hw] = inv * (dbce(pc * m, m) * m + 0.5f * dbce(pc * nm, 0.f) * nm) * (1.f - pc) * pc;
            for (int k = 0; k < nc; ++k) {
                float gk = 0.f;
                if (m == 1.f) {
                    const float pk = sigm(t[(5 + k) * hw]), tk = dense[(LT_C0 + k) * E + e];
                    gk = dbce(pk, tk) * (1.f - pk) * pk / ((float)nc * npos);
                }
                g[(5 + k) * hw] = gk;
            }
            return;
        }
    }
    if constexpr (!GRAD) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            double v = s[k];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
            if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v;
        }
        __syncthreads();
        if (threadIdx.x < 8) atomicAdd(&acc[threadIdx.x], red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
    }
}

// losses[0..6] = total, x, y, w, h, conf, cls (float32, combined like yolo_loss.py:90-92); losses[7] = targets outside the map
__global__ void loss_final_kernel(const double* __restrict__ acc, long E, int nc, float* __restrict__ losses)
{
    const float lx = (float)(acc[0] / (double)E), ly = (float)(acc[1] / (double)E), lw = (float)(acc[2] / (double)E), lh = (float)(acc[3] / (double)E);
    const float lconf = (float)(acc[4] / (double)E) + 0.5f * (float)(acc[5] / (double)E);
    const float lcls = (float)(acc[6] / ((double)nc * acc[7]));     // no positive cell: 0 / 0 = nan, like the mean of an empty tensor
    losses[0] = lx * 2.5f + ly * 2.5f + lw * 2.5f + lh * 2.5f + lconf * 1.0f + lcls * 1.0f;
    losses[1] = lx; losses[2] = ly; losses[3] = lw; losses[4] = lh; losses[5] = lconf; losses[6] = lcls;
    losses[7] = (float)acc[8];
}

size_t train_loss_workspace_bytes(int N, int fh, int fw, int na, int nc)
{
    return ((size_t)(LT_C0 + nc) * N * na * fh * fw) * sizeof(float) + 16 * sizeof(double) + 64;
}

void launch_train_loss(const float* head, int N, int fh, int fw, const float* anc, int na, int nc, const float* targets, int T, float ignore_thres,
                       void* work, float* losses, float* grad, hipStream_t s)
{
    const long E = (long)N * na * fh * fw;
    LossAnchors la{};
    for (int a = 0; a < na && a < LOSS_MAX_ANCHORS; ++a) { la.w[a] = anc[2 * a]; la.h[a] = anc[2 * a + 1]; }
    double* acc = reinterpret_cast<double*>(work);                       // 16 doubles first (8-byte aligned), the dense planes behind
    float* dense = reinterpret_cast<float*>(static_cast<char*>(work) + 16 * sizeof(double));
    const unsigned nb = (unsigned)((E + 255) / 256);
    hipLaunchKernelGGL(loss_init_kernel, dim3(nb), dim3(256), 0, s, dense, E, LT_C0 + nc, acc);
    hipLaunchKernelGGL(loss_targets_kernel, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, s, targets, T, dense, E, N, fh, fw, la, na, nc,
                       ignore_thres, acc);
    hipLaunchKernelGGL(loss_cells_kernel<false>, dim3(nb), dim3(256), 0, s, head, dense, E, fh, fw, nc, acc, (float*)nullptr);
    if (grad) hipLaunchKernelGGL(loss_cells_kernel<true>, dim3(nb), dim3(256), 0, s, head, dense, E, fh, fw, nc, acc, grad);
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(1), 0, s, acc, E, nc, losses);
}

}  // namespace yf
