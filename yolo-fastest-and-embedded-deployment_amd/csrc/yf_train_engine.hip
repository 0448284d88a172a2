// yf_train_engine.hip -- C ABI of the training step (include/yolo_fastest_hip.h: yf_train_*, yf_trainer_*): the per-layer operators
// of yf_train_kernels.hip behind plain-pointer entry points, and the trainer, which walks the graph of yolo_fastest.py:150-218 in C++
// so that the forward and the backward of an iteration are one call each.  (The loss end, yf_train_loss, lives in yf_engine.hip: it
// takes an engine handle.)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/yolo_fastest_hip.h"
#include "yf_kernels.h"
#include "yf_layers.h"

using namespace yf_layers;

namespace yf {
int set_error(int code, const char* msg);   // yf_engine.hip: the slot yf_last_error_string() reads
}

namespace {
int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    return yf::set_error(code, buf);
}
#define HIP_OK(expr)                                                                                     \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) return fail(YF_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));      \
    } while (0)
}  // namespace

// ---- the training iteration's forward and backward as ONE call each (yf_trainer_*): the graph of yolo_fastest.py:150-218 over
// NCHW fp32 tensors in a caller-owned workspace.  kLayers is in module-definition order, which is also the forward order; a layer's
// input is the previous layer's output except where noted in trainer_build(). ----
struct TLayer {
    int in;                 // layer whose y is this layer's input: -1 = the images, -2 = the concat buffer
    int Cin, Hin, Win, Cout, Ho, Wo;
    int res_from;           // conv3 of a BasicResBlock: layer whose INPUT is added to this layer's output (the block's conv1), else -1
    size_t z, y, st;        // float offsets PER FRAME of z and y, absolute float offset of stats (2 * Cout)
    int p0;                 // index of the layer's first parameter in parameters() order
};
struct yf_trainer_s {
    int device, H, W;
    bool graphs_on = true;   // yf_trainer_set_graphs: 0 = every pass as plain launches (what YF_TRAIN_GRAPH_OFF does process-wide)
    LayerSpec layers[kNumLayers];   // the graph for this trainer's io_params: kBaseLayers with conv0's Cin and the heads' Cout set (make_layers)
    TLayer L[kNumLayers];
    size_t act_floats;      // per frame: all z / y, the concat buffer
    size_t cat;             // per-frame offset of the concat buffer
    size_t stats_floats;    // total, independent of N
    size_t gmax;            // per frame: the largest activation
    size_t ga2, gb2, gd;    // per-frame sizes of the branch-point gradient buffers (conv4_2, conv5_2, deconv5_1 outputs)
    int n_params;
    size_t param_off[3 * kNumLayers];   // float offset of every parameter in a flat parameters()-order buffer, [n_params] = the total
    // the table of the pass's one multi-tensor sum of the split weight gradients (offsets only: the same every iteration of a batch size)
    std::vector<yf::TSumEntry> sum_tab;
    yf::TSumEntry* d_sum_tab = nullptr;
    // a pass with the same pointers as the call before it is captured once (on cap_stream: the caller's may be the legacy stream,
    // which cannot capture) and replayed as a HIP graph on the caller's stream from then on: see run_pass()
    struct PassGraph { std::vector<uintptr_t> key; hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; long uses = 0; };
    // as many graphs as remembered pointer sets (4): a loop that cycles through three or four sets (gradient accumulation, alternating
    // input buffers) replays all of them; a graph evicted before it was replayed a few times counts as thrash, and a pass that keeps
    // thrashing stops capturing (every capture is a host stall of several hundred launches)
    struct PassGraphs { std::vector<uintptr_t> seen[4]; int nseen = 0; PassGraph g[4]; int next = 0; long replays = 0, captures = 0, evictions = 0;
                        int failures = 0, thrash = 0; } gfwd, gbwd;
    hipStream_t cap_stream = nullptr;
    int i_conv4_2, i_conv4_3, i_conv5_2, i_conv5_3, i_conv5_6, i_head5, i_deconv, i_c411, i_c415, i_head4;
};

namespace {

int trainer_build(yf_trainer_s* t, int H, int W)
{
    const LayerSpec* const kLayers = t->layers;
    t->H = H; t->W = W;
    t->i_conv4_2 = find_layer("conv4_2"); t->i_conv4_3 = find_layer("conv4_3"); t->i_conv5_2 = find_layer("conv5_2");
    t->i_conv5_3 = find_layer("conv5_3"); t->i_conv5_6 = find_layer("conv5_6"); t->i_head5 = find_layer("head_5");
    t->i_deconv = find_layer("deconv5_1"); t->i_c411 = find_layer("conv4_1_1"); t->i_c415 = find_layer("conv4_1_5");
    t->i_head4 = find_layer("head_4");
    size_t off = 0, st = 0;
    int p = 0;
    t->gmax = 0;
    t->param_off[0] = 0;
    for (int i = 0; i < kNumLayers; ++i) {
        const LayerSpec& S = kLayers[i];
        TLayer& L = t->L[i];
        L.in = i - 1;
        if (i == t->i_deconv) L.in = t->i_conv5_2;              // deconv5_1(conv5_2)              yolo_fastest.py:208
        if (i == t->i_c411) L.in = -2;                          // conv4_1_1(cat(conv4_2, deconv5_1))       :209-211
        if (L.in >= 0) { L.Cin = t->L[L.in].Cout; L.Hin = t->L[L.in].Ho; L.Win = t->L[L.in].Wo; }
        else if (L.in == -1) { L.Cin = kLayers[0].cin; L.Hin = H; L.Win = W; }
        else { L.Cin = t->L[t->i_conv4_2].Cout + t->L[t->i_deconv].Cout; L.Hin = t->L[t->i_conv4_2].Ho; L.Win = t->L[t->i_conv4_2].Wo; }
        if (L.Cin != S.cin) return -1;
        L.Cout = S.cout;
        if (S.kind == K_DECONV) { L.Ho = 2 * L.Hin; L.Wo = 2 * L.Win; }
        else { const int pad = (S.k - 1) / 2; L.Ho = (L.Hin + 2 * pad - S.k) / S.stride + 1; L.Wo = (L.Win + 2 * pad - S.k) / S.stride + 1; }
        const size_t a = (size_t)L.Cout * L.Ho * L.Wo;
        L.res_from = -1;
        const size_t n = strlen(S.name);
        if (n > 6 && !strcmp(S.name + n - 6, ".conv3")) L.res_from = i - 2;
        {   // parameters() order: conv weight, then BatchNorm gamma, beta -- or the head's weight, bias
            const size_t wn = (size_t)S.cout * (S.kind == K_DW ? 1 : S.cin) * S.k * S.k;      // ConvTranspose2d: [Cin][Cout][2][2], the same count
            t->param_off[p + 1] = t->param_off[p] + wn;
            t->param_off[p + 2] = t->param_off[p + 1] + S.cout;
            if (S.kind != K_HEAD) t->param_off[p + 3] = t->param_off[p + 2] + S.cout;
        }
        if (S.kind == K_HEAD) { L.z = L.y = 0; L.st = 0; L.p0 = p; p += 2; continue; }     // the heads write into the caller's tensors
        L.z = off; off += a;
        L.y = off; off += a;
        L.st = st; st += 2 * (size_t)L.Cout;
        L.p0 = p; p += 3;
        if (a > t->gmax) t->gmax = a;
    }
    t->cat = off;
    off += (size_t)t->L[t->i_c411].Cin * t->L[t->i_c411].Hin * t->L[t->i_c411].Win;
    if ((size_t)t->L[t->i_c411].Cin * t->L[t->i_c411].Hin * t->L[t->i_c411].Win > t->gmax)
        t->gmax = (size_t)t->L[t->i_c411].Cin * t->L[t->i_c411].Hin * t->L[t->i_c411].Win;
    if ((size_t)kLayers[0].cin * H * W > t->gmax) t->gmax = (size_t)kLayers[0].cin * H * W;
    t->act_floats = off;
    t->stats_floats = st;
    t->n_params = p;
    t->ga2 = (size_t)t->L[t->i_conv4_2].Cout * t->L[t->i_conv4_2].Ho * t->L[t->i_conv4_2].Wo;
    t->gb2 = (size_t)t->L[t->i_conv5_2].Cout * t->L[t->i_conv5_2].Ho * t->L[t->i_conv5_2].Wo;
    t->gd = (size_t)t->L[t->i_deconv].Cout * t->L[t->i_deconv].Ho * t->L[t->i_deconv].Wo;
    return 0;
}

// workspace: [scratch | stats | activations x N | 4 gradient buffers x N x gmax | ga2, gb2, gd x N]
struct TWs {
    char* scratch; float* stats; float* act; float* g[4]; float* ga2; float* gb2; float* gd; float* slabs;
    size_t bytes, slab_floats;
};
// room for every layer's weight-gradient slabs at once: a layer splits into at most min(1024, pixels / 128) slices, and never needs
// more than the shared scratch holds
size_t trainer_slab_floats(const yf_trainer_s* t, int N)
{
    const LayerSpec* const kLayers = t->layers;
    size_t total = 0;
    for (int i = 0; i < kNumLayers; ++i) {
        const LayerSpec& S = kLayers[i];
        const TLayer& L = t->L[i];
        const size_t nw = (size_t)S.cout * (S.kind == K_DW ? 1 : S.cin) * S.k * S.k;
        size_t P = (size_t)N * (S.kind == K_DECONV ? (size_t)L.Hin * L.Win : (size_t)L.Ho * L.Wo), ns = (P + 127) / 128;
        if (ns > 1024) ns = 1024;
        size_t f = ns * nw;
        if (f > yf::train_scratch_bytes() / 4) f = yf::train_scratch_bytes() / 4;
        total += (f + 63) & ~(size_t)63;
    }
    return total;
}
TWs trainer_ws(const yf_trainer_s* t, int N, void* base)
{
    TWs w;
    char* p = static_cast<char*>(base);
    auto take = [&](size_t bytes) { char* r = p; p += (bytes + 255) & ~(size_t)255; return r; };
    w.scratch = take(yf::train_scratch_bytes());
    w.stats = reinterpret_cast<float*>(take(t->stats_floats * 4));
    w.act = reinterpret_cast<float*>(take(t->act_floats * N * 4));
    for (int i = 0; i < 4; ++i) w.g[i] = reinterpret_cast<float*>(take(t->gmax * N * 4));
    w.ga2 = reinterpret_cast<float*>(take(t->ga2 * N * 4));
    w.gb2 = reinterpret_cast<float*>(take(t->gb2 * N * 4));
    w.gd = reinterpret_cast<float*>(take(t->gd * N * 4));
    w.slab_floats = trainer_slab_floats(t, N);
    w.slabs = reinterpret_cast<float*>(take(w.slab_floats * 4));
    w.bytes = (size_t)(p - static_cast<char*>(base));
    return w;
}

// developer timing (YF_TRAIN_TIMING=1): an event after every layer of a pass, the table printed to stderr when the pass has drained
struct PassTimer {
    bool on;
    hipStream_t s;
    std::vector<hipEvent_t> ev;
    std::vector<std::string> label;
    PassTimer(hipStream_t s_) : on(getenv("YF_TRAIN_TIMING") != nullptr), s(s_) { tick("start"); }
    void tick(const char* what, const char* layer = "")
    {
        if (!on) return;
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) { on = false; return; }
        (void)hipEventRecord(e, s);
        ev.push_back(e);
        label.push_back(std::string(what) + " " + layer);
    }
    void report(const char* pass)
    {
        if (!on) return;
        (void)hipStreamSynchronize(s);
        for (size_t i = 1; i < ev.size(); ++i) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, ev[i - 1], ev[i]);
            fprintf(stderr, "[yf_train_timing] %s %-28s %9.1f us\n", pass, label[i].c_str(), ms * 1e3f);
        }
        for (hipEvent_t e : ev) (void)hipEventDestroy(e);
    }
};

}  // namespace

extern "C" {

// ---- training-step operators (second slice); NCHW fp32 device pointers, stream-ordered ----
#define YF_TOP(cond, call)                                                          \
    do {                                                                            \
        if (!(cond)) return fail(YF_E_INVALID, "%s: bad argument", __func__);       \
        HIP_OK(hipSetDevice(device));                                               \
        call;                                                                       \
        HIP_OK(hipGetLastError());                                                  \
        return YF_OK;                                                               \
    } while (0)

int yf_train_conv_forward(int device, const float* d_x, const float* d_w, const float* d_bias, float* d_y, int N, int Cin, int H, int W, int Cout,
                          int k, int stride, int depthwise, void* stream)
{
    YF_TOP(d_x && d_w && d_y && N > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && (k == 1 || k == 3 || k == 5) && (stride == 1 || stride == 2) &&
               (!depthwise || Cin == Cout),
           yf::launch_tconv_fwd(d_x, d_w, d_bias, d_y, N, Cin, H, W, Cout, k, stride, depthwise, (hipStream_t)stream));
}
int yf_train_conv_backward_data(int device, const float* d_dy, const float* d_w, float* d_dx, int N, int Cin, int H, int W, int Cout, int k,
                                int stride, int depthwise, void* stream)
{
    YF_TOP(d_dy && d_w && d_dx && N > 0 && (k == 1 || k == 3 || k == 5) && (stride == 1 || stride == 2) && (!depthwise || Cin == Cout),
           yf::launch_tconv_bwd_data(d_dy, d_w, d_dx, N, Cin, H, W, Cout, k, stride, depthwise, (hipStream_t)stream));
}
int yf_train_conv_backward_weight(int device, const float* d_x, const float* d_dy, float* d_dw, int N, int Cin, int H, int W, int Cout, int k,
                                  int stride, int depthwise, void* d_scratch, size_t scratch_bytes, void* stream)
{
    YF_TOP(d_x && d_dy && d_dw && N > 0 && (k == 1 || k == 3 || k == 5) && (stride == 1 || stride == 2) && (!depthwise || Cin == Cout),
           yf::launch_tconv_bwd_weight(d_x, d_dy, d_dw, N, Cin, H, W, Cout, k, stride, depthwise, d_scratch, scratch_bytes, (hipStream_t)stream));
}
int yf_train_deconv_forward(int device, const float* d_x, const float* d_w, float* d_y, int N, int Cin, int H, int W, int Cout, void* stream)
{
    YF_TOP(d_x && d_w && d_y && N > 0, yf::launch_tdeconv_fwd(d_x, d_w, d_y, N, Cin, H, W, Cout, (hipStream_t)stream));
}
int yf_train_deconv_backward_data(int device, const float* d_dy, const float* d_w, float* d_dx, int N, int Cin, int H, int W, int Cout, void* stream)
{
    YF_TOP(d_dy && d_w && d_dx && N > 0, yf::launch_tdeconv_bwd_data(d_dy, d_w, d_dx, N, Cin, H, W, Cout, (hipStream_t)stream));
}
int yf_train_deconv_backward_weight(int device, const float* d_x, const float* d_dy, float* d_dw, int N, int Cin, int H, int W, int Cout,
                                    void* d_scratch, size_t scratch_bytes, void* stream)
{
    YF_TOP(d_x && d_dy && d_dw && N > 0,
           yf::launch_tdeconv_bwd_weight(d_x, d_dy, d_dw, N, Cin, H, W, Cout, d_scratch, scratch_bytes, (hipStream_t)stream));
}
int yf_train_scratch_bytes(size_t* bytes)
{
    if (!bytes) return fail(YF_E_INVALID, "yf_train_scratch_bytes: null argument");
    *bytes = yf::train_scratch_bytes();
    return YF_OK;
}
int yf_train_bn_forward(int device, const float* d_x, const float* d_gamma, const float* d_beta, float* d_running_mean, float* d_running_var,
                        float* d_stats, float* d_y, int N, int C, long HW, int relu, void* d_scratch, void* stream)
{
    YF_TOP(d_x && d_gamma && d_beta && d_stats && d_y && d_scratch && N > 0 && C > 0 && C <= 256 && HW > 0,
           yf::launch_tbn_fwd(d_x, d_gamma, d_beta, d_running_mean, d_running_var, d_stats, d_y, N, C, HW, relu, d_scratch, (hipStream_t)stream));
}
int yf_train_bn_backward(int device, const float* d_x, const float* d_dy, const float* d_stats, const float* d_gamma, const float* d_beta,
                         float* d_dgamma, float* d_dbeta, float* d_dx, int N, int C, long HW, int relu, void* d_scratch, void* stream)
{
    YF_TOP(d_x && d_dy && d_stats && d_gamma && d_beta && d_dgamma && d_dbeta && d_dx && d_scratch && N > 0 && C > 0 && C <= 256 && HW > 0,
           yf::launch_tbn_bwd(d_x, d_dy, d_stats, d_gamma, d_beta, d_dgamma, d_dbeta, d_dx, N, C, HW, relu, d_scratch, (hipStream_t)stream));
}
int yf_train_channel_sum(int device, const float* d_dy, float* d_out, int N, int C, long HW, void* stream)
{
    YF_TOP(d_dy && d_out && N > 0 && C > 0 && HW > 0, yf::launch_tchan_sum(d_dy, d_out, N, C, HW, (hipStream_t)stream));
}
int yf_train_channel_sum_split(int device, const float* d_dy, float* d_out, int N, int C, long HW, void* d_scratch, void* stream)
{
    YF_TOP(d_dy && d_out && d_scratch && N > 0 && C > 0 && C <= 4096 && HW > 0, yf::launch_tchan_sum(d_dy, d_out, N, C, HW, (hipStream_t)stream, d_scratch));
}
int yf_train_add(int device, const float* d_a, const float* d_b, float* d_out, long total, void* stream)
{
    YF_TOP(d_a && d_b && d_out && total > 0, yf::launch_tadd(d_a, d_b, d_out, total, (hipStream_t)stream));
}
int yf_train_channel_slice(int device, const float* d_src, float* d_dst, int N, int C, long HW, int Cs, int sc0, int Cd, int dc0, void* stream)
{
    YF_TOP(d_src && d_dst && N > 0 && C > 0 && sc0 >= 0 && dc0 >= 0 && sc0 + C <= Cs && dc0 + C <= Cd,
           yf::launch_tslice(d_src, d_dst, N, C, HW, Cs, sc0, Cd, dc0, (hipStream_t)stream));
}
int yf_train_adam_step(int device, float* d_p, const float* d_g, float* d_m, float* d_v, long total, double lr, double beta1, double beta2,
                       double eps, int step, void* stream)
{
    YF_TOP(d_p && d_g && d_m && d_v && total > 0 && step >= 1,
           yf::launch_tadam(d_p, d_g, d_m, d_v, total, lr, beta1, beta2, eps, step, (hipStream_t)stream));
}
// conv / deconv + BatchNorm (+ ReLU) as one call each way: what the reference's conv_norm_relu / conv_norm / deconv_norm_relu blocks do
int yf_train_unit_forward(int device, int deconv, const float* d_x, const float* d_w, const float* d_gamma, const float* d_beta,
                          float* d_running_mean, float* d_running_var, float* d_stats, float* d_z, float* d_y, int N, int Cin, int H, int W, int Cout,
                          int k, int stride, int depthwise, int relu, void* d_scratch, void* stream)
{
    if (!d_x || !d_w || !d_gamma || !d_beta || !d_stats || !d_z || !d_y || !d_scratch || N <= 0 || Cin <= 0 || Cout <= 0 || Cout > 256 || H <= 0 ||
        W <= 0 || (depthwise && Cin != Cout))
        return fail(YF_E_INVALID, "yf_train_unit_forward: bad argument");
    if (!deconv && !((k == 1 || k == 3 || k == 5) && (stride == 1 || stride == 2))) return fail(YF_E_INVALID, "yf_train_unit_forward: bad kernel / stride");
    HIP_OK(hipSetDevice(device));
    hipStream_t s = (hipStream_t)stream;
    long HWo;
    // the conv may leave BatchNorm's partial sums behind the first MB of the scratch (yf_kernels.h: TStatPart)
    yf::TStatPart sp{reinterpret_cast<float2*>(static_cast<char*>(d_scratch) + (1 << 20)), yf::train_scratch_bytes() - (1 << 20), 0};
    if (deconv) {
        yf::launch_tdeconv_fwd(d_x, d_w, d_z, N, Cin, H, W, Cout, s);
        HWo = 4L * H * W;
    } else {
        yf::launch_tconv_fwd(d_x, d_w, nullptr, d_z, N, Cin, H, W, Cout, k, stride, depthwise, s, &sp);
        const int pad = (k - 1) / 2;
        HWo = (long)((H + 2 * pad - k) / stride + 1) * ((W + 2 * pad - k) / stride + 1);
    }
    yf::launch_tbn_fwd(d_z, d_gamma, d_beta, d_running_mean, d_running_var, d_stats, d_y, N, Cout, HWo, relu, d_scratch, s, nullptr, &sp);
    HIP_OK(hipGetLastError());
    return YF_OK;
}
int yf_train_unit_backward(int device, int deconv, const float* d_x, const float* d_z, const float* d_gy, const float* d_stats, const float* d_w,
                           const float* d_gamma, const float* d_beta, float* d_dgamma, float* d_dbeta, float* d_gz, float* d_dw, float* d_dx, int N,
                           int Cin, int H, int W, int Cout, int k, int stride, int depthwise, int relu, void* d_scratch, size_t scratch_bytes,
                           void* stream)
{
    if (!d_x || !d_z || !d_gy || !d_stats || !d_w || !d_gamma || !d_beta || !d_dgamma || !d_dbeta || !d_gz || !d_dw || !d_scratch || N <= 0 || Cin <= 0 ||
        Cout <= 0 || Cout > 256 || (depthwise && Cin != Cout))
        return fail(YF_E_INVALID, "yf_train_unit_backward: bad argument");
    if (!deconv && !((k == 1 || k == 3 || k == 5) && (stride == 1 || stride == 2))) return fail(YF_E_INVALID, "yf_train_unit_backward: bad kernel / stride");
    HIP_OK(hipSetDevice(device));
    hipStream_t s = (hipStream_t)stream;
    long HWo;
    if (deconv) {
        HWo = 4L * H * W;
    } else {
        const int pad = (k - 1) / 2;
        HWo = (long)((H + 2 * pad - k) / stride + 1) * ((W + 2 * pad - k) / stride + 1);
    }
    yf::launch_tbn_bwd(d_z, d_gy, d_stats, d_gamma, d_beta, d_dgamma, d_dbeta, d_gz, N, Cout, HWo, relu, d_scratch, s);
    if (deconv) {
        yf::launch_tdeconv_bwd_weight(d_x, d_gz, d_dw, N, Cin, H, W, Cout, d_scratch, scratch_bytes, s);
        if (d_dx) yf::launch_tdeconv_bwd_data(d_gz, d_w, d_dx, N, Cin, H, W, Cout, s);
    } else {
        yf::launch_tconv_bwd_weight(d_x, d_gz, d_dw, N, Cin, H, W, Cout, k, stride, depthwise, d_scratch, scratch_bytes, s);
        if (d_dx) yf::launch_tconv_bwd_data(d_gz, d_w, d_dx, N, Cin, H, W, Cout, k, stride, depthwise, s);
    }
    HIP_OK(hipGetLastError());
    return YF_OK;
}
int yf_train_adam_multi(int device, int ntensors, void* const* d_p, const void* const* d_g, void* const* d_m, void* const* d_v, const long* sizes,
                        double lr, double beta1, double beta2, double eps, int step, void* d_table, size_t table_bytes, void* stream)
{
    return yf_train_adam_multi_pinned(device, ntensors, d_p, d_g, d_m, d_v, sizes, lr, beta1, beta2, eps, step, d_table, table_bytes, nullptr, 1, stream);
}

int yf_train_adam_multi_pinned(int device, int ntensors, void* const* d_p, const void* const* d_g, void* const* d_m, void* const* d_v,
                               const long* sizes, double lr, double beta1, double beta2, double eps, int step, void* d_table, size_t table_bytes,
                               void* h_table_pinned, int upload, void* stream)
{
    if (!d_p || !d_g || !d_m || !d_v || !sizes || !d_table || ntensors <= 0 || step < 1 || table_bytes < (size_t)ntensors * 48)
        return fail(YF_E_INVALID, "yf_train_adam_multi: bad argument");
    for (int t = 0; t < ntensors; ++t)
        if (!d_p[t] || !d_g[t] || !d_m[t] || !d_v[t] || sizes[t] <= 0) return fail(YF_E_INVALID, "yf_train_adam_multi: tensor %d: null pointer or empty", t);
    HIP_OK(hipSetDevice(device));
    if (yf::launch_tadam_multi(ntensors, (float* const*)d_p, (const float* const*)d_g, (float* const*)d_m, (float* const*)d_v, sizes, lr, beta1, beta2,
                               eps, step, d_table, h_table_pinned, upload, (hipStream_t)stream))
        return fail(YF_E_HIP, "yf_train_adam_multi: the upload of the pointer table failed");
    HIP_OK(hipGetLastError());
    return YF_OK;
}
// ---- the trainer: forward and backward of the whole network as one call each ----
int yf_trainer_create(int H, int W, int device, yf_trainer* out) { return yf_trainer_create_ex(H, W, device, 1, 24, out); }

int yf_trainer_create_ex(int H, int W, int device, int input_channel, int num_out, yf_trainer* out)
{
    if (!out || H <= 0 || W <= 0 || H % 32 || W % 32) return fail(YF_E_INVALID, "yf_trainer_create: H and W must be positive multiples of 32");
    if (input_channel < 1 || input_channel > yf_layers::MAX_INPUT_CHANNEL || num_out < 1 || num_out > yf_layers::MAX_NUM_OUT)   // the inference engine's limits
        return fail(YF_E_INVALID, "yf_trainer_create: input_channel must be 1..%d and num_out 1..%d", (int)yf_layers::MAX_INPUT_CHANNEL, (int)yf_layers::MAX_NUM_OUT);
    yf_trainer_s* t = new yf_trainer_s();
    t->device = device;
    make_layers(t->layers, input_channel, num_out);
    if (trainer_build(t, H, W)) { delete t; return fail(YF_E_INVALID, "yf_trainer_create: layer table inconsistent"); }
    *out = t;
    return YF_OK;
}
void yf_trainer_destroy(yf_trainer t)
{
    if (t) {
        if (t->d_sum_tab) (void)hipFree(t->d_sum_tab);
        for (yf_trainer_s::PassGraphs* pg : {&t->gfwd, &t->gbwd})
            for (yf_trainer_s::PassGraph& g : pg->g)
                if (g.exec) { (void)hipGraphExecDestroy(g.exec); (void)hipGraphDestroy(g.graph); }
        if (t->cap_stream) (void)hipStreamDestroy(t->cap_stream);
    }
    delete t;
}
int yf_trainer_num_params(yf_trainer t, int* n_params, int* n_bn)
{
    if (!t || !n_params || !n_bn) return fail(YF_E_INVALID, "yf_trainer_num_params: null argument");
    *n_params = t->n_params;
    *n_bn = kNumLayers - 2;
    return YF_OK;
}
int yf_trainer_workspace_bytes(yf_trainer t, int N, size_t* bytes)
{
    if (!t || !bytes || N <= 0) return fail(YF_E_INVALID, "yf_trainer_workspace_bytes: bad argument");
    *bytes = trainer_ws(t, N, nullptr).bytes;
    return YF_OK;
}
static int trainer_forward_launches(yf_trainer t, const float* d_x, int N, const void* const* d_params, void* const* d_bn_buffers,
                                    float* d_head_large, float* d_head_small, void* d_ws, hipStream_t s)
{
    const TWs w = trainer_ws(t, N, d_ws);
    const LayerSpec* const kLayers = t->layers;
    auto P = [&](int i) { return static_cast<const float*>(d_params[i]); };
    int bn = 0;
    PassTimer tm(s);
    for (int i = 0; i < kNumLayers; ++i) {
        const LayerSpec& S = kLayers[i];
        const TLayer& L = t->L[i];
        const float* x = L.in == -1 ? d_x : L.in == -2 ? w.act + t->cat * N : w.act + t->L[L.in].y * N;
        if (S.kind == K_HEAD) {
            yf::launch_tconv_fwd(x, P(L.p0), P(L.p0 + 1), i == t->i_head4 ? d_head_large : d_head_small, N, L.Cin, L.Hin, L.Win, L.Cout, 1, 1, 0, s);
            tm.tick("conv", S.name);
            continue;
        }
        float* z = w.act + L.z * N;
        float* y = w.act + L.y * N;
        yf::TStatPart sp{reinterpret_cast<float2*>(w.scratch + (1 << 20)), yf::train_scratch_bytes() - (1 << 20), 0};
        if (S.kind == K_DECONV) yf::launch_tdeconv_fwd(x, P(L.p0), z, N, L.Cin, L.Hin, L.Win, L.Cout, s);
        else yf::launch_tconv_fwd(x, P(L.p0), nullptr, z, N, L.Cin, L.Hin, L.Win, L.Cout, S.k, S.stride, S.kind == K_DW, s, &sp);
        tm.tick("conv", S.name);
        float* rm = d_bn_buffers ? static_cast<float*>(d_bn_buffers[2 * bn]) : nullptr;
        float* rv = d_bn_buffers ? static_cast<float*>(d_bn_buffers[2 * bn + 1]) : nullptr;
        ++bn;
        const float* residual = nullptr;                                        // out += residual, fused into the BatchNorm pass   yolo_fastest.py:65
        if (L.res_from >= 0) {
            const TLayer& R = t->L[L.res_from];
            residual = R.in == -1 ? d_x : w.act + t->L[R.in].y * N;
        }
        yf::launch_tbn_fwd(z, P(L.p0 + 1), P(L.p0 + 2), rm, rv, w.stats + L.st, y, N, L.Cout, (long)L.Ho * L.Wo, S.relu, w.scratch, s, residual, &sp);
        tm.tick("bn", S.name);
        if (i == t->i_deconv) {                                                 // torch.cat((conv4_2, deconv5_1), 1)        :209
            const TLayer& A = t->L[t->i_conv4_2];
            float* cat = w.act + t->cat * N;
            const long HW = (long)A.Ho * A.Wo;
            yf::launch_tslice(w.act + A.y * N, cat, N, A.Cout, HW, A.Cout, 0, A.Cout + L.Cout, 0, s);
            yf::launch_tslice(y, cat, N, L.Cout, HW, L.Cout, 0, A.Cout + L.Cout, A.Cout, s);
            tm.tick("cat", S.name);
        }
    }
    tm.report("fwd");
    HIP_OK(hipGetLastError());
    return YF_OK;
}
static int trainer_backward_launches(yf_trainer t, const float* d_x, const float* d_grad_head_large, const float* d_grad_head_small, int N,
                                     const void* const* d_params, void* const* d_grads, void* d_ws, hipStream_t s, bool capturing)
{
    const TWs w = trainer_ws(t, N, d_ws);
    const LayerSpec* const kLayers = t->layers;
    auto P = [&](int i) { return static_cast<const float*>(d_params[i]); };
    auto G = [&](int i) { return static_cast<float*>(d_grads[i]); };
    const size_t sb = yf::train_scratch_bytes();
    auto xin = [&](const TLayer& L) { return L.in == -1 ? d_x : L.in == -2 ? (const float*)(w.act + t->cat * N) : (const float*)(w.act + t->L[L.in].y * N); };
    // gradient buffers: `cur` holds the gradient flowing backwards, `skip` a block's output gradient until the block's input is reached
    int cur = 0, skip = -1;
    PassTimer tm(s);
    // the gradients of a flat parameters()-order buffer (what training.py passes): the split weight-gradient sums of all layers become one
    // launch at the end of the pass.  Pointers that are not laid out that way: every layer sums its own slabs as before.
    bool flat = getenv("YF_TRAIN_SUM_EACH") == nullptr;
    for (int i = 1; i < t->n_params && flat; ++i) flat = (const float*)d_grads[i] == (const float*)d_grads[0] + t->param_off[i];
    yf::TSumDefer defer{w.slabs, w.slab_floats, 0, static_cast<float*>(d_grads[0]), 0, {}};
    yf::TSumDefer* dfr = flat ? &defer : nullptr;
    auto other = [&](int a, int b, int c) { for (int i = 0; i < 4; ++i) if (i != a && i != b && i != c) return i; return -1; };
    // backward of one conv + BN (+ ReLU) unit: gradient of its output in gy -> parameter gradients, gradient of its input in w.g[ret]
    // The data-gradient kernel of layer i produces dy of layer j = its input; where that IS the whole gradient of j's output (every layer
    // but the two branch points, whose second gradient is added afterwards) it also leaves j's backward BatchNorm sums (TBnRed): j's
    // backward then skips its reduction pass over dy and z.  red_for: the layer the pairs in the scratch belong to.
    yf::TBnRed red{nullptr, nullptr, nullptr, nullptr, 0, reinterpret_cast<float2*>(w.scratch + (1 << 20)), yf::train_scratch_bytes() - (1 << 20), 0};
    int red_for = -1;
    auto unit = [&](int i, const float* gy, bool need_dx, const float* addend = nullptr) {
        const LayerSpec& S = kLayers[i];
        const TLayer& L = t->L[i];
        const int iz = other(cur, skip, -1), ix = other(cur, skip, iz);
        float* gz = w.g[iz];
        yf::launch_tbn_bwd(w.act + L.z * N, gy, w.stats + L.st, P(L.p0 + 1), P(L.p0 + 2), G(L.p0 + 1), G(L.p0 + 2), gz, N, L.Cout, (long)L.Ho * L.Wo,
                           S.relu, w.scratch, s, (red_for == i && red.count > 0) ? &red : nullptr);
        red_for = -1;
        yf::TBnRed* rp = nullptr;
        if (need_dx && L.in >= 0 && L.in != t->i_conv4_2 && L.in != t->i_conv5_2 && kLayers[L.in].kind != K_HEAD && S.kind != K_DECONV) {
            const TLayer& J = t->L[L.in];
            red.z = w.act + J.z * N; red.stats = w.stats + J.st; red.gamma = P(J.p0 + 1); red.beta = P(J.p0 + 2); red.relu = kLayers[L.in].relu;
            red.count = 0;
            rp = &red;
        }
        tm.tick("bn", S.name);
        if (S.kind == K_DECONV) {
            yf::launch_tdeconv_bwd_weight(xin(L), gz, G(L.p0), N, L.Cin, L.Hin, L.Win, L.Cout, w.scratch, sb, s, dfr);
            tm.tick("wgrad", S.name);
            if (need_dx) yf::launch_tdeconv_bwd_data(gz, P(L.p0), w.g[ix], N, L.Cin, L.Hin, L.Win, L.Cout, s);
        } else if (S.kind == K_PW && need_dx && !tm.on &&
                   yf::launch_tpw_bwd_dual(xin(L), gz, P(L.p0), G(L.p0), w.g[ix], addend, N, L.Cin, L.Hin, L.Win, L.Cout, w.scratch, sb, s, dfr)) {
            // (pointwise layer on a small map: both gradients in one launch)
        } else {
            yf::launch_tconv_bwd_weight(xin(L), gz, G(L.p0), N, L.Cin, L.Hin, L.Win, L.Cout, S.k, S.stride, S.kind == K_DW, w.scratch, sb, s, dfr);
            tm.tick("wgrad", S.name);
            if (need_dx) yf::launch_tconv_bwd_data(gz, P(L.p0), w.g[ix], N, L.Cin, L.Hin, L.Win, L.Cout, S.k, S.stride, S.kind == K_DW, s, addend, rp);
            if (rp && red.count > 0) red_for = L.in;
        }
        tm.tick("dgrad", S.name);
        return ix;
    };
    auto head = [&](int i, const float* gy) {                                  // nn.Conv2d(C, 24, 1) with bias
        const TLayer& L = t->L[i];
        const int ix = other(cur, skip, -1);
        yf::launch_tconv_bwd_weight(xin(L), gy, G(L.p0), N, L.Cin, L.Hin, L.Win, L.Cout, 1, 1, 0, w.scratch, sb, s, dfr);
        yf::launch_tchan_sum(gy, G(L.p0 + 1), N, L.Cout, (long)L.Hin * L.Win, s, w.scratch);
        yf::launch_tconv_bwd_data(gy, P(L.p0), w.g[ix], N, L.Cin, L.Hin, L.Win, L.Cout, 1, 1, 0, s);
        tm.tick("head", kLayers[i].name);
        return ix;
    };
    // a run of layers hi .. lo (inclusive), backwards; the gradient of layer hi's output is in w.g[cur] on entry, the gradient of layer
    // lo's input is in w.g[cur] on exit
    auto run_back = [&](int hi, int lo, bool first_needs_dx) {
        for (int i = hi; i >= lo; --i) {
            const TLayer& L = t->L[i];
            if (L.res_from >= 0) skip = cur;                                    // conv3 of a block: its output gradient also goes to the skip
            const bool block_input = skip >= 0 && i + 2 < kNumLayers && t->L[i + 2].res_from == i;       // conv1 of that block:
            const int nx = unit(i, w.g[cur], i > lo || first_needs_dx, block_input ? w.g[skip] : nullptr);   // + the skip gradient, fused
            cur = nx;
            if (block_input) skip = -1;
        }
    };
    const TLayer& A = t->L[t->i_conv4_2];
    const TLayer& D = t->L[t->i_deconv];
    const long HWa = (long)A.Ho * A.Wo;
    // head_large branch: head_4, conv4_1_5 .. conv4_1_1, the concat                                   yolo_fastest.py:209-216
    cur = head(t->i_head4, d_grad_head_large);
    run_back(t->i_c415, t->i_c411, true);
    yf::launch_tslice(w.g[cur], w.ga2, N, A.Cout, HWa, A.Cout + D.Cout, 0, A.Cout, 0, s);
    yf::launch_tslice(w.g[cur], w.gd, N, D.Cout, HWa, A.Cout + D.Cout, A.Cout, D.Cout, 0, s);
    {   // deconv5_1: gradient of conv5_2's output, first part
        const int nx = unit(t->i_deconv, w.gd, true);
        HIP_OK(hipMemcpyAsync(w.gb2, w.g[nx], t->gb2 * N * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    // head_small branch: head_5, conv5_6 .. conv5_3                                                     :201-206
    cur = head(t->i_head5, d_grad_head_small);
    run_back(t->i_conv5_6, t->i_conv5_3, true);
    yf::launch_tadd(w.g[cur], w.gb2, w.g[cur], (long)N * t->gb2, s);
    run_back(t->i_conv5_2, t->i_conv4_3, true);                                  // conv5_2 .. conv4_3             :191-200
    yf::launch_tadd(w.g[cur], w.ga2, w.g[cur], (long)N * t->ga2, s);
    run_back(t->i_conv4_2, 0, false);                                            // conv4_2 .. conv0; the images need no gradient
    if (dfr && !defer.entries.empty()) {
        const size_t nb = defer.entries.size() * sizeof(yf::TSumEntry);
        if (t->sum_tab.size() != defer.entries.size() || memcmp(t->sum_tab.data(), defer.entries.data(), nb)) {
            // first pass at this batch size: the table goes to the device once (nothing of this trainer may still be reading the old one)
            if (capturing) return 1;                         // (another batch size ran in between: this call goes out as plain launches, which rebuild the table)
            HIP_OK(hipStreamSynchronize(s));
            for (yf_trainer_s::PassGraph& g : t->gbwd.g)         // a captured backward has the old table's address in its last node
                if (g.exec) { (void)hipGraphExecDestroy(g.exec); (void)hipGraphDestroy(g.graph); g.exec = nullptr; g.graph = nullptr; g.key.clear(); }
            if (t->d_sum_tab) (void)hipFree(t->d_sum_tab);
            t->d_sum_tab = nullptr;
            HIP_OK(hipMalloc(&t->d_sum_tab, nb));
            HIP_OK(hipMemcpy(t->d_sum_tab, defer.entries.data(), nb, hipMemcpyHostToDevice));
            t->sum_tab = defer.entries;
        }
        yf::launch_tsum_multi(t->d_sum_tab, (int)defer.entries.size(), defer.nblocks, w.slabs, defer.dst_base, s);
        tm.tick("wsum", "all");
    }
    tm.report("bwd");
    HIP_OK(hipGetLastError());
    return YF_OK;
}

// One pass = a few hundred launches of 3-4 us of host time each: at the reference's batch 16 that is as long as the kernels run.  A pass
// whose arguments are a pointer set seen before (the steady state of a training loop: the caching allocator hands the same blocks
// back every iteration, or alternates between two sets) is captured -- on the trainer's own stream, since the caller's may be the legacy default stream,
// which cannot capture; nothing executes there -- and replayed on the caller's stream from then on.  Up to two graphs per pass (a loop
// whose allocations alternate between two sets).  Any other pointer set runs as plain launches.  Small batches only (below).
// YF_TRAIN_GRAPH_OFF=1: never; YF_TRAIN_GRAPH_ALWAYS=1: at every batch size.
}  // extern "C"
template <class Body>
static int run_pass(yf_trainer_s* t, yf_trainer_s::PassGraphs& pg, std::vector<uintptr_t>& key, int N, hipStream_t s, Body body)
{
    static const bool off = getenv("YF_TRAIN_GRAPH_OFF") != nullptr || getenv("YF_TRAIN_TIMING") != nullptr;
    static const bool always = getenv("YF_TRAIN_GRAPH_ALWAYS") != nullptr;
    // only where the host is the slower side: a replayed node costs the GPU ~1 us more than a plain launch (measured at batch 256:
    // 19.5 -> 20.1 ms with ~535 nodes), which a 4 ms iteration wins back several times over on the host and a 20 ms one does not
    // the caller's stream captures (torch.cuda.graph around a train step): emit the launches into ITS capture -- no graph launch into a
    // capturing stream, and nothing host-synchronous: a pass that would have to (re)build its sum table first says so instead
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
        const int rc = body(s, true);
        if (rc == 1)
            return fail(YF_E_INVALID, "the caller's stream is capturing and this pass has to upload its weight-gradient sum table first "
                                      "(a host-synchronous step): run one eager iteration at this batch size before capturing");
        return rc;
    }
    if (off || !t->graphs_on || pg.failures >= 2 || pg.thrash >= 4 || (!always && (long)N * t->H * t->W > 40L * 256 * 320)) return body(s, false);   // (two failed captures / a thrashing pointer pattern: never again)
    for (yf_trainer_s::PassGraph& g : pg.g)
        if (g.exec && g.key == key) {
            HIP_OK(hipGraphLaunch(g.exec, s));
            ++pg.replays; ++g.uses;
            return YF_OK;
        }
    bool known = false;
    for (const std::vector<uintptr_t>& k : pg.seen) known = known || k == key;
    if (!known) {                                            // new pointers: plain launches (this also sets up attributes and tables)
        pg.seen[pg.nseen++ & 3] = key;
        return body(s, false);
    }
    if (!t->cap_stream && hipStreamCreateWithFlags(&t->cap_stream, hipStreamNonBlocking) != hipSuccess) { t->cap_stream = nullptr; return body(s, false); }
    yf_trainer_s::PassGraph& g = pg.g[pg.next];
    pg.next = (pg.next + 1) & 3;
    if (g.exec) {                                            // a fifth pointer set evicts the oldest graph (it may still be running on s)
        (void)hipStreamSynchronize(s);
        (void)hipGraphExecDestroy(g.exec); (void)hipGraphDestroy(g.graph); g.exec = nullptr; g.graph = nullptr;
        ++pg.evictions;
        if (g.uses < 3) ++pg.thrash;                         // evicted before it paid for its capture
        g.uses = 0;
    }
    static const bool dbg = getenv("YF_TRAIN_GRAPH_DEBUG") != nullptr;
    {
        const hipError_t pre = hipGetLastError();
        if (dbg && pre != hipSuccess) fprintf(stderr, "[run_pass] error pending before capture: %s\n", hipGetErrorString(pre));
    }
    if (hipStreamBeginCapture(t->cap_stream, hipStreamCaptureModeRelaxed) != hipSuccess) return body(s, false);
    const int rc = body(t->cap_stream, true);
    hipGraph_t graph = nullptr;
    const hipError_t e = hipStreamEndCapture(t->cap_stream, &graph);
    if (dbg) fprintf(stderr, "[run_pass] %s capture N=%d: body rc %d (%s), end capture %s, graph %p\n", &pg == &t->gfwd ? "forward" : "backward", N, rc, rc ? yf_last_error_string() : "", hipGetErrorString(e), (void*)graph);
    if (rc != YF_OK || e != hipSuccess || !graph) {
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        if (rc != 1) ++pg.failures;
        return body(s, false);
    }
    if (hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0) != hipSuccess) {
        (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        g.exec = nullptr;
        ++pg.failures;
        return body(s, false);
    }
    g.graph = graph;
    g.key = key;
    g.uses = 1;
    ++pg.captures;
    HIP_OK(hipGraphLaunch(g.exec, s));
    ++pg.replays;
    return YF_OK;
}
extern "C" {

int yf_trainer_forward(yf_trainer t, const float* d_x, int N, const void* const* d_params, void* const* d_bn_buffers, float* d_head_large,
                       float* d_head_small, void* d_ws, size_t ws_bytes, void* stream)
{
    if (!t || !d_x || !d_params || !d_head_large || !d_head_small || !d_ws || N <= 0) return fail(YF_E_INVALID, "yf_trainer_forward: bad argument");
    const size_t need = trainer_ws(t, N, d_ws).bytes;
    if (ws_bytes < need) return fail(YF_E_WORKSPACE, "yf_trainer_forward: workspace too small (%zu < %zu)", ws_bytes, need);
    for (int i = 0; i < t->n_params; ++i)
        if (!d_params[i]) return fail(YF_E_INVALID, "yf_trainer_forward: parameter %d is null", i);
    HIP_OK(hipSetDevice(t->device));
    std::vector<uintptr_t> key;
    key.reserve(t->n_params + 2 * kNumLayers + 8);
    key.push_back((uintptr_t)d_x); key.push_back((uintptr_t)N); key.push_back((uintptr_t)d_head_large); key.push_back((uintptr_t)d_head_small);
    key.push_back((uintptr_t)d_ws); key.push_back((uintptr_t)(d_bn_buffers != nullptr));
    for (int i = 0; i < t->n_params; ++i) key.push_back((uintptr_t)d_params[i]);
    if (d_bn_buffers)
        for (int i = 0; i < 2 * (kNumLayers - 2); ++i) key.push_back((uintptr_t)d_bn_buffers[i]);
    return run_pass(t, t->gfwd, key, N, (hipStream_t)stream, [&](hipStream_t s, bool) {
        return trainer_forward_launches(t, d_x, N, d_params, d_bn_buffers, d_head_large, d_head_small, d_ws, s);
    });
}

int yf_trainer_backward(yf_trainer t, const float* d_x, const float* d_grad_head_large, const float* d_grad_head_small, int N,
                        const void* const* d_params, void* const* d_grads, void* d_ws, size_t ws_bytes, void* stream)
{
    if (!t || !d_x || !d_grad_head_large || !d_grad_head_small || !d_params || !d_grads || !d_ws || N <= 0)
        return fail(YF_E_INVALID, "yf_trainer_backward: bad argument");
    const size_t need = trainer_ws(t, N, d_ws).bytes;
    if (ws_bytes < need) return fail(YF_E_WORKSPACE, "yf_trainer_backward: workspace too small (%zu < %zu)", ws_bytes, need);
    for (int i = 0; i < t->n_params; ++i)
        if (!d_params[i] || !d_grads[i]) return fail(YF_E_INVALID, "yf_trainer_backward: parameter / gradient %d is null", i);
    HIP_OK(hipSetDevice(t->device));
    std::vector<uintptr_t> key;
    key.reserve(2 * t->n_params + 8);
    key.push_back((uintptr_t)d_x); key.push_back((uintptr_t)N); key.push_back((uintptr_t)d_grad_head_large); key.push_back((uintptr_t)d_grad_head_small);
    key.push_back((uintptr_t)d_ws);
    for (int i = 0; i < t->n_params; ++i) { key.push_back((uintptr_t)d_params[i]); key.push_back((uintptr_t)d_grads[i]); }
    return run_pass(t, t->gbwd, key, N, (hipStream_t)stream, [&](hipStream_t s, bool capturing) {
        return trainer_backward_launches(t, d_x, d_grad_head_large, d_grad_head_small, N, d_params, d_grads, d_ws, s, capturing);
    });
}
int yf_trainer_graph_replays(yf_trainer t, long* forward, long* backward)
{
    if (!t || !forward || !backward) return fail(YF_E_INVALID, "yf_trainer_graph_replays: null argument");
    *forward = t->gfwd.replays;
    *backward = t->gbwd.replays;
    return YF_OK;
}
int yf_trainer_set_graphs(yf_trainer t, int on)
{
    if (!t) return fail(YF_E_INVALID, "yf_trainer_set_graphs: null trainer");
    t->graphs_on = on != 0;
    return YF_OK;
}
int yf_trainer_graph_stats(yf_trainer t, long* out6)
{
    if (!t || !out6) return fail(YF_E_INVALID, "yf_trainer_graph_stats: null argument");
    out6[0] = t->gfwd.replays; out6[1] = t->gbwd.replays; out6[2] = t->gfwd.captures; out6[3] = t->gbwd.captures;
    out6[4] = t->gfwd.evictions; out6[5] = t->gbwd.evictions;
    return YF_OK;
}
#undef YF_TOP

}  // extern "C"
