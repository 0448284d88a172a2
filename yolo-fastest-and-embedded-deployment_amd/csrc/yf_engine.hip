// yf_engine.hip -- C-ABI (include/yolo_fastest_hip.h) and the execution plan of the YOLO-Fastest graph.
//
// The plan restates the graph of the reference's YoloFastest.forward
// (src/model_training/model/yolo_fastest.py:150-218) as a list of kernel launches over NHWC tensors with
// liveness-based reuse of workspace slots; it is built once per engine (yf_create) so that a forward
// pass is nothing but stream-ordered launches: no allocation, no synchronisation, graph-capturable.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/yolo_fastest_hip.h"
#include "yf_kernels.h"
#include "yf_layers.h"

using namespace yf_layers;

namespace {

thread_local char g_err[512] = "";
int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace
// the same error slot for the other translation units of the library (yf_train_engine.hip)
namespace yf {
int set_error(int code, const char* msg)
{
    snprintf(g_err, sizeof g_err, "%s", msg);
    return code;
}
}  // namespace yf
namespace {
#define HIP_OK(expr)                                                                                     \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) return fail(YF_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));      \
    } while (0)



struct BlobHeader {
    char magic[8];
    uint32_t version, n_layers, num_out, input_channel, num_anchors, num_cls;
    uint64_t data_floats;
    char pad[24];
};
struct BlobLayer {
    char name[32];
    uint32_t kind, cin, cout, k, stride, relu, w_off, b_off;
};
static_assert(sizeof(BlobHeader) == 64 && sizeof(BlobLayer) == 64, "blob layout");



enum { BUF_INPUT = -1, BUF_HEAD_LARGE = -2, BUF_HEAD_SMALL = -3 };

struct Tensor {
    std::string name;
    int C, H, W;
    int slot;      // workspace slot, or BUF_*
    int last_use;  // index of the last op reading it
    size_t elems() const { return (size_t)C * H * W; }
};
enum OpType { OP_LAYER = 0, OP_FUSED_BLOCK = 1, OP_K19 = 2, OP_MRES = 3, OP_MDW = 4, OP_MDW2 = 5, OP_DCAT = 6 };
struct Op {
    int layer;         // index into kLayers (fused ops: the first of their layers)
    int in1, in2, res; // tensor ids (-1 = none)
    int out;
    int omode;         // pw: 0 NHWC, 1 NCHW head, 2 deconv
    int type = OP_LAYER;
    int l_pre = -1, l_exp = -1, l_dw = -1, l_proj = -1;  // fused block: conv0 (optional), expand, depthwise, project
    int l_head = -1;     // OP_MDW: fused head conv (or -1)
    int l_post = -1;     // OP_MRES, fusion level 2: a trailing 1x1 conv + ReLU applied on chip (conv5_2 after res5_5); `out` is ITS tensor
    long post_off = -1;  //          offset of its packed fragments + bias in d_wmfma
    int l_dw2 = -1, l_proj2 = -1;  // OP_MDW2 (fusion level 2): the small head's second pair (conv5_5, conv5_6) chained behind l_dw / l_proj
    long mfma_off2 = -1;           //          offset of the second pair's weight stream
    int out2 = -1;       // OP_MRES conv4_2 + conv4_3 + conv5_1: the expanded tensor (conv4_2) is a second output
    int nblk = 1;        // OP_MRES: > 1 = a chain of residual blocks in one launch; block k's layers are l_exp/l_dw/l_proj + 3k
    long wstride = 0;    //          floats between the packed weight streams of consecutive chained blocks
    long mfma_off = -1;  // >= 0: pointwise layer runs on the matrix cores; offset of its packed B fragments
    int branch = 0;      // 1: the small head's launches (conv5_3 .. head_5) -- independent of the large head's once conv5_2 exists, so a
                         // forward pass issues them on a side stream of the lane, beside deconv5_1 .. head_4 (yolo_fastest.py:200-216)
    int kdt = 0;         // dtype handed to this op's kernel launcher (yf::DT_*): the engine's, or DT_F32 where a DT_F16X3 engine has
                         // no split-operand instantiation of the kernel (same fp32 storage, exact fp32 arithmetic instead)
};

}  // namespace

struct Plan {
    int H = 0, W = 0;
    std::vector<Tensor> tensors;
    std::vector<Op> ops;
    std::vector<size_t> slot_elems;   // per-frame capacity of each slot
    std::vector<size_t> slot_offset;  // per-frame offset (floats), prefix sums
    size_t frame_floats = 0;          // sum of slot capacities
};

struct yf_engine {
    int device = 0, H = 0, W = 0, max_batch = 0, chunk = 0;
    int dtype = yf::DT_F32;           // yf::DT_F32 / DT_F16 (fp16 storage, fp16 MFMA) / DT_F16X3 (fp32 storage, split-operand fp16 MFMA)
    int sdt() const { return dtype == yf::DT_F16 ? yf::DT_F16 : yf::DT_F32; }   // storage type of the activations in HBM
    size_t esz() const { return dtype == yf::DT_F16 ? 2 : 4; }
    float* d_weights = nullptr;
    float* d_wmfma = nullptr;         // MFMA B fragments of the GEMM-worthy pointwise layers (fused plan)
    int split_sums = 1;               // yf_set_split_sums: 1 (default) = the few-frames split-sum launches (same bits as the large-batch kernels since round 6); 0 = never re-associate a sum by batch size (a frame's bits do not depend on how many frames travel with it)
    float* d_esplit = nullptr;        // scratch of the few-frames forms of the stride-32 chain and the small head (mres_esplit_kernel, mdw2_esplit_kernel): partial sums
    static size_t esplit_lane_floats() { return yf::mres_esplit_scratch_floats() + yf::mdw2_esplit_scratch_floats(); }
    size_t n_floats = 0;
    // io_params of the blob (yolo_fastest.py:72-78): the graph is kBaseLayers with conv0's Cin and the two heads' Cout set from them
    int input_channel = 1, num_anchors = 3, num_cls = 3, num_out = 24;
    LayerSpec layers[kNumLayers];
    uint32_t w_off[kNumLayers], b_off[kNumLayers];
    // [0] one launch per layer (bring-up / probes), [1] block-fused, [2] (default) block-fused + the per-frame deep stage's launch
    // boundaries removed: conv5_2 rides in the res5 launch, ...
    Plan plans[3];
    int fusion = 2;
    // Chunks of the batch can run on `lanes` concurrent streams (fork/join with events around the caller's stream):
    // at the deep stages one workgroup owns a CU and is latency-bound; a second chunk in flight fills the bubbles.
    int lanes = 2;
    int profile_repeats = 1;   // yf_set_profile_repeats
    hipStream_t side[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
    // per lane: the side stream of the small-head branch and its fork / join events
    int branches = 1;                 // 0: issue the small head's launches in line (yf_set_branches)
    hipStream_t bside[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_bfork[4] = {nullptr, nullptr, nullptr, nullptr}, ev_bjoin[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_l2b[4] = {nullptr, nullptr, nullptr, nullptr};   // lane -> branch, in front of a chunk's post-process on the branch stream
    // Which of the engine's seven streams serve as lanes / branches is decided per CALLER stream, by probing (assign_streams): the
    // runtime multiplexes HIP streams onto a few hardware queues, and two streams that share one run strictly one after the other
    hipStream_t assigned_for = nullptr;
    bool assigned = false;
    struct Assignment { hipStream_t caller, side[3], bside[4]; long used; } seen[4];   // what was found for the last few caller streams (LRU)
    int n_seen = 0;
    long seen_tick = 0;
    hipEvent_t ev_probe[3] = {nullptr, nullptr, nullptr};
    size_t head_l_elems = 0, head_s_elems = 0;
    // cv::resize's per-column / per-row tables of the source sizes seen so far (cv_tables; a handful per engine, LRU)
    // one pool of CV_SLOTS x (W + H) entries, allocated at the first resize; a slot is built by a kernel on the caller's stream (no host
    // synchronisation, capturable) and guarded by an event for callers on another stream
    enum { CV_SLOTS = 32 };
    struct CvTab { int sh = 0, sw = 0; bool built = false; hipStream_t stream = nullptr; hipEvent_t ready = nullptr; long used = 0; } cvtab[CV_SLOTS];
    int4* d_cvpool = nullptr;
    // post_split_kernel (dense frames: one workgroup per frame AND class): scratch rows + per-frame tickets, grown on demand
    int post_split = 0;               // yf_set_post_split: 0 = auto (K_max >= 256 and <= 8 classes), 1 = always, 2 = never
    int32_t* d_post_tmp = nullptr; size_t post_tmp_ints = 0;
    long cv_tick = 0;
    size_t cv_scratch_bytes(int N) const { return (((size_t)N * H * W * input_channel) + 255) & ~(size_t)255; }
    const Plan& plan() const { return plans[fusion]; }
    size_t frame_floats_max() const
    {
        size_t m = 0;
        for (const Plan& p : plans) m = p.frame_floats > m ? p.frame_floats : m;
        return m;
    }
};

namespace {


struct Builder {
    Plan* e;
    const LayerSpec* kLayers = nullptr;   // the engine's layer table (make_layers)
    int num_out = 24;
    int add_tensor(const std::string& name, int C, int H, int W, int slot = 0)
    {
        e->tensors.push_back(Tensor{name, C, H, W, slot, -1});
        return (int)e->tensors.size() - 1;
    }
    int unit(const char* lname, int in, const char* out_name = nullptr, int in2 = -1, int res = -1, int ext = 0)
    {
        int li = find_layer(lname);
        const LayerSpec& L = kLayers[li];
        const Tensor& ti = e->tensors[in];
        int Ho = ti.H, Wo = ti.W, omode = 0;
        if (L.kind == K_DECONV) { Ho *= 2; Wo *= 2; omode = 2; }
        else if (L.stride == 2) { Ho /= 2; Wo /= 2; }
        if (L.kind == K_HEAD) omode = 1;
        int out = add_tensor(out_name ? out_name : lname, L.cout, Ho, Wo, ext);
        e->ops.push_back(Op{li, in, in2, res, out, omode});
        return out;
    }
    int resblock(const std::string& n, int x)
    {
        if (fused) return fused_block(nullptr, (n + ".conv1").c_str(), (n + ".conv2").c_str(), (n + ".conv3").c_str(), x, n.c_str(), true);
        int a = unit((n + ".conv1").c_str(), x);
        int b = unit((n + ".conv2").c_str(), a);
        return unit((n + ".conv3").c_str(), b, n.c_str(), -1, x);
    }
    // Consecutive residual blocks of one shape.  Where a workgroup's tile is the whole frame (strides 16 and 32 of the 320x256
    // net) they need no halo exchange and run as ONE launch (yf_mres_kernels.hip, "CHAIN"); the intermediate results stay in LDS.
    int reschain(std::initializer_list<const char*> names, int x)
    {
        const LayerSpec& L0 = kLayers[find_layer((std::string(*names.begin()) + ".conv1").c_str())];
        const Tensor& tx = e->tensors[x];
        if (!fused || names.size() < 2 || !yf::mres_can_chain(L0.cin, L0.cout, L0.cin, tx.H, tx.W)) {
            for (const char* n : names) x = resblock(n, x);
            return x;
        }
        const char* first = *names.begin();
        const char* last = *(names.end() - 1);
        const int out = fused_block(nullptr, (std::string(first) + ".conv1").c_str(), (std::string(first) + ".conv2").c_str(),
                                    (std::string(first) + ".conv3").c_str(), x, last, true);
        e->ops.back().nblk = (int)names.size();
        // a scratch tensor of the chain's shape (unnamed: not probe-able): at small batches the launcher runs the blocks as separate launches on
        // quarter-frame tiles (yf_mres_kernels.hip launch_mres: "small batches") and needs a third buffer beside the chain's input and output
        e->ops.back().out2 = add_tensor("", L0.cin, tx.H, tx.W);
        return out;
    }
    // pw-expand -> dw3x3 -> pw-project in one launch (optionally conv0 in front, optionally + residual)
    int fused_block(const char* pre, const char* ex, const char* dw, const char* pj, int in, const char* out_name, bool res)
    {
        Op o{};
        o.type = OP_FUSED_BLOCK;
        o.l_pre = pre ? find_layer(pre) : -1;
        o.l_exp = find_layer(ex); o.l_dw = find_layer(dw); o.l_proj = find_layer(pj);
        o.layer = pre ? o.l_pre : o.l_exp;
        const Tensor& ti = e->tensors[in];
        int H = pre ? ti.H / 2 : ti.H, W = pre ? ti.W / 2 : ti.W;  // expansion resolution
        int st = kLayers[o.l_dw].stride;
        o.in1 = in; o.in2 = -1; o.res = res ? in : -1; o.omode = 0;
        if (!pre && yf::mres_has_kernel(kLayers[o.l_exp].cin, kLayers[o.l_exp].cout, kLayers[o.l_proj].cout, res, st, kLayers[o.l_proj].relu != 0, dtype))
            o.type = OP_MRES;  // both pointwise convs on the matrix cores
        o.out = add_tensor(out_name, kLayers[o.l_proj].cout, H / st, W / st);
        e->ops.push_back(o);
        return o.out;
    }
    // dw5x5(+ReLU) -> 1x1 conv [-> 1x1 head conv] in one launch (yf_mdw_kernels.hip)
    int dwpw(const char* dw, const char* pw, const char* head, int in, const char* out_name, int ext)
    {
        const LayerSpec &LD = kLayers[find_layer(dw)], &LP = kLayers[find_layer(pw)];
        if (!fused || !yf::mdw_has_kernel(LD.cin, LP.cout, head ? num_out : 0)) {
            int x = unit(pw, unit(dw, in));
            return head ? unit(head, x, out_name, -1, -1, ext) : x;
        }
        Op o{};
        o.type = OP_MDW;
        o.l_dw = find_layer(dw); o.l_proj = find_layer(pw); o.l_head = head ? find_layer(head) : -1;
        o.layer = o.l_dw; o.in1 = in; o.in2 = -1; o.res = -1; o.omode = head ? 1 : 0;
        const Tensor& ti = e->tensors[in];
        o.out = add_tensor(head ? out_name : pw, head ? num_out : LP.cout, ti.H, ti.W, head ? ext : 0);
        e->ops.push_back(o);
        return o.out;
    }
    // the small head as ONE launch: dw5x5 -> 1x1 -> dw5x5 -> 1x1 -> head conv (yf_mdw_kernels.hip, mdw2_kernel); frames that fit one tile
    int dwpw2(const char* dw1, const char* pw1, const char* dw2, const char* pw2, const char* head, int in, const char* out_name, int ext)
    {
        const LayerSpec &LD = kLayers[find_layer(dw1)], &LP = kLayers[find_layer(pw1)], &LQ = kLayers[find_layer(pw2)];
        const Tensor& ti = e->tensors[in];
        if (!yf::mdw2_can_chain(LD.cin, LP.cout, LQ.cout, num_out, ti.H, ti.W)) return -1;
        Op o{};
        o.type = OP_MDW2;
        o.l_dw = find_layer(dw1); o.l_proj = find_layer(pw1); o.l_dw2 = find_layer(dw2); o.l_proj2 = find_layer(pw2); o.l_head = find_layer(head);
        o.layer = o.l_dw; o.in1 = in; o.in2 = -1; o.res = -1; o.omode = 1;
        o.out = add_tensor(out_name, num_out, ti.H, ti.W, ext);
        e->ops.push_back(o);
        return o.out;
    }
    // conv4_2 (a skip tensor: the large head concatenates it) -> conv4_3 -> conv5_1 as one launch that also writes conv4_2
    int triple_keep_expansion(const char* a, const char* b, const char* c, int x, int* expanded)
    {
        const LayerSpec &LA = kLayers[find_layer(a)], &LC = kLayers[find_layer(c)];
        const int st = kLayers[find_layer(b)].stride;
        if (!fused || !yf::mres_has_kernel(LA.cin, LA.cout, LC.cout, false, st, LC.relu != 0, dtype)) {
            *expanded = unit(a, x);
            return unit(c, unit(b, *expanded));
        }
        const int H = e->tensors[x].H, W = e->tensors[x].W;
        *expanded = add_tensor(a, LA.cout, H, W);
        const int out = fused_block(nullptr, a, b, c, x, c, false);
        e->ops.back().out2 = *expanded;
        return out;
    }
    int triple(const char* a, const char* b, const char* c, int x)
    {
        if (fused) return fused_block(nullptr, a, b, c, x, c, false);
        return unit(c, unit(b, unit(a, x)));
    }
    bool fused = false;
    int dtype = yf::DT_F32;   // the engine's: some blocks are planned on a different kernel per dtype (mres_has_kernel)
};

void build_plan(Plan* e, int level, int dtype, const LayerSpec* kLayers)
{
    const bool fused = level >= 1, deep = level >= 2;
    Builder b{e};
    b.kLayers = kLayers;
    b.num_out = kLayers[find_layer("head_4")].cout;
    b.fused = fused;
    b.dtype = dtype;
    int x = b.add_tensor("input", kLayers[0].cin, e->H, e->W, BUF_INPUT);   // NCHW [N, input_channel, H, W]: planes, not NHWC
    if (fused && kLayers[0].cin <= yf_layers::MAX_STEM_INPUT_CHANNEL) {
        x = b.fused_block("conv0", "conv1_2", "conv1_3", "conv1_4", x, "conv1_4", false);
    } else if (fused) {     // more input channels than the stem kernel is instantiated for: conv0 as a launch of its own, then the block
        x = b.unit("conv0", x);
        x = b.fused_block(nullptr, "conv1_2", "conv1_3", "conv1_4", x, "conv1_4", false);
    } else {
        for (const char* n : {"conv0", "conv1_2", "conv1_3", "conv1_4"}) x = b.unit(n, x);
    }
    x = b.resblock("res1_1", x);
    if (fused) {
        Op o{};
        o.type = OP_K19;
        o.layer = find_layer("conv1_8"); o.l_exp = o.layer; o.l_dw = find_layer("conv1_9"); o.l_proj = find_layer("conv2_1");
        o.in1 = x; o.in2 = -1; o.res = -1; o.omode = 0;
        o.out = b.add_tensor("conv2_1", 8, e->tensors[x].H / 2, e->tensors[x].W / 2);
        e->ops.push_back(o);
        x = o.out;
    } else {
        for (const char* n : {"conv1_8", "conv1_9", "conv2_1"}) x = b.unit(n, x);
    }
    for (const char* n : {"res2_1", "res2_2"}) x = b.resblock(n, x);
    x = b.triple("conv2_2", "conv2_3", "conv3_1", x);
    for (const char* n : {"res3_1", "res3_2"}) x = b.resblock(n, x);
    x = b.triple("conv3_2", "conv3_3", "conv3_4", x);
    for (const char* n : {"res3_3", "res3_4", "res3_5", "res3_6"}) x = b.resblock(n, x);
    x = b.triple("conv3_5", "conv3_6", "conv4_1", x);
    x = b.reschain({"res4_1", "res4_2", "res4_3", "res4_4"}, x);
    const bool fused_deep = fused;
    // developer switch (tests, A/B): YF_DEEP_MASK bit 0 = conv5_2 in the res5 launch, bit 1 = the chained small head, bit 2 = deconv5_1 + conv4_1_1
    const char* dm_env = getenv("YF_DEEP_MASK");
    const int dmask = dm_env ? atoi(dm_env) : 7;
    int conv4_2 = -1;
    x = b.triple_keep_expansion("conv4_2", "conv4_3", "conv5_1", x, &conv4_2);
    x = b.reschain({"res5_1", "res5_2", "res5_3", "res5_4", "res5_5"}, x);
    int conv5_2 = -1;
    {   // level 2: conv5_2 (48 -> 96, ReLU) runs on the last res5 launch's result while it is in LDS; res5_5 itself is not stored
        Op& last = e->ops.back();
        const LayerSpec& L52 = kLayers[find_layer("conv5_2")];
        if (deep && (dmask & 1) && last.type == OP_MRES && last.out == x &&
            yf::mres_has_post(kLayers[last.l_exp].cin, kLayers[last.l_exp].cout, kLayers[last.l_proj].cout, L52.cout)) {
            last.l_post = find_layer("conv5_2");
            e->tensors[x].name.clear();   // never materialised: not probe-able
            conv5_2 = b.add_tensor("conv5_2", L52.cout, e->tensors[x].H, e->tensors[x].W);
            last.out = conv5_2;
        }
    }
    b.fused = false;
    if (conv5_2 < 0) conv5_2 = b.unit("conv5_2", x);
    b.fused = fused_deep;
    const size_t br0 = e->ops.size();
    if (!deep || !(dmask & 2) || b.dwpw2("conv5_3", "conv5_4", "conv5_5", "conv5_6", "head_5", conv5_2, "head_small", BUF_HEAD_SMALL) < 0) {
        x = b.dwpw("conv5_3", "conv5_4", nullptr, conv5_2, nullptr, 0);
        b.dwpw("conv5_5", "conv5_6", "head_5", x, "head_small", BUF_HEAD_SMALL);
    }
    if (fused)
        for (size_t i = br0; i < e->ops.size(); ++i) e->ops[i].branch = 1;
    b.fused = false;
    if (deep && (dmask & 4) && yf::dcat_has_kernel(e->tensors[conv5_2].C, e->tensors[conv4_2].C, kLayers[find_layer("conv4_1_1")].cout)) {
        // level 2: deconv5_1 + conv4_1_1 in one launch, the deconv result stays in registers (yf_dcat_kernels.hip)
        Op o{};
        o.type = OP_DCAT;
        o.layer = find_layer("deconv5_1"); o.l_proj = find_layer("conv4_1_1");
        o.in1 = conv5_2; o.in2 = conv4_2; o.res = -1; o.omode = 0;
        o.out = b.add_tensor("conv4_1_1", kLayers[o.l_proj].cout, e->tensors[conv4_2].H, e->tensors[conv4_2].W);
        e->ops.push_back(o);
        x = o.out;
    } else {
        int d = b.unit("deconv5_1", conv5_2);
        x = b.unit("conv4_1_1", conv4_2, nullptr, d);  // torch.cat((conv4_2, deconv5_1), 1), yolo_fastest.py:209
    }
    b.fused = fused_deep;
    x = b.dwpw("conv4_1_2", "conv4_1_3", nullptr, x, nullptr, 0);
    b.dwpw("conv4_1_4", "conv4_1_5", "head_4", x, "head_large", BUF_HEAD_LARGE);
    b.fused = false;

    // liveness
    for (size_t i = 0; i < e->ops.size(); ++i) {
        const Op& o = e->ops[i];
        for (int t : {o.in1, o.in2, o.res})
            if (t >= 0) e->tensors[t].last_use = (int)i;
    }
    // tensors the side-stream branch reads or writes stay live to the end of the pass: the main stream's later launches run
    // CONCURRENTLY with the branch and must not be handed their slots
    for (size_t i = 0; i < e->ops.size(); ++i) {
        const Op& o = e->ops[i];
        if (!o.branch) continue;
        for (int t : {o.in1, o.in2, o.res, o.out})
            if (t >= 0) e->tensors[t].last_use = (int)e->ops.size() - 1;
    }
    // slot assignment: smallest free slot that fits, else grow the largest free one, else a new slot
    std::vector<int> free_slots;
    for (size_t i = 0; i < e->ops.size(); ++i) {
      for (int ot : {e->ops[i].out, e->ops[i].out2}) {
        if (ot < 0) continue;
        Tensor& t = e->tensors[ot];
        if (t.slot >= 0) {
            size_t need = t.elems();
            int best = -1, largest = -1;
            for (size_t f = 0; f < free_slots.size(); ++f) {
                int s = free_slots[f];
                if (e->slot_elems[s] >= need && (best < 0 || e->slot_elems[s] < e->slot_elems[free_slots[best]])) best = (int)f;
                if (largest < 0 || e->slot_elems[s] > e->slot_elems[free_slots[largest]]) largest = (int)f;
            }
            int pick = best >= 0 ? best : largest;
            if (pick >= 0) {
                t.slot = free_slots[pick];
                free_slots.erase(free_slots.begin() + pick);
                if (e->slot_elems[t.slot] < need) e->slot_elems[t.slot] = need;
            } else {
                t.slot = (int)e->slot_elems.size();
                e->slot_elems.push_back(need);
            }
        }
      }
        const Op& o = e->ops[i];
        for (int in : {o.in1, o.in2, o.res}) {
            if (in < 0) continue;
            Tensor& ti = e->tensors[in];
            if (ti.slot >= 0 && ti.last_use == (int)i) {
                bool dup = false;  // the same tensor may appear as in1 and res
                for (int s : free_slots) dup |= (s == ti.slot);
                if (!dup) free_slots.push_back(ti.slot);
            }
        }
    }
    e->slot_offset.resize(e->slot_elems.size());
    // guard band before the first and after the last slot: k19m_kernel reads its input without bounds checks, up to one
    // region row beyond either end of the tensor (the values are discarded); per-frame units, so >= the guard for any batch
    const size_t guard = fused ? yf::k19m_guard_elems(e->W / 2) : 0;
    size_t off = guard;
    for (size_t s = 0; s < e->slot_elems.size(); ++s) {
        e->slot_offset[s] = off;
        off += (e->slot_elems[s] + 63) & ~(size_t)63;
    }
    e->frame_floats = off + guard;
}

int chunk_frames(const yf_engine* e, int N)
{
    int c = e->chunk > 0 ? e->chunk : N;
    // auto: one chunk per lane, once a pass is big enough for two half-passes to overlap usefully (measured: 64 frames of 640x512,
    // 256 of 320x256; below that the split only adds launches: batch 64 at 320x256 runs 98 k frames/s in one lane, 94 k in two)
    if (e->chunk == 0 && e->lanes > 1 && (long)N * e->H * e->W >= 64L * 512 * 640) c = (N + e->lanes - 1) / e->lanes;
    return c < N ? c : N;
}

// ---- do two streams run concurrently?  HIP multiplexes streams onto a few hardware queues (GPU_MAX_HW_QUEUES, default 4) in
// first-use order; two streams on one queue execute strictly in issue order, and nothing in the API tells.  Measured on MI355X with two
// half-batch lanes, one batch at a time: 272-284 k frames/s when the caller's stream and the lane's stream sit on different queues, 192
// k (even 142 k) when they share one -- decided by which torch pool stream the caller happened to be on (tools/lane_try.sh).  The probe:
// a one-wave kernel that spins 100 us on each stream, the second ordered behind the first's START; together they take ~0.115 ms when the
// streams overlap and ~0.215 ms when they are serialised.
__global__ void yf_spin_kernel(long ticks)
{
    const long t0 = wall_clock64();
    for (int i = 0; i < (1 << 20) && wall_clock64() - t0 < ticks; ++i) __builtin_amdgcn_s_sleep(16);   // bounded: always exits
}

int streams_overlap(yf_engine* e, hipStream_t a, hipStream_t b, bool* overlap)
{
    int khz = 100000;   // s_memrealtime: 100 MHz on this part
    (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, e->device);
    const long ticks = (long)khz * 100 / 1000;   // 100 us: well above the ~15 us of launch and event overhead around the two kernels
    // The verdict is RELATIVE to one spin alone on stream a, measured in the same call: on an idle device ~0.110 ms alone, ~0.115 ms for an
    // overlapping pair, ~0.215 ms for a serialised one; on a device that is busy with other work (another engine's batch in flight,
    // another process) all three stretch together, which a fixed threshold would read as "serialised".  One retry when the ratio is
    // ambiguous (1.35 .. 1.65).
    for (int attempt = 0; attempt < 2; ++attempt) {
        HIP_OK(hipEventRecord(e->ev_probe[0], a));
        hipLaunchKernelGGL(yf_spin_kernel, dim3(1), dim3(64), 0, a, ticks);
        HIP_OK(hipEventRecord(e->ev_probe[2], a));
        HIP_OK(hipEventSynchronize(e->ev_probe[2]));
        float alone = 0.f;
        HIP_OK(hipEventElapsedTime(&alone, e->ev_probe[0], e->ev_probe[2]));
        HIP_OK(hipEventRecord(e->ev_probe[0], a));
        HIP_OK(hipStreamWaitEvent(b, e->ev_probe[0], 0));
        hipLaunchKernelGGL(yf_spin_kernel, dim3(1), dim3(64), 0, a, ticks);
        hipLaunchKernelGGL(yf_spin_kernel, dim3(1), dim3(64), 0, b, ticks);
        HIP_OK(hipEventRecord(e->ev_probe[1], b));
        HIP_OK(hipStreamWaitEvent(a, e->ev_probe[1], 0));
        HIP_OK(hipEventRecord(e->ev_probe[2], a));
        HIP_OK(hipEventSynchronize(e->ev_probe[2]));
        float pair = 0.f;
        HIP_OK(hipEventElapsedTime(&pair, e->ev_probe[0], e->ev_probe[2]));
        const float ratio = pair / (alone > 1e-3f ? alone : 1e-3f);
        *overlap = ratio < 1.5f;
        if (ratio < 1.35f || ratio > 1.65f) break;
    }
    return YF_OK;
}

// Choose, for caller stream s, which of the engine's streams are lane 1 and the branches of lanes 0 and 1 (the three that matter:
// with the caller's stream that is one stream per hardware queue of the default four) so that they overlap with s and with each
// other; the rest keep their order.  Host-blocking (a few probes of ~0.2 ms), once per caller stream; never while capturing.
int assign_streams(yf_engine* e, hipStream_t s)
{
    for (int i = 0; i < e->n_seen; ++i)
        if (e->seen[i].caller == s) {   // probed before: a caller that alternates between streams does not pay again
            e->seen[i].used = ++e->seen_tick;
            for (int l = 0; l < 3; ++l) e->side[l] = e->seen[i].side[l];
            for (int l = 0; l < 4; ++l) e->bside[l] = e->seen[i].bside[l];
            e->assigned_for = s;
            return YF_OK;
        }
    hipStream_t pool[7] = {e->side[0], e->bside[0], e->bside[1], e->side[1], e->side[2], e->bside[2], e->bside[3]};
    bool used[7] = {false, false, false, false, false, false, false};
    hipStream_t chosen[4] = {s, nullptr, nullptr, nullptr};
    int nchosen = 1;
    hipStream_t out[7];
    for (int need = 0; need < 7; ++need) {
        int pick = -1;
        if (need < 3) {
            for (int c = 0; c < 7 && pick < 0; ++c) {
                if (used[c]) continue;
                bool ok = true;
                for (int k = 0; k < nchosen && ok; ++k) {
                    bool ov = false;
                    if (int rc = streams_overlap(e, chosen[k], pool[c], &ov)) return rc;
                    ok = ov;
                }
                if (ok) pick = c;
            }
        }
        if (pick < 0)
            for (int c = 0; c < 7 && pick < 0; ++c)
                if (!used[c]) pick = c;
        used[pick] = true;
        out[need] = pool[pick];
        if (need < 3) chosen[nchosen++] = pool[pick];
    }
    e->side[0] = out[0]; e->bside[0] = out[1]; e->bside[1] = out[2]; e->side[1] = out[3]; e->side[2] = out[4]; e->bside[2] = out[5]; e->bside[3] = out[6];
    e->assigned_for = s;
    e->assigned = true;
    int slot = e->n_seen;
    if (e->n_seen < 4) ++e->n_seen;
    else {   // replace the entry that was used longest ago (a caller rotating over more than four streams re-probes the coldest only)
        slot = 0;
        for (int i = 1; i < 4; ++i)
            if (e->seen[i].used < e->seen[slot].used) slot = i;
    }
    yf_engine::Assignment& a = e->seen[slot];
    a.caller = s;
    a.used = ++e->seen_tick;
    for (int l = 0; l < 3; ++l) a.side[l] = e->side[l];
    for (int l = 0; l < 4; ++l) a.bside[l] = e->bside[l];
    return YF_OK;
}

// per-op profiling state (yf_profile_forward): events recorded around every launch of a single-lane pass
struct ProfileEvents {
    std::vector<hipEvent_t> ev;  // ops.size() + 1
    int repeats = 1;             // every launch is issued this many times back to back between its two events (yf_set_profile_repeats)
};

int run_forward(yf_engine* e, const float* d_x, int N, float* d_hl, float* d_hs, void* ws, size_t ws_bytes,
                hipStream_t s, const char* probe, float* probe_dst, size_t probe_bytes, ProfileEvents* prof = nullptr,
                const uint8_t* d_u8 = nullptr, int u8_down2 = 0, const yf::PostArgs* post = nullptr)
{
    if (!e || (!d_x && !d_u8) || !d_hl || !d_hs || N <= 0) return fail(YF_E_INVALID, "yf_forward: null pointer or N <= 0");
    if (d_u8 && e->fusion < 1) return fail(YF_E_INVALID, "u8 input needs a fused plan (yf_set_fusion 1 or 2)");
    if (e->dtype == yf::DT_F16 && e->fusion < 1) return fail(YF_E_INVALID, "fp16 storage needs a fused plan (yf_set_fusion 1 or 2)");
    const size_t esz = e->esz();
    if (N > e->max_batch) return fail(YF_E_INVALID, "yf_forward: N=%d exceeds max_batch=%d", N, e->max_batch);
    const Plan& P = e->plan();
    const LayerSpec* const kLayers = e->layers;
    const int cf = prof ? N : chunk_frames(e, N);  // profiling: the whole batch in one pass on the caller's stream
    const int nchunks = (N + cf - 1) / cf;
    const int lanes = (probe || prof || nchunks < 2) ? 1 : (e->lanes < nchunks ? e->lanes : nchunks);
    size_t need = P.frame_floats * (size_t)cf * esz * lanes;
    if (!ws || ws_bytes < need) return fail(YF_E_WORKSPACE, "workspace %zu B < required %zu B", ws_bytes, need);
    HIP_OK(hipSetDevice(e->device));
    const hipStream_t s_main = s;
    int probe_t = -1;
    if (probe) {
        for (size_t t = 0; t < P.tensors.size(); ++t)
            if (P.tensors[t].name == probe) probe_t = (int)t;
        if (probe_t < 0) return fail(YF_E_NOPROBE, "no tensor named '%s' exists in device memory (fusion level %d)", probe, e->fusion);
        if (probe_bytes < P.tensors[probe_t].elems() * N * sizeof(float)) return fail(YF_E_INVALID, "probe buffer too small");
    }
    char* base = static_cast<char*>(ws);
    auto W = [&](int layer) { return e->d_weights + e->w_off[layer]; };
    auto B = [&](int layer) { return e->d_weights + e->b_off[layer]; };
    // Chunks are processed in groups of `lanes` (one chunk per lane = stream + workspace region); within a group the launches
    // are issued OP-MAJOR (op k of every lane, then op k + 1): issued lane-major, the second lane's first kernel was queued
    // only after the first lane's 33 launches had gone through the host (~0.15 ms of a 1.3 ms step), and the lanes overlapped
    // for little more than half of the step (rocprofv3 kernel trace, profiles/).
    // The small head's launches (Op::branch) go to a side stream of the lane, forked after conv5_2: two under-filled launch chains (80
    // and 320 pixels per frame) run beside each other instead of in sequence.  Not while profiling / probing (one stream, one launch
    // at a time).  A branch never joins back INTO a forked lane: it joins the caller's stream, and with a post-process (yf_detect) the
    // lane joins its branch and the chunk's decode + NMS runs there.  The same dependencies, but a topology that the HIP runtime
    // PyTorch bundles (ROCm 7.0's libamdhip64 in torch/lib) can capture: its hipStreamEndCapture recurses without bound when a stream
    // forked from a forked stream joins that stream again (tools/cap_repro.hip reproduces it with empty kernels on that runtime;
    // /opt/rocm's 7.2 runtime is fine).  So eager and captured passes issue the SAME thing; there is no capture special case.
    const bool use_branch = e->branches && !prof && !probe && e->fusion >= 1;
    if ((lanes > 1 || use_branch) && (!e->assigned || e->assigned_for != s_main)) {
        // first pass on this caller stream: which of the engine's streams overlap with it (see streams_overlap).  The probe waits on
        // the host, which a stream capture does not allow: a captured pass keeps the current assignment.
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s_main, &st) == hipSuccess && st == hipStreamCaptureStatusNone)
            if (int rc = assign_streams(e, s_main)) return rc;
    }
    for (int g0 = 0; g0 < nchunks; g0 += lanes) {
      const int gcount = (nchunks - g0) < lanes ? (nchunks - g0) : lanes;
      // fork: the side lanes wait for everything already queued on the caller's stream -- which, from the second group of chunks on,
      // includes the previous group's lanes AND branches (they all joined it below), whose workspace regions this group reuses
      if (gcount > 1) {
          HIP_OK(hipEventRecord(e->ev_fork, s_main));
          for (int l = 1; l < gcount; ++l) HIP_OK(hipStreamWaitEvent(e->side[l - 1], e->ev_fork, 0));
      }
      size_t op_idx = 0;
      bool forked[4] = {false, false, false, false};
      if (prof) HIP_OK(hipEventRecord(prof->ev[0], s_main));
      for (size_t oi = 0; oi < P.ops.size(); ++oi) {
       const Op& o = P.ops[oi];
       for (int lane_id = 0; lane_id < gcount; ++lane_id) {
        const int f0 = (g0 + lane_id) * cf;
        const int n = (N - f0) < cf ? (N - f0) : cf;
        s = lane_id == 0 ? s_main : e->side[lane_id - 1];
        if (use_branch && o.branch) {
            if (!forked[lane_id]) {   // everything queued on the lane so far (conv5_2 included) precedes the branch
                HIP_OK(hipEventRecord(e->ev_bfork[lane_id], s));
                HIP_OK(hipStreamWaitEvent(e->bside[lane_id], e->ev_bfork[lane_id], 0));
                forked[lane_id] = true;
            }
            s = e->bside[lane_id];
        }
        char* lane_base = base + P.frame_floats * (size_t)cf * esz * lane_id;
        auto ptr = [&](int t) -> float* {
            const Tensor& T = P.tensors[t];
            if (T.slot == BUF_INPUT) return d_x ? const_cast<float*>(d_x) + (size_t)f0 * T.elems() : nullptr;
            if (T.slot == BUF_HEAD_LARGE) return d_hl + (size_t)f0 * T.elems();
            if (T.slot == BUF_HEAD_SMALL) return d_hs + (size_t)f0 * T.elems();
            return reinterpret_cast<float*>(lane_base + P.slot_offset[T.slot] * (size_t)cf * esz);  // opaque: the kernels know the type
        };
        {
            const LayerSpec& L = kLayers[o.layer];
            const Tensor& ti = P.tensors[o.in1];
            const Tensor& to = P.tensors[o.out];
            int rc = 0;
            struct AtExit { ProfileEvents* p; size_t* i; hipStream_t st; ~AtExit() { if (p) { ++*i; (void)hipEventRecord(p->ev[*i], st); } } } at_exit{prof, &op_idx, s};
            // (profiling: `repeats` back-to-back launches of the op between its two events -- every op writes its whole output from inputs it
            //  does not modify, so repeating it is idempotent -- take the event and dispatch gap out of the per-launch figure)
            for (int rep_ = 0, nrep_ = prof ? prof->repeats : 1; rep_ < nrep_ && !rc; ++rep_) {
            if (o.type == OP_DCAT) {
                rc = yf::launch_dcat(ptr(o.in1), ptr(o.in2), e->d_wmfma + o.mfma_off, e->d_wmfma + o.mfma_off2, ptr(o.out), ti.H, ti.W, n, s, o.kdt);
            } else if (o.type == OP_MDW2) {
                rc = yf::launch_mdw2(ptr(o.in1), e->d_wmfma + o.mfma_off, e->d_wmfma + o.mfma_off2, ptr(o.out), ti.H, ti.W, e->num_out, n, s, o.kdt,
                                     (e->d_esplit && e->split_sums) ? e->d_esplit + (size_t)lane_id * e->esplit_lane_floats() + yf::mres_esplit_scratch_floats() : nullptr);
            } else if (o.type == OP_MDW) {
                yf::MdwArgs a{ptr(o.in1), e->d_wmfma + o.mfma_off, ptr(o.out), ti.H, ti.W, 0, 0};
                rc = yf::launch_mdw(ti.C, kLayers[o.l_proj].cout, o.l_head >= 0 ? e->num_out : 0, a, n, s, o.kdt);
            } else if (o.type == OP_MRES) {
                const LayerSpec &LE = kLayers[o.l_exp], &LP = kLayers[o.l_proj];
                yf::MresArgs a{ptr(o.in1), e->d_wmfma + o.mfma_off, ptr(o.out), ti.H, ti.W, 0, 0, o.out2 >= 0 ? ptr(o.out2) : nullptr,
                               o.nblk, o.wstride, nullptr, nullptr};
                if (o.l_post >= 0) { a.post_w = e->d_wmfma + o.post_off; a.post_out = ptr(o.out); a.out = nullptr; }
                a.esplit = (e->d_esplit && e->split_sums) ? e->d_esplit + (size_t)lane_id * e->esplit_lane_floats() : nullptr;   // one region per lane: lanes run concurrently
                rc = yf::launch_mres(LE.cin, LE.cout, LP.cout, o.res >= 0, kLayers[o.l_dw].stride, a, n, s, o.kdt);
            } else if (o.type == OP_FUSED_BLOCK) {
                const bool pre = o.l_pre >= 0;
                const LayerSpec &LE = kLayers[o.l_exp], &LD = kLayers[o.l_dw], &LP = kLayers[o.l_proj];
                yf::FbArgs a{};
                a.in = ptr(o.in1);
                if (pre) { a.w0 = W(o.l_pre); a.b0 = B(o.l_pre); }
                if (pre && d_u8) {  // pre-process fused into the stem's loads
                    a.in = nullptr;
                    a.in_u8 = d_u8 + (size_t)f0 * (u8_down2 ? 4 : 1) * P.tensors[o.in1].elems();
                    a.u8_down2 = u8_down2;
                }
                a.wp = e->d_wmfma + o.mfma_off;
                a.out = ptr(o.out);
                a.H = pre ? ti.H / 2 : ti.H; a.W = pre ? ti.W / 2 : ti.W; a.Ho = to.H; a.Wo = to.W;
                rc = yf::launch_fused_block(LE.cin, LE.cout, LP.cout, LD.stride, o.res >= 0, LP.relu != 0, pre ? e->input_channel : 0, a, n, s, o.kdt);
            } else if (o.type == OP_K19) {
                yf::K19Args a{};
                a.in = ptr(o.in1);
                a.w8 = W(o.l_exp); a.b8 = B(o.l_exp); a.w9 = W(o.l_dw); a.b9 = B(o.l_dw); a.w21 = W(o.l_proj); a.b21 = B(o.l_proj);
                a.out = ptr(o.out);
                a.H = ti.H; a.W = ti.W; a.Ho = to.H; a.Wo = to.W;
                a.wp = e->d_wmfma + o.mfma_off;
                rc = yf::launch_k19m(a, n, s, o.kdt);
            } else if (L.kind == K_HEAD && L.cout != 24) {
                // per-layer plan, a head other than the shipped 3 x (5 + 3): the generic head conv (any Cout), NHWC -> NCHW logits
                rc = yf::launch_head_conv(ptr(o.in1), W(o.layer), B(o.layer), ptr(o.out), ti.C, L.cout, (long)ti.H * ti.W, n, s, e->sdt());
            } else if (L.kind == K_PW || L.kind == K_HEAD || L.kind == K_DECONV) {
                yf::PwArgs a{ptr(o.in1), o.in2 >= 0 ? ptr(o.in2) : nullptr, W(o.layer), B(o.layer), o.res >= 0 ? ptr(o.res) : nullptr,
                             ptr(o.out), (long)n * ti.H * ti.W, (long)ti.H * ti.W, ti.W};
                int cin2 = o.in2 >= 0 ? P.tensors[o.in2].C : 0;
                if (o.mfma_off >= 0) {
                    a.w = e->d_wmfma + o.mfma_off;
                    rc = yf::launch_pw_mfma(ti.C, cin2, L.cout, L.relu != 0, o.res >= 0, o.omode, a, s, o.kdt);
                } else {
                    rc = yf::launch_pw(ti.C, cin2, L.cout, L.relu != 0, o.res >= 0, o.omode, a, s);
                }
            } else if (L.kind == K_DW) {
                yf::DwArgs a{ptr(o.in1), W(o.layer), B(o.layer), ptr(o.out), (long)n * to.H * to.W * (to.C / 4), ti.C, ti.H, ti.W, to.H, to.W};
                rc = yf::launch_dw(L.k, L.stride, a, s, e->sdt());
            } else {
                yf::DenseArgs a{ptr(o.in1), W(o.layer), B(o.layer), ptr(o.out), (long)n * to.H * to.W, ti.H, ti.W, to.H, to.W};
                // (conv0 with Cin > 1 reads the NCHW planes of the net input; with Cin > 4 it is a launch of its own in the fused plans
                //  too and writes the engine's storage type)
                rc = yf::launch_dense3x3s2(L.cin, L.cout, a, s, o.layer == 0 && e->fusion >= 1 ? e->sdt() : yf::DT_F32);
            }
            }
            if (rc) return fail(YF_E_INVALID, "no kernel for layer %s", L.name);
            if (o.out == probe_t) {
                if (to.slot == BUF_HEAD_LARGE || to.slot == BUF_HEAD_SMALL)
                    HIP_OK(hipMemcpyAsync(probe_dst + (size_t)f0 * to.elems(), ptr(o.out), to.elems() * n * sizeof(float),
                                          hipMemcpyDeviceToDevice, s));
                else
                    yf::launch_nhwc_to_nchw(ptr(o.out), probe_dst + (size_t)f0 * to.elems(), n, to.C, (long)to.H * to.W, s, e->sdt());
            }
            if (o.out2 >= 0 && o.out2 == probe_t) {
                const Tensor& t2 = P.tensors[o.out2];
                yf::launch_nhwc_to_nchw(ptr(o.out2), probe_dst + (size_t)f0 * t2.elems(), n, t2.C, (long)t2.H * t2.W, s, e->sdt());
            }
        }
       }
      }
      // decode + NMS of each chunk (yf_detect) on ITS lane -- frames are independent, so a lane's post-process overlaps the other
      // lane's tail instead of running after the join -- or, where the lane forked a branch, on the BRANCH stream once the lane has
      // joined it (see above)
      if (post)
        for (int lane_id = 0; lane_id < gcount; ++lane_id) {
            const int f0 = (g0 + lane_id) * cf;
            const int n = (N - f0) < cf ? (N - f0) : cf;
            hipStream_t sl = lane_id == 0 ? s_main : e->side[lane_id - 1];
            if (forked[lane_id]) {
                HIP_OK(hipEventRecord(e->ev_l2b[lane_id], sl));
                HIP_OK(hipStreamWaitEvent(e->bside[lane_id], e->ev_l2b[lane_id], 0));
                sl = e->bside[lane_id];
            }
            yf::PostArgs a = *post;
            a.head_large += (size_t)f0 * e->head_l_elems; a.head_small += (size_t)f0 * e->head_s_elems;
            if (a.records) a.records += (size_t)f0 * (1 + 8 * (size_t)a.kmax);
            else {
                a.boxes += (size_t)f0 * a.kmax * 4; a.scores += (size_t)f0 * a.kmax * 2;
                a.cls += (size_t)f0 * a.kmax; a.src += (size_t)f0 * a.kmax; a.counts += f0;
            }
            if (a.split_tmp) a.split_tmp += yf::post_split_tmp_ints(f0, a.nc, a.kmax);
            const int rc = yf::launch_post(a, n, sl);
            if (rc == -1) return fail(YF_E_INVALID, "frame of %dx%d has too many cells for the on-chip NMS (limit 8191 cells / 160 KiB LDS)", e->H, e->W);
            if (rc) return fail(YF_E_HIP, "hipFuncSetAttribute(post_kernel) failed");
        }
      for (int lane_id = 0; lane_id < gcount; ++lane_id)   // every branch joins the caller's stream
        if (forked[lane_id]) {
            HIP_OK(hipEventRecord(e->ev_bjoin[lane_id], e->bside[lane_id]));
            HIP_OK(hipStreamWaitEvent(s_main, e->ev_bjoin[lane_id], 0));
        }
      for (int l = 1; l < gcount; ++l) {  // join: the caller's stream continues only after every side lane has drained
          HIP_OK(hipEventRecord(e->ev_join[l - 1], e->side[l - 1]));
          HIP_OK(hipStreamWaitEvent(s_main, e->ev_join[l - 1], 0));
      }
    }
    HIP_OK(hipGetLastError());
    return YF_OK;
}

double sigmoid_host(float x) { return 1. / (1. + exp(-(double)x)); }  // detect.py:23-25 in this libm

// smallest fp32 t with 1/(1+exp(-t)) > thres (monotone in t); +inf / -inf sentinels at the extremes
float logit_threshold(double thres)
{
    auto to_ord = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u); };
    auto from_ord = [](uint32_t o) { uint32_t u = (o & 0x80000000u) ? (o ^ 0x80000000u) : ~o; float f; memcpy(&f, &u, 4); return f; };
    const float lo_f = -700.f, hi_f = 700.f;  // math.exp overflows beyond; sigmoid saturates long before
    if (sigmoid_host(lo_f) > thres) return -INFINITY;
    if (!(sigmoid_host(hi_f) > thres)) return INFINITY;
    uint32_t lo = to_ord(lo_f), hi = to_ord(hi_f);  // invariant: f(lo) false, f(hi) true
    while (hi - lo > 1) {
        uint32_t mid = lo + (hi - lo) / 2;
        if (sigmoid_host(from_ord(mid)) > thres) hi = mid; else lo = mid;
    }
    return from_ord(hi);
}

}  // namespace

extern "C" {

int yf_abi_version(void) { return YF_ABI_VERSION; }
uint16_t yf_f32_to_f16_bits(float f) { return yf::f32_to_f16_bits(f); }
const char* yf_last_error_string(void) { return g_err; }

int yf_create(const void* blob, size_t nbytes, int H, int W, int max_batch, int device, yf_handle* out)
{
    return yf_create_ex(blob, nbytes, H, W, max_batch, device, 0, out);
}

int yf_create_ex(const void* blob, size_t nbytes, int H, int W, int max_batch, int device, int dtype, yf_handle* out)
{
    if (dtype < 0 || dtype > 2)
        return fail(YF_E_INVALID, "dtype must be 0 (fp32), 1 (fp16 storage / fp16 MFMA) or 2 (fp32 storage / split-operand fp16 MFMA)");
    if (!blob || !out) return fail(YF_E_INVALID, "yf_create: null pointer");
    if (H <= 0 || W <= 0 || H % 32 || W % 32) return fail(YF_E_INVALID, "input shape %dx%d: rows and cols must be multiples of 32", H, W);
    if (max_batch <= 0) return fail(YF_E_INVALID, "max_batch must be positive");
    if (nbytes < sizeof(BlobHeader)) return fail(YF_E_BLOB, "blob too small");
    BlobHeader hd;
    memcpy(&hd, blob, sizeof hd);
    if (memcmp(hd.magic, "YFHIPW01", 8) || hd.version != 1) return fail(YF_E_BLOB, "bad magic/version");
    if (hd.n_layers != (uint32_t)kNumLayers)
        return fail(YF_E_BLOB, "blob describes n_layers=%u; YoloFastest has %d", hd.n_layers, kNumLayers);
    static_assert((int)yf::POST_MAX_ANCHORS == (int)yf_layers::MAX_NUM_ANCHORS, "one anchor limit");
    if (hd.input_channel < 1 || hd.input_channel > (uint32_t)yf_layers::MAX_INPUT_CHANNEL)
        return fail(YF_E_BLOB, "input_channel=%u: the HIP engine implements 1 .. %d input channels", hd.input_channel, (int)yf_layers::MAX_INPUT_CHANNEL);
    if (hd.num_anchors < 1 || hd.num_anchors > (uint32_t)yf::POST_MAX_ANCHORS || hd.num_cls < 1 || hd.num_cls > (uint32_t)yf_layers::MAX_NUM_CLS ||
        hd.num_out != hd.num_anchors * (5 + hd.num_cls))
        return fail(YF_E_BLOB, "blob describes num_anchors=%u num_cls=%u num_out=%u: need 1..%d anchors, >= 1 class and num_out == "
                    "num_anchors * (5 + num_cls) (yolo_fastest.py:76)", hd.num_anchors, hd.num_cls, hd.num_out, yf::POST_MAX_ANCHORS);
    size_t data_off = sizeof(BlobHeader) + sizeof(BlobLayer) * (size_t)kNumLayers;
    if (nbytes < data_off + hd.data_floats * 4) return fail(YF_E_BLOB, "blob truncated");
    yf_engine* e = new yf_engine;
    e->device = device; e->H = H; e->W = W; e->max_batch = max_batch; e->dtype = dtype;
    e->input_channel = (int)hd.input_channel; e->num_anchors = (int)hd.num_anchors; e->num_cls = (int)hd.num_cls; e->num_out = (int)hd.num_out;
    make_layers(e->layers, e->input_channel, e->num_out);
    const LayerSpec* const kLayers = e->layers;
    const BlobLayer* tab = reinterpret_cast<const BlobLayer*>(static_cast<const char*>(blob) + sizeof(BlobHeader));
    for (int i = 0; i < kNumLayers; ++i) {
        BlobLayer bl;
        memcpy(&bl, &tab[i], sizeof bl);
        const LayerSpec& L = kLayers[i];
        if (strncmp(bl.name, L.name, 32) || (int)bl.kind != L.kind || (int)bl.cin != L.cin || (int)bl.cout != L.cout ||
            (int)bl.k != L.k || (int)bl.stride != L.stride || (int)bl.relu != L.relu) {
            delete e;
            return fail(YF_E_BLOB, "layer %d: blob has '%.31s', YoloFastest expects '%s' (kind/shape mismatch)", i, bl.name, L.name);
        }
        size_t wn = L.kind == K_DW ? (size_t)L.k * L.k * L.cout
                  : L.kind == K_DENSE ? (size_t)L.k * L.k * L.cin * L.cout
                  : L.kind == K_DECONV ? (size_t)4 * L.cin * L.cout : (size_t)L.cin * L.cout;
        if (bl.w_off + wn > hd.data_floats || bl.b_off + (size_t)L.cout > hd.data_floats || (bl.w_off & 3) || (bl.b_off & 3)) {
            delete e;
            return fail(YF_E_BLOB, "layer %s: weight offsets out of range or unaligned", L.name);
        }
        e->w_off[i] = bl.w_off; e->b_off[i] = bl.b_off;
    }
    if (hipSetDevice(device) != hipSuccess) { delete e; return fail(YF_E_HIP, "hipSetDevice(%d) failed", device); }
    e->n_floats = hd.data_floats;
    if (hipMalloc(&e->d_weights, e->n_floats * 4) != hipSuccess) { delete e; return fail(YF_E_HIP, "hipMalloc(weights) failed"); }
    if (hipMemcpy(e->d_weights, static_cast<const char*>(blob) + data_off, e->n_floats * 4, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(e->d_weights); delete e;
        return fail(YF_E_HIP, "hipMemcpy(weights) failed");
    }
    for (int f = 0; f < 3; ++f) {
        e->plans[f].H = H; e->plans[f].W = W;
        build_plan(&e->plans[f], f, dtype, e->layers);
    }
    e->head_l_elems = (size_t)e->num_out * (H / 16) * (W / 16);
    e->head_s_elems = (size_t)e->num_out * (H / 32) * (W / 32);
    {   // matrix-core layers of the fused plan: pre-pack W[K][N] into MFMA B fragments (host) and upload
        const float* hw = reinterpret_cast<const float*>(static_cast<const char*>(blob) + data_off);
        const bool h16 = e->dtype == yf::DT_F16;
        const bool x3 = e->dtype == yf::DT_F16X3;
        std::vector<float> packed;
        for (Op& o : e->plans[0].ops) o.kdt = e->sdt();
      for (int lvl = 1; lvl <= 2; ++lvl) {
        const Plan& P = e->plans[lvl];
        for (Op& o : e->plans[lvl].ops) {
            o.kdt = e->sdt();
            if (o.type == OP_MRES) {
                const LayerSpec &LE = kLayers[o.l_exp], &LP = kLayers[o.l_proj];
                if (x3) o.kdt = yf::DT_F16X3;
                const int wm = o.kdt;   // WM_* == DT_*
                o.mfma_off = (long)packed.size();
                o.wstride = (long)((yf::mres_packed_floats(LE.cin, LE.cout, LP.cout, wm) + 63) & ~(size_t)63);
                packed.resize(packed.size() + (size_t)o.wstride * o.nblk);
                for (int k = 0; k < o.nblk; ++k)   // chained blocks: same shape, layers 3 apart, streams wstride apart
                    yf::mres_pack_weights(hw + e->w_off[o.l_exp + 3 * k], hw + e->b_off[o.l_exp + 3 * k], hw + e->w_off[o.l_dw + 3 * k],
                                          hw + e->b_off[o.l_dw + 3 * k], hw + e->w_off[o.l_proj + 3 * k], hw + e->b_off[o.l_proj + 3 * k],
                                          LE.cin, LE.cout, LP.cout, packed.data() + o.mfma_off + (size_t)o.wstride * k, wm);
                if (o.l_post >= 0) {   // trailing 1x1 conv: the pw GEMM's fp32 fragments, then its bias (exact fp32 in every engine dtype)
                    const LayerSpec& LQ = kLayers[o.l_post];
                    o.post_off = (long)packed.size();
                    packed.resize(packed.size() + ((yf::mres_post_packed_floats(LQ.cin, LQ.cout) + 63) & ~(size_t)63));
                    yf::mfma_pack_weights(hw + e->w_off[o.l_post], LQ.cin, 0, LQ.cout, packed.data() + o.post_off);
                    memcpy(packed.data() + o.post_off + yf::mfma_packed_floats(LQ.cin, 0, LQ.cout), hw + e->b_off[o.l_post], LQ.cout * sizeof(float));
                }
                continue;
            }
            if (o.type == OP_K19) {
                if (x3) o.kdt = yf::DT_F16X3;
                o.mfma_off = (long)packed.size();
                packed.resize(packed.size() + ((yf::k19_packed_floats(o.kdt) + 63) & ~(size_t)63));
                yf::k19_pack_weights(hw + e->w_off[o.l_dw], hw + e->w_off[o.l_proj], packed.data() + o.mfma_off, o.kdt);
                continue;
            }
            if (o.type == OP_FUSED_BLOCK) {
                const LayerSpec &LE = kLayers[o.l_exp], &LD = kLayers[o.l_dw], &LP = kLayers[o.l_proj];
                const int ec = yf::fb_chunk_channels(LE.cin, LE.cout, LP.cout, LD.stride, o.res >= 0, LP.relu != 0, o.l_pre >= 0 ? e->input_channel : 0);
                if (ec <= 0) { (void)hipFree(e->d_weights); delete e; return fail(YF_E_INVALID, "no fused kernel for block %s", LE.name); }
                o.mfma_off = (long)packed.size();
                packed.resize(packed.size() + ((yf::fb_packed_floats(LE.cin, LE.cout, LP.cout, ec) + 63) & ~(size_t)63));
                yf::fb_pack_weights(hw + e->w_off[o.l_exp], hw + e->b_off[o.l_exp], hw + e->w_off[o.l_dw], hw + e->b_off[o.l_dw],
                                    hw + e->w_off[o.l_proj], hw + e->b_off[o.l_proj], LE.cin, LE.cout, LP.cout, ec,
                                    packed.data() + o.mfma_off);
                continue;
            }
            if (o.type == OP_MDW) {
                const LayerSpec &LD = kLayers[o.l_dw], &LP = kLayers[o.l_proj];
                const int headn = o.l_head >= 0 ? e->num_out : 0;
                if (x3) o.kdt = yf::DT_F16X3;
                o.mfma_off = (long)packed.size();
                packed.resize(packed.size() + ((yf::mdw_packed_floats(LD.cin, LP.cout, headn, o.kdt) + 63) & ~(size_t)63));
                yf::mdw_pack_weights(hw + e->w_off[o.l_dw], hw + e->b_off[o.l_dw], hw + e->w_off[o.l_proj], hw + e->b_off[o.l_proj],
                                     headn ? hw + e->w_off[o.l_head] : nullptr, headn ? hw + e->b_off[o.l_head] : nullptr, LD.cin,
                                     LP.cout, headn, packed.data() + o.mfma_off, o.kdt);
                continue;
            }
            if (o.type == OP_DCAT) {   // deconv: the pw GEMM's fragments per quadrant; conv4_1_1: the LDS stream + both biases
                const LayerSpec& LD = kLayers[o.layer];
                o.kdt = x3 ? yf::DT_F16X3 : h16 ? yf::DT_F16 : yf::DT_F32;
                const size_t per = x3 ? yf::mfma_packed_floats_x3(LD.cin, 0, LD.cout) : h16 ? yf::mfma_packed_floats_f16(LD.cin, 0, LD.cout)
                                      : yf::mfma_packed_floats(LD.cin, 0, LD.cout);
                o.mfma_off = (long)packed.size();
                packed.resize(packed.size() + ((4 * per + 63) & ~(size_t)63));
                for (int qd = 0; qd < 4; ++qd) {
                    const float* wq = hw + e->w_off[o.layer] + (size_t)qd * LD.cin * LD.cout;
                    if (x3) yf::mfma_pack_weights_x3(wq, LD.cin, 0, LD.cout, packed.data() + o.mfma_off + per * qd);
                    else if (h16) yf::mfma_pack_weights_f16(wq, LD.cin, 0, LD.cout, packed.data() + o.mfma_off + per * qd);
                    else yf::mfma_pack_weights(wq, LD.cin, 0, LD.cout, packed.data() + o.mfma_off + per * qd);
                }
                o.mfma_off2 = (long)packed.size();
                packed.resize(packed.size() + (((x3 ? yf::dcat_packed_floats_x3() : h16 ? yf::dcat_packed_floats_f16() : yf::dcat_packed_floats()) + 63) & ~(size_t)63));
                if (x3) yf::dcat_pack_weights_x3(hw + e->w_off[o.l_proj], hw + e->b_off[o.layer], hw + e->b_off[o.l_proj], packed.data() + o.mfma_off2);
                else if (h16) yf::dcat_pack_weights_f16(hw + e->w_off[o.l_proj], hw + e->b_off[o.layer], hw + e->b_off[o.l_proj], packed.data() + o.mfma_off2);
                else yf::dcat_pack_weights(hw + e->w_off[o.l_proj], hw + e->b_off[o.layer], hw + e->b_off[o.l_proj], packed.data() + o.mfma_off2);
                continue;
            }
            if (o.type == OP_MDW2) {   // two mdw weight streams, one after the other
                if (x3) o.kdt = yf::DT_F16X3;
                for (int st = 0; st < 2; ++st) {
                    const int ld = st ? o.l_dw2 : o.l_dw, lp = st ? o.l_proj2 : o.l_proj, headn = st ? e->num_out : 0;
                    const LayerSpec &LD = kLayers[ld], &LP = kLayers[lp];
                    const long off = (long)packed.size();
                    (st ? o.mfma_off2 : o.mfma_off) = off;
                    packed.resize(packed.size() + ((yf::mdw_packed_floats(LD.cin, LP.cout, headn, o.kdt) + 63) & ~(size_t)63));
                    yf::mdw_pack_weights(hw + e->w_off[ld], hw + e->b_off[ld], hw + e->w_off[lp], hw + e->b_off[lp],
                                         headn ? hw + e->w_off[o.l_head] : nullptr, headn ? hw + e->b_off[o.l_head] : nullptr, LD.cin, LP.cout,
                                         headn, packed.data() + off, o.kdt);
                }
                continue;
            }
            if (o.type != OP_LAYER) continue;
            const LayerSpec& L = kLayers[o.layer];
            if (L.kind != K_PW && L.kind != K_HEAD && L.kind != K_DECONV) continue;
            int c1 = P.tensors[o.in1].C, c2 = o.in2 >= 0 ? P.tensors[o.in2].C : 0;
            if (!yf::mfma_has_kernel(c1, c2, L.cout, L.relu != 0, o.res >= 0, o.omode)) continue;
            const int nq = L.kind == K_DECONV ? 4 : 1;
            const bool ox3 = x3 && yf::mfma_has_x3_kernel(c1, c2, L.cout, L.relu != 0, o.res >= 0, o.omode);
            if (ox3) o.kdt = yf::DT_F16X3;
            size_t per = ox3 ? yf::mfma_packed_floats_x3(c1, c2, L.cout)
                       : h16 ? yf::mfma_packed_floats_f16(c1, c2, L.cout) : yf::mfma_packed_floats(c1, c2, L.cout);
            o.mfma_off = (long)packed.size();
            packed.resize(packed.size() + per * nq);
            for (int qd = 0; qd < nq; ++qd) {
                if (ox3) yf::mfma_pack_weights_x3(hw + e->w_off[o.layer] + (size_t)qd * L.cin * L.cout, c1, c2, L.cout,
                                                  packed.data() + o.mfma_off + per * qd);
                else if (h16) yf::mfma_pack_weights_f16(hw + e->w_off[o.layer] + (size_t)qd * L.cin * L.cout, c1, c2, L.cout,
                                                   packed.data() + o.mfma_off + per * qd);
                else yf::mfma_pack_weights(hw + e->w_off[o.layer] + (size_t)qd * L.cin * L.cout, c1, c2, L.cout,
                                           packed.data() + o.mfma_off + per * qd);
            }
        }
      }
        if (!packed.empty()) {
            if (hipMalloc(&e->d_wmfma, packed.size() * 4) != hipSuccess ||
                hipMemcpy(e->d_wmfma, packed.data(), packed.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
                (void)hipFree(e->d_weights); (void)hipFree(e->d_wmfma); delete e;
                return fail(YF_E_HIP, "hipMalloc/hipMemcpy(MFMA weights) failed");
            }
        }
    }
    if (dtype == yf::DT_F32 && H / 32 == 8 && W / 32 == 10) {   // the stride-32 chain's few-frames form and the small head's keep their partial sums here (9 MB per lane)
        if (hipMalloc(&e->d_esplit, 4 * e->esplit_lane_floats() * sizeof(float)) != hipSuccess) {   // x 4 lanes (yf_set_lanes' maximum)
            (void)yf_destroy(e);
            return fail(YF_E_HIP, "hipMalloc(esplit scratch) failed");
        }
    }
    for (int l = 0; l < 3; ++l) {
        if (hipStreamCreateWithFlags(&e->side[l], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&e->ev_join[l], hipEventDisableTiming) != hipSuccess) {
            (void)yf_destroy(e);  // frees the weights, whatever streams / events exist so far, and the engine
            return fail(YF_E_HIP, "hipStreamCreate/hipEventCreate failed");
        }
    }
    if (hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming) != hipSuccess) {
        (void)yf_destroy(e);
        return fail(YF_E_HIP, "hipEventCreate failed");
    }
    for (int l = 0; l < 3; ++l)
        if (hipEventCreate(&e->ev_probe[l]) != hipSuccess) {
            (void)yf_destroy(e);
            return fail(YF_E_HIP, "hipEventCreate (probe) failed");
        }
    for (int l = 0; l < 4; ++l) {
        if (hipStreamCreateWithFlags(&e->bside[l], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&e->ev_bfork[l], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&e->ev_bjoin[l], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&e->ev_l2b[l], hipEventDisableTiming) != hipSuccess) {
            (void)yf_destroy(e);
            return fail(YF_E_HIP, "hipStreamCreate/hipEventCreate (branch) failed");
        }
    }
    *out = e;
    return YF_OK;
}

int yf_destroy(yf_handle h)
{
    if (!h) return YF_OK;
    (void)hipSetDevice(h->device);
    for (int l = 0; l < 3; ++l) {
        if (h->side[l]) (void)hipStreamDestroy(h->side[l]);
        if (h->ev_join[l]) (void)hipEventDestroy(h->ev_join[l]);
    }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    for (int l = 0; l < 3; ++l)
        if (h->ev_probe[l]) (void)hipEventDestroy(h->ev_probe[l]);
    for (int l = 0; l < 4; ++l) {
        if (h->bside[l]) (void)hipStreamDestroy(h->bside[l]);
        if (h->ev_bfork[l]) (void)hipEventDestroy(h->ev_bfork[l]);
        if (h->ev_bjoin[l]) (void)hipEventDestroy(h->ev_bjoin[l]);
        if (h->ev_l2b[l]) (void)hipEventDestroy(h->ev_l2b[l]);
    }
    for (auto& t : h->cvtab) if (t.ready) (void)hipEventDestroy(t.ready);
    if (h->d_cvpool) (void)hipFree(h->d_cvpool);
    if (h->d_post_tmp) (void)hipFree(h->d_post_tmp);
    if (h->d_esplit) (void)hipFree(h->d_esplit);
    (void)hipFree(h->d_weights);
    (void)hipFree(h->d_wmfma);
    delete h;
    return YF_OK;
}

int yf_workspace_bytes(yf_handle h, int N, size_t* out)
{
    if (!h || !out || N <= 0) return fail(YF_E_INVALID, "yf_workspace_bytes: bad argument");
    // layer-chain slots + internal head buffers for yf_detect
    size_t per_pass = (size_t)chunk_frames(h, N) * h->lanes;
    if (per_pass < (size_t)N) per_pass = N;  // yf_profile_forward / yf_forward_probe run the whole batch in one pass
    *out = h->frame_floats_max() * per_pass * h->esz() + (h->head_l_elems + h->head_s_elems) * (size_t)N * sizeof(float) + 1024;
    *out = ((*out + 255) & ~(size_t)255) + h->cv_scratch_bytes(N);   // + the net-sized u8 frames of yf_forward_u8 / yf_forward_bgr_u8 at other source sizes
    return YF_OK;
}

int yf_forward(yf_handle h, const float* d_x, int N, float* d_hl, float* d_hs, void* ws, size_t ws_bytes, void* stream)
{
    return run_forward(h, d_x, N, d_hl, d_hs, ws, ws_bytes, (hipStream_t)stream, nullptr, nullptr, 0);
}

int yf_forward_probe(yf_handle h, const float* d_x, int N, const char* name, float* d_dst, size_t dst_bytes, void* ws,
                     size_t ws_bytes, void* stream)
{
    if (!h || !name || !d_dst) return fail(YF_E_INVALID, "yf_forward_probe: null pointer");
    // heads go to the tail of the workspace
    size_t chain = ((h->frame_floats_max() * (size_t)chunk_frames(h, N) * h->lanes * h->esz()) + 255) & ~(size_t)255;
    size_t heads = (h->head_l_elems + h->head_s_elems) * (size_t)N * sizeof(float);
    if (!ws || ws_bytes < chain + heads) return fail(YF_E_WORKSPACE, "workspace %zu B < required %zu B", ws_bytes, chain + heads);
    float* hl = reinterpret_cast<float*>(static_cast<char*>(ws) + chain);
    float* hs = hl + h->head_l_elems * (size_t)N;
    return run_forward(h, d_x, N, hl, hs, ws, chain, (hipStream_t)stream, name, d_dst, dst_bytes);
}

static yf::PostArgs make_post_args(yf_handle h, const float* d_hl, const float* d_hs, double conf_thres, double nms_thres,
                                   const double* anchors, int origin_h, int origin_w, int K_max, int32_t* d_boxes, float* d_scores,
                                   int32_t* d_cls, int32_t* d_src, int32_t* d_counts)
{
    yf::PostArgs a;
    a.head_large = d_hl; a.head_small = d_hs;
    a.hl = h->H / 16; a.wl = h->W / 16; a.hs = h->H / 32; a.ws = h->W / 32;
    a.in_h = h->H; a.in_w = h->W;
    a.logit_min = logit_threshold(conf_thres);
    a.nms_thres = nms_thres;
    a.na = h->num_anchors; a.nc = h->num_cls;
    for (int i = 0; i < 2 * yf::POST_MAX_ANCHORS * 2; ++i) a.anchors[i] = i < 2 * a.na * 2 ? anchors[i] : 0.0;   // [2][na][2]
    bool adj = origin_h > 0 && origin_w > 0 && (origin_h != h->H || origin_w != h->W);
    a.adj_h = adj ? (double)origin_h / h->H : 0.0;
    a.adj_w = adj ? (double)origin_w / h->W : 0.0;
    a.kmax = K_max;
    a.boxes = d_boxes; a.scores = d_scores; a.cls = d_cls; a.src = d_src; a.counts = d_counts;
    a.records = nullptr;
    a.split_tmp = nullptr;
    return a;
}

// Dense frames: the post-process of N frames as N x num_cls workgroups + a small assembly launch (yf_post_kernels.hip post_split_kernel) -- the
// same records, bit for bit.  Automatic (yf_set_post_split 0) when the caller reserves K_max >= 256 survivors per frame (it expects dense
// frames), the model has <= 8 classes and one workgroup per frame would leave at least half of the CUs idle (2 N <= #CU: with 256 dense
// 320x256 frames the per-frame form is the faster one, 0.057 against 0.075 ms).  The scratch is the engine's, grown on demand (the first dense
// call of a size allocates: not inside a stream capture); lanes of one batch use disjoint frame ranges of it.
static int post_setup(yf_handle h, yf::PostArgs& a, int N)
{
    const int n_cu = yf::device_cu_count(h->device);
    const bool on = h->post_split == 1 || (h->post_split == 0 && a.kmax >= 256 && a.nc <= 8 && n_cu > 0 && 2 * N <= n_cu);
    if (!on || a.nc > 64) return YF_OK;
    const size_t need = yf::post_split_tmp_ints(N, a.nc, a.kmax);
    if (need > h->post_tmp_ints) {
        if (h->d_post_tmp) { HIP_OK(hipDeviceSynchronize()); (void)hipFree(h->d_post_tmp); h->d_post_tmp = nullptr; h->post_tmp_ints = 0; }
        HIP_OK(hipMalloc(&h->d_post_tmp, need * sizeof(int32_t)));
        h->post_tmp_ints = need;
    }
    a.split_tmp = h->d_post_tmp;
    return YF_OK;
}

int yf_decode_nms(yf_handle h, const float* d_hl, const float* d_hs, int N, double conf_thres, double nms_thres,
                  const double* anchors, int origin_h, int origin_w, int K_max, int32_t* d_boxes, float* d_scores,
                  int32_t* d_cls, int32_t* d_src, int32_t* d_counts, void* stream)
{
    if (!h || !d_hl || !d_hs || !anchors || !d_boxes || !d_scores || !d_cls || !d_src || !d_counts || N <= 0 || K_max <= 0)
        return fail(YF_E_INVALID, "yf_decode_nms: null pointer or non-positive size");
    HIP_OK(hipSetDevice(h->device));
    yf::PostArgs a = make_post_args(h, d_hl, d_hs, conf_thres, nms_thres, anchors, origin_h, origin_w, K_max, d_boxes, d_scores,
                                    d_cls, d_src, d_counts);
    if (int rs = post_setup(h, a, N)) return rs;
    int rc = yf::launch_post(a, N, (hipStream_t)stream);
    if (rc == -1) return fail(YF_E_INVALID, "frame of %dx%d has too many cells for the on-chip NMS (limit 8191 cells / 160 KiB LDS)", h->H, h->W);
    if (rc) return fail(YF_E_HIP, "hipFuncSetAttribute(post_kernel) failed");
    HIP_OK(hipGetLastError());
    return YF_OK;
}

int yf_decode_nms_packed(yf_handle h, const float* d_hl, const float* d_hs, int N, double conf_thres, double nms_thres, const double* anchors,
                         int origin_h, int origin_w, int K_max, int32_t* d_records, void* stream)
{
    if (!h || !d_hl || !d_hs || !anchors || !d_records || N <= 0 || K_max <= 0)
        return fail(YF_E_INVALID, "yf_decode_nms_packed: null pointer or non-positive size");
    HIP_OK(hipSetDevice(h->device));
    yf::PostArgs a = make_post_args(h, d_hl, d_hs, conf_thres, nms_thres, anchors, origin_h, origin_w, K_max, nullptr, nullptr, nullptr, nullptr, nullptr);
    a.records = d_records;
    if (int rs = post_setup(h, a, N)) return rs;
    int rc = yf::launch_post(a, N, (hipStream_t)stream);
    if (rc == -1) return fail(YF_E_INVALID, "frame of %dx%d has too many cells for the on-chip NMS (limit 8191 cells / 160 KiB LDS)", h->H, h->W);
    if (rc) return fail(YF_E_HIP, "hipFuncSetAttribute(post_kernel) failed");
    HIP_OK(hipGetLastError());
    return YF_OK;
}

static int u8_mode(yf_handle h, int src_h, int src_w, int* down2)
{
    if (src_h == h->H && src_w == h->W) { *down2 = 0; return YF_OK; }
    if (src_h == 2 * h->H && src_w == 2 * h->W) { *down2 = 1; return YF_OK; }
    *down2 = -1;   // any other size: cv::resize's INTER_LINEAR in front of the stem (cv_pre)
    return YF_OK;
}

// cv::resize(INTER_LINEAR, 8-bit) reads per-column / per-row tables (yf_cv_kernels.hip: cv_tables_kernel).  A source size seen before costs
// nothing; a new one costs ONE small kernel on the caller's stream into a free slot of the engine's pool -- no hipMalloc / hipFree / synchronous
// copy on the hot path (the 4-slot host-built version paid a device-wide hipFree + 2 hipMalloc + 2 blocking hipMemcpy per evicted
// size and could not run under stream capture).  Only the 33rd distinct size evicts, behind a device synchronisation (the evicted tables may
// still be read by earlier launches on other streams).
static int cv_tables(yf_engine* e, int sh, int sw, const int4** dx, const int4** dy, hipStream_t s)
{
    const size_t per = (size_t)e->W + e->H;
    if (!e->d_cvpool) HIP_OK(hipMalloc(&e->d_cvpool, (size_t)yf_engine::CV_SLOTS * per * sizeof(int4)));
    int hit = -1, free_slot = -1, lru = 0;
    for (int i = 0; i < yf_engine::CV_SLOTS; ++i) {
        auto& t = e->cvtab[i];
        if (t.built && t.sh == sh && t.sw == sw) { hit = i; break; }
        if (!t.built && free_slot < 0) free_slot = i;
        if (t.used < e->cvtab[lru].used) lru = i;
    }
    int k = hit;
    if (hit < 0) {
        k = free_slot >= 0 ? free_slot : lru;
        auto& t = e->cvtab[k];
        if (free_slot < 0) HIP_OK(hipDeviceSynchronize());
        if (!t.ready) HIP_OK(hipEventCreateWithFlags(&t.ready, hipEventDisableTiming));
        yf::launch_cv_tables(sh, sw, e->H, e->W, e->d_cvpool + k * per, e->d_cvpool + k * per + e->W, s);
        HIP_OK(hipGetLastError());
        HIP_OK(hipEventRecord(t.ready, s));
        t.sh = sh; t.sw = sw; t.built = true; t.stream = s;
    } else if (e->cvtab[k].stream != s) {
        HIP_OK(hipStreamWaitEvent(s, e->cvtab[k].ready, 0));
    }
    e->cvtab[k].used = ++e->cv_tick;
    *dx = e->d_cvpool + k * per; *dy = e->d_cvpool + k * per + e->W;
    return YF_OK;
}

// detect.py:110-116 on the device: [cvtColor(BGR2GRAY)] + [resize] of N source frames into net-sized u8 frames
static int cv_pre(yf_engine* e, const uint8_t* d_src, int N, int src_h, int src_w, int src_c, int gray_bits, uint8_t* d_dst, hipStream_t s)
{
    if (src_h <= 0 || src_w <= 0 || src_h > 16384 || src_w > 16384) return fail(YF_E_INVALID, "source size %dx%d", src_h, src_w);
    if (src_c != 1 && src_c != 3) return fail(YF_E_INVALID, "source frames have 1 channel or 3 (cv2.imread's BGR), not %d", src_c);
    yf::CvArgs a{};
    a.src = d_src; a.dst = d_dst; a.n = N; a.sh = src_h; a.sw = src_w; a.sc = src_c; a.dh = e->H; a.dw = e->W; a.dc = e->input_channel;
    if (e->input_channel == 1 && src_c == 3) {
        if (gray_bits != 0 && gray_bits != 14 && gray_bits != 15) return fail(YF_E_INVALID, "gray_bits must be 14, 15 or 0 (= 15)");
        a.gray = gray_bits == 14 ? 14 : 15;
    } else if (e->input_channel == src_c) {
        a.gray = 0;
    } else {
        return fail(YF_E_INVALID, "a %d-channel net cannot take %d-channel frames (detect.py:110-113: gray from BGR, or the channels as they are)",
                    e->input_channel, src_c);
    }
    a.mode = (src_h == e->H && src_w == e->W) ? 0 : (src_h == 2 * e->H && src_w == 2 * e->W) ? 1 : 2;
    if (a.mode == 2)
        if (int rc = cv_tables(e, src_h, src_w, &a.xtab, &a.ytab, s)) return rc;
    if (yf::launch_cv_pre(a, s)) return fail(YF_E_INVALID, "no pre-process kernel for this combination");
    HIP_OK(hipGetLastError());
    return YF_OK;
}

int yf_cv_preprocess_u8(yf_handle h, const uint8_t* d_src, int N, int src_h, int src_w, int src_c, int gray_bits, uint8_t* d_dst, void* stream)
{
    if (!h || !d_src || !d_dst || N <= 0) return fail(YF_E_INVALID, "yf_cv_preprocess_u8: bad argument");
    HIP_OK(hipSetDevice(h->device));
    return cv_pre(h, d_src, N, src_h, src_w, src_c, gray_bits, d_dst, (hipStream_t)stream);
}

// the net-sized u8 frames of a pass whose source has another size live at the END of the caller's workspace (yf_workspace_bytes counts them)
static int forward_via_cv(yf_handle h, const uint8_t* d_src, int N, int src_h, int src_w, int src_c, int gray_bits, float* d_hl, float* d_hs, void* ws,
                          size_t ws_bytes, void* stream)
{
    const size_t scr = h->cv_scratch_bytes(N);
    if (!ws || ws_bytes < scr) return fail(YF_E_WORKSPACE, "workspace %zu B cannot hold the resized frames (%zu B)", ws_bytes, scr);
    HIP_OK(hipSetDevice(h->device));
    uint8_t* frames = static_cast<uint8_t*>(ws) + ((ws_bytes - scr) & ~(size_t)255);
    if (int rc = cv_pre(h, d_src, N, src_h, src_w, src_c, gray_bits, frames, (hipStream_t)stream)) return rc;
    return run_forward(h, nullptr, N, d_hl, d_hs, ws, (ws_bytes - scr) & ~(size_t)255, (hipStream_t)stream, nullptr, nullptr, 0, nullptr, frames, 0);
}

int yf_forward_u8(yf_handle h, const uint8_t* d_u8, int N, int src_h, int src_w, float* d_hl, float* d_hs, void* ws, size_t ws_bytes,
                  void* stream)
{
    if (!h || !d_u8) return fail(YF_E_INVALID, "yf_forward_u8: null pointer");
    if (h->input_channel > yf_layers::MAX_STEM_INPUT_CHANNEL)
        return fail(YF_E_INVALID, "yf_forward_u8: the fused u8 entry takes 1 .. %d input channels; a %d-channel net goes through yf_preprocess_u8 + yf_forward",
                    (int)yf_layers::MAX_STEM_INPUT_CHANNEL, h->input_channel);
    int down2;
    if (int rc = u8_mode(h, src_h, src_w, &down2)) return rc;
    if (down2 < 0) return forward_via_cv(h, d_u8, N, src_h, src_w, h->input_channel, 0, d_hl, d_hs, ws, ws_bytes, stream);
    return run_forward(h, nullptr, N, d_hl, d_hs, ws, ws_bytes, (hipStream_t)stream, nullptr, nullptr, 0, nullptr, d_u8, down2);
}

int yf_forward_bgr_u8(yf_handle h, const uint8_t* d_bgr, int N, int src_h, int src_w, int gray_bits, float* d_hl, float* d_hs, void* ws,
                      size_t ws_bytes, void* stream)
{
    if (!h || !d_bgr) return fail(YF_E_INVALID, "yf_forward_bgr_u8: null pointer");
    if (h->input_channel == 3) return yf_forward_u8(h, d_bgr, N, src_h, src_w, d_hl, d_hs, ws, ws_bytes, stream);   // detect.py:112-113: the frame as it is
    if (h->input_channel != 1) return fail(YF_E_INVALID, "cv2.imread's 3-channel frames feed 1- and 3-channel nets (detect.py:110-113), not %d", h->input_channel);
    return forward_via_cv(h, d_bgr, N, src_h, src_w, 3, gray_bits, d_hl, d_hs, ws, ws_bytes, stream);
}

int yf_nms_sorted(yf_handle h, const int32_t* d_boxes, int n, double nms_thres, int32_t* d_sup, void* stream)
{
    if (!h || !d_boxes || !d_sup || n < 0) return fail(YF_E_INVALID, "yf_nms_sorted: bad argument");
    if (n == 0) return YF_OK;
    HIP_OK(hipSetDevice(h->device));
    yf::launch_nms_sorted(d_boxes, n, nms_thres, d_sup, (hipStream_t)stream);
    HIP_OK(hipGetLastError());
    return YF_OK;
}

// Validation-time path.  anchors: HOST double[3][2] of THIS head, net-input pixels.
int yf_val_decode_head(yf_handle h, const float* d_head, int N, int fh, int fw, const double* anchors, int M_total, int m_off,
                       float* d_out, void* stream)
{
    const int na = h ? h->num_anchors : 0;
    if (!h || !d_head || !anchors || !d_out || N <= 0 || fh <= 0 || fw <= 0 || m_off < 0 || m_off + na * fh * fw > M_total)
        return fail(YF_E_INVALID, "yf_val_decode_head: bad argument");
    HIP_OK(hipSetDevice(h->device));
    // yolo_loss.py:52-56: strides and feature-map-scaled anchors are Python doubles, stored into FloatTensors
    const double stride_h = (double)h->H / fh, stride_w = (double)h->W / fw;
    float anc[2 * yf::POST_MAX_ANCHORS];
    for (int a = 0; a < na; ++a) { anc[2 * a] = (float)(anchors[2 * a] / stride_w); anc[2 * a + 1] = (float)(anchors[2 * a + 1] / stride_h); }
    yf::launch_val_decode(d_head, d_out, N, fh, fw, M_total, m_off, anc, na, h->num_cls, (float)stride_w, (float)stride_h, (hipStream_t)stream);
    HIP_OK(hipGetLastError());
    return YF_OK;
}

int yf_val_nms(yf_handle h, const float* d_pred, int N, int M, double conf_thres, double nms_thres, int K_max, float* d_det,
               int32_t* d_counts, void* stream)
{
    if (!h) return fail(YF_E_INVALID, "yf_val_nms: null handle");
    return yf_val_nms_ex(h, d_pred, N, M, h->num_cls, conf_thres, nms_thres, K_max, d_det, d_counts, stream);
}

int yf_val_nms_ex(yf_handle h, const float* d_pred, int N, int M, int num_classes, double conf_thres, double nms_thres, int K_max, float* d_det,
                  int32_t* d_counts, void* stream)
{
    if (!h || !d_pred || !d_det || !d_counts || N <= 0 || M <= 0 || K_max <= 0 || num_classes < 1) return fail(YF_E_INVALID, "yf_val_nms: bad argument");
    HIP_OK(hipSetDevice(h->device));
    int rc = yf::launch_val_nms(d_pred, N, M, num_classes, (float)conf_thres, (float)nms_thres, K_max, d_det, d_counts, (hipStream_t)stream);
    if (rc == -1) return fail(YF_E_INVALID, "yf_val_nms: at most 8191 boxes per image");
    if (rc) return fail(YF_E_HIP, "hipFuncSetAttribute(val_nms_kernel) failed");
    HIP_OK(hipGetLastError());
    return YF_OK;
}

// Training-time loss of ONE head (first slice).  anchors: HOST double[3][2] of this head, net-input pixels.
int yf_train_loss_workspace_bytes(yf_handle h, int N, int fh, int fw, size_t* out)
{
    if (!h) return fail(YF_E_INVALID, "yf_train_loss_workspace_bytes: null handle");
    return yf_train_head_loss_workspace_bytes_ex(N, fh, fw, h->num_anchors, h->num_cls, out);
}

int yf_train_head_loss_workspace_bytes(int N, int fh, int fw, size_t* out) { return yf_train_head_loss_workspace_bytes_ex(N, fh, fw, 3, 3, out); }

int yf_train_head_loss_workspace_bytes_ex(int N, int fh, int fw, int num_anchors, int num_classes, size_t* out)
{
    if (!out || N <= 0 || fh <= 0 || fw <= 0 || num_anchors < 1 || num_anchors > yf::POST_MAX_ANCHORS || num_classes < 1)
        return fail(YF_E_INVALID, "yf_train_head_loss_workspace_bytes: bad argument");
    *out = yf::train_loss_workspace_bytes(N, fh, fw, num_anchors, num_classes);
    return YF_OK;
}

int yf_train_loss(yf_handle h, const float* d_head, int N, int fh, int fw, const double* anchors, const float* d_targets, int T,
                  double ignore_thres, void* d_work, size_t work_bytes, float* d_losses, float* d_grad_head, void* stream)
{
    if (!h) return fail(YF_E_INVALID, "yf_train_loss: null handle");
    return yf_train_head_loss_ex(h->device, h->H, h->W, d_head, N, fh, fw, anchors, h->num_anchors, h->num_cls, d_targets, T, ignore_thres, d_work,
                                 work_bytes, d_losses, d_grad_head, stream);
}

// The same without an engine handle (the loss needs the net-input size only for the head's stride), like the other yf_train_* entries.
int yf_train_head_loss(int device, int H, int W, const float* d_head, int N, int fh, int fw, const double* anchors, const float* d_targets,
                       int T, double ignore_thres, void* d_work, size_t work_bytes, float* d_losses, float* d_grad_head, void* stream)
{
    return yf_train_head_loss_ex(device, H, W, d_head, N, fh, fw, anchors, 3, 3, d_targets, T, ignore_thres, d_work, work_bytes, d_losses, d_grad_head,
                                 stream);
}

int yf_train_head_loss_ex(int device, int H, int W, const float* d_head, int N, int fh, int fw, const double* anchors, int num_anchors,
                          int num_classes, const float* d_targets, int T, double ignore_thres, void* d_work, size_t work_bytes, float* d_losses,
                          float* d_grad_head, void* stream)
{
    if (!d_head || !anchors || !d_targets || !d_work || !d_losses || N <= 0 || fh <= 0 || fw <= 0 || T <= 0 || H <= 0 || W <= 0 ||
        num_anchors < 1 || num_anchors > yf::POST_MAX_ANCHORS || num_classes < 1)
        return fail(YF_E_INVALID, "yf_train_head_loss: bad argument");
    if (work_bytes < yf::train_loss_workspace_bytes(N, fh, fw, num_anchors, num_classes)) return fail(YF_E_WORKSPACE, "yf_train_head_loss: workspace too small");
    if (reinterpret_cast<uintptr_t>(d_work) & 7) return fail(YF_E_INVALID, "yf_train_head_loss: workspace must be 8-byte aligned");
    HIP_OK(hipSetDevice(device));
    // yolo_loss.py:52-56: strides and feature-map-scaled anchors are Python doubles; torch uses them as float32
    const double stride_h = (double)H / fh, stride_w = (double)W / fw;
    float anc[2 * yf::POST_MAX_ANCHORS];
    for (int a = 0; a < num_anchors; ++a) { anc[2 * a] = (float)(anchors[2 * a] / stride_w); anc[2 * a + 1] = (float)(anchors[2 * a + 1] / stride_h); }
    yf::launch_train_loss(d_head, N, fh, fw, anc, num_anchors, num_classes, d_targets, T, (float)ignore_thres, d_work, d_losses, d_grad_head,
                          (hipStream_t)stream);
    HIP_OK(hipGetLastError());
    return YF_OK;
}


int yf_detect(yf_handle h, const float* d_x, int N, double conf_thres, double nms_thres, const double* anchors, int origin_h,
              int origin_w, int K_max, int32_t* d_boxes, float* d_scores, int32_t* d_cls, int32_t* d_src, int32_t* d_counts,
              float* d_hl, float* d_hs, void* ws, size_t ws_bytes, void* stream)
{
    if (!h) return fail(YF_E_INVALID, "yf_detect: null handle");
    size_t chain = ((h->frame_floats_max() * (size_t)chunk_frames(h, N) * h->lanes * h->esz()) + 255) & ~(size_t)255;
    size_t heads = (h->head_l_elems + h->head_s_elems) * (size_t)N * sizeof(float);
    if (!ws || ws_bytes < chain + ((d_hl && d_hs) ? 0 : heads)) return fail(YF_E_WORKSPACE, "workspace too small");
    float* hl = d_hl ? d_hl : reinterpret_cast<float*>(static_cast<char*>(ws) + chain);
    float* hs = d_hs ? d_hs : reinterpret_cast<float*>(static_cast<char*>(ws) + chain) + h->head_l_elems * (size_t)N;
    if (!anchors || !d_boxes || !d_scores || !d_cls || !d_src || !d_counts || N <= 0 || K_max <= 0)
        return fail(YF_E_INVALID, "yf_detect: null pointer or non-positive size");
    yf::PostArgs pa = make_post_args(h, hl, hs, conf_thres, nms_thres, anchors, origin_h, origin_w, K_max, d_boxes, d_scores,
                                     d_cls, d_src, d_counts);
    if (int rs = post_setup(h, pa, N)) return rs;
    int rc = run_forward(h, d_x, N, hl, hs, ws, chain, (hipStream_t)stream, nullptr, nullptr, 0, nullptr, nullptr, 0, &pa);
    if (rc) return rc;
    HIP_OK(hipGetLastError());
    return YF_OK;
}

int yf_detect_packed(yf_handle h, const float* d_x, int N, double conf_thres, double nms_thres, const double* anchors, int origin_h,
                     int origin_w, int K_max, int32_t* d_records, float* d_hl, float* d_hs, void* ws, size_t ws_bytes, void* stream)
{
    if (!h) return fail(YF_E_INVALID, "yf_detect_packed: null handle");
    size_t chain = ((h->frame_floats_max() * (size_t)chunk_frames(h, N) * h->lanes * h->esz()) + 255) & ~(size_t)255;
    size_t heads = (h->head_l_elems + h->head_s_elems) * (size_t)N * sizeof(float);
    if (!ws || ws_bytes < chain + ((d_hl && d_hs) ? 0 : heads)) return fail(YF_E_WORKSPACE, "workspace too small");
    float* hl = d_hl ? d_hl : reinterpret_cast<float*>(static_cast<char*>(ws) + chain);
    float* hs = d_hs ? d_hs : reinterpret_cast<float*>(static_cast<char*>(ws) + chain) + h->head_l_elems * (size_t)N;
    if (!anchors || !d_records || N <= 0 || K_max <= 0) return fail(YF_E_INVALID, "yf_detect_packed: null pointer or non-positive size");
    yf::PostArgs pa = make_post_args(h, hl, hs, conf_thres, nms_thres, anchors, origin_h, origin_w, K_max, nullptr, nullptr, nullptr, nullptr, nullptr);
    pa.records = d_records;
    if (int rs = post_setup(h, pa, N)) return rs;
    int rc = run_forward(h, d_x, N, hl, hs, ws, chain, (hipStream_t)stream, nullptr, nullptr, 0, nullptr, nullptr, 0, &pa);
    if (rc) return rc;
    HIP_OK(hipGetLastError());
    return YF_OK;
}

int yf_preprocess_u8(yf_handle h, const uint8_t* d_u8, int N, int src_h, int src_w, float* d_x, void* stream)
{
    if (!h || !d_u8 || !d_x || N <= 0) return fail(YF_E_INVALID, "yf_preprocess_u8: bad argument");
    int down2;
    if (src_h == h->H && src_w == h->W) down2 = 0;
    else if (src_h == 2 * h->H && src_w == 2 * h->W) down2 = 1;
    else return fail(YF_E_INVALID, "source %dx%d: only 1x or exact 2x of the net input %dx%d is supported", src_h, src_w, h->H, h->W);
    HIP_OK(hipSetDevice(h->device));
    yf::launch_preprocess(d_u8, d_x, N, h->H, h->W, down2, (hipStream_t)stream, h->input_channel);
    HIP_OK(hipGetLastError());
    return YF_OK;
}

int yf_streams_overlap(yf_handle h, void* stream_a, void* stream_b, int* overlap)
{
    if (!h || !overlap || stream_a == stream_b) return fail(YF_E_INVALID, "yf_streams_overlap: bad argument");
    HIP_OK(hipSetDevice(h->device));
    bool ov = false;
    if (int rc = streams_overlap(h, (hipStream_t)stream_a, (hipStream_t)stream_b, &ov)) return rc;
    *overlap = ov ? 1 : 0;
    return YF_OK;
}

int yf_io_params(yf_handle h, int* input_channel, int* num_anchors, int* num_cls, int* num_out)
{
    if (!h) return fail(YF_E_INVALID, "yf_io_params: null handle");
    if (input_channel) *input_channel = h->input_channel;
    if (num_anchors) *num_anchors = h->num_anchors;
    if (num_cls) *num_cls = h->num_cls;
    if (num_out) *num_out = h->num_out;
    return YF_OK;
}

int yf_num_launches(yf_handle h, int* out)
{
    if (!h || !out) return fail(YF_E_INVALID, "bad argument");
    *out = (int)h->plan().ops.size();
    return YF_OK;
}

// ---- per-op introspection / timing (bench.py's roofline object) ----
static size_t layer_io_elems(const LayerSpec& L, size_t in_px, size_t out_px) { return in_px * L.cin + out_px * L.cout; }

int yf_op_info(yf_handle h, int op, char* name, int name_len, double* algorithmic_bytes_per_frame, double* flops_per_frame)
{
    double mfma = 0, valu = 0;
    int rc = yf_op_info_ex(h, op, name, name_len, algorithmic_bytes_per_frame, &mfma, &valu);
    if (rc == YF_OK && flops_per_frame) *flops_per_frame = mfma + valu;
    return rc;
}

int yf_op_info_ex(yf_handle h, int op, char* name, int name_len, double* algorithmic_bytes_per_frame, double* mfma_flops_per_frame,
                  double* valu_flops_per_frame)
{
    if (!h || op < 0 || op >= (int)h->plan().ops.size()) return fail(YF_E_INVALID, "bad op index");
    const Plan& P = h->plan();
    const LayerSpec* const kLayers = h->layers;
    const Op& o = P.ops[op];
    const Tensor &ti = P.tensors[o.in1], &to = P.tensors[o.out];
    std::string nm;
    double elems = 0, macs_mfma = 0, macs_valu = 0;
    // which pipe a layer's MACs run on in THIS plan: depthwise convs and everything inside the VALU block kernels are vector
    // FMAs; the pointwise / dense / deconv layers of the k19m, mres, mdw and pw_mfma / pw_ws kernels are MFMAs
    const bool op_mfma = o.type == OP_K19 || o.type == OP_MRES || o.type == OP_MDW || o.type == OP_MDW2 || o.type == OP_DCAT || (o.type == OP_LAYER && o.mfma_off >= 0);
    auto add = [&](int li, size_t in_px, size_t out_px, bool res) {
        if (li < 0) return;
        const LayerSpec& L = kLayers[li];
        if (!nm.empty()) nm += "+";
        nm += L.name;
        elems += (double)layer_io_elems(L, in_px, out_px) + (res ? (double)out_px * L.cout : 0.0);
        const double k2 = L.kind == K_DECONV ? 1.0 : (double)L.k * L.k;  // deconv 2x2 s2: each output pixel sees one tap
        const double m = (L.kind == K_DW ? (double)out_px * L.cout * k2 : (double)out_px * L.cout * L.cin * k2);
        if (op_mfma && L.kind != K_DW) macs_mfma += m; else macs_valu += m;
    };
    const size_t ipx = (size_t)ti.H * ti.W, opx = (size_t)to.H * to.W;
    if (o.type == OP_FUSED_BLOCK || o.type == OP_MRES) {
        size_t epx = ipx;
        if (o.l_pre >= 0) { epx = ipx / 4; add(o.l_pre, ipx, epx, false); }
        for (int k = 0; k < o.nblk; ++k) {
            add(o.l_exp + 3 * k, epx, epx, false); add(o.l_dw + 3 * k, epx, opx, false); add(o.l_proj + 3 * k, opx, opx, o.res >= 0);
        }
        add(o.l_post, opx, opx, false);
    } else if (o.type == OP_K19) {
        add(o.l_exp, ipx, ipx, false); add(o.l_dw, ipx, opx, false); add(o.l_proj, opx, opx, false);
    } else if (o.type == OP_DCAT) {
        add(o.layer, ipx, opx, false); add(o.l_proj, opx, opx, false);
    } else if (o.type == OP_MDW2) {
        add(o.l_dw, ipx, ipx, false); add(o.l_proj, ipx, ipx, false); add(o.l_dw2, ipx, ipx, false); add(o.l_proj2, ipx, ipx, false);
        add(o.l_head, ipx, ipx, false);
    } else if (o.type == OP_MDW) {
        add(o.l_dw, ipx, ipx, false); add(o.l_proj, ipx, ipx, false); add(o.l_head, ipx, ipx, false);
    } else {
        add(o.layer, ipx, opx, o.res >= 0);
    }
    if (name && name_len > 0) snprintf(name, (size_t)name_len, "%s", nm.c_str());
    if (algorithmic_bytes_per_frame) *algorithmic_bytes_per_frame = elems * (double)h->esz();
    if (mfma_flops_per_frame) *mfma_flops_per_frame = macs_mfma * 2.0;
    if (valu_flops_per_frame) *valu_flops_per_frame = macs_valu * 2.0;
    return YF_OK;
}

int yf_op_dtype(yf_handle h, int op, int* kernel_dtype)
{
    if (!h || !kernel_dtype || op < 0 || op >= (int)h->plan().ops.size()) return fail(YF_E_INVALID, "bad op index");
    *kernel_dtype = h->plan().ops[op].kdt;
    return YF_OK;
}

int yf_op_dispatches(yf_handle h, int op, int N, int* dispatches)
{
    if (!h || !dispatches || N <= 0 || op < 0 || op >= (int)h->plan().ops.size()) return fail(YF_E_INVALID, "yf_op_dispatches: bad argument");
    const Op& o = h->plan().ops[op];
    *dispatches = 1;
    if (o.type == OP_MRES) {
        const LayerSpec &LE = h->layers[o.l_exp], &LP = h->layers[o.l_proj];
        const Tensor& ti = h->plan().tensors[o.in1];
        *dispatches = yf::mres_dispatches(LE.cin, LE.cout, LP.cout, o.res >= 0, h->layers[o.l_dw].stride, o.nblk, o.out2 >= 0, o.l_post >= 0, ti.H, ti.W, N, o.kdt,
                                          h->d_esplit && h->split_sums);
    } else if (o.type == OP_MDW2) {
        const Tensor& ti = h->plan().tensors[o.in1];
        if (yf::mdw2_esplit_ok(ti.H, ti.W, h->num_out, N, o.kdt, h->split_sums ? h->d_esplit : nullptr)) *dispatches = 3;
    }
    return YF_OK;
}

static int profile_forward(yf_handle h, const float* d_x, const uint8_t* d_u8, int down2, int N, void* ws, size_t ws_bytes, void* stream, float* op_ms,
                           int n_ops);
int yf_profile_forward(yf_handle h, const float* d_x, int N, void* ws, size_t ws_bytes, void* stream, float* op_ms, int n_ops)
{
    return profile_forward(h, d_x, nullptr, 0, N, ws, ws_bytes, stream, op_ms, n_ops);
}

int yf_profile_forward_u8(yf_handle h, const uint8_t* d_u8, int N, int src_h, int src_w, void* ws, size_t ws_bytes, void* stream, float* op_ms,
                          int n_ops)
{
    if (!h || !d_u8) return fail(YF_E_INVALID, "yf_profile_forward_u8: null pointer");
    if (h->input_channel > yf_layers::MAX_STEM_INPUT_CHANNEL) return fail(YF_E_INVALID, "yf_profile_forward_u8: 1 .. %d input channels", (int)yf_layers::MAX_STEM_INPUT_CHANNEL);
    int down2;
    if (int rc = u8_mode(h, src_h, src_w, &down2)) return rc;
    if (down2 < 0) return fail(YF_E_INVALID, "yf_profile_forward_u8: source %dx%d -- the profiled pass takes frames of the net's size or exactly 2x", src_h, src_w);
    return profile_forward(h, nullptr, d_u8, down2, N, ws, ws_bytes, stream, op_ms, n_ops);
}

static int profile_forward(yf_handle h, const float* d_x, const uint8_t* d_u8, int down2, int N, void* ws, size_t ws_bytes, void* stream, float* op_ms,
                           int n_ops)
{
    if (!h || !op_ms) return fail(YF_E_INVALID, "yf_profile_forward: null pointer");
    const size_t nops = h->plan().ops.size();
    if (n_ops != (int)nops) return fail(YF_E_INVALID, "op_ms must hold %zu entries", nops);
    size_t chain = ((h->frame_floats_max() * (size_t)N * h->esz()) + 255) & ~(size_t)255;
    size_t heads = (h->head_l_elems + h->head_s_elems) * (size_t)N * sizeof(float);
    if (!ws || ws_bytes < chain + heads) return fail(YF_E_WORKSPACE, "workspace too small");
    float* hl = reinterpret_cast<float*>(static_cast<char*>(ws) + chain);
    float* hs = hl + h->head_l_elems * (size_t)N;
    ProfileEvents pe;
    pe.repeats = h->profile_repeats > 0 ? h->profile_repeats : 1;
    pe.ev.assign(nops + 1, nullptr);
    // the events are destroyed on EVERY exit path (HIP_OK returns from the middle of this function)
    struct EventGuard { std::vector<hipEvent_t>& ev; ~EventGuard() { for (auto& x : ev) if (x) { (void)hipEventDestroy(x); x = nullptr; } } } guard{pe.ev};
    for (auto& ev : pe.ev) HIP_OK(hipEventCreate(&ev));
    int rc = run_forward(h, d_x, N, hl, hs, ws, chain, (hipStream_t)stream, nullptr, nullptr, 0, &pe, d_u8, down2);
    if (rc == YF_OK) {
        HIP_OK(hipEventSynchronize(pe.ev[nops]));
        for (size_t i = 0; i < nops; ++i) {
            HIP_OK(hipEventElapsedTime(&op_ms[i], pe.ev[i], pe.ev[i + 1]));
            op_ms[i] /= (float)pe.repeats;
        }
    }
    return rc;
}

// Where the profiled pass left its head logits inside the caller's workspace (byte offsets; NCHW fp32 like yf_forward's outputs): lets a
// test hold the repeated-launch pass itself to yf_forward's bits.
int yf_profile_head_offsets(yf_handle h, int N, size_t* large_off, size_t* small_off)
{
    if (!h || !large_off || !small_off || N <= 0) return fail(YF_E_INVALID, "yf_profile_head_offsets: null pointer or N <= 0");
    const size_t chain = ((h->frame_floats_max() * (size_t)N * h->esz()) + 255) & ~(size_t)255;
    *large_off = chain;
    *small_off = chain + h->head_l_elems * (size_t)N * sizeof(float);
    return YF_OK;
}

int yf_set_profile_repeats(yf_handle h, int repeats)
{
    if (!h || repeats < 1 || repeats > 64) return fail(YF_E_INVALID, "yf_set_profile_repeats: repeats must be 1 .. 64");
    h->profile_repeats = repeats;
    return YF_OK;
}

int yf_set_lanes(yf_handle h, int lanes)
{
    if (!h || lanes < 1 || lanes > 4) return fail(YF_E_INVALID, "lanes must be 1..4");
    h->lanes = lanes;
    return YF_OK;
}

int yf_set_branches(yf_handle h, int on)
{
    if (!h || on < 0 || on > 1) return fail(YF_E_INVALID, "branches must be 0 or 1");
    h->branches = on;
    return YF_OK;
}

int yf_set_split_sums(yf_handle h, int on)
{
    if (!h || on < 0 || on > 1) return fail(YF_E_INVALID, "split_sums must be 0 or 1");
    h->split_sums = on;
    return YF_OK;
}

int yf_set_post_split(yf_handle h, int mode)
{
    if (!h || mode < 0 || mode > 2) return fail(YF_E_INVALID, "post_split must be 0 (auto), 1 (always) or 2 (never)");
    h->post_split = mode;
    return YF_OK;
}

int yf_set_fusion(yf_handle h, int level)
{
    if (!h || level < 0 || level > 2) return fail(YF_E_INVALID, "fusion level must be 0, 1 or 2");
    h->fusion = level;
    return YF_OK;
}

int yf_set_chunk(yf_handle h, int frames)
{
    if (!h || frames < 0) return fail(YF_E_INVALID, "bad argument");
    h->chunk = frames;
    return YF_OK;
}

}  // extern "C"
