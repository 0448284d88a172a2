// yf_layers.h -- the YoloFastest layer table (module-definition order == forward order, yolo_fastest.py:78-148), shared by the
// inference engine (yf_engine.hip) and the training engine (yf_train_engine.hip).
#pragma once
#include <string.h>

namespace yf_layers {

enum Kind { K_PW = 0, K_DW = 1, K_DENSE = 2, K_DECONV = 3, K_HEAD = 4 };

// ONE set of io_params limits for the inference engine (yf_create: blob header), the trainer (yf_trainer_create_ex) and model.py
// (MAX_* there mirror these): a model that constructs also runs inference AND trains.
enum { MAX_INPUT_CHANNEL = 64, MAX_STEM_INPUT_CHANNEL = 4 /* conv0 fused into the stem kernel; above: a launch of its own */, MAX_NUM_ANCHORS = 8, MAX_NUM_CLS = 4096, MAX_NUM_OUT = MAX_NUM_ANCHORS * (5 + MAX_NUM_CLS) };

// The YoloFastest layer table, module-definition order (yolo_fastest.py:78-148). The blob must match it.
// kBaseLayers is the graph for the shipped io_params (input_channel 1, num_out = 3 anchors x (5 + 3 classes) = 24); the reference's
// constructor is parameterised on input_channel (conv0's Cin, :78) and num_out = num_anchors * (5 + num_cls) (the two head convs,
// :138, :148): make_layers() gives an engine / trainer its own copy with those three entries set.
struct LayerSpec {
    const char* name;
    int kind, cin, cout, k, stride, relu;
};
#define RES(n, c, e) {n ".conv1", K_PW, c, e, 1, 1, 1}, {n ".conv2", K_DW, e, e, 3, 1, 1}, {n ".conv3", K_PW, e, c, 1, 1, 0}
static const LayerSpec kBaseLayers[] = {
    {"conv0", K_DENSE, 1, 8, 3, 2, 1}, {"conv1_2", K_PW, 8, 8, 1, 1, 1}, {"conv1_3", K_DW, 8, 8, 3, 1, 1},
    {"conv1_4", K_PW, 8, 4, 1, 1, 0}, RES("res1_1", 4, 8),
    {"conv1_8", K_PW, 4, 24, 1, 1, 1}, {"conv1_9", K_DENSE, 24, 24, 3, 2, 1}, {"conv2_1", K_PW, 24, 8, 1, 1, 0},
    RES("res2_1", 8, 32), RES("res2_2", 8, 32),
    {"conv2_2", K_PW, 8, 32, 1, 1, 1}, {"conv2_3", K_DW, 32, 32, 3, 2, 1}, {"conv3_1", K_PW, 32, 8, 1, 1, 0},
    RES("res3_1", 8, 48), RES("res3_2", 8, 48),
    {"conv3_2", K_PW, 8, 48, 1, 1, 1}, {"conv3_3", K_DW, 48, 48, 3, 1, 1}, {"conv3_4", K_PW, 48, 16, 1, 1, 0},
    RES("res3_3", 16, 96), RES("res3_4", 16, 96), RES("res3_5", 16, 96), RES("res3_6", 16, 96),
    {"conv3_5", K_PW, 16, 96, 1, 1, 1}, {"conv3_6", K_DW, 96, 96, 3, 2, 1}, {"conv4_1", K_PW, 96, 24, 1, 1, 0},
    RES("res4_1", 24, 136), RES("res4_2", 24, 136), RES("res4_3", 24, 136), RES("res4_4", 24, 136),
    {"conv4_2", K_PW, 24, 136, 1, 1, 1}, {"conv4_3", K_DW, 136, 136, 3, 2, 1}, {"conv5_1", K_PW, 136, 48, 1, 1, 1},
    RES("res5_1", 48, 224), RES("res5_2", 48, 224), RES("res5_3", 48, 224), RES("res5_4", 48, 224),
    RES("res5_5", 48, 224),
    {"conv5_2", K_PW, 48, 96, 1, 1, 1}, {"conv5_3", K_DW, 96, 96, 5, 1, 1}, {"conv5_4", K_PW, 96, 128, 1, 1, 0},
    {"conv5_5", K_DW, 128, 128, 5, 1, 1}, {"conv5_6", K_PW, 128, 128, 1, 1, 0}, {"head_5", K_HEAD, 128, 24, 1, 1, 0},
    {"deconv5_1", K_DECONV, 96, 96, 2, 2, 1},
    {"conv4_1_1", K_PW, 232, 96, 1, 1, 1}, {"conv4_1_2", K_DW, 96, 96, 5, 1, 1}, {"conv4_1_3", K_PW, 96, 96, 1, 1, 0},
    {"conv4_1_4", K_DW, 96, 96, 5, 1, 1}, {"conv4_1_5", K_PW, 96, 96, 1, 1, 0}, {"head_4", K_HEAD, 96, 24, 1, 1, 0},
};
constexpr int kNumLayers = sizeof(kBaseLayers) / sizeof(kBaseLayers[0]);
static_assert(kNumLayers == 86, "84 conv+BN units + 2 heads");

inline int find_layer(const char* name)
{
    for (int i = 0; i < kNumLayers; ++i)
        if (!strcmp(kBaseLayers[i].name, name)) return i;
    return -1;
}

inline void make_layers(LayerSpec* out, int input_channel, int num_out)
{
    for (int i = 0; i < kNumLayers; ++i) {
        out[i] = kBaseLayers[i];
        if (i == 0) out[i].cin = input_channel;           // conv0
        if (out[i].kind == K_HEAD) out[i].cout = num_out;  // head_5, head_4
    }
}

}  // namespace yf_layers
