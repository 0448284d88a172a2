"""ORACLE (test infrastructure, not product): CPU restatement of the YOLO-Fastest forward pass.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
The product path (yolo-fastest-and-embedded-deployment_amd/) never does.

Restates, as plain functional PyTorch on the CPU in fp32, what the reference computes in
  /root/reference/src/model_training/model/yolo_fastest.py
    conv_norm_relu   :16-26   Conv2d(bias=False, pad=(k-1)//2) -> BatchNorm2d(eval) -> ReLU
    conv_norm        :29-38   same without ReLU
    deconv_norm_relu :42-48   ConvTranspose2d(k=2, s=2, p=0, bias=False) -> BN -> ReLU
    BasicResBlock    :52-66   pw-expand+ReLU -> dw3x3+ReLU -> pw-project (linear) -> += residual
    YoloFastest.__init__ :70-148 (layer table) and .forward :150-218 (graph)
directly from the 508-key state-dict the reference loads at src/detect.py:90-91.

Parity pin: tests/test_oracle_golden.py checks this file against tests/golden/golden_{256,512}.npz,
which tests/golden/make_golden.py produced by running the reference module itself on the
shipped checkpoints (head logits and 25 per-layer probes); the train-mode graph (forward(train=True)) against
tests/golden/golden_train_256.npz: the reference's own training iteration -- its heads, every parameter gradient, the
BatchNorm running statistics.  The arithmetic below the
torch.nn.functional calls is PyTorch's (the reference pins pytorch 1.2/1.4; un-vendored).
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # nn.BatchNorm2d default, yolo_fastest.py:12-13

# (name, kind, cin, cout, k, stride, relu)   kind: 'c' dense conv, 'dw' depthwise, 'dc' deconv.  The table is for io_params'
# input_channel = 1; `cin` is only read for the depthwise layers' group count, so a state-dict built for another input_channel /
# num_cls / num_anchors (conv0 [8,Cin,3,3], heads [A*(5+C),.,1,1]: yolo_fastest.py:72-78,138,148) runs through the same code.
# Order == module definition order == state-dict order (yolo_fastest.py:78-148).
def _res(name, c, e):
    return [(f"{name}.conv1", "c", c, e, 1, 1, True), (f"{name}.conv2", "dw", e, e, 3, 1, True),
            (f"{name}.conv3", "c", e, c, 1, 1, False)]


LAYERS = (
    [("conv0", "c", 1, 8, 3, 2, True), ("conv1_2", "c", 8, 8, 1, 1, True), ("conv1_3", "dw", 8, 8, 3, 1, True),
     ("conv1_4", "c", 8, 4, 1, 1, False)]
    + _res("res1_1", 4, 8)
    + [("conv1_8", "c", 4, 24, 1, 1, True), ("conv1_9", "c", 24, 24, 3, 2, True), ("conv2_1", "c", 24, 8, 1, 1, False)]
    + _res("res2_1", 8, 32) + _res("res2_2", 8, 32)
    + [("conv2_2", "c", 8, 32, 1, 1, True), ("conv2_3", "dw", 32, 32, 3, 2, True), ("conv3_1", "c", 32, 8, 1, 1, False)]
    + _res("res3_1", 8, 48) + _res("res3_2", 8, 48)
    + [("conv3_2", "c", 8, 48, 1, 1, True), ("conv3_3", "dw", 48, 48, 3, 1, True), ("conv3_4", "c", 48, 16, 1, 1, False)]
    + _res("res3_3", 16, 96) + _res("res3_4", 16, 96) + _res("res3_5", 16, 96) + _res("res3_6", 16, 96)
    + [("conv3_5", "c", 16, 96, 1, 1, True), ("conv3_6", "dw", 96, 96, 3, 2, True), ("conv4_1", "c", 96, 24, 1, 1, False)]
    + _res("res4_1", 24, 136) + _res("res4_2", 24, 136) + _res("res4_3", 24, 136) + _res("res4_4", 24, 136)
    + [("conv4_2", "c", 24, 136, 1, 1, True), ("conv4_3", "dw", 136, 136, 3, 2, True),
       ("conv5_1", "c", 136, 48, 1, 1, True)]
    + _res("res5_1", 48, 224) + _res("res5_2", 48, 224) + _res("res5_3", 48, 224) + _res("res5_4", 48, 224)
    + _res("res5_5", 48, 224)
    + [("conv5_2", "c", 48, 96, 1, 1, True), ("conv5_3", "dw", 96, 96, 5, 1, True), ("conv5_4", "c", 96, 128, 1, 1, False),
       ("conv5_5", "dw", 128, 128, 5, 1, True), ("conv5_6", "c", 128, 128, 1, 1, False),
       ("deconv5_1", "dc", 96, 96, 2, 2, True),
       ("conv4_1_1", "c", 232, 96, 1, 1, True), ("conv4_1_2", "dw", 96, 96, 5, 1, True),
       ("conv4_1_3", "c", 96, 96, 1, 1, False), ("conv4_1_4", "dw", 96, 96, 5, 1, True),
       ("conv4_1_5", "c", 96, 96, 1, 1, False)]
)
_BY_NAME = {l[0]: l for l in LAYERS}


BN_MOMENTUM = 0.1  # nn.BatchNorm2d default


def _unit(sd, name, x, train=False, pre=None):
    """One conv+BN(+ReLU) unit with the reference's un-folded arithmetic order.  train: BatchNorm2d in train mode -- batch
    statistics, and running_mean / running_var / num_batches_tracked of `sd` are updated in place like the module's buffers.
    pre: optional dict; for units WITH a ReLU it receives name -> {z: the pre-activation (BatchNorm output), mu / sigma / absmax: per
    channel batch statistics of the conv output} (tests use it to find the ReLU decisions that lie within fp32 rounding of zero)."""
    _, kind, cin, cout, k, s, relu = _BY_NAME[name]
    w = sd[name + ".0.weight"]
    if kind == "dc":
        c = F.conv_transpose2d(x, w, None, stride=2, padding=0)
    else:
        c = F.conv2d(x, w, None, stride=s, padding=(k - 1) // 2, groups=(cin if kind == "dw" else 1))
    y = F.batch_norm(c, sd[name + ".1.running_mean"], sd[name + ".1.running_var"], sd[name + ".1.weight"],
                     sd[name + ".1.bias"], train, BN_MOMENTUM if train else 0.0, BN_EPS)
    if train and (name + ".1.num_batches_tracked") in sd:
        sd[name + ".1.num_batches_tracked"] += 1
    if pre is not None and relu:
        pre[name] = dict(z=y.detach(), mu=c.detach().mean((0, 2, 3)), sigma=c.detach().var((0, 2, 3), unbiased=False).sqrt(),
                         absmax=c.detach().abs().amax((0, 2, 3)))
    return F.relu(y) if relu else y


def _resblock(sd, name, x, train=False, pre=None):
    y = _unit(sd, name + ".conv1", x, train, pre)
    y = _unit(sd, name + ".conv2", y, train, pre)
    y = _unit(sd, name + ".conv3", y, train, pre)
    return y + x


def forward(sd, x, probes=None, train=False, pre=None):
    """yolo_fastest.py:150-218.  x: float32 [N,1,H,W].  Returns (head_large, head_small), NCHW.
    If `probes` is a dict it is filled with named intermediate activations (NCHW).
    train=True: the module in train mode (train.py:99, :114) -- BatchNorm on batch statistics, and the autograd graph is kept, so
    `torch.autograd.grad(loss, [sd[k] for k in parameter_keys(sd)])` is what `loss.backward()` (train.py:131) leaves in .grad."""
    def rec(name, t):
        if probes is not None:
            probes[name] = t
        return t

    with torch.enable_grad() if train else torch.no_grad():
        for n in ("conv0", "conv1_2", "conv1_3", "conv1_4"):
            x = rec(n, _unit(sd, n, x, train, pre))
        x = rec("res1_1", _resblock(sd, "res1_1", x, train, pre))
        for n in ("conv1_8", "conv1_9", "conv2_1"):
            x = rec(n, _unit(sd, n, x, train, pre))
        for n in ("res2_1", "res2_2"):
            x = rec(n, _resblock(sd, n, x, train, pre))
        for n in ("conv2_2", "conv2_3", "conv3_1"):
            x = rec(n, _unit(sd, n, x, train, pre))
        for n in ("res3_1", "res3_2"):
            x = rec(n, _resblock(sd, n, x, train, pre))
        for n in ("conv3_2", "conv3_3", "conv3_4"):
            x = rec(n, _unit(sd, n, x, train, pre))
        for n in ("res3_3", "res3_4", "res3_5", "res3_6"):
            x = rec(n, _resblock(sd, n, x, train, pre))
        for n in ("conv3_5", "conv3_6", "conv4_1"):
            x = rec(n, _unit(sd, n, x, train, pre))
        for n in ("res4_1", "res4_2", "res4_3", "res4_4"):
            x = rec(n, _resblock(sd, n, x, train, pre))
        conv4_2 = rec("conv4_2", _unit(sd, "conv4_2", x, train, pre))
        x = rec("conv4_3", _unit(sd, "conv4_3", conv4_2, train, pre))
        x = rec("conv5_1", _unit(sd, "conv5_1", x, train, pre))
        for n in ("res5_1", "res5_2", "res5_3", "res5_4", "res5_5"):
            x = rec(n, _resblock(sd, n, x, train, pre))
        conv5_2 = rec("conv5_2", _unit(sd, "conv5_2", x, train, pre))
        x = conv5_2
        for n in ("conv5_3", "conv5_4", "conv5_5", "conv5_6"):
            x = rec(n, _unit(sd, n, x, train, pre))
        head_small = F.conv2d(x, sd["head_5.weight"], sd["head_5.bias"])
        d = rec("deconv5_1", _unit(sd, "deconv5_1", conv5_2, train, pre))
        x = torch.cat((conv4_2, d), 1)  # yolo_fastest.py:209
        for n in ("conv4_1_1", "conv4_1_2", "conv4_1_3", "conv4_1_4", "conv4_1_5"):
            x = rec(n, _unit(sd, n, x, train, pre))
        head_large = F.conv2d(x, sd["head_4.weight"], sd["head_4.bias"])
    return head_large, head_small


def parameter_keys(sd):
    """The state-dict keys that are nn.Parameters, in `model.parameters()` order (what the optimizer of train.py:84 updates)."""
    return [k for k in sd if not k.endswith(("running_mean", "running_var", "num_batches_tracked"))]


def training_state(sd, dtype=torch.float32):
    """A state-dict copy ready for forward(train=True): parameters as leaves that require grad, buffers as plain tensors."""
    out = {}
    for k, v in sd.items():
        v = v.detach().clone()
        if v.is_floating_point():
            v = v.to(dtype)
        out[k] = v
    for k in parameter_keys(out):
        out[k].requires_grad_(True)
    return out


def preprocess(u8, input_channel=1):
    """src/detect.py:119-127: u8 [N,H,W] (or [H,W]) -> float32 [N,1,H,W], (x-128)/255; input_channel 3: u8 [N,H,W,3] as cv2.imread
    hands it over (BGR) -> `img[:, :, ::-1].transpose(2, 0, 1)` (:119) -> float32 [N,3,H,W]."""
    x = torch.as_tensor(u8)
    if input_channel != 1:
        if x.dim() == 3:
            x = x[None]
        x = x.flip(-1).permute(0, 3, 1, 2).to(torch.float32)
        return ((x - 128.0) / 255.0).contiguous()
    x = x.to(torch.float32)
    x = (x - 128.0) / 255.0
    if x.dim() == 2:
        x = x[None]
    return x[:, None].contiguous()


def load_state_dict(path):
    return torch.load(path, map_location="cpu")
