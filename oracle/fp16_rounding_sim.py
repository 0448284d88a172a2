"""TEST INFRASTRUCTURE (never imported by the product): CPU simulation of the rounding points of an fp16-storage evaluation of the
YoloFastest graph (yolo_fastest.py:150-218), used by tests/test_oracle_golden.py::test_fp16_rounding_alone_exceeds_the_surveys_tolerance
and tools/fp16_sim.py to show what SURVEY.md 8(d).3's "activations/weights fp16, fp32 accumulate" costs in logit error whoever
implements it.

Folded-BN forward in fp32 (torch CPU) with optional fp16 rounding of
  W  the pointwise / dense weights (MFMA operands),
  T  the tensors that live in HBM between launches of the fused plan (the narrow residual trunk + head-branch tensors),
  A  the activation operand of every pointwise MFMA (block input as the expansion's operand, depthwise result as the
     projection's operand),
  E  the expanded tensor kept in LDS between expansion and depthwise.
"""
import torch
import torch.nn.functional as F

from oracle import backbone_oracle as bo


def fold(sd):
    out = {}
    for name, kind, cin, cout, k, s, relu in bo.LAYERS:
        w = sd[name + ".0.weight"].double()
        sc = sd[name + ".1.weight"].double() / torch.sqrt(sd[name + ".1.running_var"].double() + 1e-5)
        b = sd[name + ".1.bias"].double() - sd[name + ".1.running_mean"].double() * sc
        w = w * (sc[None, :, None, None] if kind == "dc" else sc[:, None, None, None])
        out[name] = (w.float(), b.float())
    for h in ("head_5", "head_4"):
        out[h] = (sd[h + ".weight"].float(), sd[h + ".bias"].float())
    return out


def q(t):
    return t.half().float()


class Sim:
    # layers of the VALU block kernels (fp32 weights through the scalar path, fp32 arithmetic): never W / A rounded
    VALU = ("conv0", "conv1_2", "conv1_3", "conv1_4", "res1_1.conv1", "res1_1.conv2", "res1_1.conv3", "res2_1.conv1", "res2_1.conv2",
            "res2_1.conv3", "res2_2.conv1", "res2_2.conv2", "res2_2.conv3")

    def __init__(self, fw, W=False, T=False, A=False, E=False, split_w=False, split_a=False, valu_exact=True):
        self.fw, self.W, self.T, self.A, self.E, self.split_w, self.split_a = fw, W, T, A, E, split_w, split_a
        self.valu_exact = valu_exact

    def unit(self, name, x, store=False, is_exp=False):
        _, kind, cin, cout, k, s, relu = bo._BY_NAME[name]
        w, b = self.fw[name]
        mfma = kind in ("c", "dc") and not (self.valu_exact and name in self.VALU)
        if mfma:
            if self.W and not self.split_w:
                w = q(w)
            if self.A and not self.split_a:
                x = q(x)
        if kind == "dc":
            y = F.conv_transpose2d(x, w, b, stride=2)
        else:
            y = F.conv2d(x, w, b, stride=s, padding=(k - 1) // 2, groups=(cin if kind == "dw" else 1))
        if relu:
            y = F.relu(y)
        if is_exp and self.E and not (self.valu_exact and name in self.VALU):
            y = q(y)
        if store and self.T:
            y = q(y)
        return y

    def block(self, a, b, c, x, res):
        y = self.unit(a, x, is_exp=True)
        y = self.unit(b, y)
        y = self.unit(c, y)
        if res:
            y = y + x
        return q(y) if self.T else y

    def head(self, name, x):
        w, b = self.fw[name]
        if self.W and not self.split_w:
            w = q(w)
        if self.A and not self.split_a:
            x = q(x)
        return F.conv2d(x, w, b)

    def forward(self, x):
        with torch.no_grad():
            x = self.unit("conv0", x)
            x = self.block("conv1_2", "conv1_3", "conv1_4", x, False)
            x = self.block("res1_1.conv1", "res1_1.conv2", "res1_1.conv3", x, True)
            x = self.block("conv1_8", "conv1_9", "conv2_1", x, False)
            for n in ("res2_1", "res2_2"):
                x = self.block(n + ".conv1", n + ".conv2", n + ".conv3", x, True)
            x = self.block("conv2_2", "conv2_3", "conv3_1", x, False)
            for n in ("res3_1", "res3_2"):
                x = self.block(n + ".conv1", n + ".conv2", n + ".conv3", x, True)
            x = self.block("conv3_2", "conv3_3", "conv3_4", x, False)
            for n in ("res3_3", "res3_4", "res3_5", "res3_6"):
                x = self.block(n + ".conv1", n + ".conv2", n + ".conv3", x, True)
            x = self.block("conv3_5", "conv3_6", "conv4_1", x, False)
            for n in ("res4_1", "res4_2", "res4_3", "res4_4"):
                x = self.block(n + ".conv1", n + ".conv2", n + ".conv3", x, True)
            c42 = self.unit("conv4_2", x, store=True, is_exp=True)
            x = self.unit("conv4_3", c42)
            x = self.unit("conv5_1", x, store=True)
            for n in ("res5_1", "res5_2", "res5_3", "res5_4", "res5_5"):
                x = self.block(n + ".conv1", n + ".conv2", n + ".conv3", x, True)
            c52 = self.unit("conv5_2", x, store=True)
            x = self.unit("conv5_3", c52)
            x = self.unit("conv5_4", x, store=True)
            x = self.unit("conv5_5", x)
            x = self.unit("conv5_6", x)
            hs = self.head("head_5", x)
            d = self.unit("deconv5_1", c52, store=True)
            x = self.unit("conv4_1_1", torch.cat((c42, d), 1), store=True)
            x = self.unit("conv4_1_2", x)
            x = self.unit("conv4_1_3", x, store=True)
            x = self.unit("conv4_1_4", x)
            x = self.unit("conv4_1_5", x)
            hl = self.head("head_4", x)
        return hl, hs
