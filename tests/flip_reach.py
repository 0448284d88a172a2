"""Which parameter-gradient elements can a ReLU decision within fp32 rounding of zero reach?  (test helper, CPU; uses oracle/)

A pre-activation z (BatchNorm output in front of a ReLU, yolo_fastest.py:16-26) that lies within a few fp32 ulps of zero may come out on
the other side in another fp32 evaluation of the same graph: the mask element ReLU'(z) flips, and the gradient that flows (or does not)
through it differs by O(1) at that element.  With train-mode BatchNorm between every conv and its ReLU the change does not stay local:
BatchNorm's backward couples all pixels of the channel over the whole batch, so one flipped element of layer L, channel c reaches
  * layer L's own parameters of channel c: the conv's filter(s) for output channel c, gamma_c, beta_c;
  * EVERY parameter of every layer upstream of L (on a path from the input to L).
It cannot reach layers downstream of L, the other channels' parameters of L, or the other detection head's private layers."""
import numpy as np
import torch

from oracle import backbone_oracle as bo

_SMALL = ("conv5_3", "conv5_4", "conv5_5", "conv5_6")
_LARGE = ("deconv5_1", "conv4_1_1", "conv4_1_2", "conv4_1_3", "conv4_1_4", "conv4_1_5")


def upstream(name):
    """Layer names on a path from the input to `name` (exclusive), in bo.LAYERS order."""
    order = [l[0] for l in bo.LAYERS]
    i = order.index(name)
    before = order[:i]
    if name in _LARGE:
        before = [n for n in before if n not in _SMALL]
    return before


def near_zero_preactivations(sd64, x64, ulps=4.0):
    """fp64 train-mode forward of the oracle; returns {layer: sorted channel list} of the ReLU layers that hold a pre-activation with
    |z| < ulps * 2^-23 * A_c, A_c = |beta_c| + |gamma_c| (|mu_c| + |y|max_c) / sigma_c: the magnitude of the terms whose difference z is
    (z = gamma (y - mu) / sigma + beta), i.e. what an fp32 evaluation rounds at.  Also the total count."""
    pre = {}
    bo.forward(sd64, x64, train=True, pre=pre)
    out, total = {}, 0
    for name, d in pre.items():
        g, b = sd64[name + ".1.weight"].detach().abs(), sd64[name + ".1.bias"].detach().abs()
        A = b + g * (d["mu"].abs() + d["absmax"]) / torch.sqrt(d["sigma"] ** 2 + bo.BN_EPS)
        hit = d["z"].abs() < (ulps * 2.0 ** -23) * A[None, :, None, None]
        ch = torch.nonzero(hit.any(0).any(-1).any(-1)).ravel().tolist()
        if ch:
            out[name] = ch
            total += int(hit.sum())
    return out, total


def reach_masks(flips, param_names, param_shapes, parts=False):
    """flips: {layer: channels}.  Returns one boolean array per parameter tensor (flattened, `model.parameters()` order): True where a
    flipped ReLU decision can reach the gradient element.  parts=True: a pair per tensor instead -- (the flipped layers' OWN channel
    parameters, where one mask element is an O(1) share of the gradient; everything merely upstream of a flip)."""
    full = set()
    for name in flips:
        full.update(upstream(name))
    masks = []
    for pname, shape in zip(param_names, param_shapes):
        layer = pname.rsplit(".", 2)[0] if pname.count(".") >= 2 else pname.rsplit(".", 1)[0]
        m = np.zeros(shape, bool)
        if parts:
            own = np.zeros(shape, bool)
            if layer in flips:
                kind = bo._BY_NAME[layer][1]
                for c in flips[layer]:
                    if kind == "dc" and pname.endswith(".0.weight"):
                        own[:, c] = True
                    else:
                        own[c] = True
            up = np.zeros(shape, bool)
            if layer in full:
                up[...] = True
                up &= ~own
            masks.append((own.ravel(), up.ravel()))
            continue
        if layer in full:
            m[...] = True
        elif layer in flips:
            kind = bo._BY_NAME[layer][1]
            for c in flips[layer]:
                if kind == "dc" and pname.endswith(".0.weight"):
                    m[:, c] = True          # ConvTranspose2d weight is [in, out, kh, kw]
                else:
                    m[c] = True             # conv weight [out, ...], BatchNorm gamma / beta [out]
        masks.append(m.ravel())
    return masks
