"""Seeded random YoloFastest state-dicts, identical wherever numpy runs (PCG64 streams do not depend on the torch build).

Used by tests/golden/make_golden.py (which feeds them to the REFERENCE module in the build container and records its
outputs in golden_io.npz) and by the tests (which feed the same weights to the HIP engine / the oracle on the GPU box):
the golden file then only has to hold the seed, not 1.4 MB of weights per configuration.

`shapes` is an ordered {state-dict key: shape} -- taken from the module's own state_dict(), so the reference's module (in
make_golden.py) and this package's module (in the tests) must agree on keys, order and shapes for the same tensors to come out.
Scales keep the 86-layer chain in range (He-style, fan-in of ONE output element) and put the BatchNorm statistics away from
the identity.
"""
from collections import OrderedDict

import numpy as np


def seeded_state_dict(shapes, seed):
    g = np.random.default_rng(seed)
    out = OrderedDict()
    for key, shape in shapes.items():
        shape = tuple(int(s) for s in shape)
        if key.endswith("num_batches_tracked"):
            out[key] = np.zeros((), np.int64)
        elif key.startswith("head_") and key.endswith(".weight"):
            # logits of std ~0.6: about half of the cells pass conf > 0.5, and exp(t_w) stays above ~0.15 so that zero-area boxes
            # (the reference's ZeroDivisionError, detect.py:39) stay rare
            out[key] = (g.standard_normal(shape) * (0.16 / shape[1]) ** 0.5).astype(np.float32)
        elif key.endswith(".0.weight"):
            if key.startswith("deconv"):
                fan = shape[0]                          # ConvTranspose2d [Cin, Cout, 2, 2], stride 2: one tap x Cin per output element
            else:
                fan = shape[1] * shape[2] * shape[3]    # Conv2d [Cout, Cin / groups, k, k]
            out[key] = (g.standard_normal(shape) * (1.0 / fan) ** 0.5).astype(np.float32)
        elif key.startswith("head_") and key.endswith(".bias"):
            out[key] = (g.standard_normal(shape) * 0.3).astype(np.float32)
        elif key.endswith(".1.weight"):
            out[key] = (0.5 + g.random(shape)).astype(np.float32)
        elif key.endswith(".1.bias") or key.endswith("running_mean"):
            out[key] = (g.standard_normal(shape) * 0.2).astype(np.float32)
        elif key.endswith("running_var"):
            out[key] = (0.5 + g.random(shape)).astype(np.float32)
        else:
            raise KeyError("unexpected state-dict key %r" % key)
    return out


# (tag, num_cls, input_channel, num_anchors): the io_params the reference's constructors are parameterised on
# (yolo_fastest.py:72-78,138,148; detect.py:15-21,53-66; yolo_loss.py:28-33,58-60)
IO_CONFIGS = [("c1", 1, 1, 3), ("c5rgb", 5, 3, 3), ("c20", 20, 1, 3), ("a2", 3, 1, 2), ("c80rgb", 80, 3, 3), ("ch2", 3, 2, 3),
              ("ch4", 2, 4, 3), ("ch6", 3, 6, 3)]      # ch6 (round 5): more input channels than the fused stem kernel takes


def io_inputs(tag, input_channel, n=2, H=256, W=320):
    """u8 frames as cv2 hands them over: [n,H,W] gray or [n,H,W,C] (C = 3: BGR)."""
    g = np.random.default_rng(sum(ord(c) for c in tag) * 7919 + 1)
    shape = (n, H, W) if input_channel == 1 else (n, H, W, input_channel)
    return g.integers(0, 256, size=shape, dtype=np.uint8)


def io_targets(tag, num_cls, n, T=8):
    """[n,T,6] training / validation targets (x, y, w, h normalised, class, marker), a few per image, the list ended by marker 0."""
    g = np.random.default_rng(sum(ord(c) for c in tag) * 104729 + 3)
    t = np.zeros((n, T, 6), np.float32)
    for i in range(n):
        k = int(g.integers(1, T))
        t[i, :k, 0:2] = g.uniform(0.05, 0.95, (k, 2))
        t[i, :k, 2:4] = g.uniform(0.03, 0.4, (k, 2))
        t[i, :k, 4] = g.integers(0, num_cls, k)
        t[i, :k, 5] = 255.0
    return t
