"""Shared by the CPU oracle tests and the GPU tests of the io_params generality: the configurations of
tests/golden/seeded_weights.IO_CONFIGS, their io_params dicts and their numpy-seeded state-dicts (golden_io.npz holds what the
REFERENCE computed for exactly these weights and inputs)."""
import copy
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from seeded_weights import IO_CONFIGS, io_inputs, io_targets, seeded_state_dict  # noqa: E402,F401

CONFIG = {tag: (C, Cin, A) for tag, C, Cin, A in IO_CONFIGS}
TAGS = [c[0] for c in IO_CONFIGS]


def io_for(tag, H=256, W=320):
    import yolo_fastest_amd as yf
    C, Cin, A = CONFIG[tag]
    io = copy.deepcopy(yf.config_params["io_params"])
    io.update(num_cls=C, input_channel=Cin, num_anchors=A, input_shape=[H, W, Cin], origin_img_shape=[H, W, Cin],
              anchors=[grp[:A] for grp in yf.config_params["io_params"]["anchors"]], class_names=["c%d" % i for i in range(C)])
    return io


def state_dict_for(tag, seed):
    """The state-dict make_golden.py loaded into the reference module: keys, order and shapes from THIS package's module (a strict
    load into the reference's module succeeded with the same key set), values from the numpy stream of `seed`."""
    import yolo_fastest_amd as yf
    m = yf.YoloFastest(io_for(tag))
    shapes = OrderedDict((k, tuple(v.shape)) for k, v in m.state_dict().items())
    return OrderedDict((k, torch.from_numpy(np.asarray(v))) for k, v in seeded_state_dict(shapes, seed).items())


def unpack_lists(g, prefix, f):
    """frame f of pack_lists' arrays -> dict of trimmed arrays (count -2: the reference raised ZeroDivisionError)."""
    n = int(g[prefix + "_count"][f])
    m = max(n, 0)
    return dict(count=n, box=g[prefix + "_box"][f, :m], conf=g[prefix + "_conf"][f, :m], score=g[prefix + "_score"][f, :m],
                cls=g[prefix + "_cls"][f, :m], src=g[prefix + "_src"][f, :m])
