"""Weight packer: the reference's 508-key `.pth` state-dict -> BN-folded, NHWC-friendly flat blob
that `yf_create` (include/yolo_fastest_hip.h) consumes.

Reference facts this encodes (file:line under the reference repo):
  * state-dict layout, loaded at src/detect.py:90-91: for each of the 84 conv+BN units
    `<name>.0.weight` ([Cout,Cin/g,k,k]; the deconv is [Cin,Cout,2,2]) and
    `<name>.1.{weight,bias,running_mean,running_var,num_batches_tracked}`; the two heads are
    plain Conv2d with bias (`head_5.*`, `head_4.*`) -- src/model_training/model/yolo_fastest.py:78-148.
  * eval-mode BatchNorm2d (eps 1e-5, running stats): y = (x-mean)/sqrt(var+eps)*gamma + beta, so
    the fold is w' = w*gamma/sqrt(var+eps), b' = beta - mean*gamma/sqrt(var+eps).

Blob (little endian):
  header 64 B : magic "YFHIPW01", u32 version, n_layers, num_out, input_channel, num_anchors, num_cls,
                u64 data_floats
  table       : n_layers x 64 B : char name[32]; u32 kind, cin, cout, k, stride, relu, w_off, b_off
                (offsets in floats into the data section, 4-float aligned)
  data        : float32
Layouts:  pw/head W[cin][cout]; dw W[k*k][c]; dense kxk W[ky][kx][cin][cout];
          deconv2x2 W[dy][dx][cin][cout]; bias[cout].
"""
import struct

import numpy as np
import torch

MAGIC = b"YFHIPW01"
VERSION = 1
KIND_PW, KIND_DW, KIND_DENSE, KIND_DECONV, KIND_HEAD = 0, 1, 2, 3, 4
BN_EPS = 1e-5


def _res(name, c, e):
    return [(f"{name}.conv1", KIND_PW, c, e, 1, 1, 1), (f"{name}.conv2", KIND_DW, e, e, 3, 1, 1),
            (f"{name}.conv3", KIND_PW, e, c, 1, 1, 0)]


def layer_table(num_out=24, input_channel=1):
    """(name, kind, cin, cout, k, stride, relu) in module-definition order (yolo_fastest.py:78-148)."""
    t = [("conv0", KIND_DENSE, input_channel, 8, 3, 2, 1), ("conv1_2", KIND_PW, 8, 8, 1, 1, 1),
         ("conv1_3", KIND_DW, 8, 8, 3, 1, 1), ("conv1_4", KIND_PW, 8, 4, 1, 1, 0)]
    t += _res("res1_1", 4, 8)
    t += [("conv1_8", KIND_PW, 4, 24, 1, 1, 1), ("conv1_9", KIND_DENSE, 24, 24, 3, 2, 1),
          ("conv2_1", KIND_PW, 24, 8, 1, 1, 0)]
    t += _res("res2_1", 8, 32) + _res("res2_2", 8, 32)
    t += [("conv2_2", KIND_PW, 8, 32, 1, 1, 1), ("conv2_3", KIND_DW, 32, 32, 3, 2, 1),
          ("conv3_1", KIND_PW, 32, 8, 1, 1, 0)]
    t += _res("res3_1", 8, 48) + _res("res3_2", 8, 48)
    t += [("conv3_2", KIND_PW, 8, 48, 1, 1, 1), ("conv3_3", KIND_DW, 48, 48, 3, 1, 1),
          ("conv3_4", KIND_PW, 48, 16, 1, 1, 0)]
    t += _res("res3_3", 16, 96) + _res("res3_4", 16, 96) + _res("res3_5", 16, 96) + _res("res3_6", 16, 96)
    t += [("conv3_5", KIND_PW, 16, 96, 1, 1, 1), ("conv3_6", KIND_DW, 96, 96, 3, 2, 1),
          ("conv4_1", KIND_PW, 96, 24, 1, 1, 0)]
    t += _res("res4_1", 24, 136) + _res("res4_2", 24, 136) + _res("res4_3", 24, 136) + _res("res4_4", 24, 136)
    t += [("conv4_2", KIND_PW, 24, 136, 1, 1, 1), ("conv4_3", KIND_DW, 136, 136, 3, 2, 1),
          ("conv5_1", KIND_PW, 136, 48, 1, 1, 1)]
    for i in range(1, 6):
        t += _res(f"res5_{i}", 48, 224)
    t += [("conv5_2", KIND_PW, 48, 96, 1, 1, 1), ("conv5_3", KIND_DW, 96, 96, 5, 1, 1),
          ("conv5_4", KIND_PW, 96, 128, 1, 1, 0), ("conv5_5", KIND_DW, 128, 128, 5, 1, 1),
          ("conv5_6", KIND_PW, 128, 128, 1, 1, 0), ("head_5", KIND_HEAD, 128, num_out, 1, 1, 0),
          ("deconv5_1", KIND_DECONV, 96, 96, 2, 2, 1),
          ("conv4_1_1", KIND_PW, 232, 96, 1, 1, 1), ("conv4_1_2", KIND_DW, 96, 96, 5, 1, 1),
          ("conv4_1_3", KIND_PW, 96, 96, 1, 1, 0), ("conv4_1_4", KIND_DW, 96, 96, 5, 1, 1),
          ("conv4_1_5", KIND_PW, 96, 96, 1, 1, 0), ("head_4", KIND_HEAD, 96, num_out, 1, 1, 0)]
    return t


def expected_keys(num_out=24, input_channel=1):
    keys = []
    for name, kind, *_ in layer_table(num_out, input_channel):
        if kind == KIND_HEAD:
            keys += [name + ".weight", name + ".bias"]
        else:
            keys += [name + ".0.weight"] + [name + ".1." + s for s in
                                            ("weight", "bias", "running_mean", "running_var", "num_batches_tracked")]
    return keys


def fold_bn(sd, name):
    """Folded (w, b) of one conv+BN unit, computed in fp64 and rounded once to fp32."""
    w = sd[name + ".0.weight"].detach().to(torch.float64).cpu()
    g = sd[name + ".1.weight"].detach().to(torch.float64).cpu()
    beta = sd[name + ".1.bias"].detach().to(torch.float64).cpu()
    mean = sd[name + ".1.running_mean"].detach().to(torch.float64).cpu()
    var = sd[name + ".1.running_var"].detach().to(torch.float64).cpu()
    s = g / torch.sqrt(var + BN_EPS)
    return w, s, beta - mean * s


def pack_state_dict(sd, num_out=24, input_channel=1, num_anchors=3, num_cls=3, strict=True):
    """state-dict -> bytes.  strict=True mirrors load_state_dict's strict key check (detect.py:91)."""
    table = layer_table(num_out, input_channel)
    if strict:
        exp = expected_keys(num_out, input_channel)
        missing = [k for k in exp if k not in sd]
        unexpected = [k for k in sd.keys() if k not in set(exp)]
        if missing or unexpected:
            raise RuntimeError("Error(s) in loading state_dict for YoloFastest:\n\tMissing key(s): %s\n\t"
                               "Unexpected key(s): %s" % (missing, unexpected))
    chunks, rows, off = [], [], 0

    def push(a):
        nonlocal off
        a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1)
        pad = (-a.size) % 4
        if pad:
            a = np.concatenate([a, np.zeros(pad, np.float32)])
        chunks.append(a)
        o = off
        off += a.size
        return o

    for name, kind, cin, cout, k, stride, relu in table:
        if kind == KIND_HEAD:
            w = sd[name + ".weight"].detach().to(torch.float64).cpu()
            b = sd[name + ".bias"].detach().to(torch.float64).cpu()
            assert tuple(w.shape) == (cout, cin, 1, 1), (name, w.shape)
            wl = w[:, :, 0, 0].t()  # [cin][cout]
        else:
            w, s, b = fold_bn(sd, name)
            if kind == KIND_PW:
                assert tuple(w.shape) == (cout, cin, 1, 1), (name, w.shape)
                wl = (w[:, :, 0, 0] * s[:, None]).t()  # [cin][cout]
            elif kind == KIND_DW:
                assert tuple(w.shape) == (cout, 1, k, k), (name, w.shape)
                wl = (w[:, 0] * s[:, None, None]).permute(1, 2, 0).reshape(k * k, cout)  # [ky*k+kx][c]
            elif kind == KIND_DENSE:
                assert tuple(w.shape) == (cout, cin, k, k), (name, w.shape)
                wl = (w * s[:, None, None, None]).permute(2, 3, 1, 0)  # [ky][kx][cin][cout]
            elif kind == KIND_DECONV:
                assert tuple(w.shape) == (cin, cout, 2, 2), (name, w.shape)  # ConvTranspose2d: [in,out,kh,kw]
                wl = (w * s[None, :, None, None]).permute(2, 3, 0, 1)  # [dy][dx][cin][cout]
            else:
                raise AssertionError(kind)
        w_off = push(wl.numpy())
        b_off = push(b.numpy())
        nm = name.encode()
        assert len(nm) < 32
        rows.append(struct.pack("<32s8I", nm, kind, cin, cout, k, stride, relu, w_off, b_off))
    header = struct.pack("<8s6IQ", MAGIC, VERSION, len(table), num_out, input_channel, num_anchors, num_cls, off)
    header += b"\0" * (64 - len(header))
    return header + b"".join(rows) + np.concatenate(chunks).tobytes()


def unpack(blob):
    """Inverse of the container (not of the fold): {name: dict(kind,cin,cout,k,stride,relu,w,b)} -- for tests."""
    magic, ver, n, num_out, inch, na, nc, nfl = struct.unpack_from("<8s6IQ", blob, 0)
    assert magic == MAGIC and ver == VERSION
    data = np.frombuffer(blob, np.float32, nfl, 64 + 64 * n)
    out = {}
    sizes = {KIND_PW: lambda ci, co, k: ci * co, KIND_HEAD: lambda ci, co, k: ci * co,
             KIND_DW: lambda ci, co, k: k * k * co, KIND_DENSE: lambda ci, co, k: k * k * ci * co,
             KIND_DECONV: lambda ci, co, k: 4 * ci * co}
    for i in range(n):
        nm, kind, cin, cout, k, stride, relu, w_off, b_off = struct.unpack_from("<32s8I", blob, 64 + 64 * i)
        nm = nm.rstrip(b"\0").decode()
        out[nm] = dict(kind=kind, cin=cin, cout=cout, k=k, stride=stride, relu=relu,
                       w=data[w_off:w_off + sizes[kind](cin, cout, k)], b=data[b_off:b_off + cout])
    return out
