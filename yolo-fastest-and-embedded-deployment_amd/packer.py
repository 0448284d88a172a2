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


# ---------------------------------------------------------------------------------------------------------------------
# Alternative weight source: the reference's shipped ncnn model (models/ncnn/**/*.param + *.bin), SURVEY.md 8(f).3.
# The .param is the BN-folded graph ncnnoptimize wrote (58 Convolution, 27 ConvolutionDepthWise, 1 Deconvolution, blobs
# data -> head_large / head_small); the .bin holds, per weighted layer in file order, a 4-byte storage flag (0 = raw
# fp32), weight_data_size floats in [out][in/g][kh][kw] order and, with bias_term, num_output bias floats.
# ---------------------------------------------------------------------------------------------------------------------
_NCNN_MAGIC = "7767517"


def read_ncnn(param_path, bin_path):
    """-> list of dicts (type, name, num_output, kernel, stride, pad, group, relu, w, b) for the weighted layers, in
    file order (= execution order of the reference's forward)."""
    with open(param_path) as f:
        lines = [ln.split() for ln in f.read().splitlines() if ln.strip()]
    if lines[0][0] != _NCNN_MAGIC:
        raise ValueError("not an ncnn .param file (magic %s)" % lines[0][0])
    raw = np.fromfile(bin_path, dtype=np.uint8)
    pos, out = 0, []
    for tok in lines[2:]:
        ltype, name, nin, nout = tok[0], tok[1], int(tok[2]), int(tok[3])
        if ltype not in ("Convolution", "ConvolutionDepthWise", "Deconvolution"):
            continue
        kv = dict(t.split("=") for t in tok[4 + nin + nout:])
        p = {int(k): v for k, v in kv.items()}
        n_out, k, wsize = int(p[0]), int(p.get(1, 1)), int(p[6])
        flag = int(np.frombuffer(raw, np.uint32, 1, pos)[0]); pos += 4
        if flag != 0:
            raise ValueError("layer %s: only raw fp32 ncnn weights are supported (storage flag 0x%08x)" % (name, flag))
        w = np.frombuffer(raw, np.float32, wsize, pos).copy(); pos += 4 * wsize
        b = np.zeros(n_out, np.float32)
        if int(p.get(5, 0)):
            b = np.frombuffer(raw, np.float32, n_out, pos).copy(); pos += 4 * n_out
        out.append(dict(type=ltype, name=name, num_output=n_out, kernel=k, stride=int(p.get(3, 1)), pad=int(p.get(4, 0)),
                        group=int(p.get(7, 1)), relu=int(p.get(9, 0)) == 1, w=w, b=b))
    if pos != raw.size:
        raise ValueError("ncnn .bin has %d trailing bytes: the .param does not describe it" % (raw.size - pos))
    return out


# ---- ONNX (models/onnx/**: the reference's torch.onnx export, BN not folded, raw fp32 initializers) -------------------------
# A minimal protobuf wire-format reader: the `onnx` package is not a dependency.  Field numbers from onnx.proto3:
# ModelProto.graph = 7; GraphProto.node = 1, .initializer = 5; NodeProto.input = 1, .output = 2, .op_type = 4, .attribute = 5;
# AttributeProto.name = 1, .f = 2, .i = 3, .ints = 8; TensorProto.dims = 1, .data_type = 2, .float_data = 4, .name = 8, .raw_data = 9.

def _pb_varint(b, i):
    r = s = 0
    while True:
        c = b[i]; i += 1
        r |= (c & 0x7F) << s; s += 7
        if not c & 0x80:
            return r, i


def _pb_fields(b):
    i, n = 0, len(b)
    while i < n:
        key, i = _pb_varint(b, i)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _pb_varint(b, i)
        elif wt == 1:
            v = b[i:i + 8]; i += 8
        elif wt == 2:
            ln, i = _pb_varint(b, i)
            v = b[i:i + ln]; i += ln
        elif wt == 5:
            v = b[i:i + 4]; i += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield f, wt, v


def _pb_ints(wt, v):
    if wt == 0:
        return [v]
    out, j = [], 0
    while j < len(v):
        d, j = _pb_varint(v, j)
        out.append(d)
    return out


def _onnx_graph(path):
    with open(path, "rb") as f:
        data = f.read()
    graph = None
    for f_, wt, v in _pb_fields(data):
        if f_ == 7 and wt == 2:
            graph = v
    if graph is None:
        raise ValueError("not an ONNX ModelProto (no graph)")
    nodes, inits = [], {}
    for f_, wt, v in _pb_fields(graph):
        if f_ == 1:
            node = {"op": None, "in": [], "out": [], "attr": {}}
            for f2, w2, v2 in _pb_fields(v):
                if f2 == 1: node["in"].append(bytes(v2).decode())
                elif f2 == 2: node["out"].append(bytes(v2).decode())
                elif f2 == 4: node["op"] = bytes(v2).decode()
                elif f2 == 5:
                    name, val = None, None
                    for f3, w3, v3 in _pb_fields(v2):
                        if f3 == 1: name = bytes(v3).decode()
                        elif f3 == 2: val = struct.unpack("<f", v3)[0]
                        elif f3 == 3: val = v3
                        elif f3 == 8: val = (val or []) + _pb_ints(w3, v3)
                    node["attr"][name] = val
            nodes.append(node)
        elif f_ == 5:
            name, dims, dt, arr = None, [], None, None
            for f2, w2, v2 in _pb_fields(v):
                if f2 == 1: dims += _pb_ints(w2, v2)
                elif f2 == 2: dt = v2
                elif f2 == 8: name = bytes(v2).decode()
                elif f2 == 9: arr = np.frombuffer(bytes(v2), np.float32) if dt in (None, 1) else bytes(v2)
                elif f2 == 4: arr = np.frombuffer(bytes(v2), np.float32) if w2 == 2 else np.append(arr if arr is not None else [], struct.unpack("<f", v2)[0])
            if dt == 1 and arr is not None:
                inits[name] = np.asarray(arr, np.float32).reshape(dims).copy()
    return nodes, inits


def read_onnx(path, num_out=24, input_channel=1):
    """ONNX file -> the reference's 508-key state-dict (torch tensors), so that `model.load_state_dict(read_onnx(p))` is the
    ONNX twin of `torch.load(pth)`.  The Conv / ConvTranspose nodes are matched to the YoloFastest layer table by order
    (the export traces forward(), the table is in forward order) and checked: weight shape, stride, group, a
    BatchNormalization on every non-head layer (epsilon must be the 1e-5 this package folds with), Relu where the table has one."""
    nodes, inits = _onnx_graph(path)
    table = layer_table(num_out, input_channel)
    consumers = {}
    for n in nodes:
        for i in n["in"]:
            consumers.setdefault(i, []).append(n)
    convs = [n for n in nodes if n["op"] in ("Conv", "ConvTranspose")]
    if len(convs) != len(table):
        raise RuntimeError("ONNX model has %d convolutions, YoloFastest has %d" % (len(convs), len(table)))
    sd = {}
    for (name, kind, cin, cout, k, stride, relu), n in zip(table, convs):
        w = inits.get(n["in"][1])
        if w is None:
            raise RuntimeError("ONNX conv for %s: weight %s is not an fp32 initializer" % (name, n["in"][1]))
        group = int(n["attr"].get("group", 1) or 1)
        strides = n["attr"].get("strides", [1, 1])
        want = {KIND_PW: (cout, cin, 1, 1), KIND_HEAD: (cout, cin, 1, 1), KIND_DENSE: (cout, cin, k, k), KIND_DW: (cout, 1, k, k),
                KIND_DECONV: (cin, cout, 2, 2)}[kind]
        ok = (tuple(w.shape) == want and list(strides) == [stride, stride] and group == (cout if kind == KIND_DW else 1)
              and (n["op"] == "ConvTranspose") == (kind == KIND_DECONV))
        if not ok:
            raise RuntimeError("ONNX node %s %s (weight %s, strides %s, group %d) does not match YoloFastest layer %s"
                               % (n["op"], n["out"][0], w.shape, strides, group, name))
        out, nxt = n["out"][0], consumers.get(n["out"][0], [])
        if kind == KIND_HEAD:
            if len(n["in"]) < 3:
                raise RuntimeError("ONNX head conv %s has no bias" % name)
            sd[name + ".weight"] = torch.from_numpy(w.copy())
            sd[name + ".bias"] = torch.from_numpy(inits[n["in"][2]].copy())
            continue
        if len(nxt) != 1 or nxt[0]["op"] != "BatchNormalization":
            raise RuntimeError("ONNX layer %s: expected a BatchNormalization after the convolution (an export with folded "
                               "BN is not the reference's)" % name)
        bn = nxt[0]
        eps = bn["attr"].get("epsilon", 1e-5)
        if abs(eps - BN_EPS) > 1e-12:
            raise RuntimeError("ONNX layer %s: BatchNormalization epsilon %g, expected %g" % (name, eps, BN_EPS))
        has_relu = any(c["op"] == "Relu" for c in consumers.get(bn["out"][0], []))
        if has_relu != bool(relu):
            raise RuntimeError("ONNX layer %s: Relu %s, the YoloFastest layer has relu=%d" % (name, has_relu, relu))
        sd[name + ".0.weight"] = torch.from_numpy(w.copy())
        for key, src in zip(("weight", "bias", "running_mean", "running_var"), bn["in"][1:5]):
            sd[name + ".1." + key] = torch.from_numpy(inits[src].copy())
        sd[name + ".1.num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
    return sd


def pack_ncnn(param_path, bin_path, num_out=24, input_channel=1, num_anchors=3, num_cls=3):
    """ncnn .param/.bin -> the same blob pack_state_dict() produces (the weights are already BN-folded).
    The weighted layers are matched to the YoloFastest layer table by order and checked shape by shape."""
    table = layer_table(num_out, input_channel)
    layers = read_ncnn(param_path, bin_path)
    if len(layers) != len(table):
        raise RuntimeError("ncnn model has %d weighted layers, YoloFastest has %d" % (len(layers), len(table)))
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

    want_type = {KIND_PW: "Convolution", KIND_HEAD: "Convolution", KIND_DENSE: "Convolution",
                 KIND_DW: "ConvolutionDepthWise", KIND_DECONV: "Deconvolution"}
    for (name, kind, cin, cout, k, stride, relu), L in zip(table, layers):
        exp_w = k * k * cout if kind == KIND_DW else k * k * cin * cout
        if (L["type"] != want_type[kind] or L["num_output"] != cout or L["kernel"] != k or L["stride"] != stride
                or L["w"].size != exp_w or L["relu"] != bool(relu) or (kind == KIND_DW and L["group"] != cout)):
            raise RuntimeError("ncnn layer %s (%s, %d out, k%d s%d, %d weights) does not match YoloFastest layer %s"
                               % (L["name"], L["type"], L["num_output"], L["kernel"], L["stride"], L["w"].size, name))
        w = L["w"]
        if kind in (KIND_PW, KIND_HEAD):
            wl = w.reshape(cout, cin).T                                   # [cin][cout]
        elif kind == KIND_DW:
            wl = w.reshape(cout, k * k).T                                 # [ky*k+kx][c]
        elif kind == KIND_DENSE:
            wl = w.reshape(cout, cin, k, k).transpose(2, 3, 1, 0)         # [ky][kx][cin][cout]
        else:  # ncnn Deconvolution weights: [out][in][kh][kw]
            wl = w.reshape(cout, cin, 2, 2).transpose(2, 3, 1, 0)         # [dy][dx][cin][cout]
        w_off = push(wl)
        b_off = push(L["b"])
        rows.append(struct.pack("<32s8I", name.encode(), kind, cin, cout, k, stride, relu, w_off, b_off))
    header = struct.pack("<8s6IQ", MAGIC, VERSION, len(table), num_out, input_channel, num_anchors, num_cls, off)
    header += b"\0" * (64 - len(header))
    return header + b"".join(rows) + np.concatenate(chunks).tobytes()
