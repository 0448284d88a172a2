#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference); the GPU box never executes
this script -- it consumes the .npz files this script wrote.  Nothing of the reference's
source text is stored: only inputs and the outputs the reference computed for them.

What is executed from the reference:
  * the model:  src/model_training/model/yolo_fastest.py  (class YoloFastest, imported),
    with the shipped checkpoints models/pytorch/{256x320,512x640}/*.pth;
  * the post-process: class YOLO_post_process of src/detect.py:14-84, obtained by AST
    extraction (detect.py itself cannot be imported: `import cv2` at :2 and
    model_training.train at :10 need packages this image lacks) and exec'd with the
    module-level global `device` it reads at :44 injected;
  * the 15 glue lines of Detect_YOLO.batch_detect (src/detect.py:157-169: bucket by class,
    stable sort by conf descending, NMS per class, concatenate in class order) and
    __adjust_coord (:131-139) are methods of a class that needs cv2, so they are restated
    here around calls into the extracted class.

Image decode caveat: cv2 is absent, so the 20 JPEGs of test_data/ (mode L, 640x512) are
decoded with PIL; for the 256x320 model the 2x downscale uses the 2x2 box mean
(a+b+c+d+2)>>2 that OpenCV's INTER_LINEAR takes at an exact 2x ratio.  Goldens therefore
START at the pre-processed u8 tensor, which is stored.
"""
import ast
import math
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REF, "src"))

from model_training.model.yolo_fastest import YoloFastest  # noqa: E402  (the reference)
from model_training._config import config_params  # noqa: E402


def extract_post_process():
    src = open(os.path.join(REF, "src", "detect.py"), encoding="utf-8").read()
    tree = ast.parse(src)
    node = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "YOLO_post_process"][0]
    ns = {"math": math, "np": np, "device": "cpu"}
    exec(compile(ast.Module(body=[node], type_ignores=[]), "detect.py<YOLO_post_process>", "exec"), ns)
    return ns["YOLO_post_process"]


YOLO_post_process = extract_post_process()


def io_params_for(res):
    io = dict(config_params["io_params"])
    if res == 512:  # _config.py:9 -- 512x640 uses anchor groups 1..2, no resize
        io["input_shape"] = [512, 640, 1]
        io["anchors"] = io["anchors"][1:]
    return io


def load_model(res):
    path = {256: "models/pytorch/256x320/YOLO-Fastest_epoch_28.pth",
            512: "models/pytorch/512x640/YOLO-Fastest_epoch_27.pth"}[res]
    io = io_params_for(res)
    m = YoloFastest(io).eval()
    msg = m.load_state_dict(torch.load(os.path.join(REF, path), map_location="cpu"))
    assert str(msg) == "<All keys matched successfully>"
    return m, io


def preprocess_u8(path, res):
    from PIL import Image
    a = np.asarray(Image.open(path))
    assert a.shape == (512, 640) and a.dtype == np.uint8
    if res == 256:
        a = a.astype(np.uint16)
        a = ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    return np.ascontiguousarray(a)


def to_input(u8):
    x = torch.from_numpy(u8.astype(np.float32))
    x = (x - 128.0) / 255.0  # detect.py:124
    while x.dim() < 4:
        x = x.unsqueeze(0)
    return x


def src_indices(pred, conf_thres):
    """Flat (head, anchor, i, j) index of every candidate, in the reference's decode order
    (detect.py:54-58): the same Python-double sigmoid and strict > as the reference."""
    out, base = [], 0
    for ph in pred:
        a = ph.numpy()[0]
        h, w = a.shape[1], a.shape[2]
        a = a.reshape(3, 8, h, w)
        for pp in range(3):
            for i in range(h):
                for j in range(w):
                    if 1. / (1. + math.exp(-a[pp, 4, i, j])) > conf_thres:
                        out.append(base + (pp * h + i) * w + j)
        base += 3 * h * w
    return out


def reference_post(pred, io, adjust):
    """detect.py:157-169 (+ :181-182) around the extracted reference class."""
    pp = YOLO_post_process(conf_thres=io["conf_thre"], nms_thres=io["nms_thre"], num_anchors=io["num_anchors"],
                           num_class=io["num_cls"], anchors=io["anchors"], input_shape=io["input_shape"])
    cands = pp.decode_box(pred)
    srcs = src_indices(pred, io["conf_thre"])
    assert len(srcs) == len(cands)
    for c, s in zip(cands, srcs):
        c.append(s)  # rides along; the reference only touches [0..6]
    cand_copy = [list(c) for c in cands]
    buckets = [[] for _ in range(io["num_cls"])]
    for b in cands:
        buckets[b[6]].append(b)
    final = []
    for cls in range(io["num_cls"]):
        if not buckets[cls]:
            continue
        buckets[cls].sort(key=lambda it: it[4], reverse=True)
        final.extend(pp.non_maxium_supression(buckets[cls]))
    final = [list(f) for f in final]
    adj = [list(f) for f in final]
    if adjust:
        sh = io["origin_img_shape"][0] / io["input_shape"][0]
        sw = io["origin_img_shape"][1] / io["input_shape"][1]
        for f in adj:
            f[0] = round(f[0] * sw); f[2] = round(f[2] * sw)
            f[1] = round(f[1] * sh); f[3] = round(f[3] * sh)
    return cand_copy, final, adj


def pack_lists(lists, kmax):
    n = len(lists)
    box = np.zeros((n, kmax, 4), np.int32)
    conf = np.zeros((n, kmax), np.float64)
    score = np.zeros((n, kmax), np.float64)
    cls = np.full((n, kmax), -1, np.int32)
    src = np.full((n, kmax), -1, np.int32)
    cnt = np.zeros((n,), np.int32)
    for f, L in enumerate(lists):
        assert len(L) <= kmax, (len(L), kmax)
        cnt[f] = len(L)
        for k, e in enumerate(L):
            box[f, k] = e[0:4]; conf[f, k] = e[4]; score[f, k] = e[5]; cls[f, k] = int(e[6]); src[f, k] = e[7]
    return dict(box=box, conf=conf, score=score, cls=cls, src=src, count=cnt)


def synthetic_heads(seed, hl, wl):
    """SURVEY.md 8(d) config 5 recipe: dense candidates."""
    g = np.random.default_rng(seed)
    outs = []
    for (h, w) in ((hl, wl), (hl // 2, wl // 2)):
        t = np.empty((3, 8, h, w), np.float32)
        t[:, 0:2] = g.normal(0.0, 1.0, (3, 2, h, w))
        t[:, 2:4] = g.normal(0.0, 0.5, (3, 2, h, w))
        t[:, 4] = g.normal(-1.0, 1.5, (3, h, w))
        t[:, 5:8] = g.normal(0.0, 2.0, (3, 3, h, w))
        outs.append(torch.from_numpy(t.reshape(1, 24, h, w)))
    return outs


PROBES = ["conv0", "conv1_4", "res1_1", "conv1_9", "conv2_1", "res2_2", "conv2_3", "conv3_1", "res3_2", "conv3_4",
          "res3_3", "res3_6", "conv3_6", "conv4_1", "res4_4", "conv4_2", "conv4_3", "conv5_1", "res5_5", "conv5_2",
          "conv5_3", "conv5_6", "deconv5_1", "conv4_1_1", "conv4_1_5"]


def main():
    files = sorted(os.listdir(os.path.join(REF, "test_data")))
    assert len(files) == 20
    torch.manual_seed(0)
    for res in (256, 512):
        model, io = load_model(res)
        names = files  # all 20 bundled frames at both sizes (frame 14 = noCloud_2m_4359: no targets at 512x640)
        u8 = np.stack([preprocess_u8(os.path.join(REF, "test_data", f), res) for f in names])
        hl_all, hs_all, cands, finals, adjs = [], [], [], [], []
        with torch.no_grad():
            for k in range(len(names)):
                pred = model(to_input(u8[k]))
                hl_all.append(pred[0].numpy()[0]); hs_all.append(pred[1].numpy()[0])
                c, f, a = reference_post(pred, io, adjust=(res == 256))
                cands.append(c); finals.append(f); adjs.append(a)
        # the reference module evaluated in fp64 (model.double()): "exact arithmetic" yardstick that tells how far
        # the reference's own fp32 result is from the real-number result of its graph
        import copy
        m64 = copy.deepcopy(model).double()
        with torch.no_grad():
            p64 = m64(to_input(u8).reshape(len(names), 1, res, res * 5 // 4).double())
        out = dict(names=np.array(names), input_u8=u8, head_large=np.stack(hl_all), head_small=np.stack(hs_all),
                   head_large_f64=p64[0].numpy(), head_small_f64=p64[1].numpy(),
                   anchors=np.array(io["anchors"][:2], np.float64), input_shape=np.array(io["input_shape"][:2]))
        for tag, L in (("cand", cands), ("final", finals), ("adj", adjs)):
            for k, v in pack_lists(L, 16).items():
                out[f"{tag}_{k}"] = v
        print(res, "has-target flags:", [int(len(f) > 0) for f in finals], "cands", [len(c) for c in cands])

        # random-normal and uniform-u8 synthetic inputs through the real weights (heads only)
        g = np.random.default_rng(123 + res)
        n_syn = 4 if res == 256 else 2
        syn_u8 = g.integers(0, 256, size=(n_syn, res, res * 5 // 4), dtype=np.uint8)
        with torch.no_grad():
            p = model(to_input(syn_u8).reshape(n_syn, 1, res, res * 5 // 4))
        out["syn_input_u8"] = syn_u8
        out["syn_head_large"] = p[0].numpy(); out["syn_head_small"] = p[1].numpy()

        # per-layer probes on image 1 (NCHW fp32, as the reference module produces them)
        acts = {}
        hooks = [getattr(model, nm).register_forward_hook(
            lambda m, i, o, nm=nm: acts.__setitem__(nm, o.detach().numpy()[0].copy())) for nm in PROBES]
        with torch.no_grad():
            model(to_input(u8[1]))
        for h in hooks:
            h.remove()
        if res == 256:
            for nm in PROBES:
                out["probe_" + nm] = acts[nm]
        np.savez_compressed(os.path.join(HERE, f"golden_{res}.npz"), **out)

        # dense synthetic heads -> reference decode + NMS (with source indices)
        hl, wl = res // 16, res * 5 // 4 // 16
        d = {}
        cands, finals, heads_l, heads_s = [], [], [], []
        for seed in range(4):
            pred = synthetic_heads(seed, hl, wl)
            c, f, _ = reference_post(pred, io, adjust=False)
            cands.append(c); finals.append(f)
            heads_l.append(pred[0].numpy()[0]); heads_s.append(pred[1].numpy()[0])
        kmax = max(len(c) for c in cands)
        d["head_large"] = np.stack(heads_l); d["head_small"] = np.stack(heads_s)
        d["anchors"] = np.array(io["anchors"][:2], np.float64); d["input_shape"] = np.array(io["input_shape"][:2])
        for tag, L in (("cand", cands), ("final", finals)):
            for k, v in pack_lists(L, kmax).items():
                d[f"{tag}_{k}"] = v
        print(res, "dense: cands", [len(c) for c in cands], "survivors", [len(f) for f in finals])
        np.savez_compressed(os.path.join(HERE, f"golden_dense_{res}.npz"), **d)


def main_val():
    """Validation-path goldens: the reference's own YOLOLossV3 decode branch and
    utils.general.non_max_suppression, imported from src/model_training with an empty stub module for cv2
    (only plot_one_box dereferences it)."""
    import types
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    sys.path.insert(0, os.path.join(REF, "src", "model_training"))
    from loss.yolo_loss import YOLOLossV3            # the reference
    from utils.general import non_max_suppression    # the reference
    model, io = load_model(256)
    g = np.load(os.path.join(HERE, "golden_256.npz"))
    dense = np.load(os.path.join(HERE, "golden_dense_256.npz"))
    dev = torch.device("cpu")
    losses = [YOLOLossV3(io["anchors"][i], io["num_cls"], io["input_shape"], dev) for i in range(2)]
    out = {}
    for tag, hl, hs in (("real", g["head_large"], g["head_small"]), ("dense", dense["head_large"], dense["head_small"])):
        pred = (torch.from_numpy(hl.copy()), torch.from_numpy(hs.copy()))
        with torch.no_grad():
            dec = torch.cat([losses[i](pred[i]) for i in range(2)], 1)  # validate.py:38-42
            dets = non_max_suppression(dec.clone(), io["num_cls"], conf_thres=0.5, nms_thres=0.2)
        kmax = max([0 if d is None else d.shape[0] for d in dets] + [1])
        det = np.zeros((len(dets), kmax, 7), np.float32)
        cnt = np.zeros((len(dets),), np.int32)
        for f, d in enumerate(dets):
            if d is not None:
                cnt[f] = d.shape[0]; det[f, :d.shape[0]] = d.numpy()
        out[f"{tag}_decode"] = dec.numpy()[:4]
        out[f"{tag}_det"] = det; out[f"{tag}_count"] = cnt
        print("val", tag, "detections per frame:", cnt.tolist())
    np.savez_compressed(os.path.join(HERE, "golden_val_256.npz"), **out)


def main_map():
    """mAP goldens (validate.py:27-122): the reference's own `Validation.get_mAP` run on the 20 bundled
    frames with SYNTHETIC targets (the dataset is not shipped).  `validate.py` imports `dataloader.detect_dataset`, which
    imports cv2 and tensorboardX at module level (neither is used by the code under test): empty stub modules.
    The dataset yields u8 - 128 as float (DetectDataset.collate_fn :115 divides by 255, so the net sees detect.py's
    (u8 - 128) / 255: with plain u8 / 255 the shipped checkpoint detects nothing on these frames).  Targets are built from
    the reference's own detections on them, then perturbed so that every branch is hit: shifted boxes (IoU above / below the 0.5
    match threshold), a wrong class, images without targets, targets without detections, duplicate targets."""
    import types, logging
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    tb = types.ModuleType("tensorboardX"); tb.SummaryWriter = object
    sys.modules.setdefault("tensorboardX", tb)
    sys.path.insert(0, os.path.join(REF, "src", "model_training"))
    from loss.yolo_loss import YOLOLossV3            # the reference
    from utils.general import non_max_suppression    # the reference
    from validate import Validation                  # the reference
    model, io = load_model(256)
    g = np.load(os.path.join(HERE, "golden_256.npz"))
    u8 = g["input_u8"]                                # [20,256,320]
    dev = torch.device("cpu")
    losses = [YOLOLossV3(io["anchors"][i], io["num_cls"], io["input_shape"], dev) for i in range(2)]
    H, W = io["input_shape"][0], io["input_shape"][1]
    with torch.no_grad():
        x = (torch.from_numpy(u8.astype(np.float32))[:, None] - 128.0) / 255.0
        pred = model(x)
        dec = torch.cat([losses[i](pred[i]) for i in range(2)], 1)
        dets = non_max_suppression(dec.clone(), io["num_cls"], conf_thres=io["conf_thre"], nms_thres=io["nms_thre"])
    targets = np.zeros((len(dets), 64, 6), np.float32)
    for f, d in enumerate(dets):
        rows = []
        if d is not None and f % 7 != 6:              # every 7th image: no targets at all (all its detections are FP)
            for k, t in enumerate(d.numpy()):
                x1, y1, x2, y2 = t[:4]
                shift = [0, 1, 3, 9, 30][(f + k) % 5]  # 30 px: no overlap left
                cls = int(t[6]) if (f + k) % 6 != 5 else (int(t[6]) + 1) % 3   # sometimes the wrong class
                rows.append([(x1 + x2) / 2 + shift, (y1 + y2) / 2, x2 - x1, y2 - y1, cls])
                if (f + k) % 4 == 3:
                    rows.append(rows[-1][:])          # duplicate target: only one of them can be matched
        if f % 3 == 0:
            rows.append([40.0 + f, 200.0, 20.0, 10.0, f % 3])   # a target nobody detects (FN)
        for k, r in enumerate(rows):
            targets[f, k] = [r[0] / W, r[1] / H, r[2] / W, r[3] / H, r[4], 255.0]

    class Frames(torch.utils.data.Dataset):           # what DetectDataset yields: (h,w,1) image, (64,6) boxes
        def __len__(self): return len(u8)
        def __getitem__(self, i): return u8[i][:, :, None].astype(np.float32) - 128.0, targets[i].copy()   # collate_fn divides by 255

    params = {"train_params": {"batch_size": 4, "IOU_val_thre": 0.5},
              "io_params": dict(io, class_names=["carrier", "defender", "destroyer"])}
    logger = logging.getLogger("ref-val"); logger.addHandler(logging.NullHandler())
    torch.manual_seed(0)
    val = Validation(params, logger, Frames(), dev, losses)
    mAP = float(val.get_mAP(model, 0))
    APs = [float(val._Validation__calculate_AP(cls=c)) for c in range(3)]
    out = {"targets": targets,   # the frames are golden_256.npz's input_u8
           "mAP": np.float64(mAP), "AP": np.array(APs, np.float64),
           "target_num": val.target_num.numpy().astype(np.float32)}
    for c in range(3):
        ml = val.match_list[c]
        # the entries are np.array([tensor_scalar, 'TP']) = STRINGS like 'tensor(0.8714)': the reference's sort (:77) is a
        # lexicographic sort of that 4-decimal printed form
        out[f"match_key_{c}"] = np.array([str(m[0]) for m in ml])
        out[f"match_conf_{c}"] = np.array([float(str(m[0])[7:-1]) for m in ml], np.float64)
        out[f"match_tp_{c}"] = np.array([m[1] == "TP" for m in ml], np.bool_)
    kmax = max([0 if d is None else d.shape[0] for d in dets] + [1])
    det = np.zeros((len(dets), kmax, 7), np.float32); cnt = np.zeros((len(dets),), np.int32)
    for f, d in enumerate(dets):
        if d is not None:
            cnt[f] = d.shape[0]; det[f, :d.shape[0]] = d.numpy()
    out["det"] = det; out["count"] = cnt
    np.savez_compressed(os.path.join(HERE, "golden_map_256.npz"), **out)
    print("mAP", mAP, "AP", APs, "targets", val.target_num.tolist(), "matches", [len(m) for m in val.match_list],
          "TP", [int(out[f"match_tp_{c}"].sum()) for c in range(3)])


def main_loss():
    """Training-loss goldens (first slice): the reference's own `YOLOLossV3(...)(input, targets)`
    (loss/yolo_loss.py:48-97, get_target :144-196) on the reference's head tensors of the 20 bundled frames, with the synthetic
    targets of main_map plus rows that hit the loop's special cases (two targets in one cell: the later one wins and the one-hot
    classes accumulate; a zero-size target that is skipped; the end marker), and `loss.backward()` for d(total)/d(input)."""
    import types
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    sys.path.insert(0, os.path.join(REF, "src", "model_training"))
    from loss.yolo_loss import YOLOLossV3            # the reference
    model, io = load_model(256)
    g = np.load(os.path.join(HERE, "golden_256.npz"))
    targets = np.load(os.path.join(HERE, "golden_map_256.npz"))["targets"].copy()   # [20,64,6]
    # special cases on image 0 / 1: same cell twice with different classes; zero-width target in the middle of the list
    n0 = int((targets[0, :, 5] > 1).sum())
    targets[0, n0] = targets[0, 0]; targets[0, n0, 4] = (targets[0, 0, 4] + 1) % 3; targets[0, n0, 2] *= 1.1
    n1 = int((targets[1, :, 5] > 1).sum())
    targets[1, n1] = [0.5, 0.5, 0.0, 0.1, 1.0, 255.0]
    targets[1, n1 + 1] = [0.31, 0.62, 0.12, 0.07, 2.0, 255.0]
    dev = torch.device("cpu")
    out = {"targets": targets}
    tt = torch.from_numpy(targets)
    for i, name in enumerate(("head_large", "head_small")):
        x = torch.from_numpy(g[name].copy()).requires_grad_(True)
        res = YOLOLossV3(io["anchors"][i], io["num_cls"], io["input_shape"], dev)(x, tt)
        res[0].backward()
        out[f"{name}_losses"] = np.array([res[0].item()] + [float(v) for v in res[1:]], np.float32)
        out[f"{name}_grad"] = x.grad.numpy().copy()
        print("loss", name, out[f"{name}_losses"].tolist(), "grad max", float(np.abs(out[f"{name}_grad"]).max()))
    np.savez_compressed(os.path.join(HERE, "golden_loss_256.npz"), **out)


def main_train():
    """Training-step goldens (second slice): the reference's own iteration of train.py:111-132 -- model.train(),
    optimizer.zero_grad(), pred = model(imgs), the two YOLOLossV3 heads summed, loss.backward(), optim.Adam(lr0 = 0.001, betas
    (0.9, 0.999), eps 1e-8).step() -- run twice on the reference's batch size (16: the first 16 bundled frames at 256x320, the targets
    of main_loss) from the shipped checkpoint.  Stored: the train-mode heads, losses and every parameter's gradient of both iterations
    (for the first one also, from a float64 run of the same reference code, which of them are zero in exact arithmetic), the
    parameters and BatchNorm running statistics after the second."""
    import types
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    sys.path.insert(0, os.path.join(REF, "src", "model_training"))
    from loss.yolo_loss import YOLOLossV3            # the reference
    g = np.load(os.path.join(HERE, "golden_256.npz"))
    targets = np.load(os.path.join(HERE, "golden_loss_256.npz"))["targets"][:16].copy()
    u8 = g["input_u8"][:16]
    out = {"input_u8": u8, "targets": targets}
    dev = torch.device("cpu")

    def run(dtype, steps):
        torch.manual_seed(0)
        torch.set_default_dtype(dtype)               # the loss builds its masks with torch.zeros(...): default dtype
        model, io = load_model(256)
        model = model.to(dtype).train()
        x = ((torch.from_numpy(u8.astype(np.float32))[:, None] - 128.0) / 255.0).to(dtype)
        tt = torch.from_numpy(targets).to(dtype)
        crit = [YOLOLossV3(io["anchors"][i], io["num_cls"], io["input_shape"], dev) for i in range(2)]
        opt = torch.optim.Adam(model.parameters(), lr=config_params["train_params"]["lr0"], betas=(0.9, 0.999), eps=1e-08)
        rec = []
        for it in range(steps):
            opt.zero_grad()
            pred = model(x)
            losses = [[] for _ in range(7)]
            for i, item_pred in enumerate(pred):
                for j, v in enumerate(crit[i](item_pred, tt)):
                    losses[j].append(v)
            losses = [sum(v) for v in losses]
            for p in pred:
                p.retain_grad()
            losses[0].backward()
            rec.append(dict(heads=[p.detach().float().numpy().copy() for p in pred], head_grads=[p.grad.detach().clone() for p in pred],
                            losses=np.array([float(v) for v in losses], np.float64),
                            grads=np.concatenate([p.grad.detach().float().numpy().ravel() for p in model.parameters()])))
            opt.step()
        return model, rec

    model, rec = run(torch.float32, 2)
    _, rec64 = run(torch.float64, 1)
    torch.set_default_dtype(torch.float32)
    out["head_large_1"], out["head_small_1"] = rec[0]["heads"]
    out["head_large_2"], out["head_small_2"] = rec[1]["heads"]
    out["losses_1"], out["losses_2"], out["losses_1_f64"] = rec[0]["losses"], rec[1]["losses"], rec64[0]["losses"]
    out["grads_1"], out["grads_2"] = rec[0]["grads"], rec[1]["grads"]
    out["params_2"] = np.concatenate([p.detach().numpy().ravel() for p in model.parameters()])
    out["param_names"] = np.array([n for n, _ in model.named_parameters()])
    out["param_sizes"] = np.array([p.numel() for p in model.parameters()], np.int64)
    bufs = [(n, b) for n, b in model.named_buffers() if not n.endswith("num_batches_tracked")]
    out["buffers_2"] = np.concatenate([b.detach().numpy().ravel() for _, b in bufs])
    out["buffer_names"] = np.array([n for n, _ in bufs])
    out["num_batches_tracked_2"] = np.array([int(b) for n, b in model.named_buffers() if n.endswith("num_batches_tracked")], np.int64)
    # Per parameter tensor, max |gradient| of the float64 run.  It separates the parameters whose gradient is ZERO IN EXACT ARITHMETIC
    # (a BatchNorm bias or conv bias that only feeds train-mode BatchNorms: 1e-16 in float64, 1e-6 of rounding noise in the
    # reference's float32 -- which Adam then turns into +-lr steps of random sign) from the ones a parity test can compare.
    # NOT an accuracy yardstick: in float64 the sigmoids no longer saturate to exactly 1.0, so BCELoss's -100 clamp stops firing and
    # the class loss itself differs (losses_1_f64).
    off = np.concatenate([[0], np.cumsum(out["param_sizes"])])
    # The yardstick for the gradients: the SAME head gradients (the fp32 run's d loss / d heads) pushed back through the network in
    # float64 -- what the layers' backward gives without rounding.  (The float64 loss itself is no yardstick, see above.)
    torch.set_default_dtype(torch.float64)
    m64, _ = load_model(256)
    m64 = m64.double().train()
    p64 = m64(((torch.from_numpy(u8.astype(np.float32))[:, None] - 128.0) / 255.0).double())
    torch.autograd.backward(list(p64), [h.double() for h in rec[0]["head_grads"]])
    torch.set_default_dtype(torch.float32)
    out["grads_1_exact"] = np.concatenate([p.grad.detach().float().numpy().ravel() for p in m64.parameters()])
    out["grad_absmax_f64"] = np.array([np.abs(rec64[0]["grads"][off[i]:off[i + 1]]).max() for i in range(len(off) - 1)], np.float64)
    print("train: losses", out["losses_1"].tolist(), out["losses_2"].tolist(), "params", out["params_2"].size,
          "| zero-gradient tensors:", int((out["grad_absmax_f64"] < 1e-9).sum()), "of", len(off) - 1,
          "| reference fp32 gradients vs exact backward of the same head gradients: worst tensor",
          max(float(np.abs(out["grads_1"][off[i]:off[i + 1]] - out["grads_1_exact"][off[i]:off[i + 1]]).max() /
                    np.abs(out["grads_1_exact"][off[i]:off[i + 1]]).max()) for i in range(len(off) - 1) if out["grad_absmax_f64"][i] >= 1e-9))
    np.savez_compressed(os.path.join(HERE, "golden_train_256.npz"), **out)


def main_results():
    """Result-writer goldens (detect.py:176-192): what the reference's own detect.py run left under
    test_result/<size>/<laptop cpu (python)>_test_result/ -- DATA of the reference, not source:
      * from cpu-test.log: per image, whether the line says 'detect finished' or 'no targets' (+ the logged average time);
      * from the result_<name>.jpg images: the RGB value at the midpoint of each edge of every box the goldens predict
        (adj_box of golden_<size>.npz), i.e. where plot_one_box drew the class-coloured frame, plus one pixel 12 px INSIDE
        the frame (background) -- ties the golden boxes and the class colours to the reference's real output images."""
    import re
    from PIL import Image
    out = {}
    for res, sub in ((256, "256x320"), (512, "512x640")):
        d = [x for x in os.listdir(os.path.join(REF, "test_result", sub)) if "cpu" in x][0]
        d = os.path.join(REF, "test_result", sub, d)
        log = open(os.path.join(d, "cpu-test.log"), "rb").read().decode("latin-1")
        rows = re.findall(r"image_name:(\S+) -> (detect finished|no targets), infer time:([0-9.]+)ms, post_process time:([0-9.]+)ms, "
                          r"total time:([0-9.]+)ms", log)
        avg = float(re.search(r"detect avg_time: ([0-9.]+)ms", log).group(1))
        g = np.load(os.path.join(HERE, f"golden_{res}.npz"))
        names = [str(n) for n in g["names"]]
        assert [r[0] for r in rows] == names, "log order == sorted test_data order"
        flags = np.array([r[1] == "detect finished" for r in rows], np.bool_)
        assert flags.tolist() == [bool(c > 0) for c in g["adj_count"]], "the goldens reproduce the reference's logged flags"
        edge = np.zeros((len(names), 16, 5, 3), np.uint8)
        for f, n in enumerate(names):
            im = np.asarray(Image.open(os.path.join(d, "result_" + n)).convert("RGB"))
            assert im.shape == (512, 640, 3)
            for k in range(int(g["adj_count"][f])):
                x1, y1, x2, y2 = [int(v) for v in g["adj_box"][f, k]]
                pts = [((x1 + x2) // 2, y1), ((x1 + x2) // 2, y2), (x1, (y1 + y2) // 2), (x2, (y1 + y2) // 2),
                       (min(x1 + 12, (x1 + x2) // 2), (y1 + y2) // 2)]
                for j, (x, y) in enumerate(pts):
                    edge[f, k, j] = im[min(max(y, 0), 511), min(max(x, 0), 639)]
        out[f"names_{res}"] = np.array(names); out[f"finished_{res}"] = flags; out[f"avg_time_ms_{res}"] = np.float64(avg)
        out[f"edge_rgb_{res}"] = edge
        print(res, "logged flags", flags.astype(int).tolist(), "avg", avg)
    np.savez_compressed(os.path.join(HERE, "golden_results.npz"), **out)


def main_io():
    """io_params generality: the reference's constructors are parameterised on num_cls, input_channel and
    num_anchors (yolo_fastest.py:72-78,138,148; detect.py:15-21,53-66; yolo_loss.py:28-33,58-60).  For every configuration of
    seeded_weights.IO_CONFIGS the REFERENCE module is built, loaded with a numpy-seeded random state-dict (only the seed is stored)
    and run on seeded u8 frames prepared like detect.py:119-124 (3-channel frames: BGR -> [::-1] -> CHW); recorded: its heads in
    fp32 and -- model.double() -- fp64, the candidates / survivors of its own YOLO_post_process (with source indices), for the
    3-anchor configurations the validation-time decode + NMS (yolo_loss.py:98-141 hard-codes 3 anchors at :110-111), the training
    loss of both heads with its gradient, and -- two configurations, 64x96 frames -- one train-mode iteration (heads, losses, a
    strided sample and the per-tensor sums of every parameter gradient)."""
    import types, copy
    from collections import OrderedDict
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    sys.path.insert(0, os.path.join(REF, "src", "model_training"))
    sys.path.insert(0, HERE)
    from loss.yolo_loss import YOLOLossV3            # the reference
    from utils.general import non_max_suppression    # the reference
    from seeded_weights import seeded_state_dict, IO_CONFIGS, io_inputs, io_targets
    dev = torch.device("cpu")
    out = {}

    def build(C, Cin, A, H, W, seed):
        io = dict(config_params["io_params"])
        io.update(num_cls=C, input_channel=Cin, num_anchors=A, input_shape=[H, W, Cin], origin_img_shape=[H, W, Cin],
                  anchors=[grp[:A] for grp in config_params["io_params"]["anchors"]])
        m = YoloFastest(io).eval()
        shapes = OrderedDict((k, tuple(v.shape)) for k, v in m.state_dict().items())
        sd = {k: torch.from_numpy(np.asarray(v)) for k, v in seeded_state_dict(shapes, seed).items()}
        assert str(m.load_state_dict(sd)) == "<All keys matched successfully>"
        return m, io

    def to_x(u8, Cin):
        if Cin == 1:
            a = u8[:, None]
        else:
            a = np.ascontiguousarray(u8[:, :, :, ::-1].transpose(0, 3, 1, 2))   # detect.py:119: img[:, :, ::-1].transpose(2, 0, 1)
        return (torch.from_numpy(a.astype(np.float32)) - 128.0) / 255.0

    def src_idx(pred, conf_thres, A, attrs):
        o, base = [], 0
        for ph in pred:
            a = ph.numpy()[0]
            h, w = a.shape[1], a.shape[2]
            a = a.reshape(A, attrs, h, w)
            for pp in range(A):
                for i in range(h):
                    for j in range(w):
                        if 1. / (1. + math.exp(-a[pp, 4, i, j])) > conf_thres:
                            o.append(base + (pp * h + i) * w + j)
            base += A * h * w
        return o

    for idx, (tag, C, Cin, A) in enumerate(IO_CONFIGS):
        seed = 1000 + idx
        model, io = build(C, Cin, A, 256, 320, seed)
        u8 = io_inputs(tag, Cin)
        x = to_x(u8, Cin)
        with torch.no_grad():
            pred = model(x)
            p64 = copy.deepcopy(model).double()(x.double())
        out[f"{tag}_seed"] = np.int64(seed)
        out[f"{tag}_head_large"], out[f"{tag}_head_small"] = pred[0].numpy(), pred[1].numpy()
        out[f"{tag}_head_large_f64"], out[f"{tag}_head_small_f64"] = p64[0].numpy(), p64[1].numpy()
        # the reference's own post-process, frame by frame (it reads batch element 0, detect.py:46)
        pp = YOLO_post_process(conf_thres=io["conf_thre"], nms_thres=io["nms_thre"], num_anchors=A, num_class=C,
                               anchors=io["anchors"], input_shape=io["input_shape"])
        cands, finals = [], []
        for f in range(len(u8)):
            pf = (pred[0][f:f + 1], pred[1][f:f + 1])
            c = pp.decode_box(pf)
            s = src_idx(pf, io["conf_thre"], A, 5 + C)
            assert len(c) == len(s)
            for ci, si in zip(c, s):
                ci.append(si)
            cands.append([list(v) for v in c])
            buckets = [[] for _ in range(C)]
            for b in c:
                buckets[b[6]].append(b)
            fin = []
            try:
                for cls in range(C):
                    if buckets[cls]:
                        buckets[cls].sort(key=lambda it: it[4], reverse=True)
                        fin.extend(pp.non_maxium_supression(buckets[cls]))
            except ZeroDivisionError:      # detect.py:39: two zero-area boxes compared -- recorded as count -2 for that frame
                fin = None
            finals.append(None if fin is None else [list(v) for v in fin])
        kmax = max(len(c) for c in cands)
        zde = [f is None for f in finals]
        for t2, L in (("cand", cands), ("final", [f or [] for f in finals])):
            for k, v in pack_lists(L, kmax).items():
                out[f"{tag}_{t2}_{k}"] = v
        out[f"{tag}_final_count"][zde] = -2
        print(tag, "heads", tuple(pred[0].shape), tuple(pred[1].shape), "cands", [len(c) for c in cands], "survivors", [-2 if f is None else len(f) for f in finals],
              "fp32 vs fp64", float((pred[0].double() - p64[0]).abs().max()), float((pred[1].double() - p64[1]).abs().max()))
        # validation-time decode + NMS (3 anchors only: yolo_loss.py:110-111 repeats the grid 3 times)
        crit = [YOLOLossV3(io["anchors"][i], C, io["input_shape"], dev) for i in range(2)]
        if A == 3:
            with torch.no_grad():
                dec = torch.cat([crit[i](pred[i]) for i in range(2)], 1)
                dets = non_max_suppression(dec.clone(), C, conf_thres=0.5, nms_thres=0.2)
            km = max([0 if d is None else d.shape[0] for d in dets] + [1])
            det = np.zeros((len(dets), km, 7), np.float32); cnt = np.zeros((len(dets),), np.int32)
            for f, d in enumerate(dets):
                if d is not None:
                    cnt[f] = d.shape[0]; det[f, :d.shape[0]] = d.numpy()
            out[f"{tag}_val_decode"] = dec.numpy()[:1, ::3].copy()     # every third row of frame 0
            out[f"{tag}_val_det"] = det; out[f"{tag}_val_count"] = cnt
            print(tag, "val detections", cnt.tolist())
        # training loss of both heads + gradient
        tt = torch.from_numpy(io_targets(tag, C, len(u8)))
        for i, name in enumerate(("head_large", "head_small")):
            xh = pred[i].clone().requires_grad_(True)
            res = crit[i](xh, tt)
            res[0].backward()
            out[f"{tag}_{name}_losses"] = np.array([res[0].item()] + [float(v) for v in res[1:]], np.float32)
            out[f"{tag}_{name}_grad"] = xh.grad.numpy().copy()
            print(tag, "loss", name, out[f"{tag}_{name}_losses"].tolist())
        # one train-mode iteration (train.py:111-131 without the optimizer step) on small frames
        if tag in ("c5rgb", "a2", "ch4", "ch6"):
            torch.manual_seed(0)
            m, io2 = build(C, Cin, A, 64, 96, seed)
            m.train()
            u8t = io_inputs(tag + "_train", Cin, n=4, H=64, W=96)
            ttt = torch.from_numpy(io_targets(tag + "_train", C, 4))
            crit2 = [YOLOLossV3(io2["anchors"][i], C, io2["input_shape"], dev) for i in range(2)]
            p = m(to_x(u8t, Cin))
            losses = [[] for _ in range(7)]
            for i, item in enumerate(p):
                for j, v in enumerate(crit2[i](item, ttt)):
                    losses[j].append(v)
            losses = [sum(v) for v in losses]
            losses[0].backward()
            out[f"{tag}_train_head_large"], out[f"{tag}_train_head_small"] = p[0].detach().numpy(), p[1].detach().numpy()
            out[f"{tag}_train_losses"] = np.array([float(v) for v in losses], np.float64)
            gr = [q.grad.detach().numpy().ravel() for q in m.parameters()]
            out[f"{tag}_train_grad_sample"] = np.concatenate(gr)[::37].copy()
            out[f"{tag}_train_grad_abssum"] = np.array([np.abs(v.astype(np.float64)).sum() for v in gr])
            out[f"{tag}_train_param_names"] = np.array([n for n, _ in m.named_parameters()])
            bufs = [b.detach().numpy().ravel() for n, b in m.named_buffers() if not n.endswith("num_batches_tracked")]
            out[f"{tag}_train_buffers_sample"] = np.concatenate(bufs)[::7].copy()
            print(tag, "train losses", out[f"{tag}_train_losses"].tolist())
    np.savez_compressed(os.path.join(HERE, "golden_io.npz"), **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "map":
        main_map()
    elif len(sys.argv) > 1 and sys.argv[1] == "results":
        main_results()
    elif len(sys.argv) > 1 and sys.argv[1] == "loss":
        main_loss()
    elif len(sys.argv) > 1 and sys.argv[1] == "train":
        main_train()
    elif len(sys.argv) > 1 and sys.argv[1] == "io":
        main_io()
    else:
        main()
        main_val()
        main_map()
        main_results()
        main_loss()
        main_train()
        main_io()
