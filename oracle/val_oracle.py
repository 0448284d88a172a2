"""ORACLE (test infrastructure, not product): CPU restatement of the reference's VALIDATION-time decode + NMS
 -- a different convention from detect.py's post-process: fp32 torch arithmetic, float boxes,
`conf >= thres`, IoU with the +1 pixel convention, keep `iou < nms_thres`.

Follows (file:line under the reference repo)
    src/model_training/loss/yolo_loss.py   YOLOLossV3.forward, decode branch   :48-68, :98-141
    src/model_training/utils/general.py    bbox_iou :29-52, non_max_suppression :87-143
    src/model_training/validate.py         the caller: per-head decode, torch.cat over heads, NMS   :38-45

Parity pin: tests/test_oracle_golden.py checks this file against tests/golden/golden_val_256.npz, produced by
tests/golden/make_golden.py by importing and running the reference's own YOLOLossV3 / non_max_suppression.
"""
import torch


def decode_head(x, anchors, num_classes, input_shape):
    """yolo_loss.py:48-68, :98-141.  x: float32 [bs, A*(5+C), h, w] -> [bs, A*h*w, 5+C] = (cx, cy, w, h, conf, cls...)."""
    bs, _, in_h, in_w = x.shape
    A = len(anchors)
    stride_h = input_shape[0] / in_h
    stride_w = input_shape[1] / in_w
    scaled = [(a_w / stride_w, a_h / stride_h) for a_w, a_h in anchors]
    p = x.view(bs, A, 5 + num_classes, in_h, in_w).permute(0, 1, 3, 4, 2).contiguous()
    sx, sy = torch.sigmoid(p[..., 0]), torch.sigmoid(p[..., 1])
    w, h = p[..., 2], p[..., 3]
    conf = torch.sigmoid(p[..., 4])
    cls = torch.sigmoid(p[..., 5:])
    grid_x = torch.arange(in_w).repeat(bs, A, in_h, 1).float()
    grid_y = torch.arange(in_h).repeat(bs, A, in_w, 1).permute(0, 1, 3, 2).float()
    aw = torch.tensor([s[0] for s in scaled], dtype=torch.float32).view(1, A, 1, 1)
    ah = torch.tensor([s[1] for s in scaled], dtype=torch.float32).view(1, A, 1, 1)
    boxes = torch.empty(p[..., :4].shape, dtype=torch.float32)
    boxes[..., 0] = sx + grid_x
    boxes[..., 1] = sy + grid_y
    boxes[..., 2] = torch.exp(w) * aw
    boxes[..., 3] = torch.exp(h) * ah
    scale = torch.tensor([stride_w, stride_h] * 2, dtype=torch.float32)
    return torch.cat((boxes.view(bs, -1, 4) * scale, conf.view(bs, -1, 1), cls.view(bs, -1, num_classes)), -1)


def decode(pred, anchors2, num_classes, input_shape):
    """validate.py:38-42: decode every head with its anchor group, concatenate along the box axis."""
    return torch.cat([decode_head(p, anchors2[i], num_classes, input_shape) for i, p in enumerate(pred)], 1)


def bbox_iou(box1, box2):  # general.py:29-52 (x1y1x2y2=True branch), +1 pixel convention
    ix1 = torch.max(box1[:, 0], box2[:, 0]); iy1 = torch.max(box1[:, 1], box2[:, 1])
    ix2 = torch.min(box1[:, 2], box2[:, 2]); iy2 = torch.min(box1[:, 3], box2[:, 3])
    inter = torch.clamp(ix2 - ix1 + 1, min=0) * torch.clamp(iy2 - iy1 + 1, min=0)
    a1 = (box1[:, 2] - box1[:, 0] + 1) * (box1[:, 3] - box1[:, 1] + 1)
    a2 = (box2[:, 2] - box2[:, 0] + 1) * (box2[:, 3] - box2[:, 1] + 1)
    return inter / (a1 + a2 - inter + 1e-16)


def non_max_suppression(prediction, num_classes, conf_thres=0.5, nms_thres=0.4):
    """general.py:87-143.  prediction [bs, M, 5+C] (centre format; converted to corners like the reference, on a copy).
    Returns a list with one [n, 7] tensor (x1, y1, x2, y2, obj_conf, class_conf, class_pred) or None per image."""
    prediction = prediction.clone()
    c = prediction.new(prediction.shape)
    c[:, :, 0] = prediction[:, :, 0] - prediction[:, :, 2] / 2
    c[:, :, 1] = prediction[:, :, 1] - prediction[:, :, 3] / 2
    c[:, :, 2] = prediction[:, :, 0] + prediction[:, :, 2] / 2
    c[:, :, 3] = prediction[:, :, 1] + prediction[:, :, 3] / 2
    prediction[:, :, :4] = c[:, :, :4]
    output = [None for _ in range(len(prediction))]
    for i, ip in enumerate(prediction):
        ip = ip[(ip[:, 4] >= conf_thres)]
        if not ip.size(0):
            continue
        class_conf, class_pred = torch.max(ip[:, 5:5 + num_classes], dim=1, keepdim=True)
        det = torch.cat((ip[:, :5], class_conf.float(), class_pred.float()), 1)
        for cl in det[:, -1].unique():
            dc = det[det[:, 6] == cl]
            _, order = torch.sort(dc[:, 4], descending=True, stable=True)  # the reference's sort is unstable: ties are unspecified there
            dc = dc[order]
            keep = []
            while dc.size(0):
                keep.append(dc[0].unsqueeze(0))
                if len(dc) == 1:
                    break
                ious = bbox_iou(keep[-1], dc[1:])
                dc = dc[1:][ious < nms_thres]
            keep = torch.cat(keep)
            output[i] = keep if output[i] is None else torch.cat((output[i], keep))
    return output


# ---- mAP bookkeeping (validate.py:27-122) ----------------------------------------------------------------------------------

def recover_targets(targets, input_shape):
    """validate.py:112-122: normalised (xc, yc, w, h, cls, marker) -> input-image corners (x1, y1, x2, y2, cls, marker)."""
    t = targets.clone().float()
    in_h, in_w = input_shape[0], input_shape[1]
    t[:, :, (0, 2)] = t[:, :, (0, 2)] * in_w
    t[:, :, (1, 3)] = t[:, :, (1, 3)] * in_h
    out = t.clone()
    out[:, :, 0] = t[:, :, 0] - t[:, :, 2] / 2
    out[:, :, 1] = t[:, :, 1] - t[:, :, 3] / 2
    out[:, :, 2] = t[:, :, 0] + t[:, :, 2] / 2
    out[:, :, 3] = t[:, :, 1] + t[:, :, 3] / 2
    return out


def match_image(img_pred, img_target, num_cls, iou_thr, match_list, target_num):
    """validate.py:47-74 for one image: img_pred [n,7] or None (NMS output), img_target [64,6] recovered targets.
    Appends (sort key, is_tp) to match_list[c]; the key is the printed form of the 0-d confidence tensor, which is what
    np.array([t[4], 'TP']) stores in the reference."""
    img_target = img_target[img_target[:, 5] > 1]
    for t in img_target:
        target_num[int(t[4])] += 1
    if img_pred is None:
        return
    for c in img_pred[:, 6].unique():
        target_c = img_target[img_target[:, 4] == c]
        pred_c = img_pred[img_pred[:, 6] == c]
        c = int(c)
        for t in pred_c:
            if target_c.size(0) == 0:
                match_list[c].append((str(t[4]), False))
                continue
            ious = bbox_iou(t.unsqueeze(0), target_c)
            hit = False
            for index, iou in enumerate(ious):
                if iou > iou_thr:
                    match_list[c].append((str(t[4]), True))
                    hit = True
                    target_c = torch.cat((target_c[0:index], target_c[index + 1:]), dim=0)  # general.py:76-79
                    break
            if not hit:
                match_list[c].append((str(t[4]), False))


def calculate_ap(matches, n_targets):
    """validate.py:87-119, literally (O(n^2)).  matches: [(key, is_tp)] already sorted; n_targets: float32 tensor scalar."""
    import numpy as np
    pr = []
    for i in range(len(matches)):
        tp = fp = 0
        for index in range(i + 1):
            if matches[index][1]:
                tp += 1
            else:
                fp += 1
        fn = n_targets - tp                       # float32 tensor
        precision = tp / (tp + fp)
        recall = tp / (tp + fn)                   # float32 tensor
        if i > 0 and recall == pr[-1][1]:
            if precision > pr[-1][0]:
                pr[-1][0] = precision
        else:
            pr.append(np.array([precision, recall]))
    ap, pre = 0, 0
    pr = np.array(pr)
    for i in range(pr.shape[0]):
        ap += (pr[i][1] - pre) * np.max(pr[i:], axis=0)[0]
        pre = pr[i][1]
    return ap


def get_map(dets, targets, num_cls, input_shape, iou_thr):
    """dets: list of [n,7] tensors / None per image (NMS output); targets [N,64,6] normalised.  -> (mAP, [AP], target_num, match_list)"""
    match_list = [[] for _ in range(num_cls)]
    target_num = torch.zeros((num_cls))
    rec = recover_targets(targets, input_shape)
    for f, d in enumerate(dets):
        match_image(d, rec[f], num_cls, iou_thr, match_list, target_num)
    for c in range(num_cls):
        match_list[c].sort(key=lambda x: x[0], reverse=True)     # validate.py:77: a sort of STRINGS
    aps = [calculate_ap(match_list[c], target_num[c]) for c in range(num_cls)]
    return sum(aps) / num_cls, aps, target_num, match_list
