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
