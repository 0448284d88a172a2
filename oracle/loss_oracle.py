"""ORACLE (test infrastructure, not product): CPU restatement of the reference's TRAINING-time loss of one head
(first slice) and, through torch autograd on this restatement, its gradient with respect to the head tensor.

Follows (file:line under the reference repo)
    src/model_training/loss/yolo_loss.py   YOLOLossV3.forward with targets :48-97, get_target :144-196
    src/model_training/utils/general.py    bbox_iou :29-52 (x1y1x2y2 branch, +1 pixel convention)
    src/model_training/_config.py          train_params.IOU_loss_thre = 0.5 (:45)

Parity pin: tests/test_oracle_golden.py checks this file against tests/golden/golden_loss_256.npz, produced by
tests/golden/make_golden.py (main_loss) by importing and running the reference's own YOLOLossV3 with targets and calling
backward() on its total loss.
"""
import math

import numpy as np
import torch
import torch.nn as nn


def _bbox_iou(box1, box2):  # general.py:29-52
    ix1 = torch.max(box1[:, 0], box2[:, 0]); iy1 = torch.max(box1[:, 1], box2[:, 1])
    ix2 = torch.min(box1[:, 2], box2[:, 2]); iy2 = torch.min(box1[:, 3], box2[:, 3])
    inter = torch.clamp(ix2 - ix1 + 1, min=0) * torch.clamp(iy2 - iy1 + 1, min=0)
    a1 = (box1[:, 2] - box1[:, 0] + 1) * (box1[:, 3] - box1[:, 1] + 1)
    a2 = (box2[:, 2] - box2[:, 0] + 1) * (box2[:, 3] - box2[:, 1] + 1)
    return inter / (a1 + a2 - inter + 1e-16)


def get_target(target, anchors, in_w, in_h, ignore_threshold, num_classes=3):
    """yolo_loss.py:144-196.  target float32 [bs, T, 6]; anchors: list of (w, h) Python floats in feature-map units."""
    bs, A = target.size(0), len(anchors)
    mask = torch.zeros(bs, A, in_h, in_w); noobj_mask = torch.ones(bs, A, in_h, in_w)
    tx = torch.zeros(bs, A, in_h, in_w); ty = torch.zeros(bs, A, in_h, in_w)
    tw = torch.zeros(bs, A, in_h, in_w); th = torch.zeros(bs, A, in_h, in_w)
    tcls = torch.zeros(bs, A, in_h, in_w, num_classes)
    for b in range(bs):
        for t in range(target.shape[1]):
            if target[b, t, 5] < 1:
                break
            gx = target[b, t, 0] * in_w; gy = target[b, t, 1] * in_h
            gw = target[b, t, 2] * in_w; gh = target[b, t, 3] * in_h
            if gw <= 0 or gh <= 0:
                continue
            gi, gj = int(gx), int(gy)
            gt_box = torch.FloatTensor(np.array([0.0, 0.0, gw, gh], dtype=np.float32)).unsqueeze(0)
            anchor_shapes = torch.FloatTensor(np.concatenate((np.zeros((A, 2)), np.array(anchors)), 1))
            anch_ious = _bbox_iou(gt_box, anchor_shapes)
            noobj_mask[b, anch_ious > ignore_threshold, gj, gi] = 0
            best_n = int(np.argmax(anch_ious))
            mask[b, best_n, gj, gi] = 1
            tx[b, best_n, gj, gi] = gx - gi
            ty[b, best_n, gj, gi] = gy - gj
            tw[b, best_n, gj, gi] = math.log(gw / anchors[best_n][0] + 1e-16)
            th[b, best_n, gj, gi] = math.log(gh / anchors[best_n][1] + 1e-16)
            tcls[b, best_n, gj, gi, int(target[b, t, 4])] = 1
    return mask, noobj_mask, tx, ty, tw, th, tcls


def loss_head(x, targets, anchors, num_classes, input_shape, ignore_threshold=0.5):
    """yolo_loss.py:48-97.  x float32 [bs, A*(5+C), h, w] (may require grad) -> (loss tensor, x, y, w, h, conf, cls as floats)."""
    bs, _, in_h, in_w = x.shape
    A = len(anchors)
    stride_h, stride_w = input_shape[0] / in_h, input_shape[1] / in_w
    scaled = [(a_w / stride_w, a_h / stride_h) for a_w, a_h in anchors]
    p = x.view(bs, A, 5 + num_classes, in_h, in_w).permute(0, 1, 3, 4, 2).contiguous()
    sx, sy = torch.sigmoid(p[..., 0]), torch.sigmoid(p[..., 1])
    w, h = p[..., 2], p[..., 3]
    conf = torch.sigmoid(p[..., 4])
    pred_cls = torch.sigmoid(p[..., 5:])
    mask, noobj_mask, tx, ty, tw, th, tcls = get_target(targets, scaled, in_w, in_h, ignore_threshold, num_classes)
    bce, mse = nn.BCELoss(), nn.MSELoss()
    loss_x = bce(sx * mask, tx * mask); loss_y = bce(sy * mask, ty * mask)
    loss_w = mse(w * mask, tw * mask); loss_h = mse(h * mask, th * mask)
    loss_conf = bce(conf * mask, mask) + 0.5 * bce(conf * noobj_mask, noobj_mask * 0.0)
    loss_cls = bce(pred_cls[mask == 1], tcls[mask == 1])
    loss = loss_x * 2.5 + loss_y * 2.5 + loss_w * 2.5 + loss_h * 2.5 + loss_conf * 1.0 + loss_cls * 1.0
    return loss, loss_x.item(), loss_y.item(), loss_w.item(), loss_h.item(), loss_conf.item(), loss_cls.item()


def loss_and_grad(x, targets, anchors, num_classes, input_shape, ignore_threshold=0.5):
    """-> (float32[7] losses, grad of the total loss with respect to x, like x)."""
    x = x.detach().clone().requires_grad_(True)
    out = loss_head(x, targets, anchors, num_classes, input_shape, ignore_threshold)
    out[0].backward()
    return np.array([out[0].item()] + list(out[1:]), np.float32), x.grad.detach()
