"""Validation-time decode + NMS on the GPU behind the reference's own names:
  `YOLOLossV3(anchors, num_classes, input_shape, device)(input)`   src/model_training/loss/yolo_loss.py:27-141 (decode branch)
  `non_max_suppression(prediction, num_classes, conf_thres, nms_thres)`   src/model_training/utils/general.py:87-143
With targets, `YOLOLossV3(...)(input, targets)` is the reference's TRAINING loss of that head (yolo_loss.py:70-97, :144-196) with an
analytic backward to the head tensor (yf_train_loss; SURVEY.md 8(f).4, first slice -- the layers' backward is not built).  All need an engine handle (it carries the device; the decode also needs
H and W, which come from `input_shape`).  The model is passed explicitly (`model=` / `loss.model = m`); `bind(model)` sets the
default used when none is given.  The engine is chosen by (H, W, device of the tensor) -- never "whichever exists"."""
import ctypes

import torch

from . import _lib

_bound = {"model": None}


def bind(model):
    """Default model for calls that do not name one (the reference's signatures have no model argument)."""
    _bound["model"] = model


def _model(explicit):
    m = explicit if explicit is not None else _bound["model"]
    if m is None:
        raise RuntimeError("pass model=... (or call validation.bind(model) first): the kernels need an engine handle")
    return m


def _engine(t, H, W, model=None):
    if not t.is_cuda:
        raise RuntimeError("validation path (HIP) has no CPU implementation: pass GPU tensors")
    return _model(model).engine(H, W, t.shape[0], t.device)


class YOLOLossV3(torch.nn.Module):
    def __init__(self, anchors, num_classes, input_shape, device, model=None):
        super().__init__()
        # the YoloFastest whose heads this decodes (None: validation.bind's default).  NOT registered as a submodule: the loss object's
        # .train() / .eval() / .to() / state_dict() must not reach into the network (the reference's YOLOLossV3 does not hold the model)
        object.__setattr__(self, "model", model)
        self.anchors = anchors
        self.num_anchors = len(anchors)
        self.num_classes = num_classes
        self.bbox_attrs = 5 + num_classes
        self.input_shape = input_shape
        self.device = device
        self.ignore_threshold = 0.5    # config_params["train_params"]["IOU_loss_thre"] (_config.py:45)
        if not (1 <= self.num_anchors <= 8) or int(num_classes) < 1:
            raise ValueError("1..8 anchors and at least one class")

    def forward(self, input, targets=None):
        if targets is not None:
            return self._train_loss(input, targets)
        x = input.contiguous().float()
        bs, _, fh, fw = x.shape
        if (fh * 16, fw * 16) != (int(self.input_shape[0]), int(self.input_shape[1])) and \
           (fh * 32, fw * 32) != (int(self.input_shape[0]), int(self.input_shape[1])):
            raise ValueError("head of %dx%d cells does not belong to a %s input" % (fh, fw, tuple(self.input_shape[:2])))
        m = _model(self.model)
        if x.shape[1] != self.num_anchors * self.bbox_attrs or (m.num_anchors, m.num_cls) != (self.num_anchors, self.num_classes):
            raise RuntimeError("head of %d channels / a model of %d anchors x %d classes, this loss: %d anchors x (5 + %d classes)"
                               % (x.shape[1], m.num_anchors, m.num_cls, self.num_anchors, self.num_classes))   # yolo_loss.py:58's view raises
        e = _engine(x, int(self.input_shape[0]), int(self.input_shape[1]), self.model)
        # (the reference's decode branch repeats its grid 3 times, yolo_loss.py:110-111, and only runs with 3 anchors; any count here)
        M = self.num_anchors * fh * fw
        out = torch.empty((bs, M, self.bbox_attrs), dtype=torch.float32, device=x.device)
        anc = (ctypes.c_double * (2 * self.num_anchors))(*[float(v) for a in self.anchors for v in a[:2]])
        stream = torch.cuda.current_stream(x.device).cuda_stream
        _lib.check(e.lib.yf_val_decode_head(e.handle, x.data_ptr(), bs, fh, fw, anc, M, 0, out.data_ptr(), ctypes.c_void_p(stream)))
        return out


class _TrainLossFn(torch.autograd.Function):
    """total loss of one head with its analytic gradient (yf_train_loss): `loss.backward()` reaches the head tensor."""

    @staticmethod
    def forward(ctx, x, targets, loss_mod):
        bs, _, fh, fw = x.shape
        H, W = int(loss_mod.input_shape[0]), int(loss_mod.input_shape[1])
        lib = _lib.lib()                                     # no inference engine needed: the loss takes (device, H, W)
        dev_index = x.device.index if x.device.index is not None else torch.cuda.current_device()
        need = ctypes.c_size_t()
        na, nc = loss_mod.num_anchors, loss_mod.num_classes
        _lib.check(lib.yf_train_head_loss_workspace_bytes_ex(bs, fh, fw, na, nc, ctypes.byref(need)))
        work = torch.empty((need.value + 7) // 8, dtype=torch.float64, device=x.device)     # 8-byte aligned scratch
        losses = torch.empty(8, dtype=torch.float32, device=x.device)
        grad = torch.empty_like(x)
        anc = (ctypes.c_double * (2 * na))(*[float(v) for a in loss_mod.anchors for v in a[:2]])
        stream = torch.cuda.current_stream(x.device).cuda_stream
        _lib.check(lib.yf_train_head_loss_ex(dev_index, H, W, x.data_ptr(), bs, fh, fw, anc, na, nc, targets.data_ptr(), targets.shape[1],
                                             float(loss_mod.ignore_threshold), work.data_ptr(), work.numel() * 8, losses.data_ptr(),
                                             grad.data_ptr(), ctypes.c_void_p(stream)))
        ctx.save_for_backward(grad)
        ctx.mark_non_differentiable(losses)
        return losses[0].clone(), losses

    @staticmethod
    def backward(ctx, g_total, _g_losses):
        (grad,) = ctx.saved_tensors
        return grad * g_total, None, None


def _train_loss(self, input, targets):
    """yolo_loss.py:70-97 with targets: (loss, x, y, w, h, conf, cls) -- loss a 0-dim tensor whose backward() fills input.grad, the rest
    Python floats like the reference's .item() values.  Targets outside the feature map raise IndexError like the reference's indexing."""
    if input.dim() != 4 or input.shape[1] != self.num_anchors * self.bbox_attrs:
        raise RuntimeError("input must be [batch, %d x (5 + %d), h, w]" % (self.num_anchors, self.num_classes))   # yolo_loss.py:58's view
    if not input.is_cuda:
        raise RuntimeError("training loss (HIP) has no CPU implementation: pass GPU tensors")
    x = input if (input.is_contiguous() and input.dtype == torch.float32) else input.contiguous().float()
    t = targets.to(x.device).contiguous().float()
    if t.dim() != 3 or t.shape[0] != x.shape[0] or t.shape[2] != 6:
        raise ValueError("targets must be [batch, T, 6] = (x, y, w, h, class, marker)")
    loss, parts = _TrainLossFn.apply(x, t, self)
    v = parts.tolist()
    if v[7] > 0:
        raise IndexError("%d target(s) fall outside the %dx%d feature map" % (int(v[7]), x.shape[2], x.shape[3]))
    return loss, v[1], v[2], v[3], v[4], v[5], v[6]


YOLOLossV3._train_loss = _train_loss


def non_max_suppression(prediction, num_classes, conf_thres=0.5, nms_thres=0.4, kmax=None, model=None):
    """-> list with one [n,7] tensor (x1,y1,x2,y2,obj_conf,class_conf,class_pred) or None per image, like the reference.
    (The reference also overwrites prediction[..., :4] with the corners in place; this does not touch its input.)"""
    if not prediction.is_cuda:
        raise RuntimeError("validation path (HIP) has no CPU implementation: pass GPU tensors")
    p = prediction.contiguous().float()
    bs, M, attrs = p.shape
    if int(num_classes) < 1 or attrs != 5 + int(num_classes):
        raise ValueError("prediction rows have %d values, 5 + num_classes = %d" % (attrs, 5 + int(num_classes)))
    e = _model(model).engine_on(p.device)   # yf_val_nms is size-agnostic: any engine of that model on the tensor's device
    kmax = kmax or M
    det = torch.empty((bs, kmax, 7), dtype=torch.float32, device=p.device)
    cnt = torch.empty((bs,), dtype=torch.int32, device=p.device)
    stream = torch.cuda.current_stream(p.device).cuda_stream
    _lib.check(e.lib.yf_val_nms_ex(e.handle, p.data_ptr(), bs, M, int(num_classes), float(conf_thres), float(nms_thres), kmax, det.data_ptr(),
                                   cnt.data_ptr(), ctypes.c_void_p(stream)))
    counts = cnt.cpu().tolist()
    if max(counts) > kmax:
        raise OverflowError("more than kmax detections in an image")
    return [det[i, :n].clone() if n else None for i, n in enumerate(counts)]


def collate_fn(batch):
    """What the reference's DetectDataset.collate_fn does (dataloader/detect_dataset.py:106-117): stack (h,w,c) images and
    (64,6) boxes, NHWC -> NCHW, divide the images by 255."""
    import numpy as np
    images = np.concatenate([[img] for img, _ in batch], axis=0).transpose(0, 3, 1, 2)
    bboxes = np.concatenate([[box] for _, box in batch], axis=0)
    return torch.from_numpy(images).div(255.0), torch.from_numpy(bboxes)


class Validation:
    """Mirror of the reference's `Validation` (src/model_training/validate.py:8-122): same constructor, `get_mAP(model,
    epoch)`, `clear()`, same log lines.  Model forward, decode and NMS run on the GPU (this package's kernels, one batch at a
    time); the matching and the AP arithmetic are the reference's host bookkeeping with its exact conventions:
      * targets: normalised (xc, yc, w, h, cls, marker > 1) -> input-image corners (:112-122);
      * a prediction matches the FIRST remaining same-class target with IoU (+1 pixel convention, general.py:29-52) > the
        threshold; the matched target is removed (:62-70);
      * each class's list is sorted by the PRINTED form of the confidence ('tensor(0.8714)': the reference stores
        np.array([t[4], 'TP']), a string array, and sorts that, :77) -- 4 decimals, lexicographic, stable;
      * recall is float32 (target_num is a float32 tensor), precision a Python float; equal consecutive recalls keep the
        larger precision; AP = sum over the P-R points of (recall step) x (max precision from that point on) (:87-119).
    `dataset` is anything a DataLoader accepts; items are ((h,w,c) image, (64,6) boxes) like DetectDataset's, batched with
    `collate_fn` above (images / 255)."""

    def __init__(self, params, logger, dataset, device, model_loss):
        from torch.utils.data import DataLoader
        self.logger = logger
        self.model_loss = model_loss
        self.device = device
        self.bs = params["train_params"]["batch_size"]
        self.input_shape = params["io_params"]["input_shape"]
        self.num_cls = params["io_params"]["num_cls"]
        self.cls_name = params["io_params"]["class_names"]
        self.IOU_threshold = params["train_params"]["IOU_val_thre"]
        self.conf_thres = params["io_params"]["conf_thre"]
        self.nms_thres = params["io_params"]["nms_thre"]
        self.dataloader = DataLoader(dataset, batch_size=self.bs, num_workers=0, drop_last=True, pin_memory=True, shuffle=True,
                                     collate_fn=collate_fn)
        self.target_num = torch.zeros((self.num_cls))
        self.match_list = [[] for _ in range(self.num_cls)]

    def clear(self):
        self.match_list = [[] for _ in range(self.num_cls)]
        self.target_num.zero_()

    def _recover_targets(self, targets):
        in_h, in_w = self.input_shape[0], self.input_shape[1]
        t = targets.clone()
        t[:, :, (0, 2)] = t[:, :, (0, 2)] * in_w
        t[:, :, (1, 3)] = t[:, :, (1, 3)] * in_h
        out = t.clone()
        out[:, :, 0] = t[:, :, 0] - t[:, :, 2] / 2
        out[:, :, 1] = t[:, :, 1] - t[:, :, 3] / 2
        out[:, :, 2] = t[:, :, 0] + t[:, :, 2] / 2
        out[:, :, 3] = t[:, :, 1] + t[:, :, 3] / 2
        return out

    @staticmethod
    def _iou(t, targets):  # general.py:29-52, float32, one prediction against all remaining targets
        ix1 = torch.max(t[0], targets[:, 0]); iy1 = torch.max(t[1], targets[:, 1])
        ix2 = torch.min(t[2], targets[:, 2]); iy2 = torch.min(t[3], targets[:, 3])
        inter = torch.clamp(ix2 - ix1 + 1, min=0) * torch.clamp(iy2 - iy1 + 1, min=0)
        a1 = (t[2] - t[0] + 1) * (t[3] - t[1] + 1)
        a2 = (targets[:, 2] - targets[:, 0] + 1) * (targets[:, 3] - targets[:, 1] + 1)
        return inter / (a1 + a2 - inter + 1e-16)

    def _match_image(self, img_pred, img_target):
        """validate.py:47-74 for one image (host tensors): img_pred [n,7] NMS output or None, img_target [64,6] recovered."""
        img_target = img_target[img_target[:, 5] > 1]
        for t in img_target:
            self.target_num[int(t[4])] += 1
        if img_pred is None:
            return
        for c in img_pred[:, 6].unique():
            target_c = img_target[img_target[:, 4] == c]
            pred_c = img_pred[img_pred[:, 6] == c]
            c = int(c)
            for t in pred_c:
                hit = False
                if target_c.size(0):
                    over = (self._iou(t, target_c) > self.IOU_threshold).nonzero()
                    if over.numel():
                        index = int(over[0])
                        target_c = torch.cat((target_c[:index], target_c[index + 1:]), dim=0)
                        hit = True
                self.match_list[c].append((str(t[4]), hit))

    def get_mAP(self, model, epoch):
        self.clear()
        model.eval()
        for loss in self.model_loss:
            object.__setattr__(loss, "model", model)      # (not a registered submodule, see YOLOLossV3.__init__)
        with torch.no_grad():
            for imgs, targets in self.dataloader:
                targets = self._recover_targets(targets.float())                      # host: the bookkeeping stays on the CPU
                imgs = imgs.to(self.device).float()
                pred = model(imgs)
                output = torch.cat([self.model_loss[i](p) for i, p in enumerate(pred)], 1)
                output = non_max_suppression(output, self.num_cls, conf_thres=self.conf_thres, nms_thres=self.nms_thres, model=model)
                output = [None if o is None else o.cpu() for o in output]
                for img_id, img_pred in enumerate(output):
                    self._match_image(img_pred, targets[img_id])
            for c in range(self.num_cls):
                self.match_list[c].sort(key=lambda x: x[0], reverse=True)
            mAP = 0
            self.logger.info("—————— epoch: %d validation results —————" % (epoch))
            for c in range(self.num_cls):
                AP = self._calculate_AP(c)
                self.logger.info("class: %s, target_num = %d, AP = %.3f" % (self.cls_name[c], self.target_num[c], AP))
                mAP += AP
            mAP /= self.num_cls
            self.logger.info("mean AP: %.3f" % (mAP))
            self.logger.info("——————————————————————————")
        return mAP

    def _calculate_AP(self, cls):
        """validate.py:87-119 in one pass (running TP / FP counts instead of the reference's re-count per prefix)."""
        import numpy as np
        pr = []
        tp = fp = 0
        for _, is_tp in self.match_list[cls]:
            if is_tp:
                tp += 1
            else:
                fp += 1
            fn = self.target_num[cls] - tp
            if fn < 0:
                self.logger.error("error: FN less than 0!")
            precision = tp / (tp + fp)
            recall = float(tp / (tp + fn))          # float32 division, like the reference's tensor arithmetic
            if pr and recall == pr[-1][1]:
                if precision > pr[-1][0]:
                    pr[-1][0] = precision
            else:
                pr.append([precision, recall])
        ap, pre = 0, 0
        best = 0.0
        suffix_max = [0.0] * len(pr)
        for i in range(len(pr) - 1, -1, -1):
            best = max(best, pr[i][0])
            suffix_max[i] = best
        for i in range(len(pr)):
            ap += (np.float64(pr[i][1]) - pre) * suffix_max[i]
            pre = np.float64(pr[i][1])
        return ap
