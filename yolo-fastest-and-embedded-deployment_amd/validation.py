"""Validation-time decode + NMS on the GPU behind the reference's own names:
  `YOLOLossV3(anchors, num_classes, input_shape, device)(input)`   src/model_training/loss/yolo_loss.py:27-141 (decode branch)
  `non_max_suppression(prediction, num_classes, conf_thres, nms_thres)`   src/model_training/utils/general.py:87-143
Training (targets given) is out of scope and raises.  Both need the engine handle of the model that produced the heads
(`bind(model)`), which carries H, W and the device."""
import ctypes

import torch

from . import _lib

_bound = {"model": None}


def bind(model):
    _bound["model"] = model


def _engine(t, H, W):
    m = _bound["model"]
    if m is None:
        raise RuntimeError("call validation.bind(model) first")
    if not t.is_cuda:
        raise RuntimeError("validation path (HIP) has no CPU implementation: pass GPU tensors")
    return m.engine(H, W, t.shape[0], t.device)


class YOLOLossV3(torch.nn.Module):
    def __init__(self, anchors, num_classes, input_shape, device):
        super().__init__()
        self.anchors = anchors
        self.num_anchors = len(anchors)
        self.num_classes = num_classes
        self.bbox_attrs = 5 + num_classes
        self.input_shape = input_shape
        self.device = device
        if self.num_anchors != 3 or num_classes != 3:
            raise NotImplementedError("the HIP decode implements 3 anchors x 3 classes")

    def forward(self, input, targets=None):
        if targets is not None:
            raise NotImplementedError("the training loss is out of scope of this inference path (SURVEY.md 8f.4)")
        x = input.contiguous().float()
        bs, _, fh, fw = x.shape
        e = _engine(x, int(self.input_shape[0]), int(self.input_shape[1]))
        M = 3 * fh * fw
        out = torch.empty((bs, M, self.bbox_attrs), dtype=torch.float32, device=x.device)
        anc = (ctypes.c_double * 6)(*[float(v) for a in self.anchors for v in a])
        stream = torch.cuda.current_stream(x.device).cuda_stream
        _lib.check(e.lib.yf_val_decode_head(e.handle, x.data_ptr(), bs, fh, fw, anc, M, 0, out.data_ptr(), ctypes.c_void_p(stream)))
        return out


def non_max_suppression(prediction, num_classes, conf_thres=0.5, nms_thres=0.4, kmax=None):
    """-> list with one [n,7] tensor (x1,y1,x2,y2,obj_conf,class_conf,class_pred) or None per image, like the reference.
    (The reference also overwrites prediction[..., :4] with the corners in place; this does not touch its input.)"""
    if num_classes != 3:
        raise NotImplementedError("3 classes")
    p = prediction.contiguous().float()
    bs, M, _ = p.shape
    m = _bound["model"]
    if m is None or not m._engines:
        raise RuntimeError("call validation.bind(model) and run the model once first")
    e = next(iter(m._engines.values()))
    kmax = kmax or M
    det = torch.empty((bs, kmax, 7), dtype=torch.float32, device=p.device)
    cnt = torch.empty((bs,), dtype=torch.int32, device=p.device)
    stream = torch.cuda.current_stream(p.device).cuda_stream
    _lib.check(e.lib.yf_val_nms(e.handle, p.data_ptr(), bs, M, float(conf_thres), float(nms_thres), kmax, det.data_ptr(), cnt.data_ptr(),
                                ctypes.c_void_p(stream)))
    counts = cnt.cpu().tolist()
    if max(counts) > kmax:
        raise OverflowError("more than kmax detections in an image")
    return [det[i, :n].clone() if n else None for i, n in enumerate(counts)]
