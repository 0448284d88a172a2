"""`YOLO_post_process` -- the reference's post-process API (src/detect.py:14-84) executed on the GPU,
plus the batched entry the reference lacks (it handles batch element 0 only, :46).

Same constructor, same method names, same return conventions:
  decode_box(pred) -> list of [x1:int, y1:int, x2:int, y2:int, conf:float, cls_score:float, cls_index:int]
                      for batch element 0, in (head, anchor, row, col) order             (:41-67)
  non_maxium_supression(sorted_list) -> list; consumes its argument like the reference  (:69-84)
and `detect(pred)` = decode + class bucketing + stable sort + per-class NMS (+ optional __adjust_coord)
for EVERY frame of the batch in one kernel launch (yf_decode_nms).
"""
import ctypes

import numpy as np
import torch

from . import _lib


class YOLO_post_process:
    def __init__(self, conf_thres, nms_thres, num_anchors, num_class, anchors, input_shape):
        self.conf_thres = conf_thres
        self.nms_thres = nms_thres
        self.num_anchors = num_anchors
        self.bbox_attrs = 5 + num_class
        self.anchors = anchors
        self.input_shape = input_shape
        if not (1 <= int(num_anchors) <= 8) or int(num_class) < 1:
            raise ValueError("num_anchors must be 1..8 and num_class >= 1")
        if len(anchors) < 2 or any(len(head) < num_anchors for head in anchors[:2]):
            raise ValueError("anchors must hold at least num_anchors [w, h] pairs for each of the two heads (detect.py:51,63-64)")
        # [2][num_anchors][2]: the heads use anchors[head][0 .. num_anchors - 1] (detect.py:51,54,63-64)
        self._anc = (ctypes.c_double * (4 * num_anchors))(*[float(v) for head in anchors[:2] for a in head[:num_anchors] for v in a[:2]])
        self._model = None

    # the kernels need an engine handle (it carries H, W and the device); bind the model that made `pred`
    def bind(self, model):
        self._model = model
        return self

    def _engine(self, pred, slot=0):
        hl = pred[0]
        if not hl.is_cuda:
            raise RuntimeError("YOLO_post_process (HIP) has no CPU path: pass the GPU tensors the model returned")
        if self._model is None:
            raise RuntimeError("call post_process.bind(model) first (the engine handle carries H, W and the device)")
        H, W = hl.shape[2] * 16, hl.shape[3] * 16
        if [H, W] != list(self.input_shape[:2]):
            raise ValueError("pred is for a %dx%d input, input_shape says %s" % (H, W, self.input_shape[:2]))
        if hl.shape[1] != self.num_anchors * self.bbox_attrs or pred[1].shape[1] != hl.shape[1]:
            raise ValueError("pred has %d channels per cell, num_anchors x (5 + num_class) = %d"
                             % (hl.shape[1], self.num_anchors * self.bbox_attrs))   # detect.py:53's reshape would raise
        if (self._model.num_anchors, self._model.num_cls) != (self.num_anchors, self.bbox_attrs - 5):
            raise ValueError("the bound model was built for %d anchors x %d classes" % (self._model.num_anchors, self._model.num_cls))
        return self._model.engine(H, W, hl.shape[0], hl.device, slot)

    @staticmethod
    def record_views(rec, kmax):
        """The five result tensors as VIEWS of a packed record block int32 [N, 1 + 8 kmax] (yf_decode_nms_packed's output, the
        multi-GPU exchange's layout: count | boxes | scores as float bits | cls | src)."""
        n = rec.shape[0]
        return dict(counts=rec[:, 0], boxes=rec[:, 1:1 + 4 * kmax].view(n, kmax, 4),
                    scores=rec[:, 1 + 4 * kmax:1 + 6 * kmax].view(torch.float32).view(n, kmax, 2),
                    cls=rec[:, 1 + 6 * kmax:1 + 7 * kmax], src=rec[:, 1 + 7 * kmax:1 + 8 * kmax], records=rec)

    def detect_raw(self, pred, kmax=64, nms_thres=None, origin_shape=None, slot=0, packed=False):
        """Batched. Returns dict of GPU tensors: boxes [N,kmax,4] i32, scores [N,kmax,2] f32, cls, src [N,kmax] i32,
        counts [N] i32 (see include/yolo_fastest_hip.h for the conventions).  packed=True: the kernel writes ONE record block
        (`records`, int32 [N, 1 + 8 kmax]) and the five tensors are views of it -- what dist.all_gather_detections sends as is."""
        hl, hs = pred[0].contiguous(), pred[1].contiguous()
        e = self._engine(pred, slot)
        N, dev = hl.shape[0], hl.device
        if packed:
            rec = torch.empty((N, 1 + 8 * kmax), dtype=torch.int32, device=dev)
            oh, ow = (origin_shape[0], origin_shape[1]) if origin_shape is not None else (0, 0)
            _lib.check(e.lib.yf_decode_nms_packed(e.handle, hl.data_ptr(), hs.data_ptr(), N, float(self.conf_thres),
                                                  float(self.nms_thres if nms_thres is None else nms_thres), self._anc, int(oh), int(ow), kmax,
                                                  rec.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
            return self.record_views(rec, kmax)
        out = dict(boxes=torch.empty((N, kmax, 4), dtype=torch.int32, device=dev),
                   scores=torch.empty((N, kmax, 2), dtype=torch.float32, device=dev),
                   cls=torch.empty((N, kmax), dtype=torch.int32, device=dev),
                   src=torch.empty((N, kmax), dtype=torch.int32, device=dev),
                   counts=torch.empty((N,), dtype=torch.int32, device=dev))
        oh, ow = (origin_shape[0], origin_shape[1]) if origin_shape is not None else (0, 0)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(e.lib.yf_decode_nms(e.handle, hl.data_ptr(), hs.data_ptr(), N, float(self.conf_thres),
                                       float(self.nms_thres if nms_thres is None else nms_thres), self._anc, int(oh),
                                       int(ow), kmax, out["boxes"].data_ptr(), out["scores"].data_ptr(),
                                       out["cls"].data_ptr(), out["src"].data_ptr(), out["counts"].data_ptr(),
                                       ctypes.c_void_p(stream)))
        return out

    def detect_raw_from_input(self, x, kmax=64, origin_shape=None, packed=False):
        """model forward + decode + NMS in ONE C call (yf_detect): x float32 GPU tensor [N,input_channel,H,W] -> the same dict as
        detect_raw, plus the two head tensors.  packed: as in detect_raw (yf_detect_packed)."""
        if self._model is None:
            raise RuntimeError("call post_process.bind(model) first")
        if not x.is_cuda:
            raise RuntimeError("YOLO_post_process (HIP) has no CPU path")
        x = x.contiguous().float()
        N, _, H, W = x.shape
        e = self._model.engine(H, W, N, x.device)
        dev = x.device
        out = dict(boxes=torch.empty((N, kmax, 4), dtype=torch.int32, device=dev),
                   scores=torch.empty((N, kmax, 2), dtype=torch.float32, device=dev),
                   cls=torch.empty((N, kmax), dtype=torch.int32, device=dev),
                   src=torch.empty((N, kmax), dtype=torch.int32, device=dev),
                   counts=torch.empty((N,), dtype=torch.int32, device=dev),
                   head_large=torch.empty((N, self._model.num_out, H // 16, W // 16), dtype=torch.float32, device=dev),
                   head_small=torch.empty((N, self._model.num_out, H // 32, W // 32), dtype=torch.float32, device=dev))
        ws = e.workspace(N, dev)
        oh, ow = (origin_shape[0], origin_shape[1]) if origin_shape is not None else (0, 0)
        stream = torch.cuda.current_stream(dev).cuda_stream
        if packed:
            rec = torch.empty((N, 1 + 8 * kmax), dtype=torch.int32, device=dev)
            _lib.check(e.lib.yf_detect_packed(e.handle, x.data_ptr(), N, float(self.conf_thres), float(self.nms_thres), self._anc, int(oh),
                                              int(ow), kmax, rec.data_ptr(), out["head_large"].data_ptr(), out["head_small"].data_ptr(),
                                              ws.data_ptr(), ws.numel(), ctypes.c_void_p(stream)))
            return dict(self.record_views(rec, kmax), head_large=out["head_large"], head_small=out["head_small"])
        _lib.check(e.lib.yf_detect(e.handle, x.data_ptr(), N, float(self.conf_thres), float(self.nms_thres), self._anc, int(oh),
                                   int(ow), kmax, out["boxes"].data_ptr(), out["scores"].data_ptr(), out["cls"].data_ptr(),
                                   out["src"].data_ptr(), out["counts"].data_ptr(), out["head_large"].data_ptr(),
                                   out["head_small"].data_ptr(), ws.data_ptr(), ws.numel(), ctypes.c_void_p(stream)))
        return out

    @staticmethod
    def to_lists(raw, with_src=False, on_error="raise"):
        """GPU result -> per-frame Python lists in the reference's element format.
        A frame's count says how the frame ended (include/yolo_fastest_hip.h): 0 .. kmax survivors; more than kmax = the TRUE number, of
        which the first kmax are stored (capacity overflow); -2 = the reference's own ZeroDivisionError (detect.py:39: two zero-area boxes
        of one class compared).  on_error="raise" (default) behaves like the reference's loop, which dies at such a frame:
        ZeroDivisionError / OverflowError for the whole batch.  on_error="mark": the frame's entry is the exception INSTANCE and the other
        frames are returned -- what a caller that gathers many ranks' frames wants (dist.all_gather_detections: the count travels in the
        record block, so every rank sees the same status for every frame)."""
        if on_error not in ("raise", "mark"):
            raise ValueError("on_error must be 'raise' or 'mark'")
        counts = raw["counts"].cpu().numpy()
        kmax = raw["boxes"].shape[1]
        if on_error == "raise":
            if (counts == -2).any():
                raise ZeroDivisionError("division by zero")  # detect.py:39, two zero-area boxes compared
            if (counts > kmax).any():
                raise OverflowError("more than kmax=%d survivors in a frame (max %d): raise kmax" % (kmax, counts.max()))
        boxes, scores = raw["boxes"].cpu().numpy(), raw["scores"].cpu().numpy()
        cls, src = raw["cls"].cpu().numpy(), raw["src"].cpu().numpy()
        frames = []
        for f, n in enumerate(counts):
            if n == -2:
                frames.append(ZeroDivisionError("division by zero"))
                continue
            if n > kmax:
                frames.append(OverflowError("frame %d: %d survivors, capacity kmax=%d" % (f, n, kmax)))
                continue
            L = []
            for k in range(int(n)):
                e = [int(boxes[f, k, 0]), int(boxes[f, k, 1]), int(boxes[f, k, 2]), int(boxes[f, k, 3]),
                     float(scores[f, k, 0]), float(scores[f, k, 1]), int(cls[f, k])]
                if with_src:
                    e.append(int(src[f, k]))
                L.append(e)
            frames.append(L)
        return frames

    def detect(self, pred, kmax=64, origin_shape=None, with_src=False):
        """All frames: [[x1,y1,x2,y2,conf,cls_score,cls], ...] per frame, class-major like detect.py:162-169.  kmax is the capacity the
        first attempt reserves per frame; a frame with more survivors (the count is always the true number) makes this run the
        post-process once more with room for all of them -- the reference's lists have no capacity."""
        raw = self.detect_raw(pred, kmax=kmax, origin_shape=origin_shape)
        most = int(raw["counts"].max().item()) if raw["counts"].numel() else 0
        if most > kmax:
            raw = self.detect_raw(pred, kmax=most, origin_shape=origin_shape)
        return self.to_lists(raw, with_src=with_src)

    # -- reference-named methods ------------------------------------------------------------------
    def decode_box(self, pred):
        """detect.py:41-67: candidates of batch element 0 in decode order (pre-NMS)."""
        hl, hs = pred[0][:1], pred[1][:1]
        ncell = self.num_anchors * (hl.shape[2] * hl.shape[3] + hs.shape[2] * hs.shape[3])
        raw = self.detect_raw((hl, hs), kmax=ncell, nms_thres=float("inf"))  # iou > inf never: nothing suppressed
        lst = self.to_lists(raw, with_src=True)[0]
        lst.sort(key=lambda e: e[7])  # back to (head, anchor, row, col) order
        return [e[:7] for e in lst]

    def non_maxium_supression(self, bbox_list):
        """detect.py:69-84 on one class's conf-sorted list (mutates it like the reference's pop loop)."""
        n = len(bbox_list)
        if n == 0:
            return []
        if self._model is None:
            raise RuntimeError("call post_process.bind(model) first (the engine handle carries the device)")
        p = next(self._model.parameters())
        if not p.is_cuda:
            raise RuntimeError("YOLO_post_process (HIP) has no CPU path: the bound model is not on a GPU")
        e = self._model.engine_on(p.device)   # the bound model's device; yf_nms_sorted is size-agnostic
        dev = torch.device("cuda", e.device_index)
        boxes = torch.tensor([[int(b[0]), int(b[1]), int(b[2]), int(b[3])] for b in bbox_list], dtype=torch.int32,
                             device=dev)
        sup = torch.empty((n,), dtype=torch.int32, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(e.lib.yf_nms_sorted(e.handle, boxes.data_ptr(), n, float(self.nms_thres), sup.data_ptr(),
                                       ctypes.c_void_p(stream)))
        sup = sup.cpu().numpy()
        if n and sup[0] == -2:
            raise ZeroDivisionError("division by zero")
        results = [bbox_list[i] for i in range(n) if sup[i] == -1]
        last = max(i for i in range(n) if sup[i] == -1)
        # the reference leaves [last result] behind iff that result was the only element left when picked
        tail = [] if np.any(sup == last) else [bbox_list[last]]
        bbox_list[:] = tail
        return results
