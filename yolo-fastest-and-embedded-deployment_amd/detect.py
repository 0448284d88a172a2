"""`Detect_YOLO` -- the reference's PC inference driver (src/detect.py:87-192), batched and on the GPU.

Same constructor and `batch_detect(data_path, result_path)`; one log line per image in the reference's format
(:177,190,192).  Differences that follow from the design, not from taste:
  * frames are processed `batch_size` at a time (the reference loops one image per iteration, :146);
  * the whole of `__pre_process` behind the file decode runs on the device: cvtColor(BGR2GRAY) + cv2.resize for ANY frame size
    (yf_cv_preprocess_u8: OpenCV's 8-bit arithmetic restated -- include/yolo_fastest_hip.h; exactly 2x = the 2x2 box mean) and (u8-128)/255
    (yf_preprocess_u8); image decode uses PIL (this image has no cv2) and hands over what cv2.imread would: HWC, BGR;
  * model + post-process are stream-ordered launches (yf_forward, yf_decode_nms); times in the log are
    per-batch wall times divided by the batch size.
Result writer: `result_<name>` images with the reference's boxes and labels (`plot.plot_one_box`, the
reference's general.py:56-67 without cv2) and the reference's log lines; `self.last_labels` keeps the label strings per image.
"""
import ctypes
import os
import time

import numpy as np
import torch

from . import _lib
from .model import YoloFastest
from .plot import plot_one_box
from .post_process import YOLO_post_process


def preprocess_u8(model, u8, input_shape):
    """Detect_YOLO.__pre_process arithmetic (detect.py:115-127) on device.
    u8: uint8 GPU tensor [N,h,w] (h,w == net input or exactly 2x) -> float32 [N,1,H,W]; a 3-channel model takes [N,h,w,3] as
    cv2.imread returns frames (BGR) -> float32 [N,3,H,W], channels reversed like `img[:, :, ::-1].transpose(2, 0, 1)` (:119)."""
    cin = model.input_channel
    want = 3 if cin == 1 else 4
    if not u8.is_cuda or u8.dtype != torch.uint8 or u8.dim() != want or (want == 4 and u8.shape[3] != cin):
        raise ValueError("expected a uint8 GPU tensor [N,h,w]" + ("" if cin == 1 else " + [%d] (HWC)" % cin))
    H, W = int(input_shape[0]), int(input_shape[1])
    N = u8.shape[0]
    e = model.engine(H, W, N, u8.device)
    x = torch.empty((N, cin, H, W), dtype=torch.float32, device=u8.device)
    u8 = u8.contiguous()
    stream = torch.cuda.current_stream(u8.device).cuda_stream
    _lib.check(e.lib.yf_preprocess_u8(e.handle, u8.data_ptr(), N, u8.shape[1], u8.shape[2], x.data_ptr(),
                                      ctypes.c_void_p(stream)))
    return x


class Detect_YOLO():
    def __init__(self, device, model_path, config_params, logger):
        self.model = YoloFastest(config_params["io_params"]).to(device).eval()
        net_param = torch.load(model_path, map_location=device)
        self.model.load_state_dict(net_param)
        self.logger = logger
        self.device = torch.device(device)
        io = config_params["io_params"]
        self.class_names = io["class_names"]
        self.num_cls = io["num_cls"]
        self.nms_thres = io["nms_thre"]
        self.conf_thres = io["conf_thre"]
        self.input_shape = io["input_shape"]
        self.origin_img_shape = io["origin_img_shape"]
        self.post_process = YOLO_post_process(conf_thres=self.conf_thres, nms_thres=self.nms_thres,
                                              num_anchors=io["num_anchors"], anchors=io["anchors"],
                                              input_shape=self.input_shape, num_class=self.num_cls).bind(self.model)
        self.colors = [[106, 90, 205], [199, 97, 20], [112, 128, 105]]

    def _read_bgr(self, path):
        """-> (what cv2.imread(path) returns: uint8 [h,w,3] in BGR order, detect.py:108; the RGB original for drawing).  BGR2GRAY for a
        1-channel net (:110-111) and the resize (:115-116) happen on the device (`_pre_process`)."""
        from PIL import Image
        if self.model.input_channel not in (1, 3):   # cv2.imread as detect.py:108-113 calls it yields 3 channels: nothing to mirror
            raise ValueError("image files decode to 3 channels; feed a %d-channel model through detect_u8" % self.model.input_channel)
        ori = np.asarray(Image.open(path).convert("RGB"))
        return np.ascontiguousarray(ori[:, :, ::-1]), ori

    def _pre_process(self, bgr):
        """detect.py:107-127 for a batch: uint8 GPU tensor [N,h,w,3] (BGR, any size) -> float32 [N,C,H,W].  The reference resizes when its
        CONFIGURED shapes differ (:115); a frame whose actual size is not the net's is resized as well (cv2.resize there would be the only
        way to feed it)."""
        u8 = self.model.cv_preprocess_u8(bgr, self.input_shape)
        return preprocess_u8(self.model, u8, self.input_shape)

    def detect_bgr_u8(self, bgr, kmax=64, gray_bits=None):
        """bgr: uint8 GPU tensor [N,h,w,3] as cv2.imread returns frames, any size.  The whole of detect.py:108-182 on the device; per-frame
        lists in the coordinates of `origin_img_shape` (after __adjust_coord, :131-139)."""
        pred = self.model.forward_bgr_u8(bgr, self.input_shape, gray_bits=gray_bits)
        origin = None
        if list(self.input_shape[0:2]) != list(self.origin_img_shape[0:2]):
            origin = self.origin_img_shape
        return self.post_process.detect(pred, kmax=kmax, origin_shape=origin)

    def detect_u8(self, u8, kmax=64):
        """u8: uint8 GPU tensor [N,h,w] in the ORIGINAL image geometry (any size: cv2.resize's arithmetic brings it to the net's). Returns
        per-frame lists in original coordinates (after __adjust_coord, detect.py:181-182)."""
        pred = self.model.forward_u8(u8, self.input_shape)  # pre-process in front of / fused into the first kernel's loads
        origin = None
        if list(self.input_shape[0:2]) != list(self.origin_img_shape[0:2]):
            origin = self.origin_img_shape
        return self.post_process.detect(pred, kmax=kmax, origin_shape=origin)

    def batch_detect(self, data_path, result_path, batch_size=256, in_flight=2):
        """detect.py:141-192 over a directory, `batch_size` images per pass.  More than one batch: the batches go through `BatchPipeline`
        (`in_flight` of them on the GPU at a time, each on its own stream and engine -- the throughput mode bench.py measures) while the host
        decodes the next batch's files and writes the previous batch's result images; the per-image times in the log lines are then the
        batch's DEVICE times (HIP events around model and post-process, divided by the frames of the batch).  Same results, same log format
        and order as one batch at a time (`in_flight=1`)."""
        img_list = sorted(os.listdir(data_path))   # the reference iterates os.listdir order; its logs show it sorted
        num = len(img_list)
        self.last_labels = {}
        origin = None
        if list(self.input_shape[0:2]) != list(self.origin_img_shape[0:2]):
            origin = self.origin_img_shape
        batches = [img_list[b0:b0 + batch_size] for b0 in range(0, num, batch_size)]
        state = {"avg": 0.0}

        def report(names, oris, results, infer_time, post_process_time):
            total_time = infer_time + post_process_time
            state["avg"] += total_time * len(names)
            for filename, ori, boxes in zip(names, oris, results):
                if len(boxes) == 0:
                    self.last_labels[filename] = self._save(os.path.join(result_path, "result_" + filename), ori, [])
                    self.logger.info("image_name:%s -> no targets, infer time:%.2fms, post_process time:%.2fms, "
                                     "total time:%.2fms" % (filename, infer_time, post_process_time, total_time))
                    continue
                self.last_labels[filename] = self._save(os.path.join(result_path, "result_" + filename), ori, boxes)
                self.logger.info("image_name:%s -> detect finished, infer time:%.2fms, post_process time:%.2fms, "
                                 "total time:%.2fms" % (filename, infer_time, post_process_time, total_time))

        def load(names):
            bgrs, oris = zip(*[self._read_bgr(os.path.join(data_path, n)) for n in names])
            return bgrs, oris, len({b.shape for b in bgrs}) == 1

        if len(batches) > 1 and in_flight > 1:
            from .pipeline import BatchPipeline
            saved = (self.model.lanes, self.model.branches)
            pipe = BatchPipeline(self.model, self.post_process, depth=in_flight, kmax=64, origin_shape=origin, lanes=1, branches=0)
            pending = []

            def finish(item):
                names, oris, ticket, ev = item
                raw = ticket.synchronize()
                most = int(raw["counts"].max().item()) if raw["counts"].numel() else 0
                if most > raw["boxes"].shape[1]:      # more survivors than the first attempt reserved: once more with room for all (post.detect does the same)
                    raw = self.post_process.detect_raw((raw["head_large"], raw["head_small"]), kmax=most, origin_shape=origin)
                results = self.post_process.to_lists(raw)
                report(names, oris, results, ev[0].elapsed_time(ev[1]) / len(names), ev[1].elapsed_time(ev[2]) / len(names))

            try:
                for names in batches:
                    bgrs, oris, same = load(names)
                    if same:       # cv2.imread-shaped frames of one size: cvtColor + resize + (v - 128) / 255 inside the batch's own stream
                        x = torch.from_numpy(np.stack(bgrs)).to(self.device)
                    else:          # frames of different sizes: the resize brings them to one
                        x = torch.cat([self._pre_process(torch.from_numpy(b[None]).to(self.device)) for b in bgrs])
                    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                    ev[0].record()

                    def mid(pred, ev=ev):
                        ev[1].record()
                        return pred
                    ticket = pipe.submit(x, mid=mid, then=lambda out, ev=ev: ev[2].record())
                    pending.append((names, oris, ticket, ev))
                    if len(pending) >= in_flight:
                        finish(pending.pop(0))
                while pending:
                    finish(pending.pop(0))
            finally:
                pipe.drain()
                self.model.lanes, self.model.branches = saved
        else:
            for names in batches:
                bgrs, oris, same = load(names)
                if same:
                    x = self._pre_process(torch.from_numpy(np.stack(bgrs)).to(self.device))
                else:
                    x = torch.cat([self._pre_process(torch.from_numpy(b[None]).to(self.device)) for b in bgrs])
                torch.cuda.synchronize(self.device)
                start_time = time.time()
                with torch.no_grad():
                    pred = self.model(x)
                torch.cuda.synchronize(self.device)
                time_mark = time.time()
                infer_time = (time_mark - start_time) * 1000 / len(names)
                results = self.post_process.detect(pred, origin_shape=origin)
                post_process_time = (time.time() - time_mark) * 1000 / len(names)
                report(names, oris, results, infer_time, post_process_time)
        self.logger.info("detect avg_time: %.2fms" % (state["avg"] / max(num, 1)))

    def _save(self, path, ori, boxes):
        """detect.py:184-190: one box + label per detection (plot.plot_one_box), then the image file.  Returns the label
        strings it drew ('%s %.2f' % (class name, conf * cls_score), :186)."""
        img = np.ascontiguousarray(ori).copy()
        labels = []
        for *xyxy, conf, cls_score, cls_pred in boxes:
            label = '%s %.2f' % (self.class_names[int(cls_pred)], conf * cls_score)
            # (the reference has three colours, :105, and would raise IndexError from the fourth class on: they repeat here)
            plot_one_box(xyxy, img, label=label, color=self.colors[int(cls_pred) % len(self.colors)], line_thickness=3)
            labels.append(label)
        if path is not None and os.path.isdir(os.path.dirname(path)):
            from PIL import Image
            Image.fromarray(img).save(path, quality=95)   # cv2.imwrite's default JPEG quality
        return labels
