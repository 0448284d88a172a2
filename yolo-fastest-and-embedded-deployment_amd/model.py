"""`YoloFastest(io_params)` -- the reference's module API (src/model_training/model/yolo_fastest.py:69-231)
with the forward pass executed by the HIP engine (csrc/, C ABI include/yolo_fastest_hip.h).

Drop-in surface kept:
  * ctor `YoloFastest(io_params)` reading num_cls / input_channel / num_anchors (:70-76);
  * the 508-key state-dict: `load_state_dict(torch.load(path, map_location=device))` works strictly
    (src/detect.py:90-91), `.to(device)` / `.eval()` chain (:89);
  * `model(x)`: x float32 [N,input_channel,H,W] NCHW, H and W multiples of 32 -> `(head_large, head_small)` float32
    NCHW [N, num_anchors * (5 + num_cls), ...] on the same device (:218).  Every num_cls and num_anchors (up to 8) of io_params is
    implemented, and every input_channel up to 64 (1 gray, 3 = cv2 BGR frames, detect.py:109-119; above 4 conv0 is a launch of its own).
The torch.nn modules below are parameter CONTAINERS only (they give the state-dict its names and shapes);
they are never called.  In eval mode forward() packs them once (BN fold, packer.py) and runs the tuned HIP engine; in train mode it
runs the training operators (training.py: batch-statistics BatchNorm, differentiable).  There is no CPU path: a non-GPU tensor or
a missing extension raises.
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib, packer


MAX_NUM_ANCHORS, MAX_NUM_CLS, MAX_INPUT_CHANNEL = 8, 4096, 64   # csrc/yf_layers.h (one set of limits for yf_create and yf_trainer_create_ex)
DEFAULT_FUSION = 2   # yf_set_fusion: 0 = one launch per layer (bring-up, every probe); 1 = block-fused kernels (round 2's plan);
                     # 2 = + the per-frame deep stage's launch boundaries removed (conv5_2 in the res5 launch, ...)


def _container(kind, cin, cout, k, stride, relu):
    if kind == packer.KIND_HEAD:
        return nn.Conv2d(cin, cout, kernel_size=1, stride=1)
    if kind == packer.KIND_DECONV:
        conv = nn.ConvTranspose2d(cin, cout, kernel_size=2, stride=2, padding=0, bias=False)
    else:
        conv = nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=(k - 1) // 2, bias=False,
                         groups=(cin if kind == packer.KIND_DW else 1))
    mods = [conv, nn.BatchNorm2d(cout)]
    if relu:
        mods.append(nn.ReLU())
    return nn.Sequential(*mods)


class _Engine:
    """One yf_handle (+ its workspace) for a given (H, W, device)."""

    def __init__(self, blob, H, W, max_batch, device_index, dtype=0):
        self.lib = _lib.lib()
        self.handle = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(blob, len(blob))
        _lib.check(self.lib.yf_create_ex(buf, len(blob), H, W, max_batch, device_index, dtype, ctypes.byref(self.handle)))
        self.dtype = dtype
        self.H, self.W, self.max_batch, self.device_index = H, W, max_batch, device_index
        self._ws = None
        self.chunk, self.lanes, self.branches = 0, 2, 1     # the C side's defaults (yf_engine: chunk 0, lanes 2, branches on)
        self.split_sums = 1
        self.post_split = 0

    def workspace(self, N, device):
        need = ctypes.c_size_t()
        _lib.check(self.lib.yf_workspace_bytes(self.handle, N, ctypes.byref(need)))
        if self._ws is None or self._ws.numel() < need.value or self._ws.device != device:
            self._ws = torch.empty(need.value, dtype=torch.uint8, device=device)
        return self._ws

    def set_chunk(self, frames):
        _lib.check(self.lib.yf_set_chunk(self.handle, int(frames)))
        self.chunk = int(frames)
        self._ws = None

    def set_lanes(self, lanes):
        _lib.check(self.lib.yf_set_lanes(self.handle, int(lanes)))
        self.lanes = int(lanes)
        self._ws = None

    def set_fusion(self, level):
        _lib.check(self.lib.yf_set_fusion(self.handle, int(level)))

    def set_branches(self, on):
        _lib.check(self.lib.yf_set_branches(self.handle, int(on)))
        self.branches = int(on)

    def set_split_sums(self, on):
        _lib.check(self.lib.yf_set_split_sums(self.handle, int(on)))
        self.split_sums = int(on)

    def set_post_split(self, mode):
        _lib.check(self.lib.yf_set_post_split(self.handle, int(mode)))
        self.post_split = int(mode)

    def close(self):
        if self.handle:
            self.lib.yf_destroy(self.handle)
            self.handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class YoloFastest(nn.Module):
    def __init__(self, io_params):
        super().__init__()
        self.num_cls = io_params["num_cls"]
        self.input_channel = io_params["input_channel"]
        num_anchor = io_params["num_anchors"]
        self.num_anchors = num_anchor
        self.num_out = num_anchor * (5 + self.num_cls)
        if not isinstance(self.input_channel, int) or not (1 <= self.input_channel <= MAX_INPUT_CHANNEL):
            raise NotImplementedError("the HIP engine implements input_channel 1 .. %d (1 .. 4: conv0 fused into the first kernel; more: a launch "
                                      "of its own), not %r" % (MAX_INPUT_CHANNEL, self.input_channel))
        if not (1 <= int(num_anchor) <= MAX_NUM_ANCHORS) or not (1 <= int(self.num_cls) <= MAX_NUM_CLS):   # csrc/yf_layers.h: engine AND trainer
            raise ValueError("num_anchors must be 1..%d and num_cls 1..%d" % (MAX_NUM_ANCHORS, MAX_NUM_CLS))
        # parameter containers, module-definition order of the reference (state-dict order follows it)
        blocks = {}
        for name, kind, cin, cout, k, stride, relu in packer.layer_table(self.num_out, self.input_channel):
            m = _container(kind, cin, cout, k, stride, relu)
            if "." in name:  # BasicResBlock: resX_Y.conv{1,2,3}
                blk, sub = name.split(".")
                if blk not in blocks:
                    blocks[blk] = nn.Module()
                    setattr(self, blk, blocks[blk])
                setattr(blocks[blk], sub, m)
            else:
                setattr(self, name, m)
        self._engines = {}
        self._blob = None
        self.chunk = 0  # frames per pass of the layer chain (0 = whole batch); see yf_set_chunk
        self.fusion = DEFAULT_FUSION
        self.lanes = 2   # concurrent streams over chunks of the batch (chunk 0 = one chunk per lane); see yf_set_lanes
        self.branches = 1  # 1: the small head's launches run on a side stream beside the large head's; see yf_set_branches
        # True (default): at <= 9 frames the stride-32 chain and the small head split their channel sums over several workgroups (batch-1 latency
        # 0.33 -> 0.29 ms).  Since round 6 the large-batch kernels form the same per-chunk partial sums in the same order, so a frame's bits never
        # depend on the batch it travels in EITHER WAY; False only selects the one-workgroup launches (A/B, profiling)
        # (yf_set_split_sums; DESIGN.md section 4 "Small batches").
        self.split_sums = True
        # cv2.cvtColor(BGR2GRAY)'s fixed-point coefficient set on the device (yf_cv_preprocess_u8): 15 = OpenCV 4.x's RGB2Gray<uchar> (9798 / 19235 /
        # 3735, shift 15: what a current `pip install opencv-python` gives the reference), 14 = OpenCV 2.x / 3.x (4899 / 9617 / 1868, shift 14).  They
        # differ by at most 1 LSB on colour frames; gray frames (the bundled test_data) come out identical.  io_params["gray_bits"] overrides.
        self.gray_bits = int(io_params.get("gray_bits", 15)) if isinstance(io_params, dict) else 15
        # decode + NMS as one workgroup per frame and class (dense frames) instead of one per frame: 0 = automatically when the caller reserves
        # kmax >= 256 survivors per frame, 1 = always, 2 = never (yf_set_post_split); the records are the same bit for bit
        self.post_split = 0
        # activation storage / pointwise-GEMM operand type: torch.float32, or torch.float16 (BASELINE configs[2]: fp16 in HBM,
        # fp16 MFMA, fp32 accumulate).  `model.half()` selects fp16 like it would for the reference module; setting
        # `model.storage_dtype = torch.float16` keeps the fp32 master weights for the BN fold (more accurate).
        self.storage_dtype = torch.float32
        # "f16x3": fp32 storage, the pointwise GEMMs on the fp16 matrix pipe with split (hi + lo) operands -- as accurate as fp32
        # (yf_create_ex dtype 2; include/yolo_fastest_hip.h).  None: decided by storage_dtype / the parameters' dtype.
        self.precision = None

    # -- weight packing -------------------------------------------------------------------------
    def _invalidate(self):
        self._blob = None
        self._blob_from_ncnn = False
        for e in self._engines.values():
            e.close()
        self._engines = {}

    def load_state_dict(self, state_dict, strict=True, **kw):
        r = super().load_state_dict(state_dict, strict=strict, **kw)
        self._invalidate()
        return r

    def _apply(self, fn, *a, **kw):
        r = super()._apply(fn, *a, **kw)
        keep = self._blob if getattr(self, "_blob_from_ncnn", False) else None
        self._invalidate()
        if keep is not None:
            self._blob, self._blob_from_ncnn = keep, True
        return r

    def load_ncnn(self, param_path, bin_path):
        """Alternative weight source: the reference's shipped ncnn model (models/ncnn/**, already BN-folded).
        The engine then runs these weights; the torch parameter containers are left untouched."""
        self._invalidate()
        self._blob = packer.pack_ncnn(param_path, bin_path, self.num_out, self.input_channel, self.num_anchors, self.num_cls)
        self._blob_from_ncnn = True
        return self

    def load_onnx(self, path):
        """Alternative weight source: the reference's ONNX export (models/onnx/**).  Unlike the ncnn file it still has the
        un-folded BatchNorm parameters, so this is a plain strict load_state_dict of the 508 keys read from the file."""
        self.load_state_dict(packer.read_onnx(path, self.num_out, self.input_channel))
        return self

    def refresh(self):
        """Re-pack after editing parameters in place (load_state_dict / .to() do it automatically)."""
        self._invalidate()

    def initialize_weights(self):  # yolo_fastest.py:220-231
        for m in self.modules():
            if type(m) is nn.Conv2d:
                torch.nn.init.kaiming_normal_(m.weight.data, nonlinearity="relu")
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif type(m) is nn.BatchNorm2d:
                m.weight.data.normal_(1.0, 0.02)
                m.bias.data.fill_(0)
        self._invalidate()

    def _dtype_code(self):
        p = self.conv0[0].weight
        if self.precision not in (None, "f32", "f16", "f16x3"):
            raise ValueError("precision must be None, 'f32', 'f16' or 'f16x3'")
        if self.precision == "f16x3":
            return 2
        return 1 if (self.precision == "f16" or self.storage_dtype == torch.float16 or p.dtype == torch.float16) else 0

    def engine(self, H, W, N, device, slot=0):
        """The engine (yf_handle + workspace + its side streams) for this input size on this device.  `slot` > 0: a further engine
        of the same kind, so that several batches can be in flight on different streams (pipeline.BatchPipeline): one engine serves
        one stream at a time."""
        if getattr(self, "_weights_dirty", False):        # a training forward has run since the last pack: fold the CURRENT parameters
            self._invalidate()
            self._weights_dirty = False
        key = (H, W, device.index if device.index is not None else torch.cuda.current_device(), self._dtype_code(), slot)
        e = self._engines.get(key)
        if e is None or e.max_batch < N:
            if e is not None:
                e.close()
            if self._blob is None:
                self._blob = packer.pack_state_dict(self.state_dict(), self.num_out, self.input_channel,
                                                    self.num_anchors, self.num_cls)
            e = _Engine(self._blob, H, W, max(N, 256), key[2], key[3])
            self._engines[key] = e
        # the knobs are re-applied on every call, so changing model.chunk / .lanes / .fusion after an engine exists takes effect
        if e.chunk != self.chunk:
            e.set_chunk(self.chunk)
        if e.lanes != self.lanes:
            e.set_lanes(self.lanes)
        e.set_fusion(self.fusion)
        if e.branches != self.branches:
            e.set_branches(self.branches)
        if e.split_sums != int(bool(self.split_sums)):
            e.set_split_sums(int(bool(self.split_sums)))
        if e.post_split != int(self.post_split):
            e.set_post_split(int(self.post_split))
        return e

    def engine_on(self, device):
        """Some engine of this model on `device` (for the size-agnostic entry points: yf_nms_sorted, yf_val_nms); one is created
        at the smallest legal input size if the model has not run there yet."""
        idx = device.index if device.index is not None else torch.cuda.current_device()
        for (H, W, d, dt, slot), e in self._engines.items():
            if d == idx and dt == self._dtype_code() and slot == 0:
                return e
        return self.engine(32, 32, 1, torch.device("cuda", idx))

    # -- forward --------------------------------------------------------------------------------
    def forward(self, x, slot=0):
        if self.training:
            # train.py:114 `pred = model(imgs)` after model.train(): batch-statistics BatchNorm, differentiable (training.py)
            from . import training
            self._weights_dirty = True           # the optimizer will move the parameters: re-pack at the next eval forward
            return training.forward(self, x)
        if not x.is_cuda:
            raise RuntimeError("YoloFastest (HIP engine) has no CPU path: move the model and input to the GPU")
        if x.dim() != 4 or x.shape[1] != self.input_channel or x.shape[2] % 32 or x.shape[3] % 32:
            raise ValueError("expected [N,%d,H,W] with H and W multiples of 32, got %s" % (self.input_channel, tuple(x.shape)))
        in_dtype = x.dtype
        x = x.contiguous().float()
        N, _, H, W = x.shape
        e = self.engine(H, W, N, x.device, slot)
        hl = torch.empty((N, self.num_out, H // 16, W // 16), dtype=torch.float32, device=x.device)
        hs = torch.empty((N, self.num_out, H // 32, W // 32), dtype=torch.float32, device=x.device)
        ws = e.workspace(N, x.device)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        _lib.check(e.lib.yf_forward(e.handle, x.data_ptr(), N, hl.data_ptr(), hs.data_ptr(), ws.data_ptr(), ws.numel(),
                                    ctypes.c_void_p(stream)))
        if in_dtype == torch.float16:  # a .half() reference module returns half heads
            return hl.half(), hs.half()
        return hl, hs

    def forward_bgr_u8(self, bgr, input_shape, gray_bits=None, slot=0):
        """model(__pre_process(frame)) for cv2.imread's frames (detect.py:108-127): uint8 GPU tensor [N,h,w,3] in BGR order and of ANY size
        -> cvtColor(BGR2GRAY) for a 1-channel net + cv2.resize to the net's input (yf_cv_preprocess_u8, OpenCV's 8-bit arithmetic) +
        (v - 128) / 255 fused into the first kernel -> (head_large, head_small).  gray_bits: OpenCV's 14- or 15-bit gray coefficients."""
        if self.training:
            raise RuntimeError("YoloFastest (HIP engine) is inference-only: call .eval() first")
        if not bgr.is_cuda or bgr.dtype != torch.uint8 or bgr.dim() != 4 or bgr.shape[3] != 3:
            raise ValueError("expected a uint8 GPU tensor [N,h,w,3] (cv2.imread's BGR frames)")
        H, W = int(input_shape[0]), int(input_shape[1])
        bgr = bgr.contiguous()
        N = bgr.shape[0]
        e = self.engine(H, W, N, bgr.device, slot)
        hl = torch.empty((N, self.num_out, H // 16, W // 16), dtype=torch.float32, device=bgr.device)
        hs = torch.empty((N, self.num_out, H // 32, W // 32), dtype=torch.float32, device=bgr.device)
        ws = e.workspace(N, bgr.device)
        stream = torch.cuda.current_stream(bgr.device).cuda_stream
        _lib.check(e.lib.yf_forward_bgr_u8(e.handle, bgr.data_ptr(), N, bgr.shape[1], bgr.shape[2], int(self.gray_bits if gray_bits is None else gray_bits), hl.data_ptr(), hs.data_ptr(),
                                           ws.data_ptr(), ws.numel(), ctypes.c_void_p(stream)))
        return hl, hs

    def cv_preprocess_u8(self, src, input_shape, gray_bits=None):
        """detect.py:110-116 alone (yf_cv_preprocess_u8): uint8 GPU frames [N,h,w] or [N,h,w,3] of any size -> the uint8 frames
        [N,H,W] ([N,H,W,3] for a 3-channel net) that `(img - 128.0) / 255.0` is applied to next."""
        if not src.is_cuda or src.dtype != torch.uint8 or src.dim() not in (3, 4):
            raise ValueError("expected a uint8 GPU tensor [N,h,w] or [N,h,w,3]")
        H, W = int(input_shape[0]), int(input_shape[1])
        src = src.contiguous()
        N, sc = src.shape[0], (1 if src.dim() == 3 else src.shape[3])
        e = self.engine(H, W, N, src.device)
        shape = (N, H, W) if self.input_channel == 1 else (N, H, W, self.input_channel)
        dst = torch.empty(shape, dtype=torch.uint8, device=src.device)
        stream = torch.cuda.current_stream(src.device).cuda_stream
        _lib.check(e.lib.yf_cv_preprocess_u8(e.handle, src.data_ptr(), N, src.shape[1], src.shape[2], sc, int(self.gray_bits if gray_bits is None else gray_bits), dst.data_ptr(),
                                             ctypes.c_void_p(stream)))
        return dst

    def forward_u8(self, u8, input_shape, slot=0):
        """model(preprocess(u8)) with the pre-process in front of the first kernel: u8 GPU tensor [N,h,w] of ANY size (the net's input size or
        exactly 2x: fused into the first kernel's loads; otherwise cv2.resize's INTER_LINEAR runs as one extra pass; input_channel 3:
        [N,h,w,3] as cv2.imread returns frames, BGR) -> (head_large, head_small)."""
        if self.training:
            raise RuntimeError("YoloFastest (HIP engine) is inference-only: call .eval() first")
        want = 3 if self.input_channel == 1 else 4
        if not u8.is_cuda or u8.dtype != torch.uint8 or u8.dim() != want or (want == 4 and u8.shape[3] != self.input_channel):
            raise ValueError("expected a uint8 GPU tensor [N,h,w]" + ("" if want == 3 else " + [%d] (HWC)" % self.input_channel))
        H, W = int(input_shape[0]), int(input_shape[1])
        u8 = u8.contiguous()
        N = u8.shape[0]
        e = self.engine(H, W, N, u8.device, slot)
        hl = torch.empty((N, self.num_out, H // 16, W // 16), dtype=torch.float32, device=u8.device)
        hs = torch.empty((N, self.num_out, H // 32, W // 32), dtype=torch.float32, device=u8.device)
        ws = e.workspace(N, u8.device)
        stream = torch.cuda.current_stream(u8.device).cuda_stream
        _lib.check(e.lib.yf_forward_u8(e.handle, u8.data_ptr(), N, u8.shape[1], u8.shape[2], hl.data_ptr(), hs.data_ptr(),
                                       ws.data_ptr(), ws.numel(), ctypes.c_void_p(stream)))
        return hl, hs

    def profile(self, x, reps=5, launch_repeats=1, return_heads=False):
        """Per-launch timing of one forward pass (HIP events on the launch stream around every kernel; launch_repeats > 1: around that
        many back-to-back launches of it, divided -- the event packets and the dispatch gap of a lone launch, 5-7 us, drop out).
        Returns a list of dicts: name, ms (mean over reps), algorithmic_bytes, flops (= mfma_flops + valu_flops, by the pipe
        the layer runs on in this plan) -- for the whole batch.  x uint8 [N,H,W] ([N,H,W,3]): the pass from u8 frames of the net's
        size (forward_u8's fused pre-process in the first launch)."""
        u8 = x.dtype == torch.uint8
        x = x.contiguous() if u8 else x.contiguous().float()
        N, H, W = (x.shape[0], x.shape[1], x.shape[2]) if u8 else (x.shape[0], x.shape[2], x.shape[3])
        e = self.engine(H, W, N, x.device)
        ws = e.workspace(N, x.device)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        n = ctypes.c_int()
        _lib.check(e.lib.yf_num_launches(e.handle, ctypes.byref(n)))
        acc = [0.0] * n.value
        buf = (ctypes.c_float * n.value)()
        _lib.check(e.lib.yf_set_profile_repeats(e.handle, int(launch_repeats)))
        for _ in range(reps):
            if u8:
                _lib.check(e.lib.yf_profile_forward_u8(e.handle, x.data_ptr(), N, H, W, ws.data_ptr(), ws.numel(), ctypes.c_void_p(stream),
                                                       buf, n.value))
            else:
                _lib.check(e.lib.yf_profile_forward(e.handle, x.data_ptr(), N, ws.data_ptr(), ws.numel(), ctypes.c_void_p(stream),
                                                    buf, n.value))
            for i in range(n.value):
                acc[i] += buf[i] / reps
        out = []
        for i in range(n.value):
            name = ctypes.create_string_buffer(512)
            b, fm, fv = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
            _lib.check(e.lib.yf_op_info_ex(e.handle, i, name, 512, ctypes.byref(b), ctypes.byref(fm), ctypes.byref(fv)))
            kdt, nd = ctypes.c_int(), ctypes.c_int()
            _lib.check(e.lib.yf_op_dtype(e.handle, i, ctypes.byref(kdt)))
            _lib.check(e.lib.yf_op_dispatches(e.handle, i, N, ctypes.byref(nd)))
            out.append(dict(name=name.value.decode(), ms=acc[i], dispatches=nd.value, algorithmic_bytes=b.value * N, flops=(fm.value + fv.value) * N,
                            mfma_flops=fm.value * N, valu_flops=fv.value * N, kernel_dtype=("f32", "f16", "f16x3")[kdt.value]))
        if return_heads:
            lo, so = ctypes.c_size_t(), ctypes.c_size_t()
            _lib.check(e.lib.yf_profile_head_offsets(e.handle, N, ctypes.byref(lo), ctypes.byref(so)))
            torch.cuda.current_stream(x.device).synchronize()
            raw = ws.view(torch.uint8)
            nl, ns = N * self.num_out * (H // 16) * (W // 16), N * self.num_out * (H // 32) * (W // 32)
            hl = raw[lo.value:lo.value + 4 * nl].view(torch.float32).view(N, self.num_out, H // 16, W // 16).clone()
            hs = raw[so.value:so.value + 4 * ns].view(torch.float32).view(N, self.num_out, H // 32, W // 32).clone()
            return out, (hl, hs)
        return out

    def probe(self, x, name):
        """Test hook: the activation the reference module attribute `name` produces, NCHW."""
        x = x.contiguous().float()
        N, _, H, W = x.shape
        e = self.engine(H, W, N, x.device)
        C, h, w = self.probe_shape(name, H, W)
        dst = torch.empty((N, C, h, w), dtype=torch.float32, device=x.device)
        ws = e.workspace(N, x.device)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        _lib.check(e.lib.yf_forward_probe(e.handle, x.data_ptr(), N, name.encode(), dst.data_ptr(),
                                          dst.numel() * 4, ws.data_ptr(), ws.numel(), ctypes.c_void_p(stream)))
        return dst

    def probe_shape(self, name, H, W):
        # spatial size by stage, channels from the layer table
        tab = {n: (cout, ) for n, _, _, cout, *_ in packer.layer_table(self.num_out, self.input_channel)}
        lname = name if name in tab else name + ".conv3"
        C = tab[lname][0]
        h, w = H, W
        for n, kind, cin, cout, k, stride, relu in packer.layer_table(self.num_out, self.input_channel):
            if n == "deconv5_1":
                h, w = H // 16, W // 16
            elif n == "conv5_3":
                h, w = H // 32, W // 32
            elif stride == 2:
                h, w = h // 2, w // 2
            if n == lname:
                break
        return C, h, w
