"""The reference's TRAINING step on the GPU (second slice): `src/model_training/train.py:98-132`

    model.train()
    optimizer.zero_grad()
    pred = model(imgs)                                   # train-mode forward: BatchNorm on batch statistics, running stats updated
    loss = sum(model_loss[i](pred[i], targets)[0] ...)   # validation.YOLOLossV3 (yf_train_loss)
    loss.backward()                                      # backward of every layer -> param.grad
    optimizer.step()                                     # training.Adam (yf_train_adam_multi) -- or any torch optimizer

`YoloFastest.forward` routes here when the module is in train mode.  Forward and backward of the whole network are ONE C call each
(`yf_trainer_forward` / `yf_trainer_backward`: the graph walked in C++ over a workspace that is the pass's tape); `model.train_impl =
"ops"` selects the same computation orchestrated from Python, one C call per block (bring-up, probing).  Every operator is a HIP
kernel behind the C ABI (`yf_train_*`, csrc/yf_train_kernels.hip): Conv2d / ConvTranspose2d forward, backward-data, backward-weight; BatchNorm2d in train
mode with its backward (ReLU fused); channel slices for the torch.cat; Adam.  torch.autograd only carries the gradient across the
boundary (one Function for the whole network whose inputs are the parameters) -- no torch operator computes anything.
NCHW fp32 like the reference -- not the tuned inference engine (which folds BatchNorm into the weights and so cannot train); kernels
and measurements: DESIGN.md section 4 "The training step".  `train()` is the reference's epoch loop around the iteration,
`data_parallel()` the one-all-reduce gradient exchange for one process per GPU.  No CPU path.
"""
import ctypes

import torch

from . import _lib

_SEQ1 = ["conv0", "conv1_2", "conv1_3", "conv1_4", "res1_1", "conv1_8", "conv1_9", "conv2_1", "res2_1", "res2_2", "conv2_2", "conv2_3",
         "conv3_1", "res3_1", "res3_2", "conv3_2", "conv3_3", "conv3_4", "res3_3", "res3_4", "res3_5", "res3_6", "conv3_5", "conv3_6",
         "conv4_1", "res4_1", "res4_2", "res4_3", "res4_4", "conv4_2"]                               # yolo_fastest.py:151-190
_SEQ2 = ["conv4_3", "conv5_1", "res5_1", "res5_2", "res5_3", "res5_4", "res5_5", "conv5_2"]          # :191-200
_SEQ3 = ["conv5_3", "conv5_4", "conv5_5", "conv5_6"]                                                 # :201-204
_SEQ4 = ["conv4_1_1", "conv4_1_2", "conv4_1_3", "conv4_1_4", "conv4_1_5"]                            # :211-215


class _Ops:
    """ctypes front of the yf_train_* entry points for one device / stream."""

    def __init__(self, device):
        self.lib = _lib.lib()
        self.device = device
        self.dev = device.index if device.index is not None else torch.cuda.current_device()
        self.stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        sc = _scratch(self.lib, device)
        self.scratch, self.scratch_bytes = sc.data_ptr(), sc.numel()
        self.bn_scratch = self.scratch

    def new(self, *shape):
        return torch.empty(shape, dtype=torch.float32, device=self.device)

    def call(self, name, *args):
        rc = getattr(self.lib, name)(self.dev, *args, self.stream)
        if rc:
            _lib.check(rc)


_SCRATCH = {}


def _scratch(lib, device):
    """The scratch of the split reductions (include/yolo_fastest_hip.h: yf_train_scratch_bytes), one per (device, stream)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream)
    if key not in _SCRATCH:
        n = ctypes.c_size_t()
        _lib.check(lib.yf_train_scratch_bytes(ctypes.byref(n)))
        _SCRATCH[key] = torch.empty(n.value, dtype=torch.uint8, device=device)
    return _SCRATCH[key]


def _ptr(t):
    return t.data_ptr() if t is not None else None


def _conv_geom(conv, x):
    k, stride = conv.kernel_size[0], conv.stride[0]
    N, Cin, H, W = x.shape
    Cout = conv.out_channels
    dw = 1 if conv.groups > 1 else 0
    if dw and not (conv.groups == Cin == Cout):
        raise NotImplementedError("grouped convolution other than depthwise")
    pad = (k - 1) // 2
    return N, Cin, H, W, Cout, k, stride, dw, (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1


def _unit_forward(ops, mod, x, tape, name):
    """conv_norm[_relu] / deconv_norm_relu (yolo_fastest.py:16-48) in train mode: one C call (yf_train_unit_forward)."""
    conv, bn = mod[0], mod[1]
    relu = 1 if len(mod) == 3 else 0
    N, Cin, H, W = x.shape
    Cout = conv.out_channels
    if isinstance(conv, torch.nn.ConvTranspose2d):
        deconv, k, stride, dw, Ho, Wo = 1, 2, 2, 0, 2 * H, 2 * W
    else:
        deconv = 0
        _, _, _, _, _, k, stride, dw, Ho, Wo = _conv_geom(conv, x)
    zy = ops.new(2, N, Cout, Ho, Wo)
    z, y = zy[0], zy[1]
    stats = ops.new(2 * Cout)
    ops.call("yf_train_unit_forward", deconv, x.data_ptr(), conv.weight.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(), _ptr(bn.running_mean),
             _ptr(bn.running_var), stats.data_ptr(), z.data_ptr(), y.data_ptr(), N, Cin, H, W, Cout, k, stride, dw, relu, ops.scratch)
    if bn.num_batches_tracked is not None:
        tape.setdefault("_nbt", []).append(bn.num_batches_tracked)
    tape[name] = (x, z, y, stats, relu, (deconv, k, stride, dw))
    return y


def _unit_backward(ops, mod, tape, name, gy, grads, need_dx=True):
    conv, bn = mod[0], mod[1]
    x, z, y, stats, relu, (deconv, k, stride, dw) = tape[name]
    N, Cin, H, W = x.shape
    Cout = z.shape[1]
    gz = torch.empty_like(z)
    dgb = ops.new(2, Cout)
    dw_ = torch.empty_like(conv.weight)
    gx = torch.empty_like(x) if need_dx else None
    ops.call("yf_train_unit_backward", deconv, x.data_ptr(), z.data_ptr(), gy.data_ptr(), stats.data_ptr(), conv.weight.data_ptr(),
             bn.weight.data_ptr(), bn.bias.data_ptr(), dgb[0].data_ptr(), dgb[1].data_ptr(), gz.data_ptr(), dw_.data_ptr(), _ptr(gx), N, Cin, H, W, Cout,
             k, stride, dw, relu, ops.scratch, ops.scratch_bytes)
    grads[bn.weight], grads[bn.bias], grads[conv.weight] = dgb[0], dgb[1], dw_
    return gx


def _res_forward(ops, blk, x, tape, name):          # BasicResBlock.forward, yolo_fastest.py:60-66
    y = _unit_forward(ops, blk.conv1, x, tape, name + ".conv1")
    y = _unit_forward(ops, blk.conv2, y, tape, name + ".conv2")
    y = _unit_forward(ops, blk.conv3, y, tape, name + ".conv3")
    ops.call("yf_train_add", y.data_ptr(), x.data_ptr(), y.data_ptr(), y.numel())     # out += residual (conv3 has no ReLU: its backward
    return y                                                                            # does not read y)


def _res_backward(ops, blk, tape, name, g, grads):
    gx = _unit_backward(ops, blk.conv3, tape, name + ".conv3", g, grads)
    gx = _unit_backward(ops, blk.conv2, tape, name + ".conv2", gx, grads)
    gx = _unit_backward(ops, blk.conv1, tape, name + ".conv1", gx, grads)
    ops.call("yf_train_add", gx.data_ptr(), g.data_ptr(), gx.data_ptr(), gx.numel())
    return gx


def _run(ops, model, names, x, tape):
    for n in names:
        m = getattr(model, n)
        x = _res_forward(ops, m, x, tape, n) if n.startswith("res") else _unit_forward(ops, m, x, tape, n)
    return x


def _run_back(ops, model, names, g, tape, grads, first_needs_dx=True):
    for i in range(len(names) - 1, -1, -1):
        n = names[i]
        m = getattr(model, n)
        if n.startswith("res"):
            g = _res_backward(ops, m, tape, n, g, grads)
        else:
            g = _unit_backward(ops, m, tape, n, g, grads, need_dx=(i > 0 or first_needs_dx))
    return g


def _head_forward(ops, conv, x, tape, name):        # nn.Conv2d(C, num_out, 1) with bias, yolo_fastest.py:136 / :146
    N, Cin, H, W = x.shape
    y = ops.new(N, conv.out_channels, H, W)
    ops.call("yf_train_conv_forward", x.data_ptr(), conv.weight.data_ptr(), conv.bias.data_ptr(), y.data_ptr(), N, Cin, H, W, conv.out_channels,
             1, 1, 0)
    tape[name] = x
    return y


def _head_backward(ops, conv, tape, name, gy, grads):
    x = tape[name]
    N, Cin, H, W = x.shape
    Cout = conv.out_channels
    gy = gy.contiguous()
    dw_, db, gx = torch.empty_like(conv.weight), ops.new(Cout), torch.empty_like(x)
    ops.call("yf_train_conv_backward_weight", x.data_ptr(), gy.data_ptr(), dw_.data_ptr(), N, Cin, H, W, Cout, 1, 1, 0, ops.scratch, ops.scratch_bytes)
    ops.call("yf_train_channel_sum_split", gy.data_ptr(), db.data_ptr(), N, Cout, H * W, ops.scratch)
    ops.call("yf_train_conv_backward_data", gy.data_ptr(), conv.weight.data_ptr(), gx.data_ptr(), N, Cin, H, W, Cout, 1, 1, 0)
    grads[conv.weight], grads[conv.bias] = dw_, db
    return gx


def train_forward(model, x):
    """YoloFastest.forward (yolo_fastest.py:150-218) in train mode -> (head_large, head_small, tape)."""
    ops = _Ops(x.device)
    tape = {}
    a = _run(ops, model, _SEQ1, x, tape)                       # conv4_2
    b = _run(ops, model, _SEQ2, a, tape)                       # conv5_2
    c = _run(ops, model, _SEQ3, b, tape)
    hs = _head_forward(ops, model.head_5, c, tape, "head_5")
    d = _unit_forward(ops, model.deconv5_1, b, tape, "deconv5_1")
    N, Ca, H, W = a.shape
    Cd = d.shape[1]
    cat = ops.new(N, Ca + Cd, H, W)                            # torch.cat((conv4_2, deconv5_1), 1)
    ops.call("yf_train_channel_slice", a.data_ptr(), cat.data_ptr(), N, Ca, H * W, Ca, 0, Ca + Cd, 0)
    ops.call("yf_train_channel_slice", d.data_ptr(), cat.data_ptr(), N, Cd, H * W, Cd, 0, Ca + Cd, Ca)
    e = _run(ops, model, _SEQ4, cat, tape)
    hl = _head_forward(ops, model.head_4, e, tape, "head_4")
    tape["_cat"] = (Ca, Cd, H, W)
    if tape.get("_nbt"):
        torch._foreach_add_(tape.pop("_nbt"), 1)               # the 84 num_batches_tracked counters
    return hl, hs, tape


def train_backward(model, tape, g_hl, g_hs):
    """Backward of train_forward -> {parameter: gradient}."""
    ops = _Ops(g_hl.device)
    grads = {}
    Ca, Cd, H, W = tape["_cat"]
    N = g_hl.shape[0]
    g = _head_backward(ops, model.head_4, tape, "head_4", g_hl.float(), grads)
    g_cat = _run_back(ops, model, _SEQ4, g, tape, grads)
    g_a2, g_d = ops.new(N, Ca, H, W), ops.new(N, Cd, H, W)
    ops.call("yf_train_channel_slice", g_cat.data_ptr(), g_a2.data_ptr(), N, Ca, H * W, Ca + Cd, 0, Ca, 0)
    ops.call("yf_train_channel_slice", g_cat.data_ptr(), g_d.data_ptr(), N, Cd, H * W, Ca + Cd, Ca, Cd, 0)
    g_b2 = _unit_backward(ops, model.deconv5_1, tape, "deconv5_1", g_d, grads)
    g = _head_backward(ops, model.head_5, tape, "head_5", g_hs.float(), grads)
    g_b = _run_back(ops, model, _SEQ3, g, tape, grads)
    ops.call("yf_train_add", g_b.data_ptr(), g_b2.data_ptr(), g_b.data_ptr(), g_b.numel())
    g_a = _run_back(ops, model, _SEQ2, g_b, tape, grads)
    ops.call("yf_train_add", g_a.data_ptr(), g_a2.data_ptr(), g_a.data_ptr(), g_a.numel())
    _run_back(ops, model, _SEQ1, g_a, tape, grads, first_needs_dx=False)     # the images need no gradient
    return grads


class _Trainer:
    """yf_trainer handle for one (H, W, device): the whole forward / backward as one C call each (include/yolo_fastest_hip.h)."""

    def __init__(self, H, W, device, input_channel=1, num_out=24):
        self.lib = _lib.lib()
        self.dev = device.index if device.index is not None else torch.cuda.current_device()
        self.handle = ctypes.c_void_p()
        _lib.check(self.lib.yf_trainer_create_ex(H, W, self.dev, int(input_channel), int(num_out), ctypes.byref(self.handle)))
        n, nb = ctypes.c_int(), ctypes.c_int()
        _lib.check(self.lib.yf_trainer_num_params(self.handle, ctypes.byref(n), ctypes.byref(nb)))
        self.n_params, self.n_bn = n.value, nb.value
        self._bytes = {}

    def layout(self, params):
        """sizes, shapes and byte offsets of the parameters in one flat buffer (the shapes of a YoloFastest never change)."""
        lay = self.__dict__.get("_layout")
        if lay is None or len(lay["sizes"]) != len(params):
            sizes = [p.numel() for p in params]
            offs, o = [], 0
            for n in sizes:
                offs.append(4 * o)
                o += n
            lay = self._layout = dict(sizes=sizes, shapes=[tuple(p.shape) for p in params], byte_offsets=offs, total=o)
        return lay

    def workspace(self, N, device):
        if N not in self._bytes:
            need = ctypes.c_size_t()
            _lib.check(self.lib.yf_trainer_workspace_bytes(self.handle, N, ctypes.byref(need)))
            self._bytes[N] = need.value
        return torch.empty(self._bytes[N], dtype=torch.uint8, device=device)     # one per forward: it is that pass's tape

    def __del__(self):
        try:
            if self.handle:
                self.lib.yf_trainer_destroy(self.handle)
        except Exception:
            pass


def _trainer(model, H, W, device):
    key = (H, W, device.index if device.index is not None else torch.cuda.current_device())
    cache = model.__dict__.setdefault("_trainers", {})
    if key not in cache:
        cache[key] = _Trainer(H, W, device, model.input_channel, model.num_out)
    return cache[key]


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def _structure(model):
    """The model's parameter slots and BatchNorm modules, found once: at the reference's batch 16 an iteration is 4 ms of GPU work, and
    walking 430 modules three times per forward (parameters(), named_modules(), modules()) plus 170 `Module.__getattr__` lookups cost
    2 ms of host time -- more than the forward's kernels.  The slots are (module._parameters, name) pairs in parameters() order, so a
    replaced Parameter object is still picked up; the module SET is taken to be fixed (YoloFastest builds it in __init__)."""
    st = model.__dict__.get("_yf_struct")
    if st is None:
        slots, seen = [], set()
        for mod in model.modules():
            for name, prm in mod._parameters.items():
                if prm is not None and id(prm) not in seen:
                    seen.add(id(prm))
                    slots.append((mod._parameters, name))
        st = model.__dict__["_yf_struct"] = dict(slots=slots, bns=[m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)],
                                                 bn_names=[n for n, m in model.named_modules() if isinstance(m, torch.nn.BatchNorm2d)])
    return st


class _TrainForwardFn(torch.autograd.Function):
    """The whole network as one autograd node: inputs = the parameters, outputs = the two heads."""

    @staticmethod
    def forward(ctx, x, model, *params):
        if getattr(model, "train_impl", "trainer") == "ops":          # one C call per block, orchestrated here (kept for bring-up / probing)
            hl, hs, tape = train_forward(model, x)
            ctx.model, ctx.tape, ctx.params, ctx.tr = model, tape, params, None
            return hl, hs
        hl, hs, tr, (ws, pp) = _trainer_forward(model, x, params)
        ctx.model, ctx.tape, ctx.params, ctx.tr, ctx.params_ptr = model, (x, ws), params, tr, pp
        return hl, hs

    @staticmethod
    def backward(ctx, g_hl, g_hs):
        if ctx.tape is None:
            raise RuntimeError("backward through the training forward a second time: its saved activations were freed")
        if ctx.tr is None:
            grads = train_backward(ctx.model, ctx.tape, g_hl.contiguous(), g_hs.contiguous())
            ctx.tape = None
            return (None, None) + tuple(grads[p] for p in ctx.params)
        x, ws = ctx.tape
        tr, params = ctx.tr, ctx.params
        g_hl, g_hs = g_hl.contiguous().float(), g_hs.contiguous().float()
        # The C call first, the 256 gradient views after it: this runs right behind the loss's device-to-host read (the reference's
        # .item() calls), with the GPU idle until the first backward kernel is queued.  The gradients are one flat parameters()-order
        # buffer (what the trainer's single multi-tensor sum of the split weight gradients needs): pointers = base + cached offsets.
        lay = tr.layout(params)
        flat = torch.empty(lay["total"], dtype=torch.float32, device=x.device)
        base = flat.data_ptr()
        gp = (ctypes.c_void_p * len(params))(*[base + o for o in lay["byte_offsets"]])
        _lib.check(tr.lib.yf_trainer_backward(tr.handle, x.data_ptr(), g_hl.data_ptr(), g_hs.data_ptr(), x.shape[0], ctx.params_ptr,
                                              gp, ws.data_ptr(), ws.numel(),
                                              ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)))
        grads = [g.view(shp) for g, shp in zip(flat.split(lay["sizes"]), lay["shapes"])]
        ctx.tape = None
        sync = getattr(ctx.model, "_grad_sync", None)
        if sync is not None:                                     # data_parallel(): one all-reduce of the flat gradient buffer
            from . import dist as yfd
            yfd.all_reduce_mean_(flat, sync[0])
        return (None, None) + tuple(grads)


def _trainer_forward(model, x, params):
    N, _, H, W = x.shape
    tr = _trainer(model, H, W, x.device)
    if len(params) != tr.n_params:
        raise RuntimeError("the model has %d parameters, the trainer expects %d" % (len(params), tr.n_params))
    bns = _structure(model)["bns"]
    if len(bns) != tr.n_bn:
        raise RuntimeError("the model has %d BatchNorm layers, the trainer expects %d" % (len(bns), tr.n_bn))
    bufs = None
    bb = [b._buffers for b in bns]                           # (the dicts, not the attributes: Module.__getattr__ is 3 us a lookup)
    if all(d.get("running_mean") is not None for d in bb):
        bufs = (ctypes.c_void_p * (2 * len(bns)))(*[d[k].data_ptr() for d in bb for k in ("running_mean", "running_var")])
    hl = torch.empty((N, model.num_out, H // 16, W // 16), dtype=torch.float32, device=x.device)
    hs = torch.empty((N, model.num_out, H // 32, W // 32), dtype=torch.float32, device=x.device)
    ws = tr.workspace(N, x.device)
    pp = _ptr_array(params)
    _lib.check(tr.lib.yf_trainer_forward(tr.handle, x.data_ptr(), N, pp, bufs, hl.data_ptr(), hs.data_ptr(), ws.data_ptr(),
                                         ws.numel(), ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)))
    nbt = [d["num_batches_tracked"] for d in bb if d.get("num_batches_tracked") is not None]
    if nbt:
        torch._foreach_add_(nbt, 1)
    return hl, hs, tr, (ws, pp)


def forward(model, x):
    """model(imgs) in train mode (train.py:114)."""
    if not x.is_cuda:
        raise RuntimeError("YoloFastest training (HIP) has no CPU path: move the model and input to the GPU")
    if x.dim() != 4 or x.shape[1] != model.input_channel or x.shape[2] % 32 or x.shape[3] % 32:
        raise ValueError("expected [N,%d,H,W] with H and W multiples of 32, got %s" % (model.input_channel, tuple(x.shape)))
    st = _structure(model)
    params = [d[n] for d, n in st["slots"]]
    f32 = torch.float32
    for p in params:
        if p.dtype is not f32 or not p.is_cuda or not p.is_contiguous():
            raise RuntimeError("training runs on contiguous float32 GPU parameters")
    x = x.contiguous().float()
    for name, b in zip(st["bn_names"], st["bns"]):
        # the BatchNorm kernels (and the backward's bit-exact recomputation of the ReLU mask) are built for nn.BatchNorm2d's defaults,
        # which is what the reference uses (yolo_fastest.py:12-13): anything else would train silently wrong
        if b.eps != 1e-5 or b.momentum != 0.1 or not b.affine or not b.track_running_stats:
            raise NotImplementedError("%s: the training kernels implement BatchNorm2d(eps=1e-5, momentum=0.1, affine=True, "
                                      "track_running_stats=True) only (got eps=%r, momentum=%r, affine=%r, track_running_stats=%r)"
                                      % (name, b.eps, b.momentum, b.affine, b.track_running_stats))
    if torch.is_grad_enabled() and any(p.requires_grad for p in params):
        return _TrainForwardFn.apply(x, model, *params)
    if getattr(model, "train_impl", "trainer") == "ops":
        hl, hs, _ = train_forward(model, x)
    else:
        hl, hs, _, _ = _trainer_forward(model, x, tuple(params))
    return hl, hs


def data_parallel(model, process_group=None, broadcast=True):
    """One process per GPU (torch.distributed, backend "nccl" = RCCL): every rank trains on its shard of the batch; after this call
    `loss.backward()` averages the gradients over the ranks with ONE all-reduce of the flat gradient buffer before they reach
    `param.grad`, and the parameters and buffers are broadcast from rank 0 once.  BatchNorm uses each rank's own batch statistics
    (like torch's DistributedDataParallel without SyncBatchNorm), and -- UNLIKE DistributedDataParallel, which re-broadcasts rank 0's
    buffers before every forward -- the running_mean / running_var buffers then drift apart between the ranks: call
    `sync_buffers(model)` before validating or saving so that every rank holds rank 0's.  `train()` below does that, shards the
    data with a DistributedSampler and writes checkpoints on rank 0 only when torch.distributed is initialised.
    The reference has no distributed training; with one rank this changes nothing."""
    import torch.distributed as tdist
    if not tdist.is_initialized():
        raise RuntimeError("initialise torch.distributed first (one process per GPU)")
    from . import dist as yfd
    if broadcast:
        yfd.broadcast_model_(model, 0, process_group)
    model._grad_sync = (process_group,)
    model.train_impl = "trainer"
    return model


def sync_buffers(model, process_group=None):
    """Every rank takes rank 0's BatchNorm running statistics (what DistributedDataParallel's per-forward buffer broadcast amounts to
    at the points where the buffers are used: validation and checkpoints)."""
    import torch.distributed as tdist
    if tdist.is_initialized() and tdist.get_world_size(process_group) > 1:
        with torch.no_grad():
            for b in model.buffers():
                tdist.broadcast(b, src=0, group=process_group)
    return model


class Adam(torch.optim.Optimizer):
    """`optim.Adam(model.parameters(), lr=lr0, betas=(0.9, 0.999), eps=1e-08)` (train.py:84) with the update done by yf_train_adam_multi (one launch for all tensors).
    A torch.optim.Optimizer, so `param_groups[..]['lr']` edits (train.py:106-109) and `lr_scheduler.LambdaLR` (:89) work unchanged; the
    state uses torch's names (step, exp_avg, exp_avg_sq)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self.__dict__.pop("_groups", None)               # the cached state tensors may have been replaced

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("closure")
        lib = _lib.lib()
        f32 = torch.float32
        for gi, group in enumerate(self.param_groups):
            b1, b2 = group["betas"]
            ps = group["params"]
            gs = [p.grad for p in ps]
            if any(g is None for g in gs):               # (not `None in gs`: that compares tensors with ==)
                ps = [p for p, g in zip(ps, gs) if g is not None]
                gs = [g for g in gs if g is not None]
            if not ps:
                continue
            # per group, cached while the same Parameter objects take part: the state tensors, their pointers, the ctypes arrays of
            # everything that does not change between steps (at batch 16 the host, not the GPU, was the slower side of an iteration)
            cg = self.__dict__.setdefault("_groups", {}).get(gi)
            if cg is None or len(cg["ps"]) != len(ps) or any(a is not b for a, b in zip(cg["ps"], ps)):
                for q in ps:
                    if not q.is_cuda or q.dtype is not f32 or not q.is_contiguous():
                        raise RuntimeError("training.Adam (HIP) has no CPU path: contiguous float32 GPU parameters only")
                    st = self.state[q]
                    if not st:
                        st["step"], st["exp_avg"], st["exp_avg_sq"] = 0, torch.zeros_like(q), torch.zeros_like(q)
                sts = [self.state[q] for q in ps]
                n = len(ps)
                cg = self._groups[gi] = dict(ps=list(ps), sts=sts, m=[st["exp_avg"] for st in sts], v=[st["exp_avg_sq"] for st in sts],
                                             sizes=(ctypes.c_long * n)(*[q.numel() for q in ps]))
                cg["mp"] = tuple(t.data_ptr() for t in cg["m"])
                cg["vp"] = tuple(t.data_ptr() for t in cg["v"])
                cg["ma"], cg["va"] = (ctypes.c_void_p * n)(*cg["mp"]), (ctypes.c_void_p * n)(*cg["vp"])
                cg["pp"] = None
            n = len(ps)
            pp = tuple(q.data_ptr() for q in ps)
            if pp != cg["pp"]:                               # (a .to() / .float() of the model moves the parameters)
                cg["pp"], cg["pa"] = pp, (ctypes.c_void_p * n)(*pp)
            gs = [g if g.is_contiguous() else g.contiguous() for g in gs]
            gp = tuple(g.data_ptr() for g in gs)
            klist = []
            for st in cg["sts"]:
                k = st["step"] = int(st["step"]) + 1         # (a loaded torch.optim.Adam state carries `step` as a tensor)
                klist.append(k)
            device = ps[0].device
            if min(klist) == max(klist) and all(q.device == device for q in ps):
                launches = [(device, klist[0], None)]        # the usual case: one launch for the whole group
            else:                                            # mixed devices / step counts: one launch per (device, step)
                parts = {}
                for i, (q, k) in enumerate(zip(ps, klist)):
                    parts.setdefault((q.device, k), []).append(i)
                launches = [(d, k, idx) for (d, k), idx in parts.items()]
            for device, step, idx in launches:
                if idx is None:
                    n, ptrs = len(ps), (pp, gp, cg["mp"], cg["vp"])
                    arrs = (cg["pa"], (ctypes.c_void_p * n)(*gp), cg["ma"], cg["va"], cg["sizes"])
                else:
                    n = len(idx)
                    ptrs = tuple(tuple(t[i] for i in idx) for t in (pp, gp, cg["mp"], cg["vp"]))
                    arrs = tuple((ctypes.c_void_p * n)(*t) for t in ptrs) + ((ctypes.c_long * n)(*[cg["sizes"][i] for i in idx]),)
                # The pointer table lives on the device, with a pinned host copy, per (device, tensor count); it is re-uploaded only when
                # a pointer changed (p, exp_avg, exp_avg_sq never do; the gradients are views of the trainer's flat buffer, which the
                # caching allocator hands back every iteration) -- so a step is normally ONE asynchronous launch and the host runs ahead.
                cache = self.__dict__.setdefault("_tables", {})
                ent = cache.get((device, n))
                if ent is None:
                    ent = cache[(device, n)] = dict(dev=torch.empty(48 * n, dtype=torch.uint8, device=device),
                                                    host=torch.empty(48 * n, dtype=torch.uint8).pin_memory(), key=None, event=None)
                upload = ent["key"] != ptrs
                if upload and ent["event"] is not None:
                    ent["event"].synchronize()           # the previous upload has left the pinned buffer
                dev = device.index if device.index is not None else torch.cuda.current_device()
                stream = torch.cuda.current_stream(device)
                _lib.check(lib.yf_train_adam_multi_pinned(dev, n, arrs[0], arrs[1], arrs[2], arrs[3], arrs[4], float(group["lr"]), float(b1),
                                                          float(b2), float(group["eps"]), int(step), ent["dev"].data_ptr(), ent["dev"].numel(),
                                                          ent["host"].data_ptr(), int(upload), ctypes.c_void_p(stream.cuda_stream)))
                if upload:
                    ent["key"] = ptrs
                    ent["event"] = torch.cuda.Event()
                    ent["event"].record(stream)
            self._keep = gs                      # alive until the next step: the launch is asynchronous


def train_step(model, model_loss, optimizer, imgs, targets):
    """One iteration of train.py:111-132: returns the summed losses [total, x, y, w, h, conf, cls] (total a 0-dim tensor)."""
    optimizer.zero_grad()
    pred = model(imgs)
    losses = [[] for _ in range(7)]
    for i, item_pred in enumerate(pred):
        for j, v in enumerate(model_loss[i](item_pred, targets)):
            losses[j].append(v)
    losses = [sum(v) for v in losses]
    losses[0].backward()
    optimizer.step()
    return losses


def cosine_factor(epoch, total_epochs):
    """train.py:87-88 `lf`: the per-epoch learning-rate factor, 1.0 -> 0.2 over the run."""
    import math
    return ((1 + math.cos(epoch * math.pi / total_epochs)) / 2) * 0.8 + 0.2


def train(params, device, tbwriter=None, train_dataset=None, val_dataset=None, logger=None):
    """The reference's `train(params, device, tbwriter)` (train.py:44-160) with the iteration on the GPU kernels of this package.
    Same control flow and arithmetic: one YOLOLossV3 per stride (:50-53), the model from `pretrained_pth` or initialize_weights()
    (:58-65), DataLoader(batch_size, drop_last, shuffle) (:71-73), Adam(lr0, betas (0.9, 0.999), eps 1e-8) (:84), the cosine
    LambdaLR stepped per epoch (:87-91), the linear warm-up of the first max(3 epochs, 1000 iterations) written into param_groups per
    iteration (:103-109), the log line every 10 iterations (:134-150, same format), get_mAP after epoch 4 (:153-154) and one
    `YOLO-Fastest_epoch_N.pth` state-dict per epoch (:155).  Differences: the datasets are passed in (the reference builds its
    DetectDataset -- cv2 augmentation over files that are not shipped -- from paths in the config); items are DetectDataset's
    ((h, w, c) uint8-range image, (64, 6) boxes), batched by validation.collate_fn (= DetectDataset.collate_fn).  tbwriter may be None.
    Returns the trained model."""
    import logging
    import os
    import time
    import numpy as np
    from torch.optim import lr_scheduler
    from torch.utils.data import DataLoader
    from . import validation
    from .model import YoloFastest
    if train_dataset is None:
        raise ValueError("pass train_dataset (the reference's DetectDataset needs its un-shipped data and cv2)")
    logger = logger or logging.getLogger(__name__)
    io, tp = params["io_params"], params["train_params"]
    save_path = io["save_path"]
    os.makedirs(save_path, exist_ok=True)
    total_epochs, batch_size = tp["total_epochs"], tp["batch_size"]
    model = YoloFastest(io).to(device)
    model_loss = [validation.YOLOLossV3(anchors=io["anchors"][i], num_classes=io["num_cls"], input_shape=io["input_shape"], device=device,
                                        model=model) for i in range(len(io["strides"]))]
    for ml in model_loss:
        ml.ignore_threshold = tp.get("IOU_loss_thre", 0.5)
    if tp.get("pretrained_pth") and os.path.exists(tp["pretrained_pth"]):
        logger.info("Load pretrained model %s" % tp["pretrained_pth"])
        model.load_state_dict(torch.load(tp["pretrained_pth"], map_location=device))
    else:
        logger.info("initialize model")
        model.initialize_weights()
    import torch.distributed as tdist
    world = tdist.get_world_size() if tdist.is_initialized() else 1
    rank = tdist.get_rank() if tdist.is_initialized() else 0
    sampler = None
    if world > 1:            # one process per GPU: every rank its own shard of each epoch, gradients averaged by data_parallel()
        from torch.utils.data.distributed import DistributedSampler
        sampler = DistributedSampler(train_dataset, num_replicas=world, rank=rank, shuffle=True, drop_last=True)
        data_parallel(model)
    dataloader = DataLoader(train_dataset, batch_size=batch_size, num_workers=0, drop_last=True, pin_memory=True, shuffle=sampler is None,
                            sampler=sampler, collate_fn=validation.collate_fn)
    val = validation.Validation(params=params, logger=logger, dataset=val_dataset, device=device, model_loss=model_loss) \
        if val_dataset is not None else None
    batch_per_epoch = len(dataloader)
    num_warm = max(3 * batch_per_epoch, 1000)
    optimizer = Adam(model.parameters(), lr=tp["lr0"], betas=(0.9, 0.999), eps=1e-08)

    def lf(epoch):
        return cosine_factor(epoch, total_epochs)
    scheduler = lr_scheduler.LambdaLR(optimizer, lr_lambda=lf)
    start_epoch = 0
    scheduler.last_epoch = start_epoch - 1
    total_steps = (total_epochs - start_epoch) * batch_per_epoch
    step_count = 0
    logger.info("Start training.")
    losses_name = ["total_loss", "x", "y", "w", "h", "conf", "cls"]
    for epoch in range(start_epoch, total_epochs):
        model.train()
        if sampler is not None:
            sampler.set_epoch(epoch)
        for batch_id, (imgs, targets) in enumerate(dataloader):
            start_time = time.time()
            imgs = imgs.to(device).float()
            targets = targets.to(device).float()
            iteration = batch_id + batch_per_epoch * epoch
            if iteration <= num_warm:
                for x in optimizer.param_groups:
                    x["lr"] = np.interp(iteration, [0, num_warm], [0.0, x["initial_lr"] * lf(epoch)])
            losses = train_step(model, model_loss, optimizer, imgs, targets)
            step_count += 1
            if step_count > 0 and step_count % 10 == 0:
                _loss = losses[0].item()
                duration = float(time.time() - start_time)
                example_per_second = batch_size / duration
                _remain = (total_steps - step_count) * duration
                _m, _s = divmod(_remain, 60)
                _h, _m = divmod(_m, 60)
                lr = optimizer.param_groups[0]["lr"]
                logger.info("epoch [%d]: current_batch = %d/%d, total_iter = %d, loss = %.5f, example/sec = %.3f, lr = %.5f, remain = %d:%02d:%02d" %
                            (epoch, batch_id + 1, batch_per_epoch, step_count, _loss, example_per_second, lr, _h, _m, _s))
                if tbwriter is not None:
                    tbwriter.add_scalar("lr", lr, step_count)
                    tbwriter.add_scalar("example/sec", example_per_second, step_count)
                    for i, name in enumerate(losses_name):
                        tbwriter.add_scalar(name, _loss if i == 0 else losses[i], step_count)
        scheduler.step()
        if world > 1:
            sync_buffers(model)                      # validation and the checkpoint see rank 0's running statistics on every rank
        if epoch > 4 and val is not None:
            val.get_mAP(epoch=epoch, model=model)
        if rank == 0:
            torch.save(model.state_dict(), os.path.join(save_path, "YOLO-Fastest_epoch_{}.pth".format(str(epoch))))
    return model
