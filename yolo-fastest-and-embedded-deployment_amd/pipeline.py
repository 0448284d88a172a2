"""Several batches in flight: `BatchPipeline` runs consecutive batches of the hot path (model -> decode -> per-class NMS) round-robin on
`depth` HIP streams, each with its own engine (weights copy, workspace, side streams), so that a batch's late, per-frame,
latency-bound stages (strides 16 / 32 and the two heads: one 80- or 320-pixel frame per workgroup) run beside the next batch's
early, machine-filling stages (stem .. stride 8) instead of beside more of the same.

The reference has no counterpart (it loops one image per iteration, src/detect.py:146); this is the throughput mode of the batched
driver.  Measured on MI355X, 320x256, batch 256 (tools/inflight_try.py): one batch in flight as two half-batch lanes 243 k frames/s;
two batches in flight, one lane each, 279 k (fp32); 302 k -> 354 k with the f16x3 variant; a third batch in flight loses again.
Results are bitwise those of the one-at-a-time path (tests/test_gpu_parity.py::test_batch_pipeline_is_identical).
"""
import torch


class _Ticket:
    """One submitted batch.  `result()` makes the CALLER's current stream wait for the batch (a stream-level dependency, the host does
    not block) and returns what was submitted for: the raw detection dict (plus the head tensors under 'head_large' / 'head_small')."""

    def __init__(self, event, out, extra=None):
        self._event, self._out, self.extra = event, out, extra     # extra: what submit()'s `then` callback returned

    def result(self):
        cur = torch.cuda.current_stream(self._out["counts"].device)
        cur.wait_event(self._event)
        for t in self._out.values():
            t.record_stream(cur)     # the caching allocator must not hand the buffers out again before the caller's stream is done
        return self._out

    def synchronize(self):
        self._event.synchronize()
        return self._out


class BatchPipeline:
    def __init__(self, model, post, depth=2, kmax=64, origin_shape=None, lanes=1, branches=0, packed=False):
        """model: yolo_fastest_amd.YoloFastest on a GPU; post: YOLO_post_process bound to it.  depth: batches in flight (2 is the
        measured optimum).  lanes / branches: the per-engine concurrency knobs while the pipeline is used -- with two batches in
        flight the best setting is one lane and the small head in line (the other batch fills the machine instead)."""
        if depth < 1:
            raise ValueError("depth must be >= 1")
        p = next(model.parameters())
        if not p.is_cuda:
            raise RuntimeError("BatchPipeline (HIP) has no CPU path: move the model to the GPU")
        self.model, self.post, self.depth, self.kmax, self.origin_shape = model, post, depth, kmax, origin_shape
        self.packed = packed     # True: the post-process writes one packed record block per batch (`records`: what the exchange sends as is)
        model.lanes, model.branches = lanes, branches
        self.device = p.device
        self.streams = [torch.cuda.Stream(self.device) for _ in range(depth)]
        self._n = 0

    def submit(self, x, then=None, mid=None):
        """x: float32 GPU tensor [N,input_channel,H,W] -- or uint8 frames [N,h,w] ([N,h,w,3] for a 3-channel model), which take the
        device pre-process (model.forward_u8 / forward_bgr_u8: any h, w; BGR frames [N,h,w,3] for a 1-channel net go through cvtColor + resize) -- ready on the caller's current stream.  Returns a
        ticket at once.
        then(out): optional, called with the batch's result dict INSIDE the batch's stream context -- work it queues (e.g. the
        asynchronous all-gather of the records, dist.all_gather_detections_async) is ordered behind this batch only, not behind the
        caller's stream; its return value is kept in the ticket's `extra`.
        mid(pred): optional, called between the model and the post-process inside the batch's stream context; what it returns is
        post-processed instead of the model's heads (bench.py splices its synthetic dense logit field in this way)."""
        k = self._n % self.depth
        self._n += 1
        s = self.streams[k]
        s.wait_stream(torch.cuda.current_stream(self.device))      # the input (and everything the caller queued before) first
        with torch.cuda.stream(s), torch.no_grad():
            x.record_stream(s)
            # engine slots 1 .. depth: slot 0 stays the engine of plain `model(x)` calls on the caller's own stream, so those may
            # be mixed with batches in flight
            if x.dtype == torch.uint8 and x.dim() == 4 and self.model.input_channel == 1:     # cv2.imread's BGR frames for a 1-channel net: detect.py:108-127
                pred = self.model.forward_bgr_u8(x, self.post.input_shape, slot=k + 1)
            elif x.dtype == torch.uint8:
                pred = self.model.forward_u8(x, self.post.input_shape, slot=k + 1)
            else:
                pred = self.model(x, slot=k + 1)
            if mid is not None:
                pred = mid(pred)
            out = self.post.detect_raw(pred, kmax=self.kmax, origin_shape=self.origin_shape, slot=k + 1, packed=self.packed)
            out["head_large"], out["head_small"] = pred
            extra = then(out) if then is not None else None
            ev = torch.cuda.Event()
            ev.record(s)
        return _Ticket(ev, out, extra)

    def drain(self):
        for s in self.streams:
            s.synchronize()

    def _fix_collisions(self, tries=8):
        """Replace stream k by fresh pool streams until it overlaps with streams 0 .. k - 1 (yf_streams_overlap)."""
        import ctypes
        from . import _lib
        e = self.model.engine_on(next(self.model.parameters()).device)
        replaced = 0
        for k in range(1, self.depth):
            for _ in range(tries):
                ok = True
                for j in range(k):
                    ov = ctypes.c_int()
                    _lib.check(e.lib.yf_streams_overlap(e.handle, ctypes.c_void_p(self.streams[j].cuda_stream),
                                                        ctypes.c_void_p(self.streams[k].cuda_stream), ctypes.byref(ov)))
                    ok = ok and bool(ov.value)
                if ok:
                    break
                self.streams[k] = torch.cuda.Stream(self.device)
                replaced += 1
        return replaced

    def tune_streams(self, x=None, candidates=3, batches=16):
        """Make sure the batches in flight run on streams that really overlap.  The HIP runtime multiplexes streams onto a few hardware
        queues (GPU_MAX_HW_QUEUES, default 4) in first-use order, and HOW WELL two batches overlap is a property of the pair of streams
        they are issued on -- measured on MI355X, same engines, same process, 20 steps of batch 256 (tools/queue_try2.py): 253-261 k
        frames/s on a pair that shares a hardware queue (the two batches simply run one after the other), ~274 k on pairs that overlap
        but share some dispatch resource, 291-295 k on the others.  Nothing in the HIP API tells them apart, so:
          1. streams that do not overlap at all with the ones before them are replaced (yf_streams_overlap: a 100 us spin kernel on
             each stream, ~0.2 ms per pair) -- this alone removes the worst case and needs no input;
          2. with a batch `x`: `candidates` such sets of streams are timed on `batches` real submissions each and the fastest is kept
             (~15 ms per candidate at batch 256).  The candidates differ by 1-10 %, the device's clocks take tens of milliseconds of
             load to settle after an idle period, and a candidate timed while they still ramp loses to one timed after: the first
             candidate is therefore preceded by `batches` untimed submissions (round 3 timed 6 submissions per candidate from cold).
        Call once, before the first real batch.  Returns the measured frames/s per candidate (empty without x)."""
        if self.depth < 2:
            return []
        self._fix_collisions()
        if x is None:
            return []
        rates, sets = [], []
        for _ in range(batches):      # settle (see above); not timed
            self.submit(x)
        self.drain()
        for c in range(candidates):
            if c > 0:
                self.streams = [torch.cuda.Stream(self.device) for _ in range(self.depth)]
                self._fix_collisions()
            self._n = 0
            for _ in range(2 * self.depth):
                self.submit(x)
            self.drain()
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(self.device)
            t0.record()
            for _ in range(batches):
                self.submit(x)
            self.drain()
            t1.record()
            torch.cuda.synchronize(self.device)
            rates.append(1e3 * batches * x.shape[0] / t0.elapsed_time(t1))
            sets.append(self.streams)
        self.streams = sets[max(range(len(rates)), key=rates.__getitem__)]
        self._n = 0
        return rates
