"""Data-parallel sharding of a batch of frames over the GPUs of one node (one process per GPU) and the
single exchange step: an all-gather of fixed-capacity detection records (RCCL over xGMI with backend "nccl";
gloo on CPU in tests).  The reference has no distributed code; parity is defined as: the concatenation of
the per-rank results equals the single-GPU result for the same frames, in frame order."""
import torch
import torch.distributed as dist


def shard_range(n_frames, rank, world_size):
    """Contiguous split: rank r owns frames [lo, hi). Sizes differ by at most one."""
    base, rem = divmod(n_frames, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_records(raw):
    """dict of tensors (post_process.detect_raw) -> one int32 tensor [n, 1 + 8*kmax]:
    count | boxes(4k) | scores bit-cast (2k) | cls (k) | src (k).  One buffer -> one collective."""
    n, kmax = raw["cls"].shape
    return torch.cat([raw["counts"].view(n, 1), raw["boxes"].reshape(n, 4 * kmax),
                      raw["scores"].contiguous().view(torch.int32).reshape(n, 2 * kmax), raw["cls"], raw["src"]], dim=1)


def unpack_records(buf, kmax):
    """-> the five tensors as VIEWS of the record block (no copies: `records` keeps the block itself)."""
    n = buf.shape[0]
    return dict(counts=buf[:, 0], boxes=buf[:, 1:1 + 4 * kmax].view(n, kmax, 4),
                scores=buf[:, 1 + 4 * kmax:1 + 6 * kmax].view(torch.float32).view(n, kmax, 2),
                cls=buf[:, 1 + 6 * kmax:1 + 7 * kmax], src=buf[:, 1 + 7 * kmax:1 + 8 * kmax], records=buf)


class _PendingGather:
    """Handle of an exchange in flight: `wait()` makes the caller's stream wait for the collective (RCCL: a stream-level
    dependency, the host does not block) and returns the unpacked records of all frames in frame order."""

    def __init__(self, work, out, parts, world, per, n_total, kmax):
        self._work, self._out, self._parts = work, out, parts
        self._world, self._per, self._n_total, self._kmax = world, per, n_total, kmax

    def wait(self):
        if self._work is not None:
            self._work.wait()
            self._work = None
        out = torch.cat(self._parts, dim=0) if self._parts is not None else self._out
        if self._per * self._world == self._n_total:      # even shards: the gathered block IS the result, in frame order
            return unpack_records(out, self._kmax)
        keep = []
        for r in range(self._world):
            lo, hi = shard_range(self._n_total, r, self._world)
            keep.append(out[r * self._per:r * self._per + (hi - lo)])
        return unpack_records(torch.cat(keep, dim=0), self._kmax)


def all_gather_detections_async(raw, n_total, group=None):
    """Start the exchange and return at once: the collective runs on the backend's own stream, so the next batch's kernels
    (queued right after this call) overlap with it; `wait()` on the returned handle when the gathered records are needed.
    Shards may differ by one frame: each rank pads to the largest shard, the pad is dropped after the gather."""
    world = dist.get_world_size(group)
    kmax = raw["cls"].shape[1]
    # zero-copy: a post-process that ran with packed=True already wrote its results as one record block (yf_decode_nms_packed) --
    # the collective sends that buffer; otherwise the five tensors are packed here (a torch.cat per step)
    rec = raw["records"] if raw.get("records") is not None else pack_records(raw)
    per = -(-n_total // world)
    if rec.shape[0] < per:
        rec = torch.cat([rec, rec.new_zeros((per - rec.shape[0], rec.shape[1]))], dim=0)
    rec = rec.contiguous()
    if dist.get_backend(group) == "nccl":
        out = rec.new_empty((world * per, rec.shape[1]))
        work = dist.all_gather_into_tensor(out, rec, group=group, async_op=True)
        return _PendingGather(work, out, None, world, per, n_total, kmax)
    parts = [rec.new_empty(rec.shape) for _ in range(world)]
    work = dist.all_gather(parts, rec, group=group, async_op=True)
    return _PendingGather(work, None, parts, world, per, n_total, kmax)


def all_gather_detections(raw, n_total, group=None):
    """Every rank contributes its shard's records; every rank gets all n_total frames in frame order."""
    return all_gather_detections_async(raw, n_total, group).wait()


def all_reduce_mean_(flat, group=None):
    """Data-parallel TRAINING's one exchange step: the gradients of all parameters live in one flat buffer (training.py allocates them
    so), so a step needs ONE all-reduce (RCCL: ring over xGMI; 1.3 MB for this network -- latency-bound) and a scale by 1 / world.
    In place; returns `flat`."""
    world = dist.get_world_size(group)
    if world > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.mul_(1.0 / world)
    return flat


def broadcast_model_(model, src=0, group=None):
    """Every rank starts from rank `src`'s parameters and buffers (what torch's DistributedDataParallel does at construction)."""
    with torch.no_grad():
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t, src=src, group=group)
    return model
