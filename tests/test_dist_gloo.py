"""N > 1 path on CPU: world_size-2 gloo.  Frames are sharded contiguously; the single exchange step is an
all-gather of fixed-capacity detection records; the gathered result must equal the unsharded one, in frame order."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from yolo_fastest_amd import dist as yfd


def _fake_raw(n, kmax, seed):
    g = torch.Generator().manual_seed(seed)
    counts = torch.randint(0, kmax + 1, (n,), generator=g, dtype=torch.int32)
    return dict(counts=counts,
                boxes=torch.randint(-50, 700, (n, kmax, 4), generator=g, dtype=torch.int32),
                scores=torch.rand((n, kmax, 2), generator=g, dtype=torch.float32),
                cls=torch.randint(0, 3, (n, kmax), generator=g, dtype=torch.int32),
                src=torch.randint(0, 1200, (n, kmax), generator=g, dtype=torch.int32))


def test_shard_range_covers_everything():
    for n in (1, 7, 8, 255, 256, 2048):
        for w in (1, 2, 3, 8):
            spans = [yfd.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_pack_unpack_roundtrip():
    raw = _fake_raw(5, 7, 0)
    back = yfd.unpack_records(yfd.pack_records(raw), 7)
    for k in raw:
        assert torch.equal(raw[k], back[k]), k


def test_packed_record_block_is_what_the_exchange_sends():
    """The kernel's packed output (yf_decode_nms_packed: one int32 row per frame) has exactly pack_records' layout, its five result
    tensors are views of it (YOLO_post_process.record_views), and the exchange takes the block itself."""
    from yolo_fastest_amd import YOLO_post_process
    raw = _fake_raw(6, 9, 1)
    rec = yfd.pack_records(raw)
    v = YOLO_post_process.record_views(rec, 9)
    assert v["records"] is rec and all(torch.equal(v[k], raw[k]) for k in raw)
    assert all(v[k].untyped_storage().data_ptr() == rec.untyped_storage().data_ptr() for k in raw)      # views, not copies
    u = yfd.unpack_records(rec, 9)
    assert all(torch.equal(u[k], raw[k]) for k in raw) and u["records"] is rec


def _worker(rank, world, port, n_total, kmax, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = _fake_raw(n_total, kmax, 123)
    lo, hi = yfd.shard_range(n_total, rank, world)
    mine = {k: v[lo:hi].contiguous() for k, v in full.items()}
    got = yfd.all_gather_detections(mine, n_total)
    pending = yfd.all_gather_detections_async(mine, n_total)   # the overlapped form bench.py uses
    got2 = pending.wait()
    assert all(torch.equal(got[k], got2[k]) for k in got)
    # zero-copy form: the shard's records already are one packed block (what the post-process writes with packed=True)
    from yolo_fastest_amd import YOLO_post_process
    packed = YOLO_post_process.record_views(yfd.pack_records(mine), kmax)
    got3 = yfd.all_gather_detections_async(packed, n_total).wait()
    assert all(torch.equal(got[k], got3[k]) for k in got)
    ok = all(torch.equal(got[k], full[k]) for k in full)
    q.put((rank, ok))
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_all_gather_two_ranks_equals_unsharded():
    ctx = mp.get_context("spawn")
    for n_total in (8, 7):  # even and ragged split
        q = ctx.Queue()
        port = _free_port()
        ps = [ctx.Process(target=_worker, args=(r, 2, port, n_total, 5, q)) for r in range(2)]
        for p in ps:
            p.start()
        res = sorted(q.get(timeout=120) for _ in ps)
        for p in ps:
            p.join(60)
        assert res == [(0, True), (1, True)]


def _worker8(rank, world, port, n_total, kmax, q):
    """world size 8, ragged shards, and two frames that did NOT end normally: frame 3 (rank 0's shard) overflowed its capacity -- count =
    the true number of survivors > kmax -- and frame n_total - 2 (the last rank's shard) hit the reference's ZeroDivisionError (count -2).
    What every rank must see: the gathered counts are bit-identical to the unsharded ones (status included), to_lists(on_error='mark') names
    exactly those two frames with the same exception types on every rank, every other frame's list equals the unsharded one, and the
    default on_error='raise' raises like the single-GPU call does."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from yolo_fastest_amd import YOLO_post_process
    full = _fake_raw(n_total, kmax, 321)
    full["counts"][3] = kmax + 17
    full["counts"][n_total - 2] = -2
    lo, hi = yfd.shard_range(n_total, rank, world)
    mine = YOLO_post_process.record_views(yfd.pack_records({k: v[lo:hi].contiguous() for k, v in full.items()}), kmax)
    got = yfd.all_gather_detections_async(mine, n_total).wait()
    ok = all(torch.equal(got[k], full[k]) for k in full) and got["counts"].shape[0] == n_total
    lists = YOLO_post_process.to_lists(got, with_src=True, on_error="mark")
    want = YOLO_post_process.to_lists(full, with_src=True, on_error="mark")
    bad = [f for f, L in enumerate(lists) if isinstance(L, Exception)]
    ok = ok and bad == [3, n_total - 2] and isinstance(lists[3], OverflowError) and isinstance(lists[n_total - 2], ZeroDivisionError)
    ok = ok and all(a == b for f, (a, b) in enumerate(zip(lists, want)) if f not in bad)
    try:
        YOLO_post_process.to_lists(got)
        ok = False
    except ZeroDivisionError:      # the reference's loop dies at that frame (detect.py:39); -2 wins over an overflow elsewhere
        pass
    q.put((rank, ok, hi - lo))
    dist.destroy_process_group()


def test_all_gather_eight_ranks_ragged_with_error_frames():
    """the N = 8 exchange on gloo.  29 frames over 8 ranks = shards of 4, 4, 4, 4, 4, 3, 3, 3."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker8, args=(r, 8, port, 29, 6, q)) for r in range(8)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=300) for _ in ps)
    for p in ps:
        p.join(60)
    assert [r[:2] for r in res] == [(r, True) for r in range(8)], res
    assert [r[2] for r in res] == [4, 4, 4, 4, 4, 3, 3, 3]


def _worker_grad(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(7)
    both = torch.randn((world, 1000), generator=g)                 # every rank can compute what the others hold
    flat = both[rank].clone()
    views = flat.split([300, 700])                                  # parameter gradients are views of the flat buffer
    yfd.all_reduce_mean_(flat)
    ok = torch.allclose(flat, both.mean(0), atol=1e-7) and torch.allclose(torch.cat(views), both.mean(0), atol=1e-7)
    lin = torch.nn.Linear(3, 2)
    with torch.no_grad():
        lin.weight.fill_(float(rank + 1))
    yfd.broadcast_model_(lin)
    ok = ok and bool((lin.weight == 1.0).all())
    q.put((rank, ok))
    dist.destroy_process_group()


def test_training_gradient_exchange_two_ranks():
    """training.data_parallel's exchange step on gloo, world size 2: one all-reduce of the flat gradient buffer leaves the mean in every
    parameter's view on both ranks; parameters start from rank 0's."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker_grad, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
    assert res == [(0, True), (1, True)]
