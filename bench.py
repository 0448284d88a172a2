#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path (model forward -> decode -> per-class NMS [-> all-gather of the decoded
boxes when N > 1]) over one batch of synthetic frames that is already resident in HBM.
Workload at N = 1: BASELINE.json configs[1], "YOLO-Fastest 320x256 batch=256 fp32, synthetic frames"
(u8 ~ Uniform{0..255} i.i.d., x = (u8-128)/255, seed = rank).  N > 1: the same per GPU (weak
scaling, frames are independent units; one RCCL all-gather of the fixed-capacity box records).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_FRAME = {256: 42_575_360, 512: 170_301_440}   # layer-granular algorithmic bytes, fp32 (fp16: half)
FLOPS_PER_FRAME = {256: 236_442_880, 512: 945_771_520}
HBM_PEAK_GBS = 8000.0                                   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def cpu_baseline(seconds=12.0):
    """The oracle (CPU restatement of the reference's detect.py path; kind "port") timed on this host:
    batch 1 over the 20 bundled frames, model and post-process like detect.py:151-171, repeated for ~`seconds`."""
    import numpy as np
    import torch
    from oracle import backbone_oracle as bo
    from oracle import post_oracle as po
    import yolo_fastest_amd as yf
    io = yf.io_params_for(256)
    sd = bo.load_state_dict(os.path.join(ROOT, "yolo-fastest-and-embedded-deployment_amd", "assets", "weights",
                                         "yolo_fastest_256x320_epoch28.pth"))
    g = np.load(os.path.join(ROOT, "tests", "golden", "golden_256.npz"))
    xs = [bo.preprocess(g["input_u8"][i]) for i in range(20)]
    default_threads = torch.get_num_threads()
    runs = []
    for threads in sorted({1, 8, default_threads}):  # batch-1 convs of this size do not scale with cores
        torch.set_num_threads(threads)
        bo.forward(sd, xs[0])  # warm-up
        n, t_model, t_post = 0, 0.0, 0.0
        t_end = time.time() + seconds / 3
        while time.time() < t_end:
            for x in xs:
                t0 = time.time()
                hl, hs = bo.forward(sd, x)
                t1 = time.time()
                po.post_process((hl.numpy()[0], hs.numpy()[0]), io["anchors"], io["input_shape"][:2])
                t2 = time.time()
                t_model += t1 - t0; t_post += t2 - t1; n += 1
        runs.append((n / (t_model + t_post), threads, n, 1e3 * t_model / n, 1e3 * t_post / n))
    torch.set_num_threads(default_threads)
    best = max(runs)
    return {"value": round(best[0], 2), "unit": "frames/s", "cores": best[1], "kind": "port",
            "sample": "the 20 bundled test_data frames at 320x256, batch 1 (detect.py:151-171: model, then decode+sort+NMS), "
                      f"repeated ~{seconds / 3:.0f} s per thread setting; best of "
                      + "; ".join(f"{t} threads: {f:.1f} frames/s ({n} frames, model {m:.2f} ms + post {p:.2f} ms)"
                                  for f, t, n, m, p in runs)
                      + f"; host has {os.cpu_count()} logical CPUs"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256, help="frames per GPU per step")
    ap.add_argument("--res", type=int, default=256, choices=[256, 512])
    ap.add_argument("--chunk", type=int, default=0, help="frames per pass of the layer chain (0 = whole batch)")
    ap.add_argument("--lanes", type=int, default=2, help="concurrent streams over the chunks of the batch (1..4)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f16"], help="f16 = BASELINE configs[2]: fp16 storage + fp16 MFMA")
    ap.add_argument("--frames", default="noise", choices=["noise", "fixtures"],
                    help="noise: u8 ~ U{0..255} (BASELINE configs[1]); fixtures: the reference's 20 bundled frames tiled to the "
                         "batch (SURVEY.md 8(d) config 2 'realistic': dark IR frames, >= 1 box per frame)")
    ap.add_argument("--dense", action="store_true",
                    help="SURVEY.md 8(d) config 5: synthetic dense head logits (~1200 candidates, ~260 survivors per 640x512 frame) are "
                         "added to the heads of the noise frames, to stress decode + sort + NMS; use with --res 512 --batch 64 --kmax 1024")
    ap.add_argument("--kmax", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dump-ops", default=None, help="write the launch names of one forward pass to this JSON file")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import yolo_fastest_amd as yf
    from yolo_fastest_amd import dist as yfd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP extension is the only implementation (no CPU fallback)")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node == --gpus"

    io = yf.io_params_for(args.res)
    wname = {256: "yolo_fastest_256x320_epoch28.pth", 512: "yolo_fastest_512x640_epoch27.pth"}[args.res]
    model = yf.YoloFastest(io).to(dev).eval()
    model.chunk = args.chunk
    if args.dtype == "f16":
        model.storage_dtype = torch.float16
    model.lanes = args.lanes
    model.load_state_dict(torch.load(os.path.join(ROOT, "yolo-fastest-and-embedded-deployment_amd", "assets",
                                                  "weights", wname), map_location=dev))
    post = yf.YOLO_post_process(io["conf_thre"], io["nms_thre"], io["num_anchors"], io["num_cls"], io["anchors"],
                                io["input_shape"]).bind(model)
    H, W = io["input_shape"][:2]
    g = torch.Generator(device="cpu").manual_seed(rank)
    if args.frames == "fixtures":
        fx = np.load(os.path.join(ROOT, "tests", "golden", "golden_%d.npz" % args.res))["input_u8"]
        u8 = torch.from_numpy(fx[np.arange(args.batch) % len(fx)])
    else:
        u8 = torch.randint(0, 256, (args.batch, H, W), generator=g, dtype=torch.uint8)
    x = ((u8.float() - 128.0) / 255.0)[:, None].contiguous().to(dev)   # resident in HBM before timing
    n_total = args.batch * world

    syn = None
    if args.dense:  # per cell/anchor: t_conf ~ N(-1, 1.5^2) (~25 % pass), t_xy ~ N(0,1), t_wh ~ N(0,0.5^2), classes ~ N(0,2^2); seed = frame
        parts = ([], [])
        for f in range(args.batch):
            gg = np.random.default_rng(rank * args.batch + f)
            for i, (h, w) in enumerate(((H // 16, W // 16), (H // 32, W // 32))):
                t = np.empty((3, 8, h, w), np.float32)
                t[:, 0:2] = gg.normal(0.0, 1.0, (3, 2, h, w)); t[:, 2:4] = gg.normal(0.0, 0.5, (3, 2, h, w))
                t[:, 4] = gg.normal(-1.0, 1.5, (3, h, w)); t[:, 5:8] = gg.normal(0.0, 2.0, (3, 3, h, w))
                parts[i].append(t.reshape(24, h, w))
        syn = tuple(torch.from_numpy(np.stack(p)).to(dev) for p in parts)

    def forward():
        with torch.no_grad():
            pred = model(x)
        if syn is not None:   # the net's own logits on noise are ~no detections: the synthetic field replaces them
            pred = (pred[0] * 0 + syn[0], pred[1] * 0 + syn[1])
        return pred

    def step():
        pred = forward()
        raw = post.detect_raw(pred, kmax=args.kmax)
        if world > 1:
            raw = yfd.all_gather_detections(raw, n_total)
        return raw

    for _ in range(args.warmup):
        raw = step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    # HIP events on the stream the kernels are launched on (torch's current stream is the one handed to the C ABI)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    t0 = time.perf_counter()
    pending = None  # the exchange of step k runs on RCCL's stream while step k + 1 computes; all K exchanges end inside the timed region
    for k in range(args.steps):
        ev[k][0].record()
        pred = forward()
        ev[k][1].record()
        raw = post.detect_raw(pred, kmax=args.kmax)
        ev[k][2].record()
        if world > 1:
            if pending is not None:
                gathered = pending.wait()
            pending = yfd.all_gather_detections_async(raw, n_total)
    if pending is not None:
        raw = pending.wait()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    fwd_ms = sum(e[0].elapsed_time(e[1]) for e in ev) / args.steps
    post_ms = sum(e[1].elapsed_time(e[2]) for e in ev) / args.steps

    if rank == 0:
        counts = raw["counts"].cpu().numpy()
        fps = n_total * args.steps / elapsed
        # per-launch timing, HIP events on the launch stream around every kernel of one forward pass (mean of 5 passes)
        ops = model.profile(x, reps=5)
        if args.dump_ops:
            with open(args.dump_ops, "w") as f:
                json.dump([{"name": o["name"], "ms": o["ms"], "algorithmic_bytes": o["algorithmic_bytes"]} for o in ops], f)
        chain_ms = sum(o["ms"] for o in ops)
        dom = max(ops, key=lambda o: o["ms"])
        bytes_sum = sum(o["algorithmic_bytes"] for o in ops) / args.batch
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            traffic = tj.get("kernels", {}).get(dom["name"], {}).get("hbm_bytes_per_launch")
        achieved = dom["algorithmic_bytes"] / (dom["ms"] * 1e-3) / 1e9
        bpf = BYTES_PER_FRAME[args.res] // (2 if args.dtype == "f16" else 1)
        chain_achieved = args.batch * bpf / (fwd_ms * 1e-3) / 1e9
        out = {
            "metric": "frames/sec end-to-end (model forward + decode + per-class NMS), 320x256 batch=256 per GPU"
                      if args.res == 256 else "frames/sec end-to-end, 640x512",
            "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "ms_per_frame": round(1e3 * elapsed / args.steps / n_total, 6),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic" if args.frames == "noise" else "the reference's 20 bundled frames tiled to the batch",
            "config": {"workload": f"YOLO-Fastest {W}x{H} batch={args.batch} fp32 per GPU, synthetic uniform-u8 frames "
                                   f"(BASELINE.json configs[1])" if args.res == 256 and args.batch == 256 and args.dtype == "f32" and args.frames == "noise" else
                                   f"YOLO-Fastest {W}x{H} batch={args.batch} {args.dtype} per GPU, {args.frames} frames"
                                   + (", dense synthetic head logits (SURVEY.md 8(d) config 5)" if args.dense else ""),
                       "global_batch": n_total, "weights": wname, "kmax": args.kmax, "chunk": args.chunk, "lanes": args.lanes,
                       "parallelism": f"dp{world} (frames sharded, one RCCL all-gather of box records)" if world > 1 else "single GPU",
                       "survivors_per_frame_mean": round(float(np.clip(counts, 0, None).mean()), 3)},
            # dominant kernel of the forward pass: algorithmic (layer-granular) bytes of ITS layers per launch
            # over ITS average launch duration
            "roofline": {"bound": "hbm", "kernel": dom["name"], "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "launch_ms": round(dom["ms"], 4), "algorithmic_bytes_per_launch": int(dom["algorithmic_bytes"]),
                         "compute_frac_fp32_peak": round(dom["flops"] / (dom["ms"] * 1e-3) / 157.3e12, 4),
                         "share_of_forward": round(dom["ms"] / chain_ms, 4)},
            # the whole forward pass (all launches) against the same definition
            "forward_chain": {"launches": len(ops), "forward_ms": round(fwd_ms, 4), "post_ms": round(post_ms, 4),
                              "sum_of_launch_ms_single_stream": round(chain_ms, 4),
                              "algorithmic_bytes_per_frame": bpf,
                              "algorithmic_bytes_per_frame_sum_over_launches": int(bytes_sum),
                              "achieved_GBps": round(chain_achieved, 1), "frac_of_hbm_peak": round(chain_achieved / HBM_PEAK_GBS, 4),
                              "compute_frac_fp32_peak": round(args.batch * FLOPS_PER_FRAME[args.res] / (fwd_ms * 1e-3) / 157.3e12, 4),
                              "top_launches": [{"name": o["name"], "ms": round(o["ms"], 4)} for o in sorted(ops, key=lambda o: -o["ms"])[:6]]},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
