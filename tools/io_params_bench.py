#!/usr/bin/env python3
"""Per-launch times and frames/s of the engine for io_params other than the shipped ones (tests/golden/seeded_weights.IO_CONFIGS):
what a COCO-style model (80 classes, 3-channel frames: 255 head channels through the run-time head loop, conv0 over 27 taps) costs
beside the shipped 3-class gray model.      python tools/io_params_bench.py [tag ...]     (on the GPU box, from the repo root)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import __graft_entry__  # noqa: F401  (puts the package on the path)
import yolo_fastest_amd as yf
import io_cfg

dev = torch.device("cuda:0")
N = 256
for tag in (sys.argv[1:] or ["c1", "c5rgb", "c20", "c80rgb"]):
    C, Cin, A = io_cfg.CONFIG[tag]
    io = io_cfg.io_for(tag)
    m = yf.YoloFastest(io).to(dev).eval()
    m.load_state_dict({k: v.to(dev) for k, v in io_cfg.state_dict_for(tag, 1000 + io_cfg.TAGS.index(tag)).items()})
    post = yf.YOLO_post_process(io["conf_thre"], io["nms_thre"], A, C, io["anchors"], io["input_shape"]).bind(m)
    g = torch.Generator(device="cpu").manual_seed(0)
    u8 = torch.randint(0, 256, (N, 256, 320) if Cin == 1 else (N, 256, 320, Cin), generator=g, dtype=torch.uint8).to(dev)
    x = yf.preprocess_u8(m, u8, io["input_shape"])
    m.profile(x, reps=3)                      # warm-up (clocks, first-touch)
    ops = m.profile(x, reps=10)
    ops8 = m.profile(u8, reps=10)             # the same pass entered from the u8 frames (pre-process fused into the stem)
    kmax = 64 * A
    for _ in range(5):
        raw = post.detect_raw_from_input(x, kmax=kmax)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        raw = post.detect_raw_from_input(x, kmax=kmax)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    cnt = raw["counts"].cpu().numpy()
    print(f"{tag}: num_cls {C}, input_channel {Cin}, anchors {A}: launch sum {sum(o['ms'] for o in ops):.3f} ms, detect (one batch at a time) "
          f"{dt * 1e3:.3f} ms = {N / dt / 1e3:.1f} k frames/s; survivors/frame mean {cnt.clip(0).mean():.1f} max {cnt.max()} (kmax {kmax})")
    for o in ops:
        if "conv0" in o["name"] or "head" in o["name"]:
            print(f"    {o['name'][:50]:50s} {o['ms'] * 1e3:7.1f} us" + (f"   (from u8 frames: {ops8[0]['ms'] * 1e3:.1f} us)" if "conv0" in o["name"] else ""))
