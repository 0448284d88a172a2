#!/usr/bin/env python3
"""Times the decode + sort + NMS kernel on the dense synthetic heads (SURVEY.md 8d config 5 recipe), batch 64 at 640x512
and batch 256 at 320x256, and checks the result against the C oracle on a few frames."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import yolo_fastest_amd as yf
from oracle import post_oracle_c as poc

dev = torch.device("cuda:0")
for res, batch in ((512, 64), (256, 256)):
    io = yf.io_params_for(res)
    m = yf.YoloFastest(io).to(dev).eval()
    m.load_state_dict(torch.load(os.path.join(ROOT, "yolo-fastest-and-embedded-deployment_amd/assets/weights",
                                              {256: "yolo_fastest_256x320_epoch28.pth", 512: "yolo_fastest_512x640_epoch27.pth"}[res]), map_location=dev))
    post = yf.YOLO_post_process(io["conf_thre"], io["nms_thre"], 3, 3, io["anchors"], io["input_shape"]).bind(m)
    H, W = io["input_shape"][:2]
    m(torch.zeros(1, 1, H, W, device=dev))
    hl, hs = [], []
    for f in range(batch):
        g = np.random.default_rng(f)
        for (h, w), dst in (((H // 16, W // 16), hl), ((H // 32, W // 32), hs)):
            t = np.empty((3, 8, h, w), np.float32)
            t[:, 0:2] = g.normal(0.0, 1.0, (3, 2, h, w)); t[:, 2:4] = g.normal(0.0, 0.5, (3, 2, h, w))
            t[:, 4] = g.normal(-1.0, 1.5, (3, h, w)); t[:, 5:8] = g.normal(0.0, 2.0, (3, 3, h, w))
            dst.append(t.reshape(24, h, w))
    pred = (torch.from_numpy(np.stack(hl)).to(dev), torch.from_numpy(np.stack(hs)).to(dev))
    kmax = 1024
    for mode, what in ((2, "one workgroup per frame"), (1, "one workgroup per frame and class")):
        m.post_split = mode
        for _ in range(3):
            raw = post.detect_raw(pred, kmax=kmax)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            raw = post.detect_raw(pred, kmax=kmax)
        e1.record(); torch.cuda.synchronize()
        counts = raw["counts"].cpu().numpy()
        ok = True
        for f in range(3):
            r = poc.post_process(hl[f], hs[f], io["anchors"], io["input_shape"][:2])
            n = r["count"]
            ok &= n == counts[f] and np.array_equal(raw["src"][f, :n].cpu().numpy(), r["src"]) and np.array_equal(raw["boxes"][f, :n].cpu().numpy(), r["box"])
        print(f"{W}x{H} batch {batch}, {what}: post kernel {e0.elapsed_time(e1) / 20:.4f} ms/batch, survivors/frame mean {counts.mean():.1f}, "
              f"candidates ~{(np.stack(hl)[:, 4::8] > 0).sum() / batch + (np.stack(hs)[:, 4::8] > 0).sum() / batch:.0f}, oracle match: {ok}", flush=True)
    # the sparse case beside it (the headline's frames give ~0-2 survivors): the model's own logits on noise, kmax 64
    x = torch.rand(batch, 1, H, W, device=dev) - 0.5
    with torch.no_grad():
        p2 = m(x)
    for mode, what in ((2, "one workgroup per frame"), (1, "one workgroup per frame and class")):
        m.post_split = mode
        for _ in range(3):
            post.detect_raw(p2, kmax=64)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            post.detect_raw(p2, kmax=64)
        e1.record(); torch.cuda.synchronize()
        print(f"{W}x{H} batch {batch}, sparse (net on noise), {what}: {e0.elapsed_time(e1) / 20:.4f} ms/batch", flush=True)
