#!/usr/bin/env python3
"""Developer stress (GPU box): decode + NMS on random logit fields of random frame sizes / batch sizes / thresholds / densities -- the
one-workgroup-per-frame launch (yf_set_post_split 2) against the per-(frame, class) launch (1) bit for bit, and one frame per case against the
reference's loop restated in C (oracle/post_oracle.c).  Exits non-zero on any difference."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import yolo_fastest_amd as yf
from oracle import post_oracle_c as poc
dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    H = 32 * int(rng.integers(1, 17)); W = 32 * int(rng.integers(1, 21)); N = int(rng.integers(1, 13))
    io = dict(yf.io_params_for(256)); io["input_shape"] = [H, W, 1]
    m = yf.YoloFastest(io).to(dev).eval()
    m.load_state_dict(torch.load(os.path.join(ROOT, "yolo-fastest-and-embedded-deployment_amd", "assets", "weights", "yolo_fastest_256x320_epoch28.pth"), map_location=dev))
    m(torch.zeros(1, 1, H, W, device=dev))
    conf, nms = float(rng.choice([0.5, 0.2, 0.9, 0.05])), float(rng.choice([0.2, 0.0, 0.5, 1.0, -0.1]))
    mu, sd, whmu = float(rng.uniform(-3, 1)), float(rng.uniform(0.5, 2.5)), float(rng.choice([0.5, 0.0, -6.0]))
    hl, hs = [], []
    for f in range(N):
        for (h, w), dst in (((H // 16, W // 16), hl), ((H // 32, W // 32), hs)):
            t = np.empty((3, 8, h, w), np.float32)
            t[:, 0:2] = rng.normal(0.0, 1.0, (3, 2, h, w)); t[:, 2:4] = rng.normal(whmu, 0.5, (3, 2, h, w))
            t[:, 4] = rng.normal(mu, sd, (3, h, w)); t[:, 5:8] = rng.normal(0.0, 2.0, (3, 3, h, w))
            if rng.random() < 0.3:
                t[:, 4:8] = np.round(t[:, 4:8] * 2) / 2     # exact ties
            dst.append(t.reshape(24, h, w))
    post = yf.YOLO_post_process(conf, nms, 3, 3, io["anchors"], io["input_shape"]).bind(m)
    pred = (torch.from_numpy(np.stack(hl)).to(dev), torch.from_numpy(np.stack(hs)).to(dev))
    ncell = 3 * (hl[0].shape[1] * hl[0].shape[2] + hs[0].shape[1] * hs[0].shape[2])
    kmax = int(rng.choice([ncell, 64, 7]))
    out = {}
    for mode in (2, 1):
        m.post_split = mode
        out[mode] = {k: v.cpu().numpy() for k, v in post.detect_raw(pred, kmax=kmax, packed=bool(rng.integers(0, 2))).items() if k in ("counts", "boxes", "scores", "cls", "src")}
    ok = np.array_equal(out[1]["counts"], out[2]["counts"])
    for f in range(N):
        n = min(max(int(out[2]["counts"][f]), 0), kmax)
        ok = ok and all(np.array_equal(out[1][k][f, :n], out[2][k][f, :n]) for k in ("boxes", "scores", "cls", "src"))
    f = int(rng.integers(0, N))
    try:
        r = poc.post_process(hl[f], hs[f], io["anchors"], [H, W], conf_thres=conf, nms_thres=nms)
        n = min(r["count"], kmax)
        ok = ok and int(out[1]["counts"][f]) == r["count"] and np.array_equal(out[1]["src"][f, :n], r["src"][:n]) and np.array_equal(out[1]["boxes"][f, :n], r["box"][:n])
    except ZeroDivisionError:
        ok = ok and int(out[1]["counts"][f]) == -2
    bad += not ok
    print(f"{H:4d}x{W:<4d} N={N:2d} conf {conf} nms {nms} kmax {kmax:5d} survivors/frame {np.clip(out[2]['counts'], 0, None).mean():7.1f}  {'ok' if ok else 'MISMATCH'}", flush=True)
    del m
sys.exit(1 if bad else 0)
