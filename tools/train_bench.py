#!/usr/bin/env python3
"""Training-step throughput (train.py:111-132: zero_grad, train-mode forward, two-head loss, backward, Adam) on synthetic frames.
GPU box: python tools/train_bench.py [--batch 16] [--steps 20] [--cpu]   (--cpu also times oracle/ on the host cores: the baseline)"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import yolo_fastest_amd as yf  # noqa: E402
from yolo_fastest_amd import training, validation as val  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=8)
ap.add_argument("--res", type=int, default=256)
ap.add_argument("--cpu", action="store_true")
ap.add_argument("--json", action="store_true", help="also print one JSON line (metric, value, ms per iteration, cpu baseline)")
a = ap.parse_args()
dev = torch.device("cuda:0")
io = yf.io_params_for(a.res)
H, W = io["input_shape"][0], io["input_shape"][1]
torch.manual_seed(0)
m = yf.YoloFastest(io)
m.initialize_weights()
sd0 = {k: v.clone() for k, v in m.state_dict().items()}
m = m.to(dev).train()
x = torch.rand(a.batch, 1, H, W) - 0.5
rng = np.random.default_rng(0)
t = np.zeros((a.batch, 64, 6), np.float32)
for b in range(a.batch):
    k = 1 + b % 6
    t[b, :k, 0:2] = rng.uniform(0.05, 0.95, (k, 2)); t[b, :k, 2:4] = rng.uniform(0.03, 0.4, (k, 2))
    t[b, :k, 4] = rng.integers(0, 3, k); t[b, :k, 5] = 255.0
tt = torch.from_numpy(t)
crit = [val.YOLOLossV3(io["anchors"][i], 3, io["input_shape"], dev, model=m) for i in range(2)]
opt = training.Adam(m.parameters(), lr=0.001)
xd, td = x.to(dev), tt.to(dev)
for _ in range(a.warmup):
    training.train_step(m, crit, opt, xd, td)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    loss = training.train_step(m, crit, opt, xd, td)[0]
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
print("GPU  batch %d %dx%d: %.2f ms / iteration, %.0f examples/s, loss %.4f" % (a.batch, H, W, dt * 1e3, a.batch / dt, float(loss.detach())))
line = {"metric": "training_examples_per_second", "value": round(a.batch / dt, 1), "unit": "examples/s", "ms_per_iteration": round(dt * 1e3, 3),
        "steps": a.steps, "warmup": a.warmup, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "train.py:111-132 iteration (zero_grad, train-mode forward, two-head loss, backward, Adam)", "batch": a.batch,
                   "input": [H, W]}, "cpu_baseline": None}
if a.cpu:
    from oracle import backbone_oracle as bo, loss_oracle as lo
    sd = bo.training_state(sd0)
    keys = bo.parameter_keys(sd)
    params = [sd[k] for k in keys]
    copt = torch.optim.Adam(params, lr=0.001)
    n = 3
    t0 = time.perf_counter()
    for _ in range(n):
        copt.zero_grad()
        hl, hs = bo.forward(sd, x, train=True)
        total = sum(lo.loss_head(h, tt, io["anchors"][i], 3, io["input_shape"])[0] for i, h in enumerate((hl, hs)))
        total.backward()
        copt.step()
    dc = (time.perf_counter() - t0) / n
    print("CPU oracle (torch, %d threads): %.1f ms / iteration, %.1f examples/s" % (torch.get_num_threads(), dc * 1e3, a.batch / dc))
    line["cpu_baseline"] = {"value": round(a.batch / dc, 2), "unit": "examples/s", "cores": torch.get_num_threads(), "kind": "port",
                            "sample": "%d iterations of the same batch through oracle/backbone_oracle.py + loss_oracle.py + torch.optim.Adam" % n}
if a.json:
    import json
    print(json.dumps(line))
