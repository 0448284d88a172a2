#!/usr/bin/env python3
"""Small random case of the training network against oracle/backbone_oracle.py in float64: per-tensor gradient error, in backward
order (a ReLU flip shows as a step in the error from one layer on).  GPU box: python tools/train_oracle_report.py [seed] [N] [H] [W]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import yolo_fastest_amd as yf  # noqa: E402
from yolo_fastest_amd import validation as val  # noqa: E402
from oracle import backbone_oracle as bo  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
N, H, W = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (3, 64, 96)
dev = torch.device("cuda:0")
torch.manual_seed(seed)
io = yf.io_params_for(256)
m = yf.YoloFastest(io)
m.initialize_weights()
with torch.no_grad():
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.bias.normal_(0, 0.2); mod.running_mean.normal_(0, 0.1); mod.running_var.uniform_(0.5, 1.5)
    m.head_4.bias.normal_(0, 0.5); m.head_5.bias.normal_(0, 0.5)
sd0 = {k: v.clone() for k, v in m.state_dict().items()}
m = m.to(dev).train()
x = torch.rand(N, 1, H, W) - 0.5
rng = np.random.default_rng(seed)
t = np.zeros((N, 8, 6), np.float32)
for b in range(N):
    k = 2 + b % 5
    t[b, :k, 0:2] = rng.uniform(0.05, 0.95, (k, 2)); t[b, :k, 2:4] = rng.uniform(0.05, 0.6, (k, 2))
    t[b, :k, 4] = rng.integers(0, 3, k); t[b, :k, 5] = 255.0
pred = m(x.to(dev))
for p in pred:
    p.retain_grad()
loss = sum(val.YOLOLossV3(io["anchors"][i], 3, [H, W, 1], dev, model=m)(p, torch.from_numpy(t).to(dev))[0] for i, p in enumerate(pred))
loss.backward()
for dt in (torch.float64, torch.float32):
    sd = bo.training_state(sd0, dt)
    keys = bo.parameter_keys(sd)
    want = bo.forward(sd, x.to(dt), train=True)
    print(dt, "heads: max |d|", [float(np.abs(g.detach().cpu().numpy() - w.detach().numpy()).max()) for g, w in zip(pred, want)])
    g = torch.autograd.grad(list(want), [sd[k] for k in keys], [p.grad.cpu().to(dt) for p in pred])
    if dt == torch.float64:
        g64 = g
    else:
        g32 = g
for (name, p), w, w32 in zip(m.named_parameters(), g64, g32):
    w = w.numpy(); s = np.abs(w).max()
    if s < 1e-9:
        print("%-24s zero: ours %.1e torch-fp32 %.1e" % (name, p.grad.abs().max().item(), w32.abs().max().item()))
        continue
    print("%-24s ours %.2e  torch-fp32 %.2e   (scale %.2e)" % (name, np.abs(p.grad.cpu().numpy() - w).max() / s, np.abs(w32.numpy() - w).max() / s, s))
