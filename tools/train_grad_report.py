#!/usr/bin/env python3
"""Per parameter tensor: how far our training-step gradient and the reference's own fp32 gradient are from the exact backward
(tests/golden/golden_train_256.npz: grads_1 / grads_1_exact).  GPU box: python tools/train_grad_report.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import yolo_fastest_amd as yf  # noqa: E402
from yolo_fastest_amd import training, validation as val  # noqa: E402

gt = np.load(os.path.join(ROOT, "tests", "golden", "golden_train_256.npz"))
dev = torch.device("cuda:0")
io = yf.io_params_for(256)
m = yf.YoloFastest(io).to(dev)
m.load_state_dict(torch.load(os.path.join(ROOT, "yolo-fastest-and-embedded-deployment_amd", "assets", "weights",
                                          "yolo_fastest_256x320_epoch28.pth"), map_location=dev))
m.train()
x = ((torch.from_numpy(gt["input_u8"].astype(np.float32))[:, None] - 128.0) / 255.0).to(dev)
targets = torch.from_numpy(gt["targets"]).to(dev)
crit = [val.YOLOLossV3(io["anchors"][i], io["num_cls"], io["input_shape"], dev, model=m) for i in range(2)]
pred = m(x)
loss = sum(crit[i](p, targets)[0] for i, p in enumerate(pred))
loss.backward()
sizes = gt["param_sizes"]
off = np.concatenate([[0], np.cumsum(sizes)])
rows = []
for i, (name, p) in enumerate(m.named_parameters()):
    if gt["grad_absmax_f64"][i] < 1e-9:
        continue
    g = p.grad.detach().cpu().numpy().ravel()
    ref, ex = gt["grads_1"][off[i]:off[i + 1]], gt["grads_1_exact"][off[i]:off[i + 1]]
    s = np.abs(ex).max()
    rows.append((np.abs(g - ex).max() / s, np.abs(ref - ex).max() / s, np.abs(g - ref).max() / s, name))
o, t = np.array([r[0] for r in rows]), np.array([r[1] for r in rows])
print("tensors %d | ours: median %.2e max %.2e | reference fp32: median %.2e max %.2e | ours > theirs in %d" %
      (len(rows), np.median(o), o.max(), np.median(t), t.max(), (o > t).sum()))
for r in sorted(rows, key=lambda r: -r[0] / max(r[1], 1e-6))[:25]:
    print("%-24s ours %.2e theirs %.2e ours-vs-theirs %.2e" % (r[3], r[0], r[1], r[2]))
