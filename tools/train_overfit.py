#!/usr/bin/env python3
"""End-to-end sanity of the training path beyond two iterations: from initialize_weights(), overfit the first 16 bundled frames (targets
of the mAP golden) with the reference's optimizer settings, and follow the same run in the CPU oracle (torch autograd on
oracle/backbone_oracle.py + loss_oracle.py + torch.optim.Adam) for the first iterations: the two loss trajectories should track each
other until rounding noise has had time to grow.  GPU box: python tools/train_overfit.py [iterations] [oracle iterations]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import yolo_fastest_amd as yf  # noqa: E402
from yolo_fastest_amd import training, validation as val  # noqa: E402
from oracle import backbone_oracle as bo, loss_oracle as lo  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
oiters = int(sys.argv[2]) if len(sys.argv) > 2 else 25
g = np.load(os.path.join(ROOT, "tests", "golden", "golden_256.npz"))
gm = np.load(os.path.join(ROOT, "tests", "golden", "golden_map_256.npz"))
dev = torch.device("cuda:0")
io = yf.io_params_for(256)
torch.manual_seed(0)
m = yf.YoloFastest(io)
m.initialize_weights()
sd0 = {k: v.clone() for k, v in m.state_dict().items()}
m = m.to(dev).train()
x = (torch.from_numpy(g["input_u8"][:16].astype(np.float32))[:, None] - 128.0) / 255.0
t = torch.from_numpy(gm["targets"][:16].astype(np.float32))
crit = [val.YOLOLossV3(io["anchors"][i], 3, io["input_shape"], dev, model=m) for i in range(2)]
opt = training.Adam(m.parameters(), lr=0.001)
xd, td = x.to(dev), t.to(dev)
gpu = []
t0 = time.time()
for it in range(iters):
    gpu.append(float(training.train_step(m, crit, opt, xd, td)[0].detach()))
torch.cuda.synchronize()
print("GPU: %d iterations in %.1f s; loss %s" % (iters, time.time() - t0, " ".join("%d:%.4f" % (i, gpu[i]) for i in range(0, iters, max(1, iters // 12)))),
      "last %.4f" % gpu[-1])
torch.set_num_threads(16)
sd = bo.training_state(sd0)
params = [sd[k] for k in bo.parameter_keys(sd)]
copt = torch.optim.Adam(params, lr=0.001)
cpu = []
for it in range(oiters):
    copt.zero_grad()
    hl, hs = bo.forward(sd, x, train=True)
    total = sum(lo.loss_head(h, t, io["anchors"][i], 3, io["input_shape"])[0] for i, h in enumerate((hl, hs)))
    total.backward()
    copt.step()
    cpu.append(float(total.detach()))
print("iteration: GPU loss | CPU-oracle loss | relative difference")
for i in range(oiters):
    print("%3d: %.5f | %.5f | %.1e" % (i, gpu[i], cpu[i], abs(gpu[i] - cpu[i]) / abs(cpu[i])))
# after training: the inference engine on the trained weights detects the training targets
m.eval()
post = yf.YOLO_post_process(io["conf_thre"], io["nms_thre"], io["num_anchors"], io["num_cls"], io["anchors"], io["input_shape"]).bind(m)
with torch.no_grad():
    lists = post.detect(m(xd))
ntar = [int((gm["targets"][f, :, 5] > 1).sum()) for f in range(16)]
print("targets per frame   ", ntar)
print("detections per frame", [len(l) for l in lists])
