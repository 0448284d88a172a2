"""Trainer graph replay across batch-size changes (debug): YF_TRAIN_GRAPH_DEBUG=1 python tools/graph_switch_try.py"""
import os, sys, ctypes, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import yolo_fastest_amd as yf
from yolo_fastest_amd import training, validation as val
dev = torch.device('cuda:0'); io = yf.io_params_for(256); torch.manual_seed(3)
m = yf.YoloFastest(io); m.initialize_weights(); m = m.to(dev).train()
x = (torch.rand(4, 1, 128, 160) - 0.5).to(dev)
t = np.zeros((4, 8, 6), np.float32); t[:, 0] = (0.4, 0.6, 0.3, 0.2, 1, 255.0); td = torch.from_numpy(t).to(dev)
crit = [val.YOLOLossV3(io['anchors'][i], 3, [128, 160, 1], dev, model=m) for i in range(2)]
opt = training.Adam(m.parameters(), lr=0.001)
for n, k in ((4, 6), (2, 4), (4, 5)):
    for it in range(k):
        print("batch", n, "iteration", it, flush=True, file=sys.stderr)
        loss = training.train_step(m, crit, opt, x[:n], td[:n])[0]
print(float(loss.detach()))
