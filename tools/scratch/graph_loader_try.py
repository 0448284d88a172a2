"""Do the trainer's graph replays engage in a data-loader style loop (a NEW batch tensor moved to the GPU every iteration)?  GPU only."""
import os, sys, ctypes, time, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import yolo_fastest_amd as yf
from yolo_fastest_amd import training, validation as val
dev = torch.device('cuda:0'); io = yf.io_params_for(256); torch.manual_seed(3)
m = yf.YoloFastest(io); m.initialize_weights(); m = m.to(dev).train()
crit = [val.YOLOLossV3(io['anchors'][i], 3, io['input_shape'], dev, model=m) for i in range(2)]
opt = training.Adam(m.parameters(), lr=0.001)
t = np.zeros((16, 64, 6), np.float32); t[:, 0] = (0.4, 0.6, 0.3, 0.2, 1, 255.0)
batches = [(torch.rand(16, 1, 256, 320) - 0.5, torch.from_numpy(t.copy())) for _ in range(8)]
for it in range(40):
    if it == 10:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    imgs, tg = batches[it % 8]
    imgs, tg = imgs.to(dev), tg.to(dev)                   # train.py:111-112
    loss = training.train_step(m, crit, opt, imgs, tg)[0]
torch.cuda.synchronize()
tr = training._trainer(m, 256, 320, dev); f, b = ctypes.c_long(), ctypes.c_long()
tr.lib.yf_trainer_graph_replays(tr.handle, ctypes.byref(f), ctypes.byref(b))
print("40 iterations with a fresh batch tensor each: %d forward and %d backward replays; %.2f ms / iteration (incl. the host-to-device copies)"
      % (f.value, b.value, (time.perf_counter() - t0) / 30 * 1e3))
