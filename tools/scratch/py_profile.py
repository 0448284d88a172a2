"""cProfile of the training iteration's host side (batch 16, where the host is the slower side): top functions by own time.  GPU only."""
import cProfile, os, pstats, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import yolo_fastest_amd as yf
from yolo_fastest_amd import training, validation as val
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0"); io = yf.io_params_for(256)
torch.manual_seed(0)
m = yf.YoloFastest(io); m.initialize_weights(); m = m.to(dev).train()
x = (torch.rand(batch, 1, 256, 320) - 0.5).to(dev)
t = np.zeros((batch, 64, 6), np.float32); t[:, 0] = (0.5, 0.5, 0.2, 0.2, 1, 255.0)
td = torch.from_numpy(t).to(dev)
crit = [val.YOLOLossV3(io["anchors"][i], 3, io["input_shape"], dev, model=m) for i in range(2)]
opt = training.Adam(m.parameters(), lr=0.001)
for _ in range(5):
    training.train_step(m, crit, opt, x, td)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    training.train_step(m, crit, opt, x, td)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(22)
