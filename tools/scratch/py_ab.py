"""Same-box A/B of the training host path: tools/train_bench-like loop with another version of training.py and the working one,
alternating; prints wall ms and process CPU ms per iteration (at batch 16 the host is the slower side, and the box's other tenants move
the wall time by +-30 %).  First:  mkdir -p tools/variants && git show <rev>:yolo-fastest-and-embedded-deployment_amd/training.py > tools/variants/training_prev.py
   python tools/py_ab.py [batch] [steps] [rounds]"""
import importlib.util, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import yolo_fastest_amd as yf
from yolo_fastest_amd import validation as val
import yolo_fastest_amd.training as new_mod
spec = importlib.util.spec_from_file_location("yolo_fastest_amd.training_prev", os.path.join(ROOT, "tools", "variants", "training_prev.py"))
old_mod = importlib.util.module_from_spec(spec); old_mod.__package__ = "yolo_fastest_amd"; sys.modules["yolo_fastest_amd.training_prev"] = old_mod
spec.loader.exec_module(old_mod)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0"); io = yf.io_params_for(256)


def setup(mod):
    sys.modules["yolo_fastest_amd.training"] = mod; yf.training = mod
    torch.manual_seed(0)
    m = yf.YoloFastest(io); m.initialize_weights(); m = m.to(dev).train()
    x = (torch.rand(batch, 1, 256, 320) - 0.5).to(dev)
    t = np.zeros((batch, 64, 6), np.float32); t[:, 0] = (0.5, 0.5, 0.2, 0.2, 1, 255.0)
    td = torch.from_numpy(t).to(dev)
    crit = [val.YOLOLossV3(io["anchors"][i], 3, io["input_shape"], dev, model=m) for i in range(2)]
    opt = mod.Adam(m.parameters(), lr=0.001)
    return mod, m, crit, opt, x, td


def run(c, n):
    mod, m, crit, opt, x, td = c
    sys.modules["yolo_fastest_amd.training"] = mod; yf.training = mod
    for _ in range(3):
        mod.train_step(m, crit, opt, x, td)
    torch.cuda.synchronize()
    t0, c0 = time.perf_counter(), time.process_time()
    for _ in range(n):
        mod.train_step(m, crit, opt, x, td)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, (time.process_time() - c0) / n * 1e3


cs = {"prev": setup(old_mod), "new": setup(new_mod)}
for r in range(rounds):
    for k in ("prev", "new"):
        w, c = run(cs[k], steps)
        print(f"round {r} {k:5s} batch {batch}: wall {w:6.2f} ms  cpu {c:6.2f} ms per iteration", flush=True)
