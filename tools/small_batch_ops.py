"""Per-launch times of one pass at small batch sizes: where the 0.43 ms of a batch-1 pass go.
   python tools/small_batch_ops.py [res] [dtype]      (run on the GPU box)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import yolo_fastest_amd as yf

res = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
dev = torch.device("cuda:0")
io = yf.io_params_for(res)
W = {256: "yolo_fastest_256x320_epoch28.pth", 512: "yolo_fastest_512x640_epoch27.pth"}[res]
m = yf.YoloFastest(io).to(dev).eval()
m.load_state_dict(torch.load(os.path.join(ROOT, "yolo-fastest-and-embedded-deployment_amd", "assets", "weights", W), map_location=dev))
if dtype != "f32":
    m.precision = dtype
H, Wd = io["input_shape"][:2]
cols = {}
for n in [int(v) for v in os.environ.get("YF_SMALL_NS", "1,2,4,8,16,256").split(",")]:
    x = ((torch.randint(0, 256, (n, 1, H, Wd), generator=torch.Generator().manual_seed(0)).float() - 128.0) / 255.0).to(dev)
    m.profile(x, reps=2, launch_repeats=4)
    cols[n] = m.profile(x, reps=5, launch_repeats=4)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    with torch.no_grad():
        for _ in range(5): m(x)
        torch.cuda.synchronize(); t0.record()
        for _ in range(20): m(x)
        t1.record(); torch.cuda.synchronize()
    cols[n].append(dict(name="model(x), eager, per call", ms=t0.elapsed_time(t1) / 20))
names = [o["name"] for o in cols[next(iter(cols))]]
print("%-44s" % "launch (us)" + "".join("%9s" % ("N=%d" % n) for n in cols))
for i, nm in enumerate(names):
    print("%-44s" % nm[:44] + "".join("%9.1f" % (cols[n][i]["ms"] * 1e3) for n in cols))
print("%-44s" % "sum of launches" + "".join("%9.1f" % (sum(o["ms"] for o in cols[n][:-1]) * 1e3) for n in cols))
