#!/usr/bin/env python3
"""Developer tool (GPU box): which intermediate tensors of the fused plan differ from the one-launch-per-layer plan?
   python tools/probe_diff.py conv4_2 conv5_1 res5_5 conv5_2"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import yolo_fastest_amd as yf
res = 256
io = yf.io_params_for(res)
dev = torch.device("cuda:0")
m = yf.YoloFastest(io).to(dev).eval()
m.load_state_dict(torch.load(os.path.join(ROOT, "yolo-fastest-and-embedded-deployment_amd", "assets", "weights", "yolo_fastest_256x320_epoch28.pth"), map_location=dev))
torch.manual_seed(0)
x = (torch.rand(2, 1, io["input_shape"][0], io["input_shape"][1], device=dev) - 0.5)
for name in sys.argv[1:]:
    out = []
    for f in (0, 1):
        m.fusion = f
        try:
            out.append(m.probe(x, name).cpu().numpy())
        except Exception as e:
            out.append(None); print(name, "fusion", f, "->", str(e)[:80])
    if out[0] is not None and out[1] is not None:
        d = np.abs(out[0] - out[1])
        print(f"{name:12s} shape {out[0].shape} max|ref| {np.abs(out[0]).max():.4f} max diff {d.max():.3e} at {np.unravel_index(d.argmax(), d.shape)}")
    if out[0] is not None and out[1] is not None and d.max() > 1e-2:
        bad = d > 1e-2
        print("   bad per frame", bad.reshape(bad.shape[0], -1).sum(1), "per channel", bad.sum((0, 2, 3)), "\n   per row", bad.sum((0, 1, 3)), "per col", bad.sum((0, 1, 2)))
