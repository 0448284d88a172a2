#!/usr/bin/env python3
"""Prints, for every golden set, how far the HIP heads and the reference's fp32 heads are from the reference graph
evaluated in fp64 (needs a GPU).  Numbers quoted in DESIGN.md."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import yolo_fastest_amd as yf
from oracle import backbone_oracle as bo
dev = torch.device("cuda:0")
W = {256: "yolo_fastest_256x320_epoch28.pth", 512: "yolo_fastest_512x640_epoch27.pth"}
sig = lambda a: 1 / (1 + np.exp(-a.astype(np.float64)))
for res in (256, 512):
    m = yf.YoloFastest(yf.io_params_for(res)).to(dev).eval()
    m.load_state_dict(torch.load(os.path.join(ROOT, "yolo-fastest-and-embedded-deployment_amd/assets/weights", W[res]), map_location=dev))
    g = np.load(os.path.join(ROOT, f"tests/golden/golden_{res}.npz"))
    with torch.no_grad():
        hl, hs = m(bo.preprocess(g["input_u8"]).to(dev))
    for nm, got in (("large", hl.cpu().numpy()), ("small", hs.cpu().numpy())):
        r32, r64 = g[f"head_{nm}"], g[f"head_{nm}_f64"]
        print(f"{res} head_{nm}: |hip-ref64|={np.abs(got-r64).max():.3e} |ref32-ref64|={np.abs(r32-r64).max():.3e} "
              f"|hip-ref32|={np.abs(got-r32).max():.3e} score|hip-ref32|={np.abs(sig(got)-sig(r32)).max():.3e} "
              f"rms hip={np.sqrt(((got-r64)**2).mean()):.3e} rms ref32={np.sqrt(((r32-r64)**2).mean()):.3e}")
