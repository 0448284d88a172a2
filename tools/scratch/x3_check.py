"""f16x3 (split-operand fp16 MFMA, fp32 storage) against the reference goldens, beside fp32 and plain f16.  GPU only."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import yolo_fastest_amd as yf
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WDIR = os.path.join(ROOT, "yolo-fastest-and-embedded-deployment_amd", "assets", "weights")
WEIGHTS = {256: os.path.join(WDIR, "yolo_fastest_256x320_epoch28.pth"), 512: os.path.join(WDIR, "yolo_fastest_512x640_epoch27.pth")}
for res in (256, 512):
    g = np.load(os.path.join(ROOT, "tests", "golden", f"golden_{res}.npz"))
    io = yf.io_params_for(res)
    x = ((torch.from_numpy(g["input_u8"]).to("cuda:0").float() - 128.0) / 255.0).unsqueeze(1).contiguous()
    for prec in ("f32", "f16x3", "f16"):
        m = yf.YoloFastest(io).to("cuda:0").eval()
        m.load_state_dict(torch.load(WEIGHTS[res], map_location="cuda:0"))
        m.precision = prec
        with torch.no_grad():
            hl, hs = m(x)
        for name, got in (("head_large", hl), ("head_small", hs)):
            d = np.abs(got.cpu().numpy() - g[name]); d64 = np.abs(got.cpu().numpy() - g[name + "_f64"])
            print(f"{res} {prec:6s} {name}: vs ref fp32 max {d.max():.3e} mean {d.mean():.2e} | vs fp64 max {d64.max():.3e} "
                  f"(reference fp32 vs fp64 {np.abs(g[name] - g[name + '_f64']).max():.3e})")
