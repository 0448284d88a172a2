"""Times yf_cv_preprocess_u8 (cvtColor + cv2.resize on the device) on a few frame geometries.  Run on the GPU box."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import yolo_fastest_amd as yf
dev = torch.device("cuda:0"); io = yf.io_params_for(256)
m = yf.YoloFastest(io).to(dev).eval()
def t(x, n=20):
    for _ in range(3): m.cv_preprocess_u8(x, io["input_shape"])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): m.cv_preprocess_u8(x, io["input_shape"])
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
for name, shape in (("bgr 480x640", (256, 480, 640, 3)), ("bgr 512x640 (exact 1/2: area)", (256, 512, 640, 3)), ("bgr 256x320 (cvtColor only)", (256, 256, 320, 3)),
                    ("bgr 720x1280", (128, 720, 1280, 3)), ("gray 480x640", (256, 480, 640)), ("gray 128x160 (up-scaling)", (256, 128, 160)), ("bgr 481x641 (unaligned rows: byte gather)", (64, 481, 641, 3))):
    x = torch.randint(0, 256, shape, dtype=torch.uint8).to(dev); ms = t(x)
    b = x.numel() + shape[0] * 256 * 320
    print("%-44s %7.1f us  %6.0f GB/s (source + destination bytes)" % (name, ms * 1e3, b / ms / 1e6))
