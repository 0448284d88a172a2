import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import yolo_fastest_amd as yf
dev = torch.device("cuda:0"); io = yf.io_params_for(256)
m = yf.YoloFastest(io).to(dev).eval()
m.load_state_dict(torch.load(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "yolo-fastest-and-embedded-deployment_amd", "assets", "weights", "yolo_fastest_256x320_epoch28.pth"), map_location=dev))
x = ((torch.randint(0, 256, (1, 1, 256, 320)).float() - 128) / 255).to(dev)
def t(n=200):
    with torch.no_grad():
        for _ in range(20): m(x)
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
        for _ in range(n): m(x)
        e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for br in (1, 0, 1, 0):
    m.branches = br
    print("branches", br, "%.1f us per call" % t())
