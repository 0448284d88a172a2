import os, sys, numpy as np, torch
if not os.environ.get('YF_SEGV_TRACE'):
    import faulthandler; faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import yolo_fastest_amd as yf
dev = torch.device("cuda:0")
io = yf.io_params_for(256)
lanes, branches = int(sys.argv[1]), int(sys.argv[2])
m = yf.YoloFastest(io).to(dev).eval(); m.lanes = lanes; m.branches = branches
m.chunk = 32 if lanes == 2 else 0
m.load_state_dict(torch.load(os.path.join(ROOT, "yolo-fastest-and-embedded-deployment_amd/assets/weights/yolo_fastest_256x320_epoch28.pth"), map_location=dev))
x = torch.randn(64, 1, 256, 320, device=dev)
with torch.no_grad(): ref = m(x)
torch.cuda.synchronize()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s), torch.no_grad(): m(x)
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
if os.environ.get("YF_SEGV_TRACE"):
    import ctypes
    ctypes.CDLL(os.path.join(ROOT, "tools", "libsegv_trace.so")).segv_trace_install()
print("capturing", lanes, branches, flush=True)
with torch.cuda.graph(g), torch.no_grad():
    out = m(x)
print("captured", flush=True)
g.replay(); torch.cuda.synchronize()
print("replay equal:", torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]), flush=True)
