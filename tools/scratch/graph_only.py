"""30 forward passes of batch 256 on two lanes, either eager or as hipGraph replays (argv[1] = eager | graph): run under
rocprofv3 --kernel-trace and feed the trace to tools/trace_overlap.py to see how the two lanes' kernels overlap in each mode."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import yolo_fastest_amd as yf
mode = sys.argv[1] if len(sys.argv) > 1 else "eager"
dev = torch.device("cuda:0")
io = yf.io_params_for(256)
m = yf.YoloFastest(io).to(dev).eval(); m.lanes = 2
m.load_state_dict(torch.load(os.path.join(ROOT, "yolo-fastest-and-embedded-deployment_amd/assets/weights/yolo_fastest_256x320_epoch28.pth"), map_location=dev))
x = torch.randn(256, 1, 256, 320, device=dev)
with torch.no_grad():
    for _ in range(3): m(x)
torch.cuda.synchronize()
if mode == "graph":
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.no_grad(): m(x)
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    with torch.cuda.graph(g), torch.no_grad(): out = m(x)
    for _ in range(30): g.replay()
else:
    with torch.no_grad():
        for _ in range(30): m(x)
torch.cuda.synchronize()
