"""The engine's forward pass captured with the HIP runtime's own calls (no torch.cuda.graph): hipStreamBeginCapture on a fresh stream,
yf_forward, hipStreamEndCapture, hipGraphInstantiate, one replay.  Separates torch's capture machinery from the runtime's.
   python tools/cap_try2.py LANES BRANCHES   (env YF_SEGV_TRACE=1: native backtrace on SIGSEGV through tools/libsegv_trace.so)"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import yolo_fastest_amd as yf
from yolo_fastest_amd import _lib
dev = torch.device("cuda:0")
io = yf.io_params_for(256)
lanes, branches = int(sys.argv[1]), int(sys.argv[2])
m = yf.YoloFastest(io).to(dev).eval(); m.lanes = lanes; m.branches = branches
m.chunk = 32 if lanes == 2 else 0
m.load_state_dict(torch.load(os.path.join(ROOT, "yolo-fastest-and-embedded-deployment_amd/assets/weights/yolo_fastest_256x320_epoch28.pth"), map_location=dev))
x = torch.randn(64, 1, 256, 320, device=dev)
with torch.no_grad(): ref = m(x)
torch.cuda.synchronize()
hip = ctypes.CDLL("libamdhip64.so")
def ck(rc, what):
    if rc != 0: raise SystemExit(f"{what} -> hip error {rc}")
s = ctypes.c_void_p()
ck(hip.hipStreamCreateWithFlags(ctypes.byref(s), 1), "hipStreamCreateWithFlags")
e = m.engine(256, 320, 64, dev)
hl = torch.empty_like(ref[0]); hs = torch.empty_like(ref[1])
ws = e.workspace(64, dev)
def fwd():
    _lib.check(e.lib.yf_forward(e.handle, x.data_ptr(), 64, hl.data_ptr(), hs.data_ptr(), ws.data_ptr(), ws.numel(), s))
fwd(); torch.cuda.synchronize()
if os.environ.get("YF_SEGV_TRACE"):
    ctypes.CDLL(os.path.join(ROOT, "tools", "libsegv_trace.so")).segv_trace_install()
print("capturing (raw HIP)", lanes, branches, flush=True)
ck(hip.hipStreamBeginCapture(s, 0), "hipStreamBeginCapture")
fwd()
g = ctypes.c_void_p()
print("ending capture", flush=True)
ck(hip.hipStreamEndCapture(s, ctypes.byref(g)), "hipStreamEndCapture")
print("captured", flush=True)
ge = ctypes.c_void_p()
ck(hip.hipGraphInstantiate(ctypes.byref(ge), g, None, None, 0), "hipGraphInstantiate")
print("instantiated", flush=True)
hl.zero_(); hs.zero_(); torch.cuda.synchronize()
ck(hip.hipGraphLaunch(ge, s), "hipGraphLaunch")
ck(hip.hipStreamSynchronize(s), "hipStreamSynchronize")
print("replay equal:", torch.equal(hl, ref[0]) and torch.equal(hs, ref[1]), flush=True)
