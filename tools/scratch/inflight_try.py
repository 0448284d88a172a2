"""Throughput with several batches in flight: S streams, each with its own engine, steps issued round-robin.  GPU only.
   python tools/inflight_try.py [dtype]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import yolo_fastest_amd as yf
dev = torch.device("cuda:0")
io = yf.io_params_for(256)
prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
W = os.path.join(ROOT, "yolo-fastest-and-embedded-deployment_amd/assets/weights/yolo_fastest_256x320_epoch28.pth")
x = ((torch.randint(0, 256, (256, 256, 320), dtype=torch.uint8).float() - 128.0) / 255.0)[:, None].contiguous().to(dev)
CH = int(os.environ.get("YF_CHUNK", "0"))
for S, lanes, stagger, branches in ((1, 2, 0, 1), (2, 1, 0, 0), (2, 1, 0, 1), (3, 1, 0, 0)):
    ms, ps, ss = [], [], [torch.cuda.Stream() for _ in range(S)]
    for i in range(S):
        m = yf.YoloFastest(io).to(dev).eval(); m.lanes = lanes; m.precision = prec; m.branches = branches; m.chunk = CH if S > 1 else 0
        m.load_state_dict(torch.load(W, map_location=dev))
        ms.append(m)
        ps.append(yf.YOLO_post_process(io["conf_thre"], io["nms_thre"], io["num_anchors"], io["num_cls"], io["anchors"], io["input_shape"]).bind(m))
    def step(i):
        with torch.cuda.stream(ss[i]), torch.no_grad():
            return ps[i].detect_raw(ms[i](x), kmax=64)
    for k in range(10 * S): step(k % S)
    torch.cuda.synchronize()
    if stagger:   # put stream 0 half a pass ahead
        with torch.cuda.stream(ss[0]), torch.no_grad():
            ms[0](x[:128])
    K = 100
    t0 = time.perf_counter()
    for k in range(K): step(k % S)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{prec} chunk={CH} streams={S} lanes={lanes} stagger={stagger} branches={branches}: {256 * K / dt:9.0f} frames/s  ({1e3 * dt / K:.3f} ms per step)", flush=True)
    del ms, ps
