"""Two batches in flight, SHORT runs: do the two streams need a phase offset?  After a device sync both start at the stem together and
compete kernel by kernel; a delay on the second stream's first batch (torch.cuda._sleep: a one-workgroup spin) puts it half a pass
behind.  K timed steps after a sync, like bench.py.  GPU only.   python tools/stagger_try.py [K] [rounds]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import yolo_fastest_amd as yf
dev = torch.device("cuda:0")
io = yf.io_params_for(256)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
R = int(sys.argv[2]) if len(sys.argv) > 2 else 5
W = os.path.join(ROOT, "yolo-fastest-and-embedded-deployment_amd/assets/weights/yolo_fastest_256x320_epoch28.pth")
x = ((torch.randint(0, 256, (256, 256, 320), dtype=torch.uint8).float() - 128.0) / 255.0)[:, None].contiguous().to(dev)
m = yf.YoloFastest(io).to(dev).eval(); m.load_state_dict(torch.load(W, map_location=dev))
p = yf.YOLO_post_process(io["conf_thre"], io["nms_thre"], io["num_anchors"], io["num_cls"], io["anchors"], io["input_shape"]).bind(m)
m1 = yf.YoloFastest(io).to(dev).eval(); m1.load_state_dict(torch.load(W, map_location=dev)); m1.lanes, m1.branches = 2, 0
p1 = yf.YOLO_post_process(io["conf_thre"], io["nms_thre"], io["num_anchors"], io["num_cls"], io["anchors"], io["input_shape"]).bind(m1)
pipe = yf.BatchPipeline(m, p, depth=2, kmax=64, lanes=1, branches=0)
# cycles per ms of torch.cuda._sleep
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record(); torch.cuda._sleep(10_000_000); t1.record(); torch.cuda.synchronize()
cyc_per_ms = 10_000_000 / t0.elapsed_time(t1)


def run_two(delay_ms):
    for _ in range(6):
        pipe.submit(x)
    pipe.drain(); torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(K):
        if i == 1 and delay_ms > 0:
            with torch.cuda.stream(pipe.streams[1]):
                torch.cuda._sleep(int(delay_ms * cyc_per_ms))
        tk = pipe.submit(x)
    pipe.drain(); tk.synchronize(); torch.cuda.synchronize()
    return 256 * K / (time.perf_counter() - t)


def run_one():
    with torch.no_grad():
        for _ in range(5):
            p1.detect_raw(m1(x), kmax=64)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(K):
            p1.detect_raw(m1(x), kmax=64)
        torch.cuda.synchronize()
    return 256 * K / (time.perf_counter() - t)


res = {}
for r in range(R):
    res.setdefault("one at a time, 2 lanes", []).append(run_one())
    for d in (0.0, 0.2, 0.35, 0.5, 0.65):
        res.setdefault("two in flight, delay %.2f ms" % d, []).append(run_two(d))
for k, v in res.items():
    print(f"K={K} {k:32s} " + " ".join(f"{a / 1e3:7.1f}" for a in v) + f"   median {sorted(v)[len(v) // 2] / 1e3:.1f} k frames/s")
