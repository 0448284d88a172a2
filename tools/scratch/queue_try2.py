"""queue_try.py after ANOTHER engine has run two lanes on the default stream (what bench.py's process looks like when it measures the
second scheduling mode): is the two-batches-in-flight rate then a property of the stream pair?  GPU only."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import yolo_fastest_amd as yf
dev = torch.device("cuda:0")
io = yf.io_params_for(256)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
P = int(sys.argv[2]) if len(sys.argv) > 2 else 10
W = os.path.join(ROOT, "yolo-fastest-and-embedded-deployment_amd/assets/weights/yolo_fastest_256x320_epoch28.pth")
x = ((torch.randint(0, 256, (256, 256, 320), dtype=torch.uint8).float() - 128.0) / 255.0)[:, None].contiguous().to(dev)


def mk(lanes, branches):
    m = yf.YoloFastest(io).to(dev).eval(); m.load_state_dict(torch.load(W, map_location=dev)); m.lanes, m.branches = lanes, branches
    return m, yf.YOLO_post_process(io["conf_thre"], io["nms_thre"], io["num_anchors"], io["num_cls"], io["anchors"], io["input_shape"]).bind(m)


m1, p1 = mk(2, 0)
m, p = mk(1, 0)


def one():
    with torch.no_grad():
        for _ in range(5):
            p1.detect_raw(m1(x), kmax=64)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(K):
            p1.detect_raw(m1(x), kmax=64)
        torch.cuda.synchronize()
    return 256 * K / (time.perf_counter() - t) / 1e3


def two(pipe):
    for _ in range(6):
        pipe.submit(x)
    pipe.drain(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(K):
        tk = pipe.submit(x)
    pipe.drain(); tk.synchronize(); torch.cuda.synchronize()
    return 256 * K / (time.perf_counter() - t) / 1e3


order = sys.argv[3] if len(sys.argv) > 3 else "one-first"
if order == "one-first":
    print(f"one at a time first: {one():.1f}")
for i in range(P):
    pipe = yf.BatchPipeline(m, p, depth=2, kmax=64, lanes=1, branches=0)
    a = two(pipe)
    rep = pipe._fix_collisions()       # replaces streams that share a hardware queue (yf_streams_overlap)
    b = two(pipe)
    rates = pipe.tune_streams(x)       # ... and times candidate sets on real batches
    c = two(pipe)
    print(f"pipeline {i}: as created {a:.1f}, collisions fixed ({rep} replaced) {b:.1f}, tuned {c:.1f} k frames/s (candidates: "
          + " ".join("%.0f" % (r / 1e3) for r in rates) + f")   (one at a time again: {one():.1f})", flush=True)
