"""Does the throughput with two batches in flight depend on WHICH HIP streams the two batches run on?  HIP multiplexes streams onto a few
hardware queues (GPU_MAX_HW_QUEUES, default 4); two streams on one queue do not overlap.  Builds the pipeline several times (fresh
torch streams each time) and times K steps each.  GPU only.   GPU_MAX_HW_QUEUES=8 python tools/queue_try.py [K] [pipelines]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import yolo_fastest_amd as yf
dev = torch.device("cuda:0")
io = yf.io_params_for(256)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 100
P = int(sys.argv[2]) if len(sys.argv) > 2 else 6
W = os.path.join(ROOT, "yolo-fastest-and-embedded-deployment_amd/assets/weights/yolo_fastest_256x320_epoch28.pth")
x = ((torch.randint(0, 256, (256, 256, 320), dtype=torch.uint8).float() - 128.0) / 255.0)[:, None].contiguous().to(dev)
m = yf.YoloFastest(io).to(dev).eval(); m.load_state_dict(torch.load(W, map_location=dev))
p = yf.YOLO_post_process(io["conf_thre"], io["nms_thre"], io["num_anchors"], io["num_cls"], io["anchors"], io["input_shape"]).bind(m)
out = []
for i in range(P):
    pipe = yf.BatchPipeline(m, p, depth=2, kmax=64, lanes=1, branches=0)
    for _ in range(6):
        pipe.submit(x)
    pipe.drain(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(K):
        tk = pipe.submit(x)
    pipe.drain(); tk.synchronize(); torch.cuda.synchronize()
    out.append(256 * K / (time.perf_counter() - t) / 1e3)
    print(f"pipeline {i}: streams {[hex(s.cuda_stream)[-6:] for s in pipe.streams]}  {out[-1]:.1f} k frames/s", flush=True)
print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', '(default)')} K={K}: " + " ".join(f"{v:.1f}" for v in out))
