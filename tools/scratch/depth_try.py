"""Batches in flight: 2 vs 3 (vs 4), each on tuned streams (BatchPipeline.tune_streams).  GPU only.  python tools/depth_try.py [K] [dtype]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import yolo_fastest_amd as yf
dev = torch.device("cuda:0"); io = yf.io_params_for(256)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 40
prec = sys.argv[2] if len(sys.argv) > 2 else "f32"
W = os.path.join(ROOT, "yolo-fastest-and-embedded-deployment_amd/assets/weights/yolo_fastest_256x320_epoch28.pth")
x = ((torch.randint(0, 256, (256, 256, 320), dtype=torch.uint8).float() - 128.0) / 255.0)[:, None].contiguous().to(dev)
for depth in (2, 3, 4, 2, 3):
    m = yf.YoloFastest(io).to(dev).eval(); m.precision = prec; m.load_state_dict(torch.load(W, map_location=dev))
    p = yf.YOLO_post_process(io["conf_thre"], io["nms_thre"], io["num_anchors"], io["num_cls"], io["anchors"], io["input_shape"]).bind(m)
    pipe = yf.BatchPipeline(m, p, depth=depth, kmax=64, lanes=1, branches=0)
    rates = pipe.tune_streams(x)
    out = []
    for r in range(3):
        for _ in range(2 * depth):
            pipe.submit(x)
        pipe.drain(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(K):
            tk = pipe.submit(x)
        pipe.drain(); torch.cuda.synchronize()
        out.append(256 * K / (time.perf_counter() - t) / 1e3)
    print(f"{prec} GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', 'default')} depth {depth}: " + " ".join("%.1f" % v for v in out)
          + "  (tuning candidates: " + " ".join("%.0f" % (r / 1e3) for r in rates) + ")", flush=True)
    del pipe, m, p
