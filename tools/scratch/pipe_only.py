"""60 batches through BatchPipeline (two in flight, one lane each): run under rocprofv3 --kernel-trace and feed the trace to
tools/trace_overlap.py to see which kernels still run alone."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import yolo_fastest_amd as yf
prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
dev = torch.device("cuda:0")
io = yf.io_params_for(256)
m = yf.YoloFastest(io).to(dev).eval(); m.precision = prec
m.load_state_dict(torch.load(os.path.join(ROOT, "yolo-fastest-and-embedded-deployment_amd/assets/weights/yolo_fastest_256x320_epoch28.pth"), map_location=dev))
post = yf.YOLO_post_process(io["conf_thre"], io["nms_thre"], io["num_anchors"], io["num_cls"], io["anchors"], io["input_shape"]).bind(m)
x = ((torch.randint(0, 256, (256, 256, 320), dtype=torch.uint8).float() - 128.0) / 255.0)[:, None].contiguous().to(dev)
pipe = yf.BatchPipeline(m, post, depth=2)
for _ in range(60): pipe.submit(x)
pipe.drain()
torch.cuda.synchronize()
