#!/bin/bash
# GPU box: one batch at a time on two lanes -- the rate as a function of the streams involved (the caller's: the default stream, a torch
# pool stream, a high-priority pool stream), for several engines in one process.  Before the engine probed which of its streams overlap
# with the caller's (yf_engine.hip assign_streams) this printed 272-284 k frames/s for most combinations and 192 / 142 k for the pairs
# that happened to share a hardware queue.
for cfg in "probed"; do
python - "$cfg" <<'PY' 2>&1 | grep -v amdgpu
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import yolo_fastest_amd as yf
dev = torch.device("cuda:0"); io = yf.io_params_for(256)
W = "yolo-fastest-and-embedded-deployment_amd/assets/weights/yolo_fastest_256x320_epoch28.pth"
x = ((torch.randint(0, 256, (256, 256, 320), dtype=torch.uint8).float() - 128.0) / 255.0)[:, None].contiguous().to(dev)
res = []
for rep in range(4):
    m = yf.YoloFastest(io).to(dev).eval(); m.load_state_dict(torch.load(W, map_location=dev)); m.lanes, m.branches = 2, 0
    p = yf.YOLO_post_process(io["conf_thre"], io["nms_thre"], io["num_anchors"], io["num_cls"], io["anchors"], io["input_shape"]).bind(m)
    for S in (None, torch.cuda.Stream(), torch.cuda.Stream(priority=-1)):
        with torch.cuda.stream(S) if S is not None else torch.no_grad():
            with torch.no_grad():
                for _ in range(8): p.detect_raw(m(x), kmax=64)
                torch.cuda.synchronize(); t = time.perf_counter()
                for _ in range(20): p.detect_raw(m(x), kmax=64)
                torch.cuda.synchronize()
        res.append(256 * 20 / (time.perf_counter() - t) / 1e3)
print(sys.argv[1].ljust(36), "(default stream | pool stream | high-priority pool stream) x 4 engines:", " ".join("%.0f" % r for r in res))
PY
done
