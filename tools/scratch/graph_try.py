import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import yolo_fastest_amd as yf
dev = torch.device("cuda:0")
io = yf.io_params_for(256)
for lanes, NB in ((1, 256), (2, 256), (1, 1), (1, 8)):
    m = yf.YoloFastest(io).to(dev).eval(); m.lanes = lanes
    m.load_state_dict(torch.load(os.path.join(ROOT, "yolo-fastest-and-embedded-deployment_amd/assets/weights/yolo_fastest_256x320_epoch28.pth"), map_location=dev))
    x = torch.randn(NB, 1, 256, 320, device=dev)
    with torch.no_grad():
        ref = m(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.no_grad():
        m(x)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    try:
        with torch.cuda.graph(g), torch.no_grad():
            out = m(x)
        g.replay(); torch.cuda.synchronize()
        ok = torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])
        t0 = time.perf_counter()
        for _ in range(20): g.replay()
        torch.cuda.synchronize()
        tg = (time.perf_counter() - t0) / 20 * 1e3
        t0 = time.perf_counter()
        with torch.no_grad():
            for _ in range(20): m(x)
        torch.cuda.synchronize()
        te = (time.perf_counter() - t0) / 20 * 1e3
        print(f"batch={NB} lanes={lanes}: capture ok, replay equals eager: {ok}; graph {tg:.3f} ms/forward vs eager {te:.3f} ms")
    except Exception as e:
        print(f"lanes={lanes}: capture FAILED: {type(e).__name__}: {str(e)[:300]}")
