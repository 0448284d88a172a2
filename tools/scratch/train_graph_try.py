"""Is the batch-16 training forward (one C call, ~250 small launches) faster as a HIP graph replay than as eager launches?  GPU only."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import yolo_fastest_amd as yf
from yolo_fastest_amd import training
dev = torch.device("cuda:0"); io = yf.io_params_for(256)
for batch in (16, 64):
    torch.manual_seed(0)
    m = yf.YoloFastest(io); m.initialize_weights(); m = m.to(dev).train()
    x = (torch.rand(batch, 1, 256, 320) - 0.5).to(dev)
    with torch.no_grad():
        for _ in range(3):
            ref = m(x)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(20):
            m(x)
        torch.cuda.synchronize()
        te = (time.perf_counter() - t) / 20 * 1e3
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            m(x)
        torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g):
                out = m(x)
            g.replay(); torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(20):
                g.replay()
            torch.cuda.synchronize()
            tg = (time.perf_counter() - t) / 20 * 1e3
            print(f"batch {batch}: train-mode forward eager {te:.3f} ms, graph replay {tg:.3f} ms; same heads: {torch.allclose(out[0], ref[0], atol=1e-3)}")
        except Exception as e:
            print(f"batch {batch}: eager {te:.3f} ms; capture failed: {type(e).__name__}: {str(e)[:200]}")
