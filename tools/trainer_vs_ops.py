"""Per-tensor difference between the trainer's gradients and the per-block path's (model.train_impl = "ops"), in backward order.  GPU only.
   python tools/trainer_vs_ops.py [N] [H] [W] [seed]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import yolo_fastest_amd as yf
N, H, W, seed = (int(v) for v in (sys.argv[1:5] + ["5", "192", "320", "11"][len(sys.argv) - 1:]))
dev = torch.device("cuda:0")
xr = (torch.rand(N, 1, H, W, generator=torch.Generator().manual_seed(4)) - 0.5).to(dev)
out = {}
for impl in ("trainer", "ops"):
    torch.manual_seed(seed)
    m = yf.YoloFastest(yf.io_params_for(256)); m.initialize_weights(); m = m.to(dev).train(); m.train_impl = impl
    hl, hs = m(xr)
    torch.manual_seed(1)
    ghl, ghs = torch.randn(hl.shape, device=dev), torch.randn(hs.shape, device=dev)
    torch.autograd.backward([hl, hs], [ghl, ghs])
    out[impl] = [(n, p.grad.clone()) for n, p in m.named_parameters()]
for (n, ga), (_, gb) in reversed(list(zip(out["trainer"], out["ops"]))):
    d = float((ga - gb).abs().max()); s = float(gb.abs().max())
    print(f"{n:28s} max|d| {d:9.3e}  max|g| {s:9.3e}  rel {d / max(s, 1e-30):9.2e}")
