"""Interleaved A/B of the fusion levels in ONE process (yf_set_fusion 1 = round 2's block-fused plan, 2 = + the deep-stage fusions):
per-launch times of both plans, then throughput -- one batch at a time (two lanes, small head on its side stream) and two batches in
flight (BatchPipeline) -- in alternating rounds.  GPU only.
   python tools/fusion_ab.py [dtype] [rounds] [steps] [levels]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import yolo_fastest_amd as yf
dev = torch.device("cuda:0")
io = yf.io_params_for(256)
prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
K = int(sys.argv[3]) if len(sys.argv) > 3 else 100
levels = [int(c) for c in (sys.argv[4] if len(sys.argv) > 4 else "12")]
W = os.path.join(ROOT, "yolo-fastest-and-embedded-deployment_amd/assets/weights/yolo_fastest_256x320_epoch28.pth")
x = ((torch.randint(0, 256, (256, 256, 320), dtype=torch.uint8).float() - 128.0) / 255.0)[:, None].contiguous().to(dev)


def make(level, lanes, branches):
    m = yf.YoloFastest(io).to(dev).eval(); m.lanes = lanes; m.precision = prec; m.branches = branches; m.fusion = level
    m.load_state_dict(torch.load(W, map_location=dev))
    p = yf.YOLO_post_process(io["conf_thre"], io["nms_thre"], io["num_anchors"], io["num_cls"], io["anchors"], io["input_shape"]).bind(m)
    return m, p


ONE = [(2, 1), (2, 0), (1, 1)] if os.environ.get("YF_AB_ONE_ALL") else [(2, 1)]   # (lanes, branches) of the one-at-a-time runs
models = {lvl: (make(lvl, 2, 1), make(lvl, 1, 0)) for lvl in levels}
extra = {(lvl, lb): make(lvl, *lb) for lvl in levels for lb in ONE[1:]}
ref = None
for lvl in levels:
    m, _ = models[lvl][0]
    m.profile(x, reps=2)
    ops = m.profile(x, reps=8)
    print(f"---- fusion {lvl}: {len(ops)} launches, sum {sum(o['ms'] for o in ops) * 1e3:.1f} us")
    for o in ops:
        print(f"{o['ms'] * 1e3:8.1f} us  {o['name'][:90]}")
    with torch.no_grad():
        out = m(x)
    if ref is None:
        ref = out
    else:
        print("bitwise equal heads vs the first level:", torch.equal(ref[0], out[0]) and torch.equal(ref[1], out[1]))


def one_at_a_time(lvl, lb=(2, 1)):
    m, p = models[lvl][0] if lb == (2, 1) else extra[(lvl, lb)]
    with torch.no_grad():
        for _ in range(10):
            p.detect_raw(m(x), kmax=64)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            p.detect_raw(m(x), kmax=64)
        torch.cuda.synchronize()
    return 256 * K / (time.perf_counter() - t0)


def two_in_flight(lvl):
    m, p = models[lvl][1]
    pipe = yf.BatchPipeline(m, p, depth=2, kmax=64, lanes=1, branches=0)
    for _ in range(10):
        pipe.submit(x)
    pipe.drain(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        t = pipe.submit(x)
    pipe.drain(); t.synchronize(); torch.cuda.synchronize()
    return 256 * K / (time.perf_counter() - t0)


res = {(lvl, k): [] for lvl in levels for k in ["two"] + ["one l%d b%d" % lb for lb in ONE]}
for r in range(rounds):
    for lvl in levels:
        for lb in ONE:
            res[(lvl, "one l%d b%d" % lb)].append(one_at_a_time(lvl, lb))
        res[(lvl, "two")].append(two_in_flight(lvl))
for (lvl, k), v in sorted(res.items()):
    print(f"{prec} fusion {lvl} {k:>9} in flight: " + " ".join(f"{a / 1e3:7.1f}" for a in v) + f"  k frames/s  (mean {sum(v) / len(v) / 1e3:.1f})")
