#!/usr/bin/env python3
"""Per-layer times of one training iteration (GPU box): runs tools/train_bench.py with YF_TRAIN_TIMING=1 (yf_train_engine.hip records an
event after every layer of a pass) and prints the table of the LAST iteration: conv / BatchNorm forward, BatchNorm / weight-gradient /
data-gradient backward per layer, and the sums.   python tools/train_timing.py [--batch 256] [--full]"""
import collections, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
batch = sys.argv[sys.argv.index("--batch") + 1] if "--batch" in sys.argv else "256"
env = dict(os.environ, YF_TRAIN_TIMING="1")
out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train_bench.py"), "--batch", batch, "--steps", "2", "--warmup", "2"], env=env,
                     capture_output=True, text=True)
rows = [l.split() for l in out.stderr.splitlines() if l.startswith("[yf_train_timing]")]
print(out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-2000:])
blocks = []
for r in rows:
    if r[1] == "fwd" and r[2] == "conv" and r[3] == "conv0":
        blocks.append([])
    blocks[-1].append(r)
agg = collections.OrderedDict(); by = collections.OrderedDict()
for r in blocks[-1]:
    us = float(r[-2]); agg[r[1] + "." + r[2]] = agg.get(r[1] + "." + r[2], 0.0) + us
    by.setdefault(r[3], {})[r[1] + "." + r[2]] = us
print("  ".join(f"{k} {v:.0f}" for k, v in agg.items()), f"  total {sum(agg.values()):.0f} us")
if "--full" in sys.argv:
    cols = ["fwd.conv", "fwd.bn", "bwd.bn", "bwd.wgrad", "bwd.dgrad", "bwd.head", "fwd.cat"]
    print(f"{'layer':16s}" + "".join(f"{c:>10s}" for c in cols))
    for l, d in by.items():
        print(f"{l:16s}" + "".join(f"{d.get(c, 0):10.0f}" for c in cols))
