#!/usr/bin/env python3
"""Overlap analysis of a rocprofv3 --kernel-trace csv: per kernel name, how much of its wall time ran alone vs beside another
kernel, over the steady-state part of the run (last `--frac` of the trace).
   python tools/trace_overlap.py gpurun_out/prof_default/*_kernel_trace.csv"""
import csv, sys, collections, argparse
ap = argparse.ArgumentParser(); ap.add_argument("csv"); ap.add_argument("--frac", type=float, default=0.5); ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--skip-tail", type=int, default=0, help="ignore the last N post_kernel launches (bench.py measures the one-batch-at-a-time "
                "mode AFTER the headline loop: 20 + warmup + steps of them with the default flags)")
ap.add_argument("--pairs", default=None, help="substring of a kernel name: print which kernels ran beside it (time-weighted) and its mean duration alone / shared")
a = ap.parse_args()
rows = []
for r in csv.DictReader(open(a.csv)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
posts = [r for r in rows if "post_kernel" in r[2]]
if a.skip_tail:
    posts = posts[:-a.skip_tail]
if len(posts) > a.steps:      # the last `--steps` steps: from the end of post_kernel[-steps-1] to the end of the last one
    lo, hi = posts[-a.steps - 1][1], posts[-1][1]
    rows = [r for r in rows if r[0] >= lo and r[1] <= hi]
    print(f"window: {a.steps} steps, {(hi - lo) / a.steps / 1e6:.4f} ms per step")
else:
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    cut = t1 - (t1 - t0) * a.frac
    rows = [r for r in rows if r[0] >= cut]
ev = []
for i, (s, e, n) in enumerate(rows):
    ev.append((s, 1, i)); ev.append((e, -1, i))
ev.sort()
active = set(); last = ev[0][0]
alone = collections.Counter(); shared = collections.Counter(); idle = 0; busy1 = 0; busy2 = 0
partner = collections.Counter()
for t, d, i in ev:
    dt = t - last
    if dt > 0:
        if not active: idle += dt
        elif len(active) == 1:
            busy1 += dt; alone[rows[next(iter(active))][2]] += dt
        else:
            busy2 += dt
            for j in active:
                shared[rows[j][2]] += dt
                if a.pairs and a.pairs in rows[j][2]:
                    for k in active:
                        if k != j: partner[rows[k][2]] += dt
    last = t
    if d == 1: active.add(i)
    else: active.discard(i)
span = ev[-1][0] - ev[0][0]
print(f"span {span/1e6:.3f} ms: idle {100*idle/span:.1f} %, one kernel {100*busy1/span:.1f} %, two or more {100*busy2/span:.1f} %")
names = sorted(set(alone) | set(shared), key=lambda n: -(alone[n] + shared[n]))
tot = collections.Counter()
for s, e, n in rows: tot[n] += e - s
for n in names[:24]:
    print(f"{n[:86]:86s} total {tot[n]/1e6:8.3f} ms  alone {100*alone[n]/max(tot[n],1):5.1f} %")
if a.pairs:
    me = [r for r in rows if a.pairs in r[2]]
    print(f"\nbeside {a.pairs} ({len(me)} launches, mean {sum(e - s for s, e, _ in me) / max(len(me), 1) / 1e3:.1f} us):")
    tot_p = sum(partner.values())
    for n, v in partner.most_common(12):
        print(f"  {n[:100]:100s} {100 * v / max(tot_p, 1):5.1f} % of the shared time ({v / max(len(me), 1) / 1e3:.1f} us per launch)")
