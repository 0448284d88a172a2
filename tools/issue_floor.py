#!/usr/bin/env python3
"""Issue-side floor of one forward pass from tools/pmc_bench.sh: per kernel name, launches per pass x (VALU instructions x 4
cycles + MFMA busy cycles) per SIMD -- the time the pass would take if nothing ever stalled (fp32 MFMA and VALU issue do not
overlap on gfx950).  1024 SIMDs, 2.4 GHz."""
import csv, sys, collections
path = sys.argv[1]; trace = sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(path)):
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(trace)):
    dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
passes = len(acc[next(k for k in acc if "post_kernel" in k)]["SQ_INSTS_VALU"])
tot_floor = tot_dur = 0
print(f"{'kernel':60s} {'n/pass':>6s} {'us':>7s} {'valu us':>8s} {'mfma us':>8s} {'floor':>7s} {'util':>5s}")
for k, cs in acc.items():
    if not k.startswith("void yf::") and "post_kernel" not in k: continue
    n = len(cs["SQ_INSTS_VALU"]) / passes
    valu = sum(cs["SQ_INSTS_VALU"]) / len(cs["SQ_INSTS_VALU"]) * 4 / 1024 / 2400.0
    mfma = sum(cs["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(cs["SQ_VALU_MFMA_BUSY_CYCLES"]) / 1024 / 2400.0
    d = sorted(dur[k])[len(dur[k]) // 2] / 1e3
    print(f"{k[10:70]:60s} {n:6.1f} {d:7.1f} {valu:8.1f} {mfma:8.1f} {valu+mfma:7.1f} {100*(valu+mfma)/d:5.0f}")
    tot_floor += n * (valu + mfma); tot_dur += n * d
print(f"pass: kernels {tot_dur:.0f} us, issue floor {tot_floor:.0f} us ({100*tot_floor/tot_dur:.0f} %)")
