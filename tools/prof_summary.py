#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats kernel_stats.csv: per-step time of the yf:: kernels."""
import csv, re, sys
path, steps = sys.argv[1], int(sys.argv[2])
rows = [r for r in csv.DictReader(open(path)) if 'yf::' in r['Name']]
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total yf kernel time per step: {tot/steps/1e6:.3f} ms")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    n = re.sub(r'void yf::', '', r['Name']); n = re.sub(r'\(.*', '', n)
    print(f"{n[:80]:80s} calls/step={int(r['Calls'])/steps:4.0f} avg_us={float(r['AverageNs'])/1e3:8.1f} per-step_us={float(r['TotalDurationNs'])/steps/1e3:8.1f} {100*float(r['TotalDurationNs'])/tot:5.1f}%")
