#!/usr/bin/env python3
"""Per-kernel means of the counters in rocprofv3 counter_collection csv files (tools/pmc_kbench.sh)."""
import csv, sys, collections
for path in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(k)
        for c, v in cs.items():
            print(f"    {c:28s} {sum(v)/len(v):16.0f}   (n={len(v)})")
