#!/usr/bin/env python3
"""Average per-dispatch PMC values (in millions) per kernel from rocprofv3 counter_collection.csv files."""
import csv, collections, re, sys
for f in sys.argv[1:]:
    rows = list(csv.DictReader(open(f)))
    agg = collections.OrderedDict()
    for r in rows:
        k = re.sub(r'void yf::', '', r['Kernel_Name']); k = re.sub(r'\(.*', '', k)[:70]
        agg.setdefault(k, collections.defaultdict(list))[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        if 'copyBuffer' in k: continue
        print(k, ' '.join(f"{c.replace('SQ_','')}={sum(x)/len(x)/1e6:.3f}" for c, x in v.items()))
