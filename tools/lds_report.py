#!/usr/bin/env python3
"""LDS view of every kernel of a forward pass (tools/pmc_lds.sh): bank-conflict share of the LDS-array cycles and the share of
wave time spent waiting on LDS results."""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"{'kernel':62s} {'conflict/LDS-active':>20s} {'LDS wait / wave time':>22s} {'any wait / wave time':>22s}")
for k, c in acc.items():
    if "yf::" not in k: continue
    m = lambda n: sum(c[n]) / max(len(c[n]), 1)
    idx = m("SQ_LDS_IDX_ACTIVE")
    print(f"{k[10:72]:62s} {100*m('SQ_LDS_BANK_CONFLICT')/max(idx,1):19.1f}% {100*m('SQ_WAIT_INST_LDS')/m('SQ_WAVE_CYCLES'):21.1f}% {100*m('SQ_WAIT_INST_ANY')/m('SQ_WAVE_CYCLES'):21.1f}%")
