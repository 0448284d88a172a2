#!/usr/bin/env python3
"""Wave-time breakdown per kernel from tools/pmc_wait.sh: parked (s_waitcnt / barrier), issue-stalled, issuing; VALU and LDS share."""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"{'kernel':64s} {'parked':>7s} {'stalled':>8s} {'issuing':>8s} {'valu':>6s} {'lds':>6s} {'lds-stall':>9s}")
for k, c in acc.items():
    if "yf::" not in k: continue
    m = lambda n: sum(c[n]) / max(len(c[n]), 1)
    w = m("SQ_WAVE_CYCLES")
    print(f"{k[10:74]:64s} {100*m('SQ_WAIT_ANY')/w:6.1f}% {100*m('SQ_WAIT_INST_ANY')/w:7.1f}% {100*m('SQ_ACTIVE_INST_ANY')/w:7.1f}% "
          f"{100*m('SQ_ACTIVE_INST_VALU')/w:5.1f}% {100*m('SQ_ACTIVE_INST_LDS')/w:5.1f}% {100*m('SQ_WAIT_INST_LDS')/w:8.1f}%")
