#!/usr/bin/env python3
"""Per-kernel totals from a rocprofv3 results .db (what --stats would print): python tools/rocprof_db_stats.py file.db [steps]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = list(c.execute(f"select s.kernel_name, count(*), sum(d.end-d.start)/1e3, avg(d.end-d.start)/1e3 from {kd} d join {ks} s "
                      "on d.kernel_id=s.id group by s.kernel_name order by 3 desc"))
tot = sum(r[2] for r in rows)
print("%-64s %8s %12s %10s %6s" % ("kernel", "calls", "total us", "avg us", "%"))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 20]:
    print("%-64s %8d %12.1f %10.1f %6.1f" % (r[0][:64], r[1], r[2], r[3], 100 * r[2] / tot))
print("total %.1f us over %d launches; per step (/%g): %.1f us, %.0f launches" % (tot, sum(r[1] for r in rows), steps, tot / steps,
                                                                                  sum(r[1] for r in rows) / steps))
