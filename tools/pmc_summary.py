#!/usr/bin/env python
"""Per-kernel averages of the PMC counters in a rocprofv3 rocpd .db:  python tools/pmc_summary.py run.db [kernel-substring]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
sub = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [d[0] for d in c.execute("select * from counters_collection limit 1").description]
name_col = "kernel_name" if "kernel_name" in cols else "name"
q = f"select {name_col}, counter_name, avg(value), count(*) from counters_collection group by {name_col}, counter_name"
rows = {}
for k, cn, v, n in c.execute(q):
    if sub in k:
        rows.setdefault(k.split("(")[0][:60], {})[cn] = (v, n)
for k, d in rows.items():
    print(k)
    for cn, (v, n) in sorted(d.items()):
        print(f"   {cn:32s} {v:18.1f}   (n={n})")
