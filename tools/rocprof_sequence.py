#!/usr/bin/env python
"""Launch-ordered durations of the kernels whose name contains a substring, from a rocprofv3 rocpd .db -- to tell the calls of one
kernel apart by their position in a known launch sequence (the persistent GEMM always runs 256 workgroups).

    python tools/rocprof_sequence.py run.db gemm_pq [last N]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
sub = sys.argv[2] if len(sys.argv) > 2 else ""
last = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rows = [(s, n.split("(")[0][:44], d) for n, s, d in c.execute("select name, start, duration from kernels order by start")
        if sub in n and not n.startswith(("void at::", "__amd", "at::"))]          # the library's own kernels only
if last:
    rows = rows[-last:]
t0 = rows[0][0] if rows else 0
for s, n, d in rows:
    print(f"{(s - t0) / 1e6:10.3f} ms  {n:44s} {d / 1e3:9.1f} us")
