#!/usr/bin/env python
"""Per-(kernel, grid size) averages from a rocprofv3 rocpd .db -- separates the launches of one kernel by shape
(e.g. the four projections of a decode layer all run gemv_kernel<1>, with different grids).

    python tools/rocprof_by_grid.py run.db [kernel-substring]
"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
sub = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
if gx is None:
    sys.exit(f"no grid column in kernels view; columns: {cols}")
q = (f"select name, {gx}, workgroup_x, count(*), avg(duration), min(duration), max(duration), sum(duration) from kernels "
     f"group by name, {gx} order by sum(duration) desc")
print("kernel,grid_x(threads),wg_x,calls,avg_us,min_us,max_us,total_ms")
for name, g, wg, n, avg, mn, mx, tot in c.execute(q):
    if sub in name:
        print(f"{name.split('(')[0][:50]},{g},{wg},{n},{avg / 1e3:.2f},{mn / 1e3:.2f},{mx / 1e3:.2f},{tot / 1e6:.3f}")
