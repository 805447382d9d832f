#!/usr/bin/env python
"""Condense a rocprofv3 run (rocpd .db written by `rocprofv3 --kernel-trace --stats`, or its *_kernel_stats.csv) into the
short per-kernel table committed under profiles/.

    python tools/rocprof_summary.py gpurun_out/prof_r1/bench_results.db > profiles/r01_bench_kernel_stats.csv
"""
import csv
import re
import sqlite3
import sys


def short(name, n=72):
    name = re.sub(r"\(.*", "", name)                       # drop the argument list
    name = re.sub(r"^void\s+", "", name)
    name = name.replace("at::native::", "torch::").replace("(anonymous namespace)::", "")
    return name if len(name) <= n else name[: n - 3] + "..."


def rows_from_db(path):
    c = sqlite3.connect(path)
    q = ("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), "
         "max(accum_vgpr_count), max(lds_size), max(workgroup_x) from kernels group by name order by sum(duration) desc")
    return list(c.execute(q))


def main():
    path = sys.argv[1]
    rows = rows_from_db(path)
    total = sum(r[2] for r in rows) or 1
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct", "vgpr", "agpr", "lds_bytes", "wg_size"])
    merged = {}
    for name, calls, tot, avg, mn, mx, vg, ag, lds, wg in rows:
        k = short(name)
        if k in merged:          # different template args folded into one truncated name: keep them apart by suffix
            k = k + f"#{len(merged)}"
        merged[k] = (calls, tot, avg, mn, mx, vg, ag, lds, wg)
    for k, (calls, tot, avg, mn, mx, vg, ag, lds, wg) in merged.items():
        w.writerow([k, calls, f"{tot / 1e6:.3f}", f"{avg / 1e3:.2f}", f"{mn / 1e3:.2f}", f"{mx / 1e3:.2f}", f"{100 * tot / total:.3f}",
                    vg, ag, lds, wg])
    w.writerow(["TOTAL", sum(v[0] for v in merged.values()), f"{total / 1e6:.3f}", "", "", "", "100", "", "", "", ""])


if __name__ == "__main__":
    main()
