#!/usr/bin/env python
"""Where the GPU is IDLE inside a profiled run: the gaps between consecutive kernels of a rocprofv3 rocpd .db (kernels view: name, start,
duration in ns), attributed to the pair (kernel before, kernel after), plus the busy / idle totals of the densest window (the timed region of
bench.py is the longest stretch without a > 50 ms pause).  Gaps under a threshold are launch turn-around (counted, summed separately).

    python tools/rocprof_gaps.py run.db [min_gap_us=20] > profiles/rNN_bench_gaps.txt"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r"\(.*", "", n)
    n = re.sub(r"^void\s+", "", n).replace("at::native::", "torch::").replace("(anonymous namespace)::", "")
    return n[:48]


def main():
    c = sqlite3.connect(sys.argv[1])
    thr = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 20e3
    rows = list(c.execute("select name, start, duration from kernels order by start"))
    if not rows:
        print("no kernels")
        return
    # windows separated by pauses > 50 ms (model build, host-side legs); report the longest one in GPU-busy time
    wins, cur = [], [rows[0]]
    for r in rows[1:]:
        if r[1] - (cur[-1][1] + cur[-1][2]) > 50e6:
            wins.append(cur)
            cur = []
        cur.append(r)
    wins.append(cur)
    win = max(wins, key=lambda w: sum(r[2] for r in w))
    span = win[-1][1] + win[-1][2] - win[0][1]
    busy = sum(r[2] for r in win)
    small, big = 0.0, 0.0
    nsmall = nbig = 0
    pairs = defaultdict(lambda: [0, 0.0])
    end = win[0][1] + win[0][2]
    prev = win[0][0]
    for n, s, d in win[1:]:
        gap = s - end
        if gap > 0:
            if gap < thr:
                small += gap
                nsmall += 1
            else:
                big += gap
                nbig += 1
                k = (short(prev), short(n))
                pairs[k][0] += 1
                pairs[k][1] += gap
        if s + d > end:
            end = s + d
            prev = n
    print(f"window: {len(win)} kernels over {span / 1e6:.1f} ms; kernels busy {busy / 1e6:.1f} ms ({100 * busy / span:.1f} %)")
    print(f"gaps < {thr / 1e3:.0f} us: {nsmall} totalling {small / 1e6:.2f} ms (mean {small / max(nsmall, 1) / 1e3:.2f} us); "
          f"gaps >= {thr / 1e3:.0f} us: {nbig} totalling {big / 1e6:.2f} ms")
    print("largest idle gaps by (kernel before -> kernel after): count, total ms, mean us")
    for (a, b), (cnt, tot) in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"  {cnt:6d} {tot / 1e6:9.2f} ms {tot / cnt / 1e3:9.1f} us   {a} -> {b}")


if __name__ == "__main__":
    main()
