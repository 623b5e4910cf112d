#!/usr/bin/env python
"""Where is the GPU idle?  From a rocprofv3 --kernel-trace database: the gaps between consecutive kernels of the steady
state (second half of the trace), attributed to the kernel that FOLLOWS the gap (the launch the host was late with).
Usage: python tools/gap_profile.py DB OUT.json"""
import json
import sqlite3
import sys
from collections import defaultdict


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    res = dict(columns=cols)
    try:
        rows = db.execute("select name, start, end from kernels order by start").fetchall()
    except Exception as e:
        res["error"] = str(e)
        json.dump(res, open(sys.argv[2], "w"), indent=1)
        return
    rows = rows[len(rows) // 2:]
    busy = sum(e - s for _, s, e in rows)
    span = rows[-1][2] - rows[0][1]
    gaps = defaultdict(lambda: [0, 0.0, 0.0])
    prev_end, prev_name = rows[0][2], rows[0][0]
    pair = defaultdict(lambda: [0, 0.0])
    for name, s, e in rows[1:]:
        g = s - prev_end
        if g > 0:
            a = gaps[name[:70]]
            a[0] += 1
            a[1] += g
            a[2] = max(a[2], g)
            if g > 20000:
                p = pair[(prev_name[:50], name[:50])]
                p[0] += 1
                p[1] += g
        prev_end, prev_name = max(prev_end, e), name
    n_adam = sum(1 for n, _, _ in rows if "k_adam" in n)
    res.update(span_ms=span / 1e6, busy_ms=busy / 1e6, idle_frac=1 - busy / span, n_kernels=len(rows),
               adam_calls=n_adam)
    res["gaps_by_following_kernel"] = sorted(
        [dict(kernel=k, n=v[0], total_us=v[1] / 1e3, avg_us=v[1] / 1e3 / v[0], max_us=v[2] / 1e3) for k, v in gaps.items()],
        key=lambda d: -d["total_us"])[:30]
    res["big_gaps"] = sorted([dict(prev=k[0], next=k[1], n=v[0], total_us=v[1] / 1e3) for k, v in pair.items()],
                             key=lambda d: -d["total_us"])[:15]
    json.dump(res, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
