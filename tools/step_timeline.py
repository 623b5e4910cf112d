#!/usr/bin/env python
"""One steady-state training step as a kernel timeline, from a rocprofv3 --kernel-trace database: every dispatch between two
consecutive table-Adam launches (k_adam) -- stream, start offset, duration, gap to the previous dispatch of the SAME stream.
Usage: python tools/step_timeline.py DB OUT.txt [step_index_from_end=3]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    sid = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
    rows = db.execute(f"select name, start, end, {sid or '0'} from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if r[0].startswith("_Z6k_adam") or r[0].startswith("k_adam(")]
    k = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    a, b = marks[-k - 1], marks[-k]
    seq = rows[a + 1:b + 1]
    t0 = seq[0][1]
    last = {}
    out = [f"step of {len(seq)} dispatches, {(seq[-1][2] - t0) / 1e3:.1f} us from first start to last end", "stream  start_us  dur_us  gap_us  kernel"]
    busy = {}
    for name, s, e, st in seq:
        gap = (s - last[st]) / 1e3 if st in last else 0.0
        last[st] = e
        busy[st] = busy.get(st, 0.0) + (e - s) / 1e3
        out.append(f"{st:6d} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} {gap:7.1f}  {name[:110]}")
    out.append("busy us per stream: " + ", ".join(f"{k_}: {v:.1f}" for k_, v in sorted(busy.items())))
    open(sys.argv[2], "w").write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
