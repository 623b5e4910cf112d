#!/usr/bin/env python
"""Per-step kernel time from TWO rocprofv3 summaries (tools/prof_summary.py) of the same command with different step counts:
(total_us(B) - total_us(A)) / (steps_B - steps_A) per kernel -- the set-up kernels cancel, the ATen glue kernels of a step
stay in.  Usage: tools/prof_diff.py A_stats.json stepsA B_stats.json stepsB [top]"""
import json
import sys


def load(p):
    out = {}
    for k in json.load(open(p))["kernels"]:
        out[k["name"]] = (k["calls"], k["total_us"])
    return out


def main():
    a, na, b, nb = load(sys.argv[1]), int(sys.argv[2]), load(sys.argv[3]), int(sys.argv[4])
    top = int(sys.argv[5]) if len(sys.argv) > 5 else 40
    d = nb - na
    rows = []
    for k, (cb, tb) in b.items():
        ca, ta = a.get(k, (0, 0.0))
        if cb > ca:       # (fewer calls in the longer run: a set-up kernel whose count depends on the step count)
            rows.append((k, (cb - ca) / d, (tb - ta) / d))
    rows.sort(key=lambda r: -r[2])
    tot = sum(r[2] for r in rows)
    print(f"per step: {tot:.1f} us in {sum(r[1] for r in rows):.1f} launches")
    print(f"{'kernel':100s} {'calls/step':>10s} {'us/step':>9s} {'avg us':>8s} {'%':>5s}")
    for k, c, t in rows[:top]:
        print(f"{k[:100]:100s} {c:10.2f} {t:9.1f} {t / max(c, 1e-9):8.1f} {100 * t / tot:5.1f}")


if __name__ == "__main__":
    main()
