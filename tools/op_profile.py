#!/usr/bin/env python
"""aten-op level host profile of steady-state training steps (torch.profiler, CPU activity only).  Development aid."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import bench  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    tr = bench.build_trainer(dev, 0, 1)
    for it in range(212, 268):
        tr.train_step(it)
    torch.cuda.synchronize()
    n = 10
    with profile(activities=[ProfilerActivity.CPU], record_shapes=False) as prof:
        for it in range(273, 273 + n):
            tr.train_step(it)
        torch.cuda.synchronize()
    ka = prof.key_averages()
    rows = sorted(ka, key=lambda e: -e.self_cpu_time_total)
    tot = sum(e.self_cpu_time_total for e in rows)
    print(f"total self cpu {tot / n:.0f} us/step, ops/step {sum(e.count for e in rows) / n:.0f}")
    for e in rows[:60]:
        print(f"{e.key[:60]:60s} n/step {e.count / n:6.1f} self {e.self_cpu_time_total / n:8.1f} us/step  total {e.cpu_time_total / n:8.1f}")


if __name__ == "__main__":
    main()
