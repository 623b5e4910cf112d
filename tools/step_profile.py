#!/usr/bin/env python
"""Host-side breakdown of one training step (wall clock with/without device syncs).  Development aid."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    tr = bench.build_trainer(dev, 0, 1)
    for it in range(212, 268):
        tr.train_step(it)
    torch.cuda.synchronize()
    import cProfile
    import pstats
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    n = 20
    for it in range(273, 273 + n):
        tr.train_step(it)
    torch.cuda.synchronize()
    pr.disable()
    print(f"avg step {(time.perf_counter() - t0) / n * 1e3:.3f} ms (with cProfile overhead)")
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(45)
    st.sort_stats("tottime").print_stats(35)


if __name__ == "__main__":
    main()
