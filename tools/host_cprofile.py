#!/usr/bin/env python
"""cProfile of the host side of the training step (development aid): which Python calls cost the host time between the
sample-count sync and the backward?"""
import cProfile
import pstats
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    cfg = sys.argv[1] if len(sys.argv) > 1 else "object"      # object | street | indoor | multi
    tr = bench.build_trainer(dev, 0, 1) if cfg == "object" else bench.build_config_trainer(cfg, dev, 0, 1, 16384)
    n0, n1 = (59, 48) if cfg == "object" else (10, 12)
    for it in range(241, 241 + n0):
        tr.train_step(it)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for it in range(305, 305 + n1):
        tr.train_step(it)
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(60)
    st.sort_stats("tottime").print_stats(35)


if __name__ == "__main__":
    main()
