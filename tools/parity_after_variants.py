#!/usr/bin/env python
"""Which of bench.py's side measurements moves the fp16-vs-oracle rendering check of the SAME trainer (development aid):
the check after the fused steps, after API-path steps, after an 800x800 evaluation render."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import bench  # noqa: E402


def show(tag, tr):
    p = bench.parity_check(tr)
    print(tag, p["psnr_rgb_db"], p["psnr_rgb_db_without_worst_0p1pct"], p["max_abs_rgb"], p["rays_beyond_max_tol"], p["p99_abs_rgb"],
          p["p999_abs_rgb"], p["worst_ray"], p["ok"], flush=True)


def main():
    dev = torch.device("cuda", 0)
    tr = bench.build_trainer(dev, 0, 1)
    it = 241
    for _ in range(96):
        tr.train_step(it)
        it += 1
    show("after 96 fused steps        ", tr)
    for _ in range(30):
        tr.train_step(it)
        it += 1
    show("after 30 more fused steps   ", tr)
    tr.fused_step = False
    for _ in range(30):
        tr.train_step(it)
        it += 1
    tr.fused_step = True
    show("after 30 API-path steps     ", tr)
    for _ in range(30):
        tr.train_step(it)
        it += 1
    show("after 30 more fused steps   ", tr)


if __name__ == "__main__":
    main()
