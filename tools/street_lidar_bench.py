#!/usr/bin/env python
"""The reference's own street iteration -- 8192 pixel rays + 8192 lidar beams, two optimizer steps
(withmask_withlidar_joint.240219.yaml:7-8; code_single/tools/train.py:1480-1590) -- for a kernel profile:
    rocprofv3 --kernel-trace --stats -- python tools/street_lidar_bench.py"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from neuralsim_amd import scenarios as sc      # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    tr = sc.build_street_trainer(dev, 0, 1, rays_per_gpu=8192, lidar_rays=8192)
    it = 257
    for _ in range(4):
        tr.train_step(it)
        it += 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        tr.train_step(it)
        it += 1
    torch.cuda.synchronize()
    print(f"street px8192 + lidar8192: {(time.perf_counter() - t0) / 8 * 1e3:.3f} ms per iteration")


if __name__ == "__main__":
    main()
