#!/usr/bin/env python
"""profiles/round6_convergence.json: the bench workload trained for 3000 fused steps, PSNR / SSIM of three held-out 800 x 800
views against the analytic target at 0 / 250 / 1000 / 3000 steps (bench.convergence_record).  Run on the GPU box:
    python tools/convergence_run.py > gpurun_out/convergence.json"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch   # noqa: E402

import bench   # noqa: E402

print(json.dumps(bench.convergence_record(torch.device("cuda", 0)), indent=1))
