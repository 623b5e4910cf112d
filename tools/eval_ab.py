#!/usr/bin/env python
"""Evaluation-path A/B (round 5): a full 800 x 800 view of the configs[1] model, ms per view.  NSIM_EVAL_PLANES=0|1 selects the
fused point-major forward or the level-major gather + plane decoders; NSIM_UPSAMPLE_ON_MARCHED_ONLY=0|1 the sampling mode."""
import json
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    tr = bench.build_trainer(dev, 0, 1)
    for it in range(250, 262):
        tr.train_step(it)
    from neuralsim_amd.eval import render_image
    ha = tr.appear.detach()[0:1]
    ts = []
    for k in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        render_image(tr.renderer, tr.model, tr.intr, tr.c2w, tr.WH, frame=k, rays_h_appear=ha)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(json.dumps(dict(eval_planes=os.environ.get("NSIM_EVAL_PLANES", "0"), marched_only=os.environ.get("NSIM_UPSAMPLE_ON_MARCHED_ONLY", "1"),
                          ms=[round(t, 2) for t in ts], best=round(min(ts[1:]), 2))))


if __name__ == "__main__":
    main()
