#!/usr/bin/env python
"""End-to-end sanity of the bench workload: a few hundred steps, photometric error and sample statistics.
Development aid (GPU)."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import bench  # noqa: E402


def main(n_blocks=8, per=25):
    dev = torch.device("cuda", 0)
    tr = bench.build_trainer(dev, 0, 1)
    it = 241
    for k in range(n_blocks):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        losses = []
        for _ in range(per):
            losses.append(tr.train_step(it))
            it += 1
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / per * 1e3
        with torch.no_grad():
            xy, fidx, gt = tr.sample_batch()
            tr.renderer.eval()
            ret = tr.render(xy, fidx)
            tr.renderer.train()
            mse = float(((ret["rendered"]["rgb_volume"] - gt) ** 2).mean())
            mask = float(ret["rendered"]["mask_volume"].mean())
        print(f"it {it - 241:4d}  loss {float(torch.stack(losses).mean()):.5f}  eval mse {mse:.5f}  mask {mask:.3f}  "
              f"S_f {tr.stats['S_f']}  {ms:.3f} ms/step", flush=True)


if __name__ == "__main__":
    main()
