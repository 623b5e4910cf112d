#!/usr/bin/env python
"""Per-launch time of the with-grad field kernels vs number of points (uniform points, eikonal-style loss): separates
fixed per-launch cost (weight staging, LDS accumulator flush) from the per-point slope.  Development aid."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import bench  # noqa: E402
from neuralsim_amd import _lib  # noqa: E402
from neuralsim_amd.losses import eikonal_loss  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    tr = bench.build_trainer(dev, 0, 1)
    m = tr.model
    g = torch.Generator(device=dev).manual_seed(1)
    for S in (1024, 4096, 16384, 65536, 262144, 1048576):
        x = torch.rand([S, 3], device=dev, generator=g) * 1.6 - 0.8
        for rep in range(4):
            _lib.TIMER = _lib.KernelTimer() if rep == 3 else None
            out = m.forward_sdf_nablas(x)
            loss = eikonal_loss(out["nablas"]) + out["sdf"].mean()
            tr.optim.zero_grad()
            loss.backward()
            m.query_sdf(x)
            torch.cuda.synchronize()
        s = _lib.TIMER.summary()
        _lib.TIMER = None
        print(f"S={S:8d} " + " ".join(f"{k[5:]}={v['total_ms'] * 1e3:.0f}us" for k, v in s.items()
                                      if k.startswith("nsim_field") or k.startswith("nsim_lotd")), flush=True)


if __name__ == "__main__":
    main()
