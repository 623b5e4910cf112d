#!/usr/bin/env python
"""Per-kernel timing of the field kernels on one fixed render batch (8192 rays of the bench workload), with the
NSIM_ABLATE / NSIM_DEDUP_MAX_RES profiling knobs of csrc/field.hip.  Development aid, not part of the product."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import bench  # noqa: E402
from neuralsim_amd import _lib  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    tr = bench.build_trainer(dev, 0, 1)
    xy, fidx, gt = tr.sample_batch()
    variants = [("full", {}), ("no-scatter", {"NSIM_ABLATE": "1"}), ("no-dW", {"NSIM_ABLATE": "4"}),
                ("no-scatter,no-dW", {"NSIM_ABLATE": "5"}), ("no-dedup", {"NSIM_DEDUP_MAX_RES": "0"}),
                ("dedup<=100", {"NSIM_DEDUP_MAX_RES": "100"}), ("dedup-all", {"NSIM_DEDUP_MAX_RES": "4096"})]
    for name, env in variants:
        for k in ("NSIM_ABLATE", "NSIM_DEDUP_MAX_RES"):
            os.environ.pop(k, None)
        os.environ.update(env)
        for rep in range(3):
            _lib.TIMER = _lib.KernelTimer() if rep == 2 else None
            tested, ret = tr.render(xy, fidx)
            loss, _ = tr.loss(tested, ret, gt)
            tr.optim.zero_grad()
            loss.backward()
            torch.cuda.synchronize()
        s = _lib.TIMER.summary()
        _lib.TIMER = None
        S = ret["volume_buffer"]["t"].shape[0]
        print(f"{name:20s} S_f={S} " + " ".join(f"{k[5:]}={v['total_ms']:.3f}ms/{v['calls']}" for k, v in s.items()
                                              if k.startswith("nsim_field")), flush=True)


if __name__ == "__main__":
    main()
