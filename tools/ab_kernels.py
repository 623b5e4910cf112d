#!/usr/bin/env python
"""A/B timing of kernel variants selected by environment switches on one fixed training batch of the bench workload
(development aid).  usage: python tools/ab_kernels.py NAME=ENV=VAL[,ENV=VAL] ...   (the baseline runs first)"""
import json
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
    variants = [("base", {})]
    for a in sys.argv[1:]:
        name, rest = a.split("=", 1)
        variants.append((name, dict(kv.split("=") for kv in rest.split(","))))
    keys = sorted({k for _, e in variants for k in e})
    out = {}
    for name, env in variants:
        for k in keys:
            os.environ.pop(k, None)
        os.environ.update(env)
        it = 300
        for _ in range(6):
            tr.train_step(it)
            it += 1
        torch.cuda.synchronize()
        _lib.TIMER = _lib.KernelTimer(only=bench.KERNEL_MODEL.keys())
        for _ in range(16):
            tr.train_step(it)
            it += 1
        s = _lib.TIMER.summary()
        _lib.TIMER = None
        out[name] = {k[5:]: round(v["avg_ms"], 4) for k, v in s.items()}
        print(f"{name:16s} " + " ".join(f"{k}={v:.4f}" for k, v in sorted(out[name].items())), flush=True)
    Path(ROOT / "gpurun_out").mkdir(exist_ok=True)
    (ROOT / "gpurun_out" / "ab_kernels.json").write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
