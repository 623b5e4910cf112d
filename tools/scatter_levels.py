#!/usr/bin/env python
"""Per-level timing of nsim_lotd_scatter on the tensors of a real training step (GPU box).
Captures the arguments of the last scatter launch of a few bench steps and replays it level by level
(``level_begin, level_count``) between HIP events.  Usage: python tools/scatter_levels.py OUT.json"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import bench  # noqa: E402
from neuralsim_amd import _lib  # noqa: E402


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/scatter_levels.json"
    dev = torch.device("cuda", 0)
    tr = bench.build_trainer(dev, 0, 1)
    cap = {}
    orig = _lib.call

    def spy(name, *args):
        if name == "nsim_lotd_scatter":
            cap["args"] = args
        return orig(name, *args)
    _lib.call = spy
    import neuralsim_amd.trainer as T
    T._lib.call = spy
    for it in range(241, 262):
        tr.train_step(it)
    torch.cuda.synchronize()
    _lib.call = orig
    T._lib.call = orig
    args = list(cap["args"])
    S = args[7]
    cfg = tr.model.encoding.cfg
    dgrid = args[11]
    rec = dict(points=int(S), levels=[])

    def timed(l0, n, reps=20):
        for _ in range(3):
            orig("nsim_lotd_scatter", *args[:12], l0, n)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            orig("nsim_lotd_scatter", *args[:12], l0, n)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    rec["all_us"] = timed(0, 0)
    for l in range(cfg.num_levels):
        rec["levels"].append(dict(level=l, res=cfg.lod_res[l], type=cfg.lod_types[l], us=round(timed(l, 1), 2)))
    rec["sum_levels_us"] = round(sum(x["us"] for x in rec["levels"]), 2)
    dgrid.zero_()
    Path(out).parent.mkdir(exist_ok=True)
    Path(out).write_text(json.dumps(rec, indent=1))
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
