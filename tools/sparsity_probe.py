#!/usr/bin/env python
"""Measurements behind DESIGN.md's notes on the table gradient's sparsity and the occupancy refresh (development aid):
 (1) per level, the fraction of table entries / of 256-entry blocks that receive a gradient in one training step;
 (2) step time and occupied fraction with the periodic from-net refresh at num_steps = 4 / 1 and with / without the
     render-time sample collection (update_from_samples_cfg)."""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import bench  # noqa: E402


def run(tr, it0, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(it0, it0 + n):
        tr.train_step(it)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    dev = torch.device("cuda", 0)
    out = {}
    tr = bench.build_trainer(dev, 0, 1)
    run(tr, 241, 16)
    g = tr.model.encoding.flattened_params.grad
    cfg = tr.model.encoding.cfg
    lv = []
    offs = list(cfg.lod_offsets) + [cfg.n_params]
    for l in range(cfg.num_levels):
        x = g[offs[l]:offs[l + 1]]
        nz = (x != 0)
        n = nz.numel() // 512 * 512
        blk = nz[:n].view(-1, 512).any(dim=1)          # 256 entries x 2 features
        lv.append(dict(level=l, res=int(cfg.lod_res[l]), type=cfg.lod_types[l], entries=int(nz.numel() // 2),
                       touched_entries=round(float(nz.float().mean()), 4), touched_blocks256=round(float(blk.float().mean()), 4)))
    tot = (g != 0)
    n = tot.numel() // 512 * 512
    out["gradient_sparsity"] = dict(levels=lv, touched_entries=round(float(tot.float().mean()), 4),
                                    touched_blocks256=round(float(tot[:n].view(-1, 512).any(dim=1).float().mean()), 4),
                                    S_f=tr.stats["S_f"])
    print(json.dumps(out["gradient_sparsity"]), flush=True)
    res = {}
    for name, steps, collect in (("net4+samples (reference config)", 4, True), ("net4 only", 4, False),
                                 ("net1+samples", 1, True), ("net1 only", 1, False)):
        tr = bench.build_trainer(dev, 0, 1)
        tr.model.accel.num_steps = steps
        if not collect:
            tr.model.accel.update_from_samples_cfg = None
        run(tr, 241, 16)
        ms = run(tr, 257, 96)
        res[name] = dict(ms_per_step=round(ms, 3), frac_occupied=round(tr.model.accel.frac_occupied(), 5), S_f=tr.stats["S_f"])
        print(name, res[name], flush=True)
    out["refresh"] = res
    (ROOT / "gpurun_out").mkdir(exist_ok=True)
    (ROOT / "gpurun_out" / "sparsity_probe.json").write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
