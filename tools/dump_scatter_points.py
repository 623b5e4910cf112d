#!/usr/bin/env python
"""Dump the sample set one table scatter of the bench step sees (positions as rays_o / rays_d / t / ridx), so that the
sector-request count of scatter variants can be modelled on the host (tools/scatter_sector_model.py): the float-atomic
rate of the MI355X is paid per distinct 64-byte sector per wave instruction (profiles/round4_atomic_line_bench.txt), which
is a function of the lane -> sample mapping and the sample positions alone.   -> gpurun_out/scatter_points.npz"""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np   # noqa: E402
import torch         # noqa: E402

import bench         # noqa: E402
from neuralsim_amd import _lib    # noqa: E402

dev = torch.device("cuda", 0)
config = sys.argv[1] if len(sys.argv) > 1 else "object"
tr = bench.build_trainer(dev, 0, 1) if config == "object" else bench.build_config_trainer(config, dev, 0, 1, 16384)
for it in range(20):
    tr.train_step(300 + it)
grabbed = {}
real = _lib.call


def spy(name, *args):
    if name == "nsim_lotd_scatter" and "o" not in grabbed:
        meta, x, o, d, t, ridx, goff, S = args[:8]
        if x is None:
            grabbed.update(o=o.detach().cpu().numpy(), d=d.detach().cpu().numpy(), t=t.detach().cpu().numpy()[:S],
                           ridx=ridx.detach().cpu().numpy()[:S])
        else:
            grabbed.update(x=x.detach().cpu().numpy()[:S], ridx=np.zeros(0, np.int64))
        grabbed["S"] = np.int64(S)
    return real(name, *args)


_lib.call = spy
import neuralsim_amd.trainer as trm      # noqa: E402
for mod in list(sys.modules.values()):
    if getattr(mod, "__name__", "").startswith("neuralsim_amd") and getattr(mod, "call", None) is real:
        mod.call = spy
tr.train_step(333)
torch.cuda.synchronize()
cfg = tr.model.encoding.cfg
out = ROOT / "gpurun_out"
out.mkdir(exist_ok=True)
np.savez_compressed(out / f"scatter_points_{config}.npz", lod_res=np.asarray(cfg.lod_res), hashmap_size=np.int64(cfg.hashmap_size),
                    aabb=tr.model.accel.aabb.detach().cpu().numpy(), **grabbed)
print("dumped", {k: getattr(v, "shape", v) for k, v in grabbed.items()})
