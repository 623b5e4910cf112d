"""Generate tests/golden/pack_fixture.json from the literal inputs of the reference's only self-contained
numeric fixture, ``test_multi_buffer_collect_and_merge`` (app/renderers/buffer_compose_renderer.py:972-1049).

The reference's pack ops (nr3d_lib) are absent, so the expected values are derived here with plain Python
(sorted()) from those literals -- the invariants the reference checks (``sorted == total[indices]``) plus the
per-ray sample counts fixed by its inputs.  Run from the repo root:  python tests/golden/make_pack_fixture.py
"""
import json
from pathlib import Path

buffers = [
    dict(type="batched", rays_inds_hit=[1, 1, 2, 6, 8], num_per_hit=3,
         t=[[0.1, 0.2, 0.3], [1.1, 1.2, 1.3], [0.1, 0.2, 0.3], [0.1, 0.2, 0.3], [0.1, 0.2, 0.3]]),
    dict(type="packed", rays_inds_hit=[1, 1, 2, 6, 8], n=[2, 3, 2, 2, 1],
         t=[0.15, 0.25, 0.11, 0.12, 0.21, 0.4, 0.5, 0.05, 0.15, 0.14]),
    dict(type="packed", rays_inds_hit=[0, 1, 2, 6, 8], n=[1, 2, 3, 2, 4],
         t=[0.05, 0.31, 0.34, 0.24, 0.26, 0.28, 0.6, 0.7, 0.5, 0.6, 0.7, 0.71]),
]
total_num_rays = 10
per_ray = {r: [] for r in range(total_num_rays)}
for b in buffers:
    if b["type"] == "batched":
        for r, row in zip(b["rays_inds_hit"], b["t"]):
            per_ray[r] += row
    else:
        k = 0
        for r, n in zip(b["rays_inds_hit"], b["n"]):
            per_ray[r] += b["t"][k:k + n]
            k += n
import numpy as np
counts = [len(per_ray[r]) for r in range(total_num_rays)]
sorted_depths = []
for r in range(total_num_rays):
    sorted_depths += [float(np.float32(v)) for v in sorted(per_ray[r])]
out = dict(buffers=buffers, total_num_rays=total_num_rays, ray_visible_samples=counts,
           total_rays_inds_hit=[r for r in range(total_num_rays) if counts[r] > 0], total_num_samples=sum(counts),
           sorted_depths=sorted_depths)
assert counts == [1, 13, 8, 0, 0, 0, 7, 0, 8, 0] and sum(counts) == 37
Path(__file__).with_name("pack_fixture.json").write_text(json.dumps(out, indent=1))
print("wrote pack_fixture.json", counts)
