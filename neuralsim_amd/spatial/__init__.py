"""``nr3d_lib.models.spatial.AABBSpace`` at the calls the reference makes on ``model.space``:
``space.aabb`` (app/models/single/nerf.py:175), ``space.get_bounding_volume()`` -> ``[..., 6] = centre + radius3d``
(app/resources/nodes.py:92-103), ``space.ray_test(**ray_input)`` (app/visualizer/gui_runner_single_cuboid.py:135-138),
``AABBSpace(bounding_size=, device=)`` (app/models/asset_base.py:117).  The slab test + compaction run in the HIP
kernels (csrc/sampling.hip ``k_aabb_ray_test``, ``k_gather_rays``)."""
from typing import Dict, Optional

import torch
import torch.nn as nn

from .. import _lib


def aabb_ray_test(aabb: torch.Tensor, meta, rays_o, rays_d, near=None, far=None, **extra) -> Dict:
    """Slab test against ``aabb`` clamped to [near, far] + compaction of the hit rays -> ``{num_rays, rays_inds [R] i64,
    rays_o/rays_d [R,3], near/far [R]}`` + the per-ray extras filtered to the hit rays (the dict the reference's
    renderers consume, app/renderers/single_volume_renderer.py:289-300)."""
    rays_o = rays_o.float().contiguous()
    rays_d = rays_d.float().contiguous()
    N, dev = rays_o.shape[0], rays_o.device
    near_t = torch.empty([N], dtype=torch.float32, device=dev)
    far_t = torch.empty([N], dtype=torch.float32, device=dev)
    hit = torch.empty([N], dtype=torch.uint8, device=dev)
    _lib.call("nsim_aabb_ray_test", _lib.ptr(rays_o.detach()), _lib.ptr(rays_d.detach()), N, meta,
              float(near) if near is not None else 0.0, float(far) if far is not None else -1.0, _lib.ptr(near_t),
              _lib.ptr(far_t), _lib.ptr(hit))
    rays_inds = hit.nonzero()[:, 0]     # host sync (the reference compacts here as well)
    R = int(rays_inds.shape[0])
    if rays_o.requires_grad or rays_d.requires_grad:        # keep the graph for callers that differentiate rays
        ret = dict(num_rays=R, rays_inds=rays_inds, rays_o=rays_o[rays_inds], rays_d=rays_d[rays_inds],
                   near=near_t[rays_inds], far=far_t[rays_inds])
    else:
        o_h, d_h = torch.empty([R, 3], dtype=torch.float32, device=dev), torch.empty([R, 3], dtype=torch.float32, device=dev)
        n_h, f_h = torch.empty([R], dtype=torch.float32, device=dev), torch.empty([R], dtype=torch.float32, device=dev)
        _lib.call("nsim_gather_rays", _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(near_t), _lib.ptr(far_t),
                  _lib.ptr(rays_inds), R, _lib.ptr(o_h), _lib.ptr(d_h), _lib.ptr(n_h), _lib.ptr(f_h))
        ret = dict(num_rays=R, rays_inds=rays_inds, rays_o=o_h, rays_d=d_h, near=n_h, far=f_h)
    for k, v in extra.items():
        if isinstance(v, torch.Tensor) and v.shape[:1] == (N,):
            if v.requires_grad and v.dim() == 2 and v.dtype == torch.float32:
                from ..losses import embedding_lookup      # row gather with a one-launch backward
                ret[k] = embedding_lookup(v, rays_inds)
            else:
                ret[k] = v[rays_inds]
        else:
            ret[k] = v
    return ret


def make_occ_meta(aabb: torch.Tensor, resolution=(1, 1, 1)):
    m = _lib.OccMeta()
    a = aabb.detach().float().cpu().reshape(2, 3)
    for i in range(3):
        m.aabb_min[i], m.aabb_max[i] = float(a[0, i]), float(a[1, i])
        m.scale[i] = float(resolution[i]) / float(a[1, i] - a[0, i])
        m.res[i] = int(resolution[i])
    return m


class AABBSpace(nn.Module):
    def __init__(self, bounding_size: float = 2.0, aabb: Optional[torch.Tensor] = None, device=None, dtype=torch.float32):
        super().__init__()
        if aabb is None:
            h = float(bounding_size) / 2.0
            aabb = torch.tensor([[-h, -h, -h], [h, h, h]])
        self.register_buffer("aabb", torch.as_tensor(aabb, dtype=torch.float32).reshape(2, 3).clone())
        self._meta = None
        if device is not None:
            self.to(device)

    @property
    def center(self) -> torch.Tensor:
        return (self.aabb[1] + self.aabb[0]) / 2.0

    @property
    def radius3d(self) -> torch.Tensor:
        return (self.aabb[1] - self.aabb[0]) / 2.0

    def get_bounding_volume(self) -> torch.Tensor:
        """[6] = centre (3) + per-axis radius (3), object coordinates (app/resources/nodes.py:92-103)."""
        return torch.cat([self.center, self.radius3d], dim=-1)

    def normalize_coords(self, x: torch.Tensor) -> torch.Tensor:
        """object coordinates -> [-1, 1]^3 of the AABB."""
        return (x - self.center) / self.radius3d

    def unnormalize_coords(self, x: torch.Tensor) -> torch.Tensor:
        return x * self.radius3d + self.center

    def contains(self, x: torch.Tensor) -> torch.Tensor:
        return ((x >= self.aabb[0]) & (x <= self.aabb[1])).all(dim=-1)

    def sample_pts_uniform(self, num_pts: int, generator=None) -> torch.Tensor:
        return self.aabb[0] + torch.rand([num_pts, 3], device=self.aabb.device, generator=generator) * (self.aabb[1] - self.aabb[0])

    def ray_test(self, rays_o, rays_d, near=None, far=None, **extra) -> Dict:
        if self._meta is None or self._meta[0] != self.aabb._version:
            self._meta = (self.aabb._version, make_occ_meta(self.aabb))
        return aabb_ray_test(self.aabb, self._meta[1], rays_o, rays_d, near=near, far=far, **extra)
