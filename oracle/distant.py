"""oracle.distant -- CPU restatement of the NeRF++ distant-view model (``LoTDNeRFDistant``) of the reference.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED: ``nr3d_lib.models.fields_distant.nerf`` is absent.
Follows the reference's wrapper and config:
* app/models/single/nerf.py:145-196 (reuses the close-range object's AABB and ray_test);
* call site app/renderers/single_volume_renderer.py:281-309 (all rays are queried, ``near`` := cr ``far`` on the rays
  that hit the close-range AABB, pose gradients detached);
* code_single/configs/object_centric/lotd_neus.dtu.230814.yaml:186-247: 4-D LoTD ``ngp4d`` (target 8 Mi params,
  min_res_xyz 8, min_res_w 4, F=2, T=2^19, scale 1.382), density decoder D1 W64 with softplus output, radiance decoder
  D2 W64 on [features, SH-4 view dirs, appearance-4] (``use_pos false``, ``use_nablas false``),
  ``include_inf_distance true``, ``radius_scale_min/max 1/1000``, ``sample_mode box``, ``max_steps 64``.

Conventions fixed here (mirrored by csrc/nerf_field.hip):
* shells: 1/r is uniform in [1/r_max, 1/r_min]: ``inv_r_k = 1/r_min + (k+u)/K (1/r_max - 1/r_min)``; the sample of shell
  k sits where the ray LEAVES the AABB scaled by r_k about its centre; shells the ray does not cross are invalid
  (alpha 0) -- the buffer stays batched [N, K];
* 4-D input: (p/r normalised to the unit cube, 1/r), each mapped to [0,1]; level l has (Rx^3 * Rw) vertices,
  Rx = ceil(min_res_xyz s^l), Rw = ceil(min_res_w s^l); Dense iff Rx^3 Rw <= T; levels are added until the parameter
  count reaches ``target_num_params``; hash primes (1, 2654435761, 805459861, 3674653429);
* sigma = softplus(raw) ; alpha_k = 1 - exp(-sigma_k * delta_k), delta_k = t_{k+1} - t_k over VALID shells, the last
  valid shell gets delta = 1e10 (``include_inf_distance``);
* hidden activations ReLU, rgb sigmoid.
"""
import math
from dataclasses import dataclass, field
from typing import List

import torch
import torch.nn.functional as F

from .field import sh4

PRIMES4 = (1, 2654435761, 805459861, 3674653429)


@dataclass
class LoTD4Spec:
    res_xyz: List[int]
    res_w: List[int]
    types: List[str]
    sizes: List[int]
    offsets: List[int]
    n_params: int
    hashmap_size: int
    n_feats: int = 2
    res3: List[List[int]] = None        # per-axis (x, y, z) vertex counts (``lotd_use_cuboid``); None = cubic res_xyz

    @property
    def num_levels(self):
        return len(self.res_xyz)

    def res_of(self, l: int):
        """(Rx, Ry, Rz, Rw) of level l."""
        r3 = self.res3[l] if self.res3 is not None else [self.res_xyz[l]] * 3
        return [*r3, self.res_w[l]]


def make_ngp4d_spec(target_num_params=8 * 2 ** 20, min_res_xyz=8, min_res_w=4, n_feats=2, log2_hashmap_size=19,
                    per_level_scale=1.382, max_levels=16, aspect=None) -> LoTD4Spec:
    """``aspect`` = the AABB's (x, y, z) extents for ``lotd_use_cuboid: true``
    (withmask_withlidar_joint.240219.yaml:256): the shortest axis gets the ``min_res_xyz`` progression, the others are
    stretched by their extent ratio (same convention as the 3-D pyramid, oracle/lotd.py cuboid_ngp_res)."""
    T = 2 ** log2_hashmap_size
    rx, rw, types, sizes, offs, r3 = [], [], [], [], [], []
    off = 0
    asp = [1.0, 1.0, 1.0] if aspect is None else [float(a) / min(float(b) for b in aspect) for a in aspect]
    for l in range(max_levels):
        Rx = int(math.ceil(min_res_xyz * per_level_scale ** l - 1e-6))
        Rw = int(math.ceil(min_res_w * per_level_scale ** l - 1e-6))
        R3 = [int(math.ceil(min_res_xyz * per_level_scale ** l * a - 1e-6)) for a in asp]
        n = R3[0] * R3[1] * R3[2] * Rw
        dense = n <= T
        rx.append(Rx)
        rw.append(Rw)
        r3.append(R3)
        types.append('Dense' if dense else 'Hash')
        sizes.append(n if dense else T)
        offs.append(off)
        off += sizes[-1] * n_feats
        if off >= target_num_params:
            break
    return LoTD4Spec(rx, rw, types, sizes, offs, off, T, n_feats, None if aspect is None else r3)


def lotd4_forward(u: torch.Tensor, params: torch.Tensor, spec: LoTD4Spec) -> torch.Tensor:
    """u [S,4] in [0,1] -> [S, L*2] f32 (quadrilinear, 16 corners per level)."""
    S = u.shape[0]
    p32 = params.float()
    outs = []
    m = 0xFFFFFFFF
    for l in range(spec.num_levels):
        R4 = spec.res_of(l)
        R = torch.tensor(R4, dtype=torch.float32)
        pos = u * (R - 1.0)
        c0 = torch.minimum(torch.floor(pos.detach()).clamp_min(0), R - 2.0).long()
        w = pos - c0.to(pos.dtype)
        table = p32[spec.offsets[l]: spec.offsets[l] + spec.sizes[l] * 2].view(-1, 2)
        feat = u.new_zeros([S, 2])
        Rx, Ry, Rz = R4[0], R4[1], R4[2]
        for corner in range(16):
            d = [(corner >> a) & 1 for a in range(4)]
            wt = torch.ones(S)
            for a in range(4):
                wt = wt * (w[:, a] if d[a] else 1.0 - w[:, a])
            c = [c0[:, a] + d[a] for a in range(4)]
            if spec.types[l] == 'Dense':
                idx = c[0] + Rx * (c[1] + Ry * (c[2] + Rz * c[3]))
            else:
                idx = ((c[0] * PRIMES4[0]) & m) ^ ((c[1] * PRIMES4[1]) & m) ^ ((c[2] * PRIMES4[2]) & m) ^ ((c[3] * PRIMES4[3]) & m)
                idx = idx % spec.hashmap_size
            feat = feat + wt.unsqueeze(-1) * table[idx]
        outs.append(feat)
    return torch.cat(outs, dim=-1)


@dataclass
class DistantParams:
    spec: LoTD4Spec
    grid: torch.Tensor                                        # fp16-representable values
    den_w: List[torch.Tensor] = field(default_factory=list)   # (64, F), (1, 64)
    den_b: List[torch.Tensor] = field(default_factory=list)
    rad_w: List[torch.Tensor] = field(default_factory=list)   # (64, F+20), (64, 64), (3, 64)
    rad_b: List[torch.Tensor] = field(default_factory=list)

    def tensors(self):
        return [self.grid, *self.den_w, *self.den_b, *self.rad_w, *self.rad_b]

    def requires_grad_(self, flag=True):
        for t in self.tensors():
            t.requires_grad_(flag)
        return self


def make_distant_params(spec: LoTD4Spec = None, seed=7, grid_bound=1e-4, W=64, use_view_dirs=True) -> DistantParams:
    spec = spec or make_ngp4d_spec()
    g = torch.Generator().manual_seed(seed)
    grid = (((torch.rand(spec.n_params, generator=g) * 2 - 1) * grid_bound).half()).float()
    Fdim = spec.num_levels * 2

    def lin(o, i):
        b = 1.0 / math.sqrt(i)
        return (torch.rand(o, i, generator=g) * 2 - 1) * b, (torch.rand(o, generator=g) * 2 - 1) * b
    dw1, db1 = lin(W, Fdim)
    dw2, db2 = lin(1, W)
    rw1, rb1 = lin(W, Fdim + (20 if use_view_dirs else 4))     # radiance_decoder_cfg.use_view_dirs (street: false)
    rw2, rb2 = lin(W, W)
    rw3, rb3 = lin(3, W)
    return DistantParams(spec, grid, [dw1, dw2], [db1, db2], [rw1, rw2, rw3], [rb1, rb2, rb3])


def distant_forward(u4, v, h_appear, p: DistantParams):
    """-> sigma [S], rgb [S,3]."""
    h = lotd4_forward(u4, p.grid, p.spec)
    a = F.relu(F.linear(h, p.den_w[0], p.den_b[0]))
    sigma = F.softplus(F.linear(a, p.den_w[1], p.den_b[1]).squeeze(-1))
    use_view_dirs = p.rad_w[0].shape[1] == h.shape[1] + 20
    rin = torch.cat([h, sh4(v), h_appear], dim=-1) if use_view_dirs else torch.cat([h, h_appear], dim=-1)
    r = F.relu(F.linear(rin, p.rad_w[0], p.rad_b[0]))
    r = F.relu(F.linear(r, p.rad_w[1], p.rad_b[1]))
    rgb = torch.sigmoid(F.linear(r, p.rad_w[2], p.rad_b[2]))
    return sigma, rgb


def distant_shells(rays_o, rays_d, aabb_min, aabb_max, near, K=64, r_min=1.0, r_max=1000.0, jitter=None):
    """-> t [N,K], inv_r [N,K], valid [N,K] (exit depth of the AABB scaled by r_k; must lie beyond ``near``)."""
    N = rays_o.shape[0]
    k = torch.arange(K, dtype=torch.float32)
    u = jitter if jitter is not None else torch.full((N, K), 0.5)
    inv_r = 1.0 / r_min + ((k[None, :] + u) / float(K)) * (1.0 / r_max - 1.0 / r_min)
    r = 1.0 / inv_r
    center = (aabb_min + aabb_max) * 0.5
    half = (aabb_max - aabb_min) * 0.5
    o = (rays_o - center)[:, None, :]                          # [N,1,3]
    d = rays_d[:, None, :]
    tiny = 1e-12
    d_safe = torch.where(d.abs() < tiny, torch.where(d < 0, -torch.full_like(d, tiny), torch.full_like(d, tiny)), d)
    inv = 1.0 / d_safe
    hr = half[None, None, :] * r[..., None]                     # [N,K,3]
    t1 = (-hr - o) * inv
    t2 = (hr - o) * inv
    tmin = torch.minimum(t1, t2).max(dim=-1).values
    tmax = torch.maximum(t1, t2).min(dim=-1).values
    valid = (tmax > tmin) & (tmax > near[:, None])
    return tmax, inv_r, valid


def shell_points_u4(rays_o, rays_d, t, inv_r, aabb_min, aabb_max):
    """4-D network input in [0,1]^4 of the shell samples."""
    center = (aabb_min + aabb_max) * 0.5
    half = (aabb_max - aabb_min) * 0.5
    x = rays_o[:, None, :] + t[..., None] * rays_d[:, None, :]
    xn = (x - center) / half * inv_r[..., None]                 # on the unit cube surface
    u = torch.cat([xn * 0.5 + 0.5, inv_r[..., None]], dim=-1)
    return u.clamp(0.0, 1.0)


def density_alpha(sigma, t, valid, include_inf=True):
    """alpha_k = 1 - exp(-sigma_k delta_k) over valid shells; the last valid shell reaches to infinity
    (``include_inf_distance``) or repeats the interval before it (0 if it has no valid predecessor)."""
    N, K = t.shape
    big = torch.full_like(t, float('inf'))
    tv = torch.where(valid, t, big)
    # next valid depth: suffix minimum of later valid depths
    nxt = torch.flip(torch.cummin(torch.flip(torch.cat([tv[:, 1:], big[:, :1]], dim=1), [1]), dim=1).values, [1])
    if include_inf:
        last = torch.full_like(t, 1e10)
    else:
        prev_t = torch.cat([t[:, :1], t[:, :-1]], dim=1)
        prev_ok = torch.cat([valid[:, :1] & False, valid[:, :-1]], dim=1)
        last = torch.where(prev_ok, t - prev_t, torch.zeros_like(t))
    delta = torch.where(torch.isinf(nxt), last, nxt - t)
    alpha = 1.0 - torch.exp(-sigma * delta)
    return torch.where(valid, alpha, torch.zeros_like(alpha))


def distant_ray_query(p: DistantParams, rays_o, rays_d, near, h_appear, aabb_min, aabb_max, K=64, r_min=1.0,
                      r_max=1000.0, jitter=None, include_inf=True):
    """``query_mode: march`` of the distant model on ALL rays -> batched volume buffer [N,K]."""
    N = rays_o.shape[0]
    with torch.no_grad():
        t, inv_r, valid = distant_shells(rays_o, rays_d, aabb_min, aabb_max, near, K, r_min, r_max, jitter)
        u4 = shell_points_u4(rays_o, rays_d, t, inv_r, aabb_min, aabb_max)
    v = rays_d[:, None, :].expand(N, K, 3)
    ha = h_appear[:, None, :].expand(N, K, 4) if h_appear is not None else torch.zeros(N, K, 4)
    sigma, rgb = distant_forward(u4.reshape(-1, 4), v.reshape(-1, 3), ha.reshape(-1, 4), p)
    sigma, rgb = sigma.view(N, K), rgb.view(N, K, 3)
    alpha = density_alpha(sigma, t, valid, include_inf)
    return dict(type='batched', rays_inds_hit=torch.arange(N), num_per_hit=K, t=t, opacity_alpha=alpha, rgb=rgb,
                sigma=sigma, valid=valid, u4=u4)
