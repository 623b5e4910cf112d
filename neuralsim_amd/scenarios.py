"""Synthetic scenarios shaped like BASELINE.json ``configs[2..4]`` (SURVEY.md sec. 8a', 8d) -- the inputs of the hot path
for the indoor, street and multi-object configurations, built without a dataset: an analytic world (ground plane,
spheres, a box room) that supplies BOTH the geometric initialisation of the SDF tables (standing in for the reference's
pre-training loops, app/models/single/neus.py:198-236) and the dataset side (images, monocular depth / normals rendered
analytically and held in HBM, gathered per batch as the reference's pixel loader does,
dataio/data_loader/pixel_loader.py:323-327).

  * ``build_street_trainer``  configs[3]  code_single/configs/waymo/streetsurf/withmask_withlidar_joint.240219.yaml
        cuboid 19-level LoTD (T = 2^20, ~33 Mi parameters), 1x64 SDF decoder, ``sdf_scale 25``, AABB 200 x 100 x 30 m
        with 1 m occupancy voxels, near .1 / far 200, step .2, ``num_coarse 128``, ``upsample_use_estimate_alpha: false``,
        distant model (``fixed_cuboid_shells``: no view directions, ``include_inf_distance: false``, cuboid 4-D pyramid
        16 Mi) + sky MLP, 6-camera rig, l1 photometric loss, 2^16 uniform eikonal points, 16384 rays per GPU.
  * ``build_indoor_trainer``  configs[2]  code_single/configs/indoor/lotd_neus.replica.230814.yaml
        ``inside_out`` geometry (a box room seen from inside), 64 x 64 image patch + pixel rays = 16384, normals and depth
        rendered WITH gradient for the monocular losses, no distant / sky model.
  * ``build_multi_trainer``   configs[4]  code_multi/configs/exps/fg_neus=hyper_lotd/no_fg_occ.221218.yaml
        street background + 8 posed instances of ONE shared batched vehicle model (``occ_grid_batched`` 32^3,
        ``num_coarse 32``, ``num_fine 8``, ``upsample_inv_s_factors [1, 4]``) + distant + sky through the
        ``BufferComposeRenderer`` mirror, 16384 rays per GPU.

``small=True`` shrinks tables, occupancy grids, images and ray counts so the SAME code runs on the host emulator of the
CPU test-suite.  Used by bench.py (``variants.street_ms / indoor_ms / multi_ms``) and tests/test_fullsize_configs.py.
"""
import math
import os
from typing import Dict, List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, backward_on_calling_thread
from .fields.neus import LoTDNeuSModel
from .graphics.cameras import selected_rays
from .grid_encodings.lotd import cuboid_ngp_res, gen_ngp_res


# ------------------------------------------------------------------------------------------------ analytic world
class AnalyticWorld:
    """Ground plane z = ``ground_z`` (or None), spheres, and optionally the inside of an axis-aligned box (``room``
    half-extents [3]: the camera is INSIDE, walls face inwards).  ``sdf`` is the exact signed distance of the union
    (positive in free space), ``trace`` the first hit of unit rays with its normal and a position-dependent colour."""

    def __init__(self, ground_z: Optional[float] = None, spheres: Optional[List] = None, room: Optional[List[float]] = None,
                 device=None):
        self.ground_z = ground_z
        sp = spheres or []
        self.centres = torch.tensor([s[0] for s in sp], dtype=torch.float32, device=device).reshape(-1, 3)
        self.radii = torch.tensor([s[1] for s in sp], dtype=torch.float32, device=device).reshape(-1)
        self.colours = torch.tensor([s[2] for s in sp], dtype=torch.float32, device=device).reshape(-1, 3)
        self.room = torch.tensor(room, dtype=torch.float32, device=device) if room is not None else None

    def to(self, device):
        w = AnalyticWorld.__new__(AnalyticWorld)
        w.ground_z = self.ground_z
        w.centres, w.radii, w.colours = self.centres.to(device), self.radii.to(device), self.colours.to(device)
        w.room = self.room.to(device) if self.room is not None else None
        return w

    def sdf(self, x: torch.Tensor) -> torch.Tensor:
        x = x.float()
        w = self if self.centres.device == x.device else self.to(x.device)
        d = torch.full(x.shape[:-1], float("inf"), device=x.device)
        if w.ground_z is not None:
            d = torch.minimum(d, x[..., 2] - w.ground_z)
        if w.room is not None:
            d = torch.minimum(d, (w.room - x.abs()).min(dim=-1).values)
        if w.centres.shape[0]:
            ds = (x[..., None, :] - w.centres).norm(dim=-1) - w.radii
            d = torch.minimum(d, ds.min(dim=-1).values)
        return d

    def trace(self, o: torch.Tensor, d: torch.Tensor) -> Dict[str, torch.Tensor]:
        """-> dict(hit [N] bool, t [N], normal [N,3], rgb [N,3]); rays that hit nothing get the sky colour."""
        w = self if self.centres.device == o.device else self.to(o.device)
        N, dev = o.shape[0], o.device
        t = torch.full([N], float("inf"), device=dev)
        nrm = torch.zeros([N, 3], device=dev)
        rgb = torch.zeros([N, 3], device=dev)
        if w.ground_z is not None:
            dz = d[:, 2]
            tp = (w.ground_z - o[:, 2]) / torch.where(dz.abs() < 1e-9, torch.full_like(dz, -1e-9), dz)
            ok = (dz < 0) & (tp > 0)
            p = o + tp[:, None] * d
            chk = ((torch.floor(p[:, 0] / 4.0) + torch.floor(p[:, 1] / 4.0)) % 2.0)
            col = torch.stack([0.30 + 0.15 * chk, 0.30 + 0.15 * chk, 0.32 + 0.13 * chk], dim=-1)
            t = torch.where(ok, tp, t)
            nrm = torch.where(ok[:, None], torch.tensor([0.0, 0.0, 1.0], device=dev).expand(N, 3), nrm)
            rgb = torch.where(ok[:, None], col, rgb)
        if w.room is not None:
            ds = torch.where(d.abs() < 1e-9, torch.full_like(d, 1e-9), d)
            t_ax = (torch.sign(ds) * w.room - o) / ds                         # exit depth per axis (origin inside)
            tw, ax = t_ax.min(dim=-1)
            n_w = -F.one_hot(ax, 3).float() * torch.sign(ds.gather(1, ax[:, None]))
            p = o + tw[:, None] * d
            base = torch.tensor([[0.8, 0.5, 0.4], [0.4, 0.7, 0.5], [0.45, 0.5, 0.8]], device=dev)[ax]
            stripe = 0.85 + 0.15 * torch.sin(6.0 * (p.sum(-1)))
            ok = tw < t
            t = torch.where(ok, tw, t)
            nrm = torch.where(ok[:, None], n_w, nrm)
            rgb = torch.where(ok[:, None], base * stripe[:, None], rgb)
        for i in range(w.centres.shape[0]):
            oc = o - w.centres[i]
            b = (oc * d).sum(-1)
            disc = b * b - ((oc * oc).sum(-1) - w.radii[i] ** 2)
            ts = -b - torch.sqrt(disc.clamp_min(0))
            ok = (disc > 0) & (ts > 0) & (ts < t)
            n_s = F.normalize(oc + ts[:, None] * d, dim=-1)
            t = torch.where(ok, ts, t)
            nrm = torch.where(ok[:, None], n_s, nrm)
            rgb = torch.where(ok[:, None], w.colours[i] * (0.55 + 0.45 * n_s[:, 2:3]), rgb)
        hit = torch.isfinite(t)
        sky = torch.stack([0.45 + 0.2 * d[:, 2], 0.6 + 0.15 * d[:, 2], 0.85 + 0.1 * d[:, 2]], dim=-1).clamp(0, 1)
        rgb = torch.where(hit[:, None], rgb, sky)
        return dict(hit=hit, t=torch.where(hit, t, torch.zeros_like(t)), normal=nrm, rgb=rgb.clamp(0, 1))


def mono_priors(depth: torch.Tensor, normals: torch.Tensor, pos: torch.Tensor):
    """Monocular cues are network PREDICTIONS of a geometry, not the geometry: depth up to an affine map with a smooth
    multiplicative error, normals with a smooth angular error (both deterministic functions of the surface point).
    Priors that coincide with the model's own geometry to the last bit would make the L1 normal term's gradient
    sign(n - n_gt) -- and the residuals of the scale-and-shift fit -- pure rounding noise."""
    wob = torch.sin(3.1 * pos[..., 0] + 1.7 * pos[..., 1]) * torch.cos(2.3 * pos[..., 2] + 0.5)
    d = (depth * 1.7 + 0.3) * (1.0 + 0.06 * wob)
    dn = torch.stack([torch.sin(2.9 * pos[..., 1] + 0.3), torch.sin(3.7 * pos[..., 2] + 1.1), torch.sin(2.3 * pos[..., 0] + 2.0)], dim=-1)
    return d, F.normalize(normals + 0.12 * dn, dim=-1)


@torch.no_grad()
def render_dataset(world: AnalyticWorld, intr, c2w, WH, with_mono: bool = False, chunk: int = 2 ** 18):
    """The synthetic dataset of a scenario: images [V,H,W,3] (+ monocular depth [V,H,W] and world-space normal
    [V,H,W,3] priors, ``mono_priors``) of the analytic world from every camera, resident on the cameras' device."""
    from .eval import all_pixel_xy
    V, dev = intr.shape[0], intr.device
    W, H = int(WH[0, 0]), int(WH[0, 1])
    xy = all_pixel_xy(W, H, dev)
    img = torch.empty([V, H, W, 3], dtype=torch.float32, device=dev)
    dep = torch.empty([V, H, W], dtype=torch.float32, device=dev) if with_mono else None
    nrm = torch.empty([V, H, W, 3], dtype=torch.float32, device=dev) if with_mono else None
    for f in range(V):
        for s in range(0, xy.shape[0], chunk):
            q = xy[s:s + chunk]
            o, d = selected_rays(q, torch.full([q.shape[0]], f, dtype=torch.long, device=dev), intr, c2w, WH)
            tr = world.trace(o, d)
            img[f].view(-1, 3)[s:s + chunk] = tr["rgb"]
            if with_mono:
                d_, n_ = mono_priors(tr["t"], tr["normal"], o + tr["t"][:, None] * d)
                dep[f].view(-1)[s:s + chunk] = d_
                nrm[f].view(-1, 3)[s:s + chunk] = n_
    return (img, dep, nrm) if with_mono else img


def _pose(eye, fwd, world_up=(0.0, 0.0, 1.0)):
    """OpenCV camera-to-world (+x right, +y down, +z forward) at ``eye`` looking along ``fwd``."""
    fwd = F.normalize(torch.as_tensor(fwd, dtype=torch.float32), dim=0)
    up = torch.tensor(world_up, dtype=torch.float32)
    right = torch.linalg.cross(fwd, up)
    if float(right.norm()) < 1e-6:
        right = torch.tensor([1.0, 0.0, 0.0])
    right = F.normalize(right, dim=0)
    down = torch.linalg.cross(fwd, right)
    m = torch.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = right, down, fwd, torch.as_tensor(eye, dtype=torch.float32)
    return m


def street_rig(n_ego: int = 10, x_range=(-60.0, 60.0), H: int = 800, W: int = 800, f: float = 500.0, device=None):
    """An ego vehicle driving along +x at z = 0 (road at z = -2: ``ego_height 2.0``, 240219.yaml:241) with a 6-camera rig
    (front, front-left/right, side-left/right, rear; slight downward pitch) -> intr [V,3,3], c2w [V,4,4], WH [V,2],
    V = 6 n_ego, frame index = ego * 6 + camera."""
    yaws = [0.0, 50.0, -50.0, 100.0, -100.0, 180.0]
    poses = []
    for e in range(n_ego):
        x = x_range[0] + (x_range[1] - x_range[0]) * (e + 0.5) / n_ego
        for k, yaw in enumerate(yaws):
            a = math.radians(yaw)
            poses.append(_pose([x, 0.3 * math.sin(1.3 * e), 0.0], [math.cos(a), math.sin(a), -0.08]))
    c2w = torch.stack(poses)
    V = c2w.shape[0]
    intr = torch.tensor([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1.0]]).repeat(V, 1, 1)
    WH = torch.tensor([[W, H]], dtype=torch.long).repeat(V, 1)
    if device is not None:
        intr, c2w, WH = intr.to(device), c2w.to(device), WH.to(device)
    return intr, c2w, WH


def indoor_rig(V: int = 40, H: int = 800, W: int = 800, f: float = 450.0, seed: int = 3, device=None):
    """V cameras near the centre of the room looking in quasi-uniform directions (a hand-held indoor capture)."""
    g = torch.Generator().manual_seed(seed)
    poses = []
    for i in range(V):
        u = torch.rand(5, generator=g)
        th, z = 2 * math.pi * (i + float(u[0])) / V * 3.0, -0.6 + 1.2 * float(u[1])
        fwd = [math.sqrt(1 - z * z) * math.cos(th), math.sqrt(1 - z * z) * math.sin(th), z]
        eye = ((u[2:5] - 0.5) * 0.5).tolist()
        poses.append(_pose(eye, fwd))
    c2w = torch.stack(poses)
    intr = torch.tensor([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1.0]]).repeat(V, 1, 1)
    WH = torch.tensor([[W, H]], dtype=torch.long).repeat(V, 1)
    if device is not None:
        intr, c2w, WH = intr.to(device), c2w.to(device), WH.to(device)
    return intr, c2w, WH


# ------------------------------------------------------------------------------------------------ street (configs[3])
STREET_AABB = [[-100.0, -50.0, -15.0], [100.0, 50.0, 15.0]]
STREET_SDF_SCALE = 25.0


def street_world(device=None) -> AnalyticWorld:
    """Road at z = -2 m and a row of 'buildings' (big spheres) on both sides."""
    g = torch.Generator().manual_seed(5)
    sp = []
    for i in range(14):
        u = torch.rand(4, generator=g)
        side = 1.0 if i % 2 == 0 else -1.0
        r = 6.0 + 5.0 * float(u[0])
        sp.append(([-85.0 + 13.0 * i + 4.0 * float(u[1]), side * (16.0 + r + 6.0 * float(u[2])), -2.0 + 0.6 * r], r,
                   [0.35 + 0.6 * float(u[3]), 0.4 + 0.5 * float(u[1]), 0.3 + 0.6 * float(u[2])]))
    return AnalyticWorld(ground_z=-2.0, spheres=sp, device=device)


def street_models(device, precision: str = "fp16", seed: int = 42, small: bool = False, world: AnalyticWorld = None,
                  with_distant: bool = True, with_sky: bool = True):
    """-> (street NeuS model, distant model, sky model) of the street config, geometry initialised to ``world``."""
    from .env import SimpleSky
    from .fields.nerf_distant import LoTDNeRFDistantModel
    aabb = torch.tensor(STREET_AABB)
    ext = (aabb[1] - aabb[0]).tolist()
    if small:
        res, l2, occ_res = cuboid_ngp_res(ext, 3, 24, 18), 12, [40, 20, 6]
        acc_n, d_auto, sky_w = dict(num_steps=2, num_pts=2 ** 13), dict(target_num_params=2 ** 14, min_res_xyz=3, min_res_w=2,
                                                                        log2_hashmap_size=10), 256
        qp = dict(num_coarse=16, num_fine=[4, 4, 8], march_cfg=dict(step_size=2.0, max_steps=256))
    else:
        # lotd_auto_compute_cfg{type: ngp, target_num_params 32 Mi, min_res 16, log2_hashmap_size 20} (yaml:161-167):
        # 19 cuboid levels up to 2048 vertices along the shortest axis land at ~33 Mi parameters
        res, l2, occ_res = cuboid_ngp_res(ext, 16, 2048, 19), 20, [200, 100, 30]       # vox_size 1.0 (yaml:197)
        acc_n, d_auto, sky_w = dict(num_steps=4, num_pts=2 ** 20), dict(target_num_params=16 * 2 ** 20, min_res_xyz=16,
                                                                        min_res_w=4, log2_hashmap_size=19), 256
        qp = dict(num_coarse=128, num_fine=[8, 8, 32], march_cfg=dict(step_size=0.2, max_steps=4096))
    qp.update(nablas_has_grad=True, upsample_inv_s=64.0, upsample_inv_s_factors=[1, 4, 16],
              upsample_use_estimate_alpha=False)                                       # yaml:71, 229
    m = LoTDNeuSModel(lod_res=res, log2_hashmap_size=l2, sdf_D=1, precision=precision, ln_inv_s_init=0.45,
                      sdf_scale=STREET_SDF_SCALE, aabb=aabb, seed=seed,
                      accel_cfg=dict(resolution=occ_res, init_cfg=acc_n, update_from_net_cfg=acc_n, update_from_samples_cfg={}),
                      ray_query_cfg=dict(query_mode="march_occ_multi_upsample_compressed", query_param=qp)).to(device)
    world = world or street_world()
    # ``sdf_scale``: one unit of SDF is 25 m (yaml:24) -- the table holds distance / 25
    m.geometric_init_fn(lambda x: world.sdf(x) / STREET_SDF_SCALE, noise_scale=0.25)
    dm = sm = None
    if with_distant:
        dm = LoTDNeRFDistantModel(aabb=aabb, precision=precision, include_inf_distance=False, use_view_dirs=False,
                                  lotd_use_cuboid=True, lotd_auto_compute_cfg=d_auto, max_steps=16 if small else 64,
                                  seed=seed + 7).to(device)
        # depths are METRES here and the shells reach 1000 x the 200 m box: the density head starts at
        # softplus(-11) = 1.7e-5 / m (optical depth ~1 over the shells), as a fitted background has -- the default
        # bias (sigma 0.7 / m) would make every first shell opaque and hide the sky from the step
        with torch.no_grad():
            dm.den_b[-1] = -11.0
            dm.den_b.add_(0)
    if with_sky:
        sm = SimpleSky(n_appear_embedding=4, W=sky_w, precision=precision, seed=seed + 11).to(device)
    return m, dm, sm


@torch.no_grad()
def street_lidar(world: AnalyticWorld, n_ego: int = 10, beams: int = 16384, x_range=(-60.0, 60.0), device=None, seed: int = 9):
    """A roof lidar on the ego vehicle (``lidar_TOP``): per ego pose ``beams`` rays, azimuth uniform, elevation in
    [-22, +2.5] degrees; ranges by tracing the analytic world (0 = no return).  -> rays_o, rays_d [F,M,3], ranges [F,M]."""
    g = torch.Generator().manual_seed(seed)
    o_l, d_l, r_l = [], [], []
    for e in range(n_ego):
        x = x_range[0] + (x_range[1] - x_range[0]) * (e + 0.5) / n_ego
        az = torch.rand(beams, generator=g) * (2 * math.pi)
        el = torch.deg2rad(-22.0 + 24.5 * torch.rand(beams, generator=g))
        d = torch.stack([torch.cos(el) * torch.cos(az), torch.cos(el) * torch.sin(az), torch.sin(el)], dim=-1)
        o = torch.tensor([x, 0.3 * math.sin(1.3 * e), 0.5]).expand(beams, 3).contiguous()
        tr = world.trace(o, d)
        o_l.append(o)
        d_l.append(d)
        r_l.append(torch.where(tr["hit"], tr["t"], torch.zeros_like(tr["t"])))
    out = torch.stack(o_l), torch.stack(d_l), torch.stack(r_l)
    return tuple(t.to(device) for t in out) if device is not None else out


def build_street_trainer(device, rank: int = 0, world_size: int = 1, precision: str = "fp16", rays_per_gpu: int = 16384,
                         seed: int = 42, small: bool = False, num_uniform: Optional[int] = None, n_ego: int = None,
                         lidar_rays: int = 0):
    from . import distributed as ndist
    from .trainer import RenderTrainer
    world = street_world()
    m, dm, sm = street_models(device, precision, seed, small, world)
    m.accel.init(m.query_sdf, generator=torch.Generator(device=device).manual_seed(seed))
    for mod in (m, dm, sm):
        ndist.broadcast_module(mod)
    hw = 24 if small else 800
    intr, c2w, WH = street_rig(n_ego=n_ego or (2 if small else 10), H=hw, W=hw, f=0.625 * hw, device=device)
    images = render_dataset(world, intr, c2w, WH)
    if num_uniform is None:
        num_uniform = 256 if small else 2 ** 16                                         # ``num_uniform 2^16`` (yaml:38)
    lidar = None
    if lidar_rays:          # the reference's street iteration: a pixel batch AND a lidar batch (yaml:7-8: 8192 + 8192)
        lo, ld, lr_ = street_lidar(world, n_ego=n_ego or (2 if small else 10), beams=512 if small else 16384, device=device)
        lidar = dict(rays_o=lo, rays_d=ld, ranges=lr_, num_rays=int(lidar_rays), w_depth=0.02, w_los=0.1, epsilon=1.5,
                     discard_toofar=80.0, near=0.1, far=200.0)
    return RenderTrainer(m, intr, c2w, WH, num_rays=rays_per_gpu, lr=1e-3, w_eikonal=0.01, num_uniform=num_uniform,
                         near=0.1, far=200.0, rank=rank, world_size=world_size, seed=seed, learn_inv_s=False,
                         distant_model=dm, sky_model=sm, target_images=images, rgb_fn="l1", lidar=lidar)


# ------------------------------------------------------------------------------------------------ indoor (configs[2])
def indoor_world(device=None) -> AnalyticWorld:
    return AnalyticWorld(room=[0.85, 0.8, 0.7], spheres=[([0.45, -0.35, -0.5], 0.2, [0.8, 0.3, 0.3]),
                                                         ([-0.4, 0.3, -0.45], 0.25, [0.3, 0.4, 0.8])], device=device)


def indoor_model(device, precision: str = "fp16", seed: int = 42, small: bool = False, world: AnalyticWorld = None):
    if small:
        res, l2, occ_res = [4, 6, 8, 11, 14, 18, 23, 29, 36, 44, 53, 63, 74, 86, 99, 113], 12, [16, 16, 16]
        acc_n = dict(num_steps=2, num_pts=2 ** 13)
        qp = dict(num_coarse=16, num_fine=[4, 4, 8], march_cfg=dict(step_size=0.02, max_steps=256))
    else:
        res, l2, occ_res = gen_ngp_res(16, 2048, 16), 19, [64, 64, 64]
        acc_n = dict(num_steps=4, num_pts=2 ** 20)
        qp = dict(num_coarse=64, num_fine=[8, 8, 32], march_cfg=dict(step_size=0.005, max_steps=4096))
    qp.update(nablas_has_grad=True, upsample_inv_s=64.0, upsample_inv_s_factors=[1, 4, 16], upsample_use_estimate_alpha=True)
    m = LoTDNeuSModel(lod_res=res, log2_hashmap_size=l2, sdf_D=2, precision=precision, ln_inv_s_init=0.5, inside_out=True,
                      seed=seed, accel_cfg=dict(resolution=occ_res, init_cfg=acc_n, update_from_net_cfg=acc_n,
                                                update_from_samples_cfg={}),
                      ray_query_cfg=dict(query_mode="march_occ_multi_upsample_compressed", query_param=qp)).to(device)
    world = world or indoor_world()
    m.geometric_init_fn(world.sdf, noise_scale=0.25)         # positive inside the room (``inside_out: true``, yaml:95)
    return m


def build_indoor_trainer(device, rank: int = 0, world_size: int = 1, precision: str = "fp16", rays_per_gpu: int = 16384,
                         seed: int = 42, small: bool = False, num_uniform: Optional[int] = None):
    from . import distributed as ndist
    from .trainer import RenderTrainer
    world = indoor_world()
    m = indoor_model(device, precision, seed, small, world)
    m.accel.init(m.query_sdf, generator=torch.Generator(device=device).manual_seed(seed))
    ndist.broadcast_module(m)
    hw = 24 if small else 800
    intr, c2w, WH = indoor_rig(V=4 if small else 40, H=hw, W=hw, f=0.56 * hw, device=device)
    img, dep, nrm = render_dataset(world, intr, c2w, WH, with_mono=True)
    patch = (8, 8) if small else (64, 64)                                               # yaml:237-238
    tr = RenderTrainer(m, intr, c2w, WH, num_rays=rays_per_gpu, lr=1e-3, w_eikonal=0.1,
                       num_uniform=(256 if small else 4096) if num_uniform is None else num_uniform, near=0.01, rank=rank,
                       world_size=world_size, seed=seed, learn_inv_s=False, target_images=img,
                       mono=dict(depth=dep, normals=nrm, patch_hw=patch, w_depth=0.1, w_normal=0.05))
    tr.renderer.config.update(depth_use_normalized_vw=False)
    return tr


# ------------------------------------------------------------------------------------------------ multi (configs[4])
VEHICLE_BOUND = 1.4          # ``bounding_size: 1.4`` (no_fg_occ.221218.yaml:317)


def vehicle_poses(B: int = 8, device=None):
    """B parked / driving 'cars' on the road: (rotation [3,3] about z, translation [3], scale) -- object space is the
    [-0.7, 0.7]^3 box of the shared model, scaled to ~4.5 m vehicles."""
    out = []
    for b in range(B):
        a = 0.25 * math.sin(2.1 * b)
        c, s = math.cos(a), math.sin(a)
        R = torch.tensor([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
        t = torch.tensor([-70.0 + 19.0 * b, 5.5 if b % 2 == 0 else -5.5, -0.2])
        out.append((R.to(device) if device is not None else R, t.to(device) if device is not None else t, 3.2))
    return out


def multi_world(poses, device=None) -> AnalyticWorld:
    """The street world + one sphere per vehicle (the image of an instance whose object-space SDF is a sphere)."""
    w = street_world()
    sp = [(c.tolist(), float(r), col.tolist()) for c, r, col in zip(w.centres, w.radii, w.colours)]
    for b, (R, t, s) in enumerate(poses):
        sp.append((t.tolist(), (0.35 + 0.03 * (b % 4)) * s, [0.9 - 0.08 * b, 0.2 + 0.08 * b, 0.25]))
    return AnalyticWorld(ground_z=-2.0, spheres=sp, device=device)


def vehicle_model(device, B: int = 8, precision: str = "fp16", seed: int = 42, small: bool = False):
    """The shared foreground model: per-instance tables (what the config's ``lotd_grower_cfg`` emits per batch item:
    dense levels [5, 8, 13, 21] + [34, 55, 89, 144], no_fg_occ.221218.yaml:322-352), shared 2x64 decoders,
    ``occ_grid_batched`` [B, 32^3] (:369-377), query ``num_coarse 32, num_fine 8, upsample_inv_s_factors [1, 4]``
    (:378-390; step .005 instead of .001 keeps max_steps 2048 meaningful on a 1.4-wide box)."""
    from .fields.batched_neus import BatchedLoTDNeuSModel
    h = VEHICLE_BOUND / 2
    if small:
        res, l2, occ_res, npts = [3, 5, 8, 13, 21], 13, [16, 16, 16], 2 ** 11
        qp = dict(num_coarse=8, num_fine=8, march_cfg=dict(step_size=0.05, max_steps=128))
    else:
        res, l2, occ_res, npts = [5, 8, 13, 21, 34, 55, 89, 144], 22, [32, 32, 32], 2 ** 16
        # ``num_fine: 8`` next to two factors -- the yaml's literal; ``fields.neus.fine_list`` reads a scalar as the TOTAL, dealt
        # evenly to the stages ([4, 4]): ONE reading of the key for the shim path and for this workload (rounds 3-4 ran [8, 8] here)
        qp = dict(num_coarse=32, num_fine=8, march_cfg=dict(step_size=0.005, max_steps=2048))
    qp.update(nablas_has_grad=True, upsample_inv_s=64.0, upsample_inv_s_factors=[1, 4], upsample_use_estimate_alpha=True)
    vm = BatchedLoTDNeuSModel(B, ins_ids=[f"car{b}" for b in range(B)], lod_res=res, log2_hashmap_size=l2, sdf_D=2,
                              precision=precision, ln_inv_s_init=0.5, aabb=torch.tensor([[-h, -h, -h], [h, h, h]]),
                              accel_cfg=dict(resolution=occ_res, num_steps=2, num_pts=npts), seed=seed + 3,
                              ray_query_cfg=dict(query_mode="march_occ_multi_upsample", query_param=qp)).to(device)
    vm.geometric_init_instances([0.35 + 0.03 * (b % 4) for b in range(B)])
    vm.accel.init(vm.query_sdf, generator=torch.Generator(device=device).manual_seed(seed))
    return vm


class ComposeTrainer:
    """One training iteration of the multi-object configuration: rays of the rig -> ``BufferComposeRenderer`` over the
    street background, the posed vehicle instances, distant model and sky -> l1 photometric + eikonal on every object's
    samples -> backward -> (gradient all-reduce) -> Adam on every model.  Mirrors the loop of code_multi/tools/train.py
    reduced to the hot path, like ``RenderTrainer`` does for code_single."""

    def __init__(self, street, vehicles, poses, distant, sky, intr, c2w, WH, images, num_rays: int, lr: float = 1e-3,
                 w_eikonal: float = 0.01, rank: int = 0, world_size: int = 1, seed: int = 42):
        from .optim import FusedAdam
        from .renderers.buffer_compose_renderer import BufferComposeRenderer, Drawable
        self.street, self.vehicles, self.distant, self.sky = street, vehicles, distant, sky
        self.intr, self.c2w, self.WH, self.images = intr, c2w, WH, images
        self.V, self.num_rays, self.w_eikonal = intr.shape[0], num_rays, w_eikonal
        self.rank, self.world_size = rank, world_size
        dev = street.device
        self.gen = torch.Generator(device=dev).manual_seed(seed + 1000 * rank)
        self.gen_shared = torch.Generator(device=dev).manual_seed(seed)
        g = torch.Generator().manual_seed(seed)
        self.appear = nn.Parameter((torch.randn(self.V, 4, generator=g) * 0.1).to(dev))
        self.drawables = [Drawable("street", "Street", street)] + \
            [Drawable(f"car{b}", "Vehicle", vehicles, rotation=R, translation=t, scale=s) for b, (R, t, s) in enumerate(poses)]
        self.renderer = BufferComposeRenderer(dict(with_rgb=True, with_normal=True, near=0.1, far=200.0, perturb=True,
                                                   depth_use_normalized_vw=False)).train()
        self.optim = FusedAdam(street, lr=lr, learn_inv_s=False)
        self.optim.groups.append(dict(p=self.appear, p16=None, betas=(0.9, 0.99), m=torch.zeros_like(self.appear),
                                      v=torch.zeros_like(self.appear)))
        self.optim.add_model(vehicles, learn_inv_s=False)
        if distant is not None:
            self.optim.add_distant_model(distant)
        if sky is not None:
            self.optim.add_sky_model(sky)
        self.skip_allreduce = False
        self._reducer = None
        self.stats: Dict[str, float] = {}

    @property
    def model(self):
        return self.street

    def sample_batch(self):
        dev, N = self.street.device, self.num_rays
        xy = torch.rand([N, 2], device=dev, generator=self.gen).clamp_(1e-6, 1 - 1e-6)
        fidx = torch.randint(0, self.V, [N], device=dev, generator=self.gen)
        H_, W_ = self.images.shape[1], self.images.shape[2]
        ix, iy = (xy[:, 0] * W_).long().clamp_(0, W_ - 1), (xy[:, 1] * H_).long().clamp_(0, H_ - 1)
        return xy, fidx, self.images[fidx, iy, ix]

    def render(self, xy, fidx, **kw):
        from .losses import embedding_lookup
        o, d = selected_rays(xy, fidx, self.intr, self.c2w, self.WH)
        ha = embedding_lookup(self.appear, fidx)
        return self.renderer(o, d, drawables=self.drawables, rays_h_appear=ha, sky_model=self.sky,
                             distant_model=self.distant, return_buffer=False, return_details=True, **kw)

    def loss(self, ret, gt):
        from .losses import eikonal_loss
        loss = (ret["rendered"]["rgb_volume"] - gt).abs().mean()
        n_s = 0
        for raw in ret["raw_per_obj_model"].values():
            vb = raw["volume_buffer"]
            if vb["type"] != "empty" and "nablas" in vb:
                loss = loss + self.w_eikonal * eikonal_loss(vb["nablas"])
                n_s += int(vb["nablas"].shape[0])
        return loss, n_s

    def train_step(self, it: int) -> torch.Tensor:
        from . import distributed as ndist
        st = self.street
        st.training_before_per_step(it)
        acc = st.accel
        if it >= acc.n_steps_warmup and it % acc.n_steps_between_update == 0:
            acc.update_from_net(st.query_sdf, generator=self.gen_shared)
        va = self.vehicles.accel
        if it >= va.n_steps_warmup and it % va.n_steps_between_update == 0:
            va.update_from_net(self.vehicles.query_sdf, generator=self.gen_shared)
        xy, fidx, gt = self.sample_batch()
        _lib.arena_begin(gt.device)      # zero-filled buffers of the step: one arena (_lib.zeros)
        ret = self.render(xy, fidx)
        loss, n_s = self.loss(ret, gt)
        self.optim.zero_grad()
        red = None
        if self.world_size > 1 and not self.skip_allreduce and os.environ.get("NSIM_OVERLAP_ALLREDUCE", "1") == "1":
            if self._reducer is None:       # the table gradients leave during the backward (ndist.BackwardReducer)
                self._reducer = ndist.BackwardReducer(self.optim.params())
            red = self._reducer
            red.begin()
        with backward_on_calling_thread():
            loss.backward()
        _lib.arena_end()
        if red is not None:
            red.reduce_and_step(self.optim, 1.0 / self.world_size)
        else:
            if not self.skip_allreduce:
                ndist.allreduce_grads(self.optim.params(), average=False)
            self.optim.step(grad_scale=1.0 if self.skip_allreduce else 1.0 / self.world_size)
        self.stats = dict(S_f=n_s, R_hit=int((ret["ray_intersections"]["samples_cnt"] > 0).sum()))
        return loss.detach()


def build_multi_trainer(device, rank: int = 0, world_size: int = 1, precision: str = "fp16", rays_per_gpu: int = 16384,
                        seed: int = 42, small: bool = False, B: int = 8):
    from . import distributed as ndist
    poses = vehicle_poses(B, device=device)
    world = multi_world(poses)
    street, dm, sm = street_models(device, precision, seed, small, street_world())
    street.accel.init(street.query_sdf, generator=torch.Generator(device=device).manual_seed(seed))
    vm = vehicle_model(device, B, precision, seed, small)
    for mod in (street, vm, dm, sm):
        ndist.broadcast_module(mod)
    hw = 24 if small else 800
    intr, c2w, WH = street_rig(n_ego=2 if small else 10, H=hw, W=hw, f=0.625 * hw, device=device)
    images = render_dataset(world, intr, c2w, WH)
    return ComposeTrainer(street, vm, poses, dm, sm, intr, c2w, WH, images, num_rays=rays_per_gpu, rank=rank,
                          world_size=world_size, seed=seed)
