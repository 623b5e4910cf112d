"""oracle.render -- CPU restatement of the sampling + compositing half of the NeuS render step.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED for everything that lived in nr3d_lib
(ray test, occupancy marching, up-sampling, sdf->alpha); pinned by reference code only for ray
generation and volume integration.  Follows:
* ray generation: app/resources/observers/cameras.py:281-310 (snap to pixel centre, pinhole lift,
  rotate by c2w with broadcast-multiply-sum -- NOT mm/bmm, cameras.py:355-359 --, normalise);
* ray test: call site app/renderers/single_volume_renderer.py:235-238 (keys :289-300);
* marching / sampling config: code_single/configs/object_centric/lotd_neus.dtu.230814.yaml:140-173
  (occ grid 64^3, occ_val_fn sdf inv_s 256, thre 0.3, ema 0.95; num_coarse 64, num_fine [8,8,32],
  upsample_inv_s 64 x [1,4,16], step_size .005, max_steps 4096, upsample_use_estimate_alpha true);
* up-sampling: NeuS (arXiv 2106.10689) ``up_sample`` + NeRF ``sample_pdf`` (deterministic u);
* volume integration: app/renderers/single_volume_renderer.py:73-102.

Conventions fixed here (mirrored by the HIP kernels):
* marched samples live on a per-ray lattice ``t_k = near + (k + jitter) * step`` and are kept iff the
  voxel containing ``o + t_k d`` is occupied ("skipping empty space" never changes the sample set);
* the with-grad query evaluates all n samples of a ray; ``opacity_alpha[i]`` belongs to the interval
  (i, i+1) and the LAST sample of every ray has alpha = 0 (so packs keep their size);
* every op order below is the one the kernels use, compiled with -ffp-contract=off, so the discrete
  decisions (voxel membership, counts) are bit-reproducible.
"""
import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import pack_ops as po
from .field import FieldParams, forward_sdf, forward_field, forward_sdf_nablas


# ------------------------------------------------------------------------------------- ray gen
def opencv_distort(x, y, dist):
    """The OpenCV radial-tangential model (k1, k2, p1, p2, k3) on normalised camera coordinates: undistorted -> distorted
    (the calibration Waymo ships, dataio/autonomous_driving/waymo/preprocess.py:172)."""
    k1, k2, p1, p2, k3 = dist.unbind(-1)
    r2 = x * x + y * y
    rad = 1.0 + ((k3 * r2 + k2) * r2 + k1) * r2
    return (x * rad + 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x),
            y * rad + p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y)


def opencv_undistort(x0, y0, dist, n_iters: int = 5):
    """distorted -> undistorted normalised coordinates by the fixed-point iteration of cv::undistortPoints (n_iters rounds,
    5 in OpenCV) -- the ``lift`` of ``camera_model: opencv`` (cameras.py:84-87).  nr3d_lib's OpenCVCameraMatHW is absent:
    parity unpinned, semantics fixed here; same operation order as the kernel (csrc/sampling.hip raygen_lift)."""
    k1, k2, p1, p2, k3 = dist.unbind(-1)
    x, y = x0, y0
    for _ in range(n_iters):
        r2 = x * x + y * y
        icd = 1.0 / (1.0 + ((k3 * r2 + k2) * r2 + k1) * r2)
        dx = 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x)
        dy = p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y
        x, y = (x0 - dx) * icd, (y0 - dy) * icd
    return x, y


def fisheye_distort(a, b, dist):
    """The OpenCV fisheye (Kannala-Brandt equidistant) model on normalised camera coordinates (a, b) = (x / z, y / z):
    theta = atan(r), theta_d = theta (1 + k1 theta^2 + k2 theta^4 + k3 theta^6 + k4 theta^8), (x_d, y_d) = theta_d / r (a, b)
    -- the reference's own formulas, app/resources/observers/fisheye.py:31-42."""
    k1, k2, k3, k4 = dist.unbind(-1)
    r = torch.sqrt(a * a + b * b)
    th = torch.atan(r)
    t2 = th * th
    thd = th * (1.0 + (((k4 * t2 + k3) * t2 + k2) * t2 + k1) * t2)
    sc = torch.where(r > 1e-8, thd / r.clamp_min(1e-12), torch.ones_like(r))
    return a * sc, b * sc


def fisheye_lift(xd, yd, dist, n_iters: int = 10):
    """distorted normalised coordinates -> direction in the camera frame (``lift`` of ``camera_model: fisheye``,
    cameras.py:88-92): theta from ``n_iters`` Newton rounds on theta_d = theta (1 + k1 theta^2 + ...) starting at theta_d
    (cv::fisheye::undistortPoints runs 10), direction (sin theta x_d / theta_d, sin theta y_d / theta_d, cos theta).
    nr3d_lib's FisheyeCameraMatHW is absent: parity unpinned, semantics fixed here; the kernel's operation order
    (csrc/sampling.hip raygen_lift_fisheye)."""
    k1, k2, k3, k4 = dist.unbind(-1)
    td = torch.sqrt(xd * xd + yd * yd)
    th = td
    for _ in range(n_iters):
        t2 = th * th
        f = th * (1.0 + (((k4 * t2 + k3) * t2 + k2) * t2 + k1) * t2) - td
        fp = 1.0 + (((9.0 * k4 * t2 + 7.0 * k3) * t2 + 5.0 * k2) * t2 + 3.0 * k1) * t2
        th = th - f / fp
    sc = torch.where(td > 1e-8, torch.sin(th) / td.clamp_min(1e-12), torch.ones_like(td))
    return xd * sc, yd * sc, torch.cos(th)


def pinhole_rays(xy: torch.Tensor, fidx: torch.Tensor, intr: torch.Tensor, c2w: torch.Tensor,
                 WH: torch.Tensor, snap_to_pixel_centers: bool = True, distortion: torch.Tensor = None, n_iters: int = 5,
                 camera_model: str = "opencv"):
    """xy [N,2] in [0,1], fidx [N] frame index, intr [V,3,3], c2w [V,4,4] (OpenCV), WH [V,2] (W,H)
    -> rays_o, rays_d [N,3].  cameras.py:281-310.  distortion [V,5]: the OpenCV camera model (``opencv_undistort``);
    [V,4] with ``camera_model='fisheye'``: ``fisheye_lift``."""
    wh_i = WH[fidx]
    if snap_to_pixel_centers:
        wh = (xy * wh_i).long().clamp(torch.zeros_like(wh_i), wh_i - 1).to(xy.dtype) + 0.5
    else:
        wh = xy * wh_i
    K = intr[fidx]
    fx, fy, cx, cy = K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2]
    dx = (wh[:, 0] - cx) / fx
    dy = (wh[:, 1] - cy) / fy
    dz = torch.ones_like(dx)
    if distortion is not None and camera_model == "fisheye":
        dx, dy, dz = fisheye_lift(dx, dy, distortion[fidx], n_iters)
    elif distortion is not None:
        dx, dy = opencv_undistort(dx, dy, distortion[fidx], n_iters)
    dirs = torch.stack([dx, dy, dz], dim=-1)
    R = c2w[fidx, :3, :3]
    rays_d = (R * dirs.unsqueeze(-2)).sum(-1)
    rays_d = rays_d / rays_d.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    rays_o = c2w[fidx, :3, 3]
    return rays_o, rays_d


# ------------------------------------------------------------------------------------ ray test
def aabb_ray_test(rays_o, rays_d, aabb_min, aabb_max, near: float, far: Optional[float]):
    """Slab test -> (near [N], far [N], hit [N] bool)."""
    d = rays_d
    tiny = 1e-12
    d_safe = torch.where(d.abs() < tiny, torch.where(d < 0, -torch.full_like(d, tiny), torch.full_like(d, tiny)), d)
    inv = 1.0 / d_safe
    t1 = (aabb_min - rays_o) * inv
    t2 = (aabb_max - rays_o) * inv
    tmin = torch.minimum(t1, t2).max(dim=-1).values
    tmax = torch.maximum(t1, t2).min(dim=-1).values
    n = torch.clamp_min(tmin, near if near is not None else 0.0)
    f = tmax if far is None else torch.clamp_max(tmax, far)
    hit = f > n
    return n, f, hit


# ------------------------------------------------------------------------------------ occ grid
def occ_val_from_sdf(sdf: torch.Tensor, inv_s: float = 256.0):
    """``occ_val_fn_cfg{type: sdf, inv_s: 256}``: 4 sig(s x)(1 - sig(s x)) (peak-normalised logistic
    density; 0.27 at |sdf| = 0.01 -- config comment '+- 0.01 sdf @ 0.3 thre')."""
    s = torch.sigmoid(sdf * inv_s)
    return 4.0 * s * (1.0 - s)


def voxel_index(p, aabb_min, scale, res):
    """p [S,3] -> (flat voxel index [S], inside [S]); g = floor((p - min) * scale), scale = res/(max-min)."""
    g = torch.floor((p - aabb_min) * scale).long()
    inside = ((g >= 0) & (g < res)).all(dim=-1)
    gc = g.clamp(torch.zeros_like(res), res - 1)
    flat = gc[:, 0] + res[0] * (gc[:, 1] + res[1] * gc[:, 2])
    return flat, inside


def occ_update(occ_val: torch.Tensor, pts, sdf, aabb_min, scale, res, decay=0.95, inv_s=256.0):
    """EMA-max update: val = max(val * decay, f(sdf)) scattered at the voxels of ``pts``."""
    flat, inside = voxel_index(pts, aabb_min, scale, res)
    new = occ_val_from_sdf(sdf, inv_s)
    out = occ_val * decay
    out = out.scatter_reduce(0, flat[inside], new[inside], reduce='amax', include_self=True)
    return out


def occ_collect(occ_val: torch.Tensor, pts, sdf, aabb_min, scale, res, inv_s=256.0):
    """``update_from_samples_cfg: {}`` (lotd_neus.dtu.230814.yaml:158): the samples of a training step's sampling pass
    folded into the value grid -- ``occ_update`` without the decay (the periodic refresh decays).  Semantics fixed here
    (nr3d_lib absent): PARITY UNPINNED."""
    return occ_update(occ_val, pts, sdf, aabb_min, scale, res, decay=1.0, inv_s=inv_s)


def build_occ_grid(p: FieldParams, aabb_min, aabb_max, res, n_pts=2 ** 18, n_steps=4, thre=0.3,
                   inv_s=256.0, seed=0):
    """``init_cfg{mode: from_net}``: EMA grid from random SDF queries -> (val f32 [res^3], occ bool)."""
    res_t = torch.tensor(res, dtype=torch.long)
    scale = res_t.float() / (aabb_max - aabb_min)
    val = torch.zeros(int(res_t.prod()))
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for _ in range(n_steps):
            pts = aabb_min + torch.rand(n_pts, 3, generator=g) * (aabb_max - aabb_min)
            sdf = forward_sdf(pts, p)
            val = occ_update(val, pts, sdf, aabb_min, scale, res_t, decay=0.95, inv_s=inv_s)
    return val, val > thre


# ------------------------------------------------------------------------------------ sampling
def march_lattice(rays_o, rays_d, near, far, jitter, occ: torch.Tensor, aabb_min, scale, res,
                  step: float, max_steps: int):
    """-> (t [M], ridx [M], counts [R]) of occupied lattice samples (ascending inside each ray)."""
    R = rays_o.shape[0]
    K = torch.ceil((far - near) / step).clamp(0, max_steps).long()
    Kmax = int(K.max()) if R > 0 else 0
    k = torch.arange(Kmax, dtype=torch.float32)
    t = near[:, None] + (k[None, :] + jitter[:, None]) * step                       # [R,Kmax]
    valid = (torch.arange(Kmax)[None, :] < K[:, None]) & (t < far[:, None])
    pts = rays_o[:, None, :] + t[..., None] * rays_d[:, None, :]
    flat, inside = voxel_index(pts.reshape(-1, 3), aabb_min, scale, res)
    keep = valid & (inside & occ[flat]).view(R, Kmax)
    counts = keep.sum(dim=1)
    ridx = torch.nonzero(keep)[:, 0]
    return t[keep], ridx, counts


def coarse_depths(near, far, num_coarse: int, jitter_c: Optional[torch.Tensor]):
    """``coarse_step_cfg{step_mode: linear}``: t_i = near + (far-near) * ((i + u_i)/C); u = 0.5 w/o perturb."""
    i = torch.arange(num_coarse, dtype=torch.float32)
    u = jitter_c if jitter_c is not None else torch.full((near.shape[0], num_coarse), 0.5)
    return near[:, None] + (far - near)[:, None] * ((i[None, :] + u) / float(num_coarse))


def merge_sorted(t_a, pi_a, t_b_batched):
    """Merge per-ray ascending packed ``t_a`` with ascending batched ``t_b`` [R,nb]; a-first on ties.
    -> (t [S], pack_infos [R,2], pos_a [Sa], pos_b [R,nb])."""
    R, nb = t_b_batched.shape
    pi_b = po.get_pack_infos_from_n(torch.full((R,), nb, dtype=torch.long))
    rays = torch.arange(R)
    pidx_a, pidx_b, pi = po.merge_two_packs_sorted(t_a, pi_a, rays, t_b_batched.reshape(-1), pi_b, rays)
    t = torch.empty(t_a.shape[0] + R * nb, dtype=t_a.dtype)
    t[pidx_a] = t_a
    t[pidx_b] = t_b_batched.reshape(-1)
    return t, pi, pidx_a, pidx_b.view(R, nb)


def neus_alpha_packed(sdf, pack_infos, inv_s):
    """opacity of interval (i,i+1) stored at i; 0 at the last sample of every pack.
    alpha = clamp((Phi(s sdf_i) - Phi(s sdf_{i+1}) + 1e-5) / (Phi(s sdf_i) + 1e-5), 0, 1)  (NeuS eq. 13)."""
    S = sdf.shape[0]
    ridx = po.pack_ridx(pack_infos, S)
    last = torch.zeros(S, dtype=torch.bool)
    ends = pack_infos[:, 0] + pack_infos[:, 1] - 1
    last[ends[pack_infos[:, 1] > 0]] = True
    nxt = torch.cat([sdf[1:], sdf[-1:]])
    c0 = torch.sigmoid(sdf * inv_s)
    c1 = torch.sigmoid(nxt * inv_s)
    alpha = ((c0 - c1 + 1e-5) / (c0 + 1e-5)).clamp(0.0, 1.0)
    return torch.where(last, torch.zeros_like(alpha), alpha)


def upsample_stage(t, sdf, pack_infos, inv_s: float, n_fine: int, use_estimate_alpha: bool = True):
    """One NeuS up-sampling stage on packed (t, sdf) -> new depths [R, n_fine] (ascending).
    Deterministic u_k = (k + 0.5)/n_fine."""
    tp, mask, _ = po.to_padded(t, pack_infos, fill=0.0)
    sp, _, _ = po.to_padded(sdf, pack_infos, fill=0.0)
    R, n = tp.shape
    cnt = pack_infos[:, 1]
    iv_valid = torch.arange(n - 1)[None, :] < (cnt - 1)[:, None]          # interval i valid
    t0, t1 = tp[:, :-1], tp[:, 1:]
    s0, s1 = sp[:, :-1], sp[:, 1:]
    if use_estimate_alpha:
        mid = (s0 + s1) * 0.5
        dist = t1 - t0
        cos = (s1 - s0) / (dist + 1e-5)
        prev = torch.cat([torch.zeros_like(cos[:, :1]), cos[:, :-1]], dim=1)
        cos = torch.minimum(prev, cos).clamp(-1e3, 0.0)
        half = cos * dist * 0.5
        pc = torch.sigmoid((mid - half) * inv_s)
        nc = torch.sigmoid((mid + half) * inv_s)
        alpha = (pc - nc + 1e-5) / (pc + 1e-5)
    else:
        c0 = torch.sigmoid(s0 * inv_s)
        c1 = torch.sigmoid(s1 * inv_s)
        alpha = ((c0 - c1 + 1e-5) / (c0 + 1e-5)).clamp(0.0, 1.0)
    alpha = torch.where(iv_valid, alpha, torch.zeros_like(alpha))
    trans = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha[:, :-1] + 1e-7], dim=1), dim=1)
    w = torch.where(iv_valid, alpha * trans + 1e-5, torch.zeros_like(alpha))
    wsum = w.sum(dim=1, keepdim=True)
    cdf = torch.cumsum(w, dim=1) / wsum                                  # cdf after interval i
    cdf = torch.cat([torch.zeros_like(cdf[:, :1]), cdf], dim=1)          # [R, n], cdf[0] = 0
    u = ((torch.arange(n_fine, dtype=torch.float32) + 0.5) / n_fine)[None, :].expand(R, n_fine).contiguous()
    # first interval i with cdf[i+1] > u, restricted to valid intervals
    inds = torch.searchsorted(cdf[:, 1:].contiguous(), u, right=True)    # in [0, n-1]
    inds = torch.minimum(inds, (cnt - 2).clamp_min(0)[:, None])
    c_lo = torch.gather(cdf, 1, inds)
    c_hi = torch.gather(cdf, 1, inds + 1)
    b_lo = torch.gather(tp, 1, inds)
    b_hi = torch.gather(tp, 1, (inds + 1).clamp_max(n - 1))
    den = c_hi - c_lo
    den = torch.where(den < 1e-5, torch.ones_like(den), den)
    frac = ((u - c_lo) / den).clamp(0.0, 1.0)
    return b_lo + frac * (b_hi - b_lo)


# ----------------------------------------------------------------------------- volume integration
def volume_integration(alpha, t, rgb, nablas, pack_infos, depth_use_normalized_vw=False):
    """single_volume_renderer.py:73-102 on the hit rays -> dict(vw, mask, depth, rgb, normals)."""
    vw = po.packed_alpha_to_vw(alpha, pack_infos)
    mask = po.packed_sum(vw, pack_infos)
    depth_w = po.packed_div(vw, mask + 1e-10, pack_infos) if depth_use_normalized_vw else vw
    out = dict(vw=vw, mask_volume=mask, depth_volume=po.packed_sum(depth_w * t, pack_infos))
    if rgb is not None:
        out['rgb_volume'] = po.packed_sum(vw[:, None] * rgb, pack_infos)
    if nablas is not None:
        out['normals_volume'] = po.packed_sum(vw[:, None] * nablas, pack_infos)
    return out


# ------------------------------------------------------------------------------------ ray query
def ray_query(p: FieldParams, rays_o, rays_d, h_appear, occ, aabb_min, aabb_max, res, *,
              near=0.01, far=None, num_coarse=64, num_fine=(8, 8, 32), upsample_inv_s=64.0,
              upsample_inv_s_factors=(1, 4, 16), step_size=0.005, max_steps=4096,
              use_estimate_alpha=True, jitter=None, jitter_c=None, forward_inv_s=None,
              depth_use_normalized_vw=False, sdf_fn=None, compress=False, compress_thre=1e-4,
              upsample_on_marched_only=True) -> Dict:
    """``query_mode = march_occ_multi_upsample`` on N rays (already in object space, AABB-normalised).
    Returns the volume buffer + per-hit-ray renderings.  ``jitter`` [N] / ``jitter_c`` [N,C] carry the
    perturbation randoms so that the HIP path can consume the identical numbers.

    ``upsample_on_marched_only`` (``ray_query_cfg.query_param``; default True): coarse depths and the up-sampling stages run
    on the rays whose occupancy march found occupied voxels only; the volume buffer lists those rays -- ``rays_inds_hit`` /
    ``pack_infos_hit`` are the [R'] subset of the R AABB-tested rays that the reference's buffers carry
    (single_volume_renderer.py:209-220,289-300: the renderer scatters ``rays_inds_hit`` rows into images of all rays; an
    object whose march found nothing returns ``type: 'empty'``).  ``rendered`` / ``debug`` stay per TESTED ray (rows of
    rays without samples are empty packs).  False: every tested ray gets num_coarse + sum(num_fine) samples (rounds 1-4)."""
    N = rays_o.shape[0]
    res_t = torch.tensor(res, dtype=torch.long)
    scale = res_t.float() / (aabb_max - aabb_min)
    near_t, far_t, hit = aabb_ray_test(rays_o, rays_d, aabb_min, aabb_max, near, far)
    rays_inds = torch.nonzero(hit)[:, 0]
    R = rays_inds.shape[0]
    ret = dict(num_rays=R, rays_inds=rays_inds)
    if R == 0:
        ret['volume_buffer'] = dict(type='empty')
        return ret
    o, d, nr, fr = rays_o[rays_inds], rays_d[rays_inds], near_t[rays_inds], far_t[rays_inds]
    jit = jitter[rays_inds] if jitter is not None else torch.full((R,), 0.5)
    jc = jitter_c[rays_inds] if jitter_c is not None else None
    query_sdf = sdf_fn if sdf_fn is not None else (lambda x: forward_sdf(x, p))
    def sample(o_, d_, nr_, fr_, jc_, t_m_, cnt_m_):
        """coarse depths + up-sampling stages of the given rays (their marched depths t_m_ packed by cnt_m_)"""
        R_ = o_.shape[0]
        pi_m_ = po.get_pack_infos_from_n(cnt_m_)
        t_c = coarse_depths(nr_, fr_, num_coarse, jc_)
        t_, pi_, _, _ = merge_sorted(t_m_, pi_m_, t_c)
        ridx_ = po.pack_ridx(pi_, t_.shape[0])
        sdf_ = query_sdf(o_[ridx_] + t_[:, None] * d_[ridx_])
        for nf, fac in zip(num_fine, upsample_inv_s_factors):
            t_new = upsample_stage(t_, sdf_, pi_, upsample_inv_s * fac, nf, use_estimate_alpha)
            ridx_new = torch.arange(R_).repeat_interleave(nf)
            sdf_new = query_sdf(o_[ridx_new] + t_new.reshape(-1, 1) * d_[ridx_new])
            t_old, sdf_old = t_, sdf_
            t_, pi_, pa, pb = merge_sorted(t_old, pi_, t_new)
            sdf_ = torch.empty_like(t_)
            sdf_[pa] = sdf_old
            sdf_[pb.reshape(-1)] = sdf_new
        return t_, sdf_, pi_

    live = torch.ones(R, dtype=torch.bool)
    with torch.no_grad():
        t_m, ridx_m, cnt_m = march_lattice(o, d, nr, fr, jit, occ, aabb_min, scale, res_t, step_size, max_steps)
        if upsample_on_marched_only:
            # the rays whose march found something, sampled on their own; rays are independent, so this IS the dense pass
            # restricted to them.  Pack infos go back to all R tested rays (the others: empty packs)
            live = cnt_m > 0
            if bool(live.any()):
                t, sdf, pi_l = sample(o[live], d[live], nr[live], fr[live], jc[live] if jc is not None else None, t_m, cnt_m[live])
                cnt_all = torch.zeros(R, dtype=torch.long)
                cnt_all[live] = pi_l[:, 1]
                pi = po.get_pack_infos_from_n(cnt_all)
            else:
                t, sdf, pi = torch.zeros(0), torch.zeros(0), po.get_pack_infos_from_n(torch.zeros(R, dtype=torch.long))
        else:
            t, sdf, pi = sample(o, d, nr, fr, jc, t_m, cnt_m)
        ridx = po.pack_ridx(pi, t.shape[0])
    ret['debug'] = dict(t=t, sdf_nograd=sdf, pack_infos=pi, march_counts=cnt_m, ridx=ridx, x_nograd=o[ridx] + t[:, None] * d[ridx],
                        live=live)
    if t.shape[0] == 0:
        ret['volume_buffer'] = dict(type='empty')
        return ret
    if compress:
        # ``march_occ_multi_upsample_compressed``: keep the samples that bound an interval with vw > thre
        with torch.no_grad():
            inv_s_c = p.inv_s().detach() if forward_inv_s is None else torch.as_tensor(float(forward_inv_s))
            vw = po.packed_alpha_to_vw(neus_alpha_packed(sdf, pi, inv_s_c), pi)
            sig = vw > compress_thre
            first = torch.zeros_like(sig)
            first[pi[:, 0][pi[:, 1] > 0]] = True
            prev_sig = torch.cat([sig[:1] & False, sig[:-1]]) & ~first
            keep = sig | prev_sig
            cnt_k = torch.zeros(R, dtype=torch.long).index_add_(0, ridx[keep], torch.ones(int(keep.sum()), dtype=torch.long))
            t, ridx, pi = t[keep], ridx[keep], po.get_pack_infos_from_n(cnt_k)
        ret['debug']['compress_counts'] = cnt_k
    x = o[ridx] + t[:, None] * d[ridx]
    v = d[ridx]
    ha = h_appear[rays_inds][ridx] if h_appear is not None else torch.zeros(x.shape[0], 4)
    sdf_g, nablas, rgb = forward_field(x, v, ha, p, x_has_grad=x.requires_grad)   # rays with grad: pose refinement
    inv_s = p.inv_s() if forward_inv_s is None else torch.as_tensor(float(forward_inv_s))
    alpha = neus_alpha_packed(sdf_g, pi, inv_s)
    ret['volume_buffer'] = dict(type='packed', rays_inds_hit=rays_inds[live], pack_infos_hit=pi[live], t=t,
                                opacity_alpha=alpha, rgb=rgb, nablas=nablas, sdf=sdf_g, net_x=x)
    ret['rendered'] = volume_integration(alpha, t, rgb, nablas, pi, depth_use_normalized_vw)       # per TESTED ray
    ret['pack_infos_tested'] = pi           # the final sample set packed per TESTED ray (rows of rays without samples: n = 0)
    ret['near'], ret['far'] = nr, fr
    return ret


def render_loss(ret: Dict, gt_rgb: torch.Tensor, N: int, w_eikonal: float = 0.1):
    """photometric mse on all N rays (app/loss/photometric.py:88-146, ``rgb_fn_type: mse``) + eikonal
    (|nablas|-1)^2 averaged over the render samples (app/loss/eikonal.py:216-251)."""
    rgb_full = torch.zeros(N, 3)
    if ret['num_rays'] > 0:
        rgb_full = rgb_full.index_put((ret['rays_inds'],), ret['rendered']['rgb_volume'])
    loss_rgb = ((rgb_full - gt_rgb) ** 2).mean()
    if ret['num_rays'] > 0:
        nab = ret['volume_buffer']['nablas']
        loss_eik = ((nab.norm(dim=-1) - 1.0) ** 2).mean()
    else:
        loss_eik = torch.zeros(())
    return loss_rgb + w_eikonal * loss_eik, dict(loss_rgb=loss_rgb, loss_eikonal=loss_eik)


def convert_rays_in_node(rays_o, rays_d, rotation, translation, scale=1.0):
    """World -> object rays ``R^-1 (o - t) / s``, ``R^-1 d / s`` (app/resources/scenes.py:686-708)."""
    Rt = rotation.transpose(-1, -2)
    return ((rays_o - translation).unsqueeze(-2) * Rt).sum(-1) / scale, (rays_d.unsqueeze(-2) * Rt).sum(-1) / scale


def compose_buffers(buffers, N: int, depth_use_normalized_vw: bool = True):
    """Joint rendering of several objects' packed volume buffers (app/renderers/buffer_compose_renderer.py:644-718),
    restated ray by ray: all samples of a ray, from every object, sorted by depth and composited.
    ``buffers``: dicts with rays_inds [R], pack_infos [R,2], t [S], alpha [S], rgb [S,3] (several entries of one buffer
    may name the same ray: batch items).  -> mask [N], depth [N], rgb [N,3], samples per ray [N]."""
    per_ray = [[] for _ in range(N)]
    for b in buffers:
        for k in range(b['rays_inds'].shape[0]):
            st, n = int(b['pack_infos'][k, 0]), int(b['pack_infos'][k, 1])
            if n > 0:
                per_ray[int(b['rays_inds'][k])].append((b['t'][st:st + n], b['alpha'][st:st + n], b['rgb'][st:st + n]))
    mask, depth, rgb = torch.zeros(N), torch.zeros(N), torch.zeros(N, 3)
    cnt = torch.zeros(N, dtype=torch.long)
    masks, depths, rgbs = [], [], []
    for r in range(N):
        if not per_ray[r]:
            masks.append(torch.zeros(())); depths.append(torch.zeros(())); rgbs.append(torch.zeros(3))
            continue
        t = torch.cat([c[0] for c in per_ray[r]])
        a = torch.cat([c[1] for c in per_ray[r]])
        c = torch.cat([c[2] for c in per_ray[r]])
        order = torch.sort(t.detach(), stable=True).indices
        t, a, c = t[order], a[order], c[order]
        cnt[r] = t.shape[0]
        trans = torch.cumprod(torch.cat([torch.ones(1), 1.0 - a + 1e-10])[:-1], dim=0)
        vw = a * trans
        m = vw.sum()
        masks.append(m)
        depths.append(((vw / (m + 1e-10)) * t).sum() if depth_use_normalized_vw else (vw * t).sum())
        rgbs.append((vw[:, None] * c).sum(0))
    return torch.stack(masks), torch.stack(depths), torch.stack(rgbs), cnt


# ------------------------------------------------------------------------- single scene: close range + distant + sky
def render_scene(p: FieldParams, rays_o, rays_d, h_appear, occ, aabb_min, aabb_max, res, *, query_kw: dict,
                 distant=None, distant_kw: dict = None, sky=None, depth_use_normalized_vw=False, with_normal=True):
    """``SingleVolumeRenderer.ray_query`` (app/renderers/single_volume_renderer.py:136-492) restated for one close-range
    object: the NeuS query on the rays that hit its AABB (:235-267); the distant NeRF++ model on ALL rays with ``near`` :=
    the close-range ``far`` where the AABB was hit (:281-309); both buffers merged per ray by depth
    (``merge_two_packs_sorted`` + scatter, :337-375) and integrated (:73-102, :412-442); the sky blended with the residual
    transmittance (:449-457).
    ``distant`` = oracle.distant.DistantParams or None, ``distant_kw`` = dict(K, jitter, include_inf, r_min, r_max);
    ``sky`` = (ws, bs) of oracle.sky or None.  -> dict(rendered, cr=<ray_query result>, dv=<distant buffer>)."""
    from . import distant as od, sky as osky
    N = rays_o.shape[0]
    cr = ray_query(p, rays_o, rays_d, h_appear, occ, aabb_min, aabb_max, res,
                   depth_use_normalized_vw=depth_use_normalized_vw, **query_kw)
    near = query_kw.get("near", 0.01)
    ri = cr["rays_inds"]
    dv = None
    if distant is not None:
        kw = dict(distant_kw or {})
        near_dv = torch.full([N], float(near))
        if cr["num_rays"] > 0:
            near_dv = near_dv.index_put((ri,), cr["far"])
        dv = od.distant_ray_query(distant, rays_o.detach(), rays_d.detach(), near_dv, h_appear, aabb_min, aabb_max, **kw)
    has_cr = cr["num_rays"] > 0 and cr["volume_buffer"]["type"] != "empty"
    rendered = dict(mask_volume=torch.zeros(N), depth_volume=torch.zeros(N), rgb_volume=torch.zeros(N, 3))
    if with_normal:
        rendered["normals_volume"] = torch.zeros(N, 3)
    if has_cr and dv is not None:
        vb = cr["volume_buffer"]
        K = dv["t"].shape[1]
        pi_dv = po.get_pack_infos_from_n(torch.full((N,), K, dtype=torch.long))
        pidx_dv, pidx_cr, pi_tot = po.merge_two_packs_sorted(dv["t"].flatten(), pi_dv, torch.arange(N), vb["t"],
                                                              vb["pack_infos_hit"], vb["rays_inds_hit"])
        S = N * K + vb["t"].shape[0]

        def place(a_dv, a_cr, tail=()):
            z = torch.zeros([S, *tail])
            if a_dv is not None:
                z = z.index_put((pidx_dv,), a_dv)
            return z.index_put((pidx_cr,), a_cr)
        tt = place(dv["t"].flatten(), vb["t"])
        aa = place(dv["opacity_alpha"].flatten(), vb["opacity_alpha"])
        cc = place(dv["rgb"].flatten(0, 1), vb["rgb"], (3,))
        nn = place(None, vb["nablas"], (3,)) if with_normal else None
        out = volume_integration(aa, tt, cc, nn, pi_tot, depth_use_normalized_vw)
        for k in rendered:
            rendered[k] = out[k]
        cr["volume_buffer"]["vw_in_total"] = out["vw"][pidx_cr]
        dv["vw_in_total"] = out["vw"][pidx_dv].view(N, K)
    elif has_cr:
        for k in rendered:
            rendered[k] = rendered[k].index_put((ri,), cr["rendered"][k])
    elif dv is not None:
        K = dv["t"].shape[1]
        pi_dv = po.get_pack_infos_from_n(torch.full((N,), K, dtype=torch.long))
        out = volume_integration(dv["opacity_alpha"].flatten(), dv["t"].flatten(), dv["rgb"].flatten(0, 1), None, pi_dv,
                                 depth_use_normalized_vw)
        for k in ("mask_volume", "depth_volume", "rgb_volume"):
            rendered[k] = out[k]
    rendered["rgb_volume_occupied"] = rendered["rgb_volume"]
    if sky is not None:
        ws, bs = sky
        rgb_sky = osky.sky_forward(F_normalize(rays_d), h_appear, ws, bs)
        rendered["rgb_sky"] = rgb_sky
        rendered["rgb_volume"] = osky.blend_sky(rendered["rgb_volume"], rendered["mask_volume"], rgb_sky)
    return dict(rendered=rendered, cr=cr, dv=dv)


def F_normalize(v):
    return torch.nn.functional.normalize(v, dim=-1)
