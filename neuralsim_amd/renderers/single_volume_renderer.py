"""Single-scene volume renderer -- mirrors ``app/renderers/single_volume_renderer.py`` of the reference for the
close-range ("cr") object, which is the part of that file that sits on the hot path:

* ``ray_query``  (reference :136-492): ``model.ray_test`` -> ``model.ray_query`` -> volume integration of the returned
  packed buffer into per-ray ``mask_volume / depth_volume / rgb_volume / normals_volume`` scattered to all N rays
  (``prepare_empty_rendered``, app/renderers/utils.py:30-43), ``ray_intersections.samples_cnt``, the in-place
  additions to the volume buffer (``vw``, ``vw_in_total``, ``rays_inds_collect``, ``pack_infos_collect``, :416-442);
* ``render`` (reference :495-581): flattening of ray batches, ``rayschunk`` batching in eval mode
  (``batchify_query``, :553-565), training/eval grad mode, normalised normals in eval (:99-101).

* the distant NeRF++ model (reference :281-375): queried on ALL rays with ``near`` := the close-range ``far`` on the
  rays that hit the AABB, pose gradients detached, its batched buffer merged with the close-range packed buffer by
  ``merge_two_packs_sorted`` and scattered into the total buffers.

* the sky blend (reference :447-457): ``rgb_volume + (1 - mask_volume) * sky(v, h_appear)``.

There is no Scene graph here: the model is passed directly (the reference looks it up through
``scene.get_drawable_groups_by_class_name``), rays are expected in the model's object space.
"""
import os
from typing import Callable, Dict, List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..fields.neus import LoTDNeuSModel, volume_integration
from ..graphics import pack_ops as po


def batchify_query(fn: Callable, *args: torch.Tensor, chunk: int, dim_batchify: int = 0, show_progress: bool = False):
    """``nr3d_lib.models.utils.batchify_query`` (reference call site single_volume_renderer.py:565): run ``fn`` on
    chunks of the leading dimension and concatenate (nested dicts of) tensors."""
    N = args[0].shape[dim_batchify]
    outs = []
    for i in range(0, N, chunk):
        outs.append(fn(*[a[i:i + chunk] if isinstance(a, torch.Tensor) else a for a in args]))

    def cat(items):
        first = items[0]
        if isinstance(first, dict):
            return {k: cat([it[k] for it in items]) for k in first}
        if isinstance(first, torch.Tensor):
            return torch.cat(items, dim=dim_batchify)
        return first
    return cat(outs)


def prepare_empty_rendered(prefix, device, with_rgb=True, with_normal=True):
    """app/renderers/utils.py:30-43 -- zero images for every requested channel (two memsets: the scalar channels are
    rows of one buffer, the vector channels column blocks of another)."""
    sc = torch.zeros([2, *prefix], dtype=torch.float32, device=device)
    r = dict(mask_volume=sc[0], depth_volume=sc[1])
    nv = int(bool(with_rgb)) + int(bool(with_normal))
    if nv:
        vec = torch.zeros([*prefix, 3 * nv], dtype=torch.float32, device=device)
        if with_rgb:
            r["rgb_volume"] = vec[..., :3]
        if with_normal:
            r["normals_volume"] = vec[..., 3 * (nv - 1):]
    return r


class SingleVolumeRenderer(nn.Module):
    def __init__(self, config: Optional[dict] = None):
        super().__init__()
        self.config = dict(config or {})
        self.image_keys = ["depth_volume", "mask_volume", "rgb_volume", "normals_volume", "rgb_sky",
                           "rgb_volume_occupied", "rgb_volume_non_occupied"]

    def forward(self, *args, **kwargs):
        return self.ray_query(*args, **kwargs)

    def ray_query(self, rays_o: torch.Tensor, rays_d: torch.Tensor, rays_ts: torch.Tensor = None,
                  rays_pix: torch.Tensor = None, *, model: LoTDNeuSModel, rays_h_appear: torch.Tensor = None,
                  near=None, far=None, with_rgb: bool = None, with_normal: bool = None, return_buffer=False,
                  return_details=False, render_per_obj_individual=False, bypass_ray_query_cfg: dict = None,
                  distant_model=None, sky_model=None, with_env: bool = None, cr_ray_tested: dict = None,
                  world_transform=None) -> Dict:
        """``cr_ray_tested``: a ``model.ray_test`` result computed ahead of time for exactly these rays (the trainer
        prefetches the next batch's AABB test while it waits on the current batch's sample count).
        ``world_transform`` = (rotation [3,3] or per-ray [N,3,3], translation [3] / [N,3], scale): the object -> world
        pose of the main object's scene node; rays are brought into the object frame before the queries
        (``scene.convert_rays_in_node``, reference :225) -- the distant model sees the same object-frame rays (:284-286),
        the sky the world directions (:455) -- and the sample normals are rotated back with the DETACHED rotation
        (:262-265).  None = identity (the single-object configs)."""
        assert rays_o.dim() == rays_d.dim() == 2, "rays_o and rays_d should have size of [N, 3]"
        config = self.config
        if with_rgb is None:
            with_rgb = config.get("with_rgb", True)
        if with_normal is None:
            with_normal = config.get("with_normal", False)
        if near is None:
            near = config.get("near", None)
        if far is None:
            far = config.get("far", None)
        N, device = rays_o.shape[0], rays_o.device
        total_num_samples_per_ray = torch.zeros(N, dtype=torch.long, device=device)
        total_rendered = None       # all-rays images: written by the fused compositing, zero images only if nothing hit
        rays_d_world = rays_d
        if world_transform is not None:
            assert cr_ray_tested is None, "a precomputed ray test is in the object frame already"
            rot, trans, scale = world_transform
            rays_o, rays_d = model.convert_rays_in_node(rays_o, rays_d, rot, trans, scale)

        cr_ray_input = dict(rays_o=rays_o, rays_d=rays_d, near=near, far=far, rays_ts=rays_ts, rays_pix=rays_pix,
                            rays_h_appear=rays_h_appear)
        if cr_ray_tested is None:
            cr_ray_tested = model.ray_test(**cr_ray_input)
        ray_query_config = dict(model.ray_query_cfg)
        ray_query_config.update({k: v for k, v in config.items()})
        ray_query_config.update(with_rgb=with_rgb, with_normal=with_normal)
        for k, v in (bypass_ray_query_cfg or {}).items():
            ray_query_config[k] = v
        cr_ret = model.ray_query(ray_input=cr_ray_input, ray_tested=cr_ray_tested, config=ray_query_config,
                                 return_buffer=True, return_details=return_details,
                                 render_per_obj_individual=render_per_obj_individual)
        # what the loss modules look up in raw_per_obj_model (reference :247; app/loss/eikonal.py:197-202)
        cr_ret.update(class_name=config.get("main_class_name", "Main"), model_id=getattr(model, "id", "main"), obj_id="main")
        vb = cr_ret["volume_buffer"]
        if vb["type"] != "empty":
            rih, pih = vb["rays_inds_hit"], vb["pack_infos_hit"]
            total_num_samples_per_ray.index_put_((rih,), pih[:, 1])      # (first writer into the zeros; hit rays are unique)
            vb.update(rays_inds_collect=rih, pack_infos_collect=pih)
            if "nablas" in vb:
                if world_transform is None:
                    vb["nablas_in_world"] = vb["nablas"]      # identity object->world rotation
                else:
                    o2w = world_transform[0].detach()
                    if o2w.dim() == 2:
                        vb["nablas_in_world"] = (o2w * vb["nablas"].unsqueeze(-2)).sum(-1)
                    else:                                     # per-ray rotation (object frozen at several frames)
                        vb["nablas_in_world"] = po.packed_matmul(vb["nablas"], o2w[rih], pih)
        # ---- distant-view model on ALL rays (reference :281-335)
        dv_vb = None
        if distant_model is not None:
            near_dv = torch.full([N], float(near) if near is not None else 0.0, dtype=torch.float32, device=device)
            if cr_ray_tested["num_rays"] > 0:
                near_dv[cr_ray_tested["rays_inds"]] = cr_ray_tested["far"]
            dv_tested = dict(rays_o=rays_o.detach(), rays_d=rays_d.detach(), near=near_dv, rays_h_appear=rays_h_appear,
                             num_rays=N, rays_inds=torch.arange(N, device=device))
            dv_cfg = dict(distant_model.ray_query_cfg)
            dv_cfg.update({k: v for k, v in config.items()})
            for k, v in (bypass_ray_query_cfg or {}).items():
                dv_cfg[k] = v
            dv_ret = distant_model.ray_query(ray_tested=dv_tested, config=dv_cfg, return_buffer=True,
                                             return_details=return_details)
            dv_ret.update(class_name="Distant", model_id=getattr(distant_model, "id", "distant"), obj_id="distant")
            dv_vb = dv_ret["volume_buffer"]
            K = dv_vb["num_per_hit"]
            total_num_samples_per_ray += K
            dv_vb.update(rays_inds_collect=dv_vb["rays_inds_hit"],
                         pack_infos_collect=po.get_pack_infos_from_n(torch.full([N], K, dtype=torch.long, device=device)))
        # ---- total volume buffer (reference :337-407)
        total_volume_buffer = dict(type="empty")
        pidx_cr = pidx_dv = None
        if vb["type"] != "empty" and dv_vb is not None:
            pidx_dv, pidx_cr, total_pi = po.merge_two_packs_sorted(
                dv_vb["t"].flatten(), dv_vb["pack_infos_collect"], dv_vb["rays_inds_collect"],
                vb["t"].flatten(), vb["pack_infos_collect"], vb["rays_inds_collect"], a_is_arange=True)
            S_tot = dv_vb["t"].numel() + vb["t"].numel()

            def place(a_dv, a_cr, tail=()):
                z = torch.zeros([S_tot, *tail], dtype=torch.float32, device=device)
                if a_dv is not None:
                    z = z.index_put((pidx_dv,), a_dv)
                if a_cr is not None:
                    z = z.index_put((pidx_cr,), a_cr)
                return z
            total_volume_buffer = dict(type="packed", rays_inds_hit=torch.arange(N, device=device), pack_infos_hit=total_pi,
                                       t=place(dv_vb["t"].flatten(), vb["t"].flatten()),
                                       opacity_alpha=place(dv_vb["opacity_alpha"].flatten(), vb["opacity_alpha"].flatten()))
            if with_rgb:
                total_volume_buffer["rgb"] = place(dv_vb["rgb"].flatten(0, -2), vb["rgb"].flatten(0, -2), (3,))
            if with_normal and "nablas_in_world" in vb:
                total_volume_buffer["nablas_in_world"] = place(None, vb["nablas_in_world"].flatten(0, -2), (3,))
        elif vb["type"] != "empty":
            total_volume_buffer = dict(type="packed", rays_inds_hit=vb["rays_inds_hit"], pack_infos_hit=vb["pack_infos_hit"],
                                       t=vb["t"], opacity_alpha=vb["opacity_alpha"])
            for k in ("rgb", "nablas_in_world"):
                if k in vb:
                    total_volume_buffer[k] = vb[k]
        elif dv_vb is not None:
            total_volume_buffer = dict(type="packed", rays_inds_hit=dv_vb["rays_inds_hit"],
                                       pack_infos_hit=dv_vb["pack_infos_collect"], t=dv_vb["t"].flatten(),
                                       opacity_alpha=dv_vb["opacity_alpha"].flatten())
            if with_rgb:
                total_volume_buffer["rgb"] = dv_vb["rgb"].flatten(0, -2)
        # ---- volume rendering (reference :73-102, :412-442) through the fused compositing kernel
        if total_volume_buffer["type"] != "empty":
            tvb = total_volume_buffer
            rih, pih = tvb["rays_inds_hit"], tvb["pack_infos_hit"]
            nab = tvb.get("nablas_in_world") if with_normal else None
            if nab is not None and not self.training:
                nab = F.normalize(nab.clamp(-1, 1), dim=-1)
            every_ray = rih.shape[0] == N          # rays_inds are sorted & unique: R == N means identity
            out = volume_integration(tvb["opacity_alpha"], tvb["t"], tvb.get("rgb") if with_rgb else None, nab, pih,
                                     config.get("depth_use_normalized_vw", True),
                                     rays_inds=None if every_ray else rih, num_rays=None if every_ray else N)
            tvb["vw"] = out["vw"]
            if pidx_cr is not None:
                vb["vw_in_total"], dv_vb["vw_in_total"] = out["vw"][pidx_cr], out["vw"][pidx_dv]
                thre = float(config.get("distant_bwd_trans_thre", os.environ.get("NSIM_DISTANT_BWD_THRE", 1e-3)))
                if thre > 0 and self.training and "_bwd_holder" in dv_ret:
                    # shells behind an (almost) opaque stretch of the joint ray: no backward, no table scatter for them.
                    # 1e-3: the compressed close-range query keeps samples down to a visibility weight of 1e-4, so the
                    # transmittance it leaves behind an opaque surface is ~1e-4 .. 1e-3, never ~0
                    dv_ret["_bwd_holder"]["keep"] = (out["trans"][pidx_dv] >= thre).to(torch.uint8)
            elif vb["type"] != "empty":
                vb["vw"] = vb["vw_in_total"] = out["vw"]
            else:
                dv_vb["vw_in_total"] = out["vw"]
            # already all-rays images (the scatter of the hit rays is fused into the compositing launch)
            total_rendered = {k: out[k] for k in ("mask_volume", "depth_volume", "rgb_volume", "normals_volume") if k in out}
            if with_normal and "normals_volume" not in total_rendered:
                total_rendered["normals_volume"] = torch.zeros([N, 3], dtype=torch.float32, device=device)
        if total_rendered is None:
            total_rendered = prepare_empty_rendered([N], device, with_rgb=with_rgb, with_normal=with_normal)
        # ---- sky model (reference :449-457): one query per ray, blended with the residual transmittance
        if with_rgb:
            total_rendered["rgb_volume_occupied"] = total_rendered["rgb_volume"]
            if with_env is None:
                with_env = config.get("with_env", True)
            if with_env and sky_model is not None:
                env_rgb = sky_model(v=F.normalize(rays_d_world, dim=-1), h_appear=rays_h_appear)
                total_rendered["rgb_sky"] = env_rgb
                total_rendered["rgb_volume_non_occupied"] = env_blend = \
                    (1.0 - total_rendered["mask_volume"][..., None]) * env_rgb
                total_rendered["rgb_volume"] = total_rendered["rgb_volume"] + env_blend
        ret = dict(ray_intersections=dict(samples_cnt=total_num_samples_per_ray), rendered=total_rendered)
        if return_buffer:
            ret["volume_buffer"] = total_volume_buffer
        if return_details:
            ret["raw_per_obj_model"] = {"main": cr_ret}
            if dv_vb is not None:
                ret["raw_per_obj_model"]["distant"] = dv_ret
        return ret

    def render(self, model: LoTDNeuSModel, *, rays: List[torch.Tensor], rays_h_appear: torch.Tensor = None, near=None,
               far=None, rayschunk: int = None, with_rgb=None, with_normal=None, return_buffer=False,
               return_details=False, render_per_obj_individual=False, bypass_ray_query_cfg: dict = None,
               distant_model=None, sky_model=None, with_env: bool = None, cr_ray_tested: dict = None,
               world_transform=None) -> Dict:
        """rays = [rays_o, rays_d(, rays_ts, rays_pix)] with arbitrary prefix shape (reference :495-581)."""
        if rayschunk is None:
            rayschunk = self.config.get("rayschunk", 0)
        with torch.set_grad_enabled(self.training):
            prefix_shape = rays[0].shape[:-1]
            flat = [r.flatten(0, len(prefix_shape) - 1) if r is not None else None for r in rays]
            ha = rays_h_appear.flatten(0, len(prefix_shape) - 1) if rays_h_appear is not None else None
            kwargs = dict(model=model, near=near, far=far, with_rgb=with_rgb, with_normal=with_normal,
                          return_buffer=return_buffer, return_details=return_details,
                          render_per_obj_individual=render_per_obj_individual, bypass_ray_query_cfg=bypass_ray_query_cfg,
                          distant_model=distant_model, sky_model=sky_model, with_env=with_env,
                          cr_ray_tested=cr_ray_tested, world_transform=world_transform)
            if self.training or (not rayschunk) or flat[0].shape[0] <= rayschunk:
                ret = self(*flat[:2], rays_h_appear=ha, **kwargs)
            else:
                assert (not return_buffer) and (not return_details), \
                    "batchify_query does not work when return_buffer=True or return_details=True"
                if ha is None:
                    fn = lambda o, d: self(o, d, **kwargs)                       # noqa: E731
                    ret = batchify_query(fn, flat[0], flat[1], chunk=rayschunk)
                else:
                    fn = lambda o, d, h: self(o, d, rays_h_appear=h, **kwargs)   # noqa: E731
                    ret = batchify_query(fn, flat[0], flat[1], ha, chunk=rayschunk)
            ret.update(rays_o=flat[0], rays_d=flat[1])
            if len(prefix_shape) > 1:
                for k in self.image_keys:
                    if k in ret["rendered"]:
                        ret["rendered"][k] = ret["rendered"][k].unflatten(0, prefix_shape)
        return ret
