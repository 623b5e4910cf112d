"""Multi-object volume renderer -- mirrors the part of ``app/renderers/buffer_compose_renderer.py`` that sits on the hot
path:

* per drawable group (reference :209-345): object-space rays (``Scene.convert_rays_in_nodes_list``,
  app/resources/scenes.py:631-683) -> ``model.ray_test`` / ``model.ray_query`` for single models or
  ``model.batched_ray_test(compact_batch=True)`` / ``set_condition`` / ``model.batched_ray_query`` for a shared batched
  model; normals rotated to world with ``packed_matmul`` (:333-345);
* collect (:644-681: every ray's samples of every object written to one packed buffer through ``interleave_linstep`` on a
  running per-ray cursor) and sort (:683-695: ``packed_sort`` by depth + a permutation of every attribute): here ONE launch
  (``nsim_compose_collect_sort``, csrc/pack_ops.hip) that returns the sorted depths and every object sample's final position;
  integrate (:697-718): ONE fused compositing launch instead of the reference's chain of packed_sum / packed_div;
* ``vw_in_total`` of every object buffer (:720-727) and the sky blend (:820-833, as in the single renderer).

There is no Scene graph in this repository (harness, out of scope): drawables are passed explicitly as
``Drawable(id, class_name, model, rotation [3,3], translation [3], scale)`` -- the object-to-world transform of the
frame being rendered.  ``render_per_class_in_scene`` / ``render_per_obj_in_scene`` (:729-806) are mirrored; not
mirrored: ``render_per_obj_individual`` and its segmentation z-buffer (:276-311), which are evaluation outputs.
"""
from dataclasses import dataclass
import os
from typing import Dict, List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..fields.neus import LoTDNeuSModel, volume_integration
from ..graphics import pack_ops as po
from .single_volume_renderer import prepare_empty_rendered

_COMPOSE_FUSED = os.environ.get("NSIM_COMPOSE_FUSED", "1") != "0"


@dataclass
class Drawable:
    id: str
    class_name: str
    model: nn.Module
    rotation: Optional[torch.Tensor] = None      # [3,3] object -> world
    translation: Optional[torch.Tensor] = None   # [3]
    scale: float = 1.0

    def rays_in_object(self, rays_o, rays_d):
        if self.rotation is None:
            return rays_o, rays_d
        return LoTDNeuSModel.convert_rays_in_node(rays_o, rays_d, self.rotation.to(rays_o), self.translation.to(rays_o),
                                                  self.scale)


def _rays_in_objects(grp, rays_o, rays_d):
    """World -> object rays of all the items of a batched group at once, [B', N, 3] each: the broadcast-multiply-sum of
    ``convert_rays_in_node`` (scenes.py:686-708) with the items' poses stacked -- the same element-wise operations in the same
    order as one call per item (bit-identical), in 7 launches instead of 14 per item (the 8-vehicle step issued 112 here)."""
    if all(dr.rotation is None for dr in grp):
        B = len(grp)
        return rays_o.unsqueeze(0).expand(B, *rays_o.shape).contiguous(), rays_d.unsqueeze(0).expand(B, *rays_d.shape).contiguous()
    eye = torch.eye(3, device=rays_o.device, dtype=rays_o.dtype)
    rot = torch.stack([dr.rotation.to(rays_o) if dr.rotation is not None else eye for dr in grp])                     # [B', 3, 3]
    tr = torch.stack([dr.translation.to(rays_o) if dr.rotation is not None else torch.zeros_like(eye[0]) for dr in grp])
    scales = [dr.scale if dr.rotation is not None else 1.0 for dr in grp]
    if any(torch.is_tensor(s_) for s_ in scales):       # a learnable / tensor scale: the per-item form
        pairs = [dr.rays_in_object(rays_o, rays_d) for dr in grp]
        return torch.stack([p_[0] for p_ in pairs]), torch.stack([p_[1] for p_ in pairs])
    Rt = rot.transpose(-1, -2)[:, None]                                                                                # [B', 1, 3, 3]
    o = ((rays_o[None] - tr[:, None]).unsqueeze(-2) * Rt).sum(-1)
    d = (rays_d[None].unsqueeze(-2) * Rt).sum(-1)
    # ``x / python_scalar`` (the per-item form) is, in ATen, on the device a multiplication by the reciprocal formed on the HOST
    # in double and rounded to f32 (BinaryDivTrueKernel: ``inv_b = opmath_t(1.0 / double(b))``), on the host a true division:
    # the same here, so that the stacked form reproduces the per-item one bit for bit on either
    if rays_o.is_cuda:
        inv = torch.tensor([1.0 / float(s_) for s_ in scales], dtype=torch.float64).to(rays_o.dtype).to(rays_o.device)
        return o * inv[:, None, None], d * inv[:, None, None]
    sc = torch.tensor([float(s_) for s_ in scales], dtype=rays_o.dtype, device=rays_o.device)
    return o / sc[:, None, None], d / sc[:, None, None]


class BufferComposeRenderer(nn.Module):
    def __init__(self, config: Optional[dict] = None):
        super().__init__()
        self.config = dict(config or {})

    def forward(self, *a, **k):
        return self.ray_query(*a, **k)

    def _query_cfg(self, model, with_rgb, with_normal, bypass):
        cfg = dict(model.ray_query_cfg)
        cfg.update(self.config)
        cfg.update(with_rgb=with_rgb, with_normal=with_normal)
        cfg.update(bypass or {})
        return cfg

    def ray_query(self, rays_o: torch.Tensor, rays_d: torch.Tensor, *, drawables: List[Drawable],
                  rays_h_appear: torch.Tensor = None, near=None, far=None, with_rgb: bool = None,
                  with_normal: bool = None, sky_model=None, distant_model=None, distant_cr_id: str = None,
                  return_buffer=False, return_details=False,
                  bypass_ray_query_cfg: Dict[str, dict] = None, render_per_obj_in_scene: bool = False,
                  render_per_class_in_scene: bool = False) -> Dict:
        """``render_per_class_in_scene`` / ``render_per_obj_in_scene``: every class's / object's share of the JOINT
        rendering -- its samples weighted by ``vw_in_total`` (reference :729-806; the per-class masks feed the
        importance sampler of code_multi/tools/train.py:589-592, the per-object images the mono / manhattan losses).
        ``distant_model``: the 'Distant' class, queried LAST (reference :162-164) on ALL rays in the frame of its
        close-range object ``distant_cr_id`` (default: the first single-model drawable), ``near`` := that object's
        ``far`` on the rays that passed its ray test, rays detached (reference :506-531); its batched [N, K] buffer joins
        the collect as one pack of K per ray (:598-606)."""
        assert rays_o.dim() == rays_d.dim() == 2
        cfgd = self.config
        with_rgb = cfgd.get("with_rgb", True) if with_rgb is None else with_rgb
        with_normal = cfgd.get("with_normal", False) if with_normal is None else with_normal
        near = cfgd.get("near", None) if near is None else near
        far = cfgd.get("far", None) if far is None else far
        N, dev = rays_o.shape[0], rays_o.device
        bypass = bypass_ray_query_cfg or {}
        total_rendered = prepare_empty_rendered([N], dev, with_rgb=with_rgb, with_normal=with_normal)
        ray_visible_samples = torch.zeros([N], dtype=torch.long, device=dev)
        raw_per_obj_model: Dict[str, Dict] = {}

        # ---- group drawables by model: one query per single model, ONE batched query per shared model
        groups: Dict[int, List[Drawable]] = {}
        for dr in drawables:
            groups.setdefault(id(dr.model), []).append(dr)
        for grp in groups.values():
            model = grp[0].model
            cls = grp[0].class_name
            qcfg = self._query_cfg(model, with_rgb, with_normal, bypass.get(cls))
            if getattr(model, "is_batched_query_supported", False):
                oo, dd = _rays_in_objects(grp, rays_o, rays_d)                               # [B', N, 3]
                extra = {}
                if rays_h_appear is not None:
                    extra["rays_h_appear"] = rays_h_appear.unsqueeze(0).expand(len(grp), *rays_h_appear.shape)
                bt = model.batched_ray_test(oo, dd, near=near, far=far, compact_batch=True, **extra)
                model.set_condition({"ins_id": [dr.id for dr in grp]})
                raw = model.batched_ray_query(batched_ray_tested=bt, config=qcfg, return_buffer=True,
                                              return_details=return_details)
                model.clean_condition()
                vb = raw["volume_buffer"]
                if vb["type"] != "empty" and "nablas" in vb:
                    rot = torch.stack([dr.rotation if dr.rotation is not None else torch.eye(3) for dr in grp]).to(dev)
                    rot_hit = rot[vb["rays_full_bidx_hit"]].detach()                         # reference :264 (detached)
                    vb["nablas_in_world"] = po.packed_matmul(vb["nablas"], rot_hit, vb["pack_infos_hit"])
                raw.update(class_name=cls, obj_id=[dr.id for dr in grp], num_rays=bt["num_rays"])
                raw_per_obj_model[cls] = raw
            else:
                for dr in grp:
                    o_o, d_o = dr.rays_in_object(rays_o, rays_d)
                    tested = model.ray_test(o_o, d_o, near=near, far=far, rays_h_appear=rays_h_appear)
                    raw = model.ray_query(ray_tested=tested, config=qcfg, return_buffer=True,
                                          return_details=return_details)
                    vb = raw["volume_buffer"]
                    if vb["type"] != "empty" and "nablas" in vb:
                        if dr.rotation is None:
                            vb["nablas_in_world"] = vb["nablas"]
                        else:
                            R = dr.rotation.to(dev).detach().expand(vb["pack_infos_hit"].shape[0], 3, 3).contiguous()
                            vb["nablas_in_world"] = po.packed_matmul(vb["nablas"], R, vb["pack_infos_hit"])
                    raw.update(class_name=dr.class_name, obj_id=dr.id, num_rays=tested["num_rays"],
                               rays_inds=tested["rays_inds"], ray_far=tested["far"], _drawable=dr)
                    raw_per_obj_model[dr.id] = raw
        if distant_model is not None:
            singles = [r for r in raw_per_obj_model.values() if "_drawable" in r]
            cr = raw_per_obj_model.get(distant_cr_id) if distant_cr_id is not None else (singles[0] if singles else None)
            near_dv = torch.full([N], float(near) if near is not None else 0.0, dtype=torch.float32, device=dev)
            o_dv, d_dv = rays_o, rays_d
            if cr is not None:
                o_dv, d_dv = cr["_drawable"].rays_in_object(rays_o, rays_d)
                if cr["num_rays"] > 0:
                    near_dv = near_dv.index_put((cr["rays_inds"],), cr["ray_far"])
            dv_tested = dict(rays_o=o_dv.detach(), rays_d=d_dv.detach(), near=near_dv, rays_h_appear=rays_h_appear,
                             num_rays=N, rays_inds=torch.arange(N, device=dev))
            raw = distant_model.ray_query(ray_tested=dv_tested, config=self._query_cfg(distant_model, with_rgb, with_normal,
                                                                                         bypass.get("Distant")),
                                          return_buffer=True, return_details=return_details)
            vb = raw["volume_buffer"]
            vb["pack_infos_hit"] = po.get_pack_infos_from_n(torch.full([N], int(vb["num_per_hit"]), dtype=torch.long, device=dev))
            raw.update(class_name="Distant", obj_id="distant", num_rays=N)
            raw_per_obj_model["distant"] = raw
        for r in raw_per_obj_model.values():
            r.pop("_drawable", None)
        # ---- per-ray sample counts over all objects (several batch items may hit the same ray: index_add)
        for raw in raw_per_obj_model.values():
            vb = raw["volume_buffer"]
            if vb["type"] == "empty":
                continue
            rih, n_hit = vb["rays_inds_hit"], vb["pack_infos_hit"][:, 1]
            # a ray that crosses several items of a batched model owns several consecutive packs: regroup them into
            # ONE pack per ray for the collect step (reference :347-368)
            ric, dup = torch.unique_consecutive(rih, return_counts=True)
            if ric.shape[0] != rih.shape[0]:
                n_col = torch.zeros([N], dtype=torch.long, device=dev).index_add_(0, rih, n_hit)[ric]
                vb["rays_inds_collect"], vb["pack_infos_collect"] = ric, po.get_pack_infos_from_n(n_col)
            else:
                vb["rays_inds_collect"], vb["pack_infos_collect"] = rih, vb["pack_infos_hit"]
            ray_visible_samples.index_add_(0, rih, n_hit)

        total_volume_buffer = dict(type="empty")
        total_rays_inds_hit = ray_visible_samples.nonzero()[:, 0]
        if total_rays_inds_hit.numel() > 0:
            # ---- collect (reference :648-681)
            pi_sparse, tot = po.get_pack_infos_from_n(ray_visible_samples, return_total=True)
            total_pack_infos = pi_sparse[total_rays_inds_hit]
            S = int(tot.item())
            f32 = dict(dtype=torch.float32, device=dev)
            # collect + sort in ONE launch (nsim_compose_collect_sort): every ray's depths in order and, per object buffer, the
            # final position of each of its samples -- the reference's cursor / interleave_linstep / packed_sort / permutation
            # bookkeeping (:660-695) without the unsorted intermediate; an attribute is one indexed store per object
            live = [raw["volume_buffer"] for raw in raw_per_obj_model.values() if raw["volume_buffer"]["type"] != "empty"]
            if len(live) <= 64 and _COMPOSE_FUSED:
                t_sorted, dsts = po.compose_collect_sort(
                    [(vb["t"], vb["rays_inds_collect"], vb["pack_infos_collect"]) for vb in live], pi_sparse, S)
            else:       # more object buffers than the kernel has lanes for (or NSIM_COMPOSE_FUSED=0, the A/B aid): the reference's steps
                depths = torch.zeros([S], **f32)
                cursor, pidx = pi_sparse[:, 0].clone(), []
                for vb in live:
                    ric, n = vb["rays_inds_collect"], vb["pack_infos_collect"][:, 1]
                    pidx.append(po.interleave_linstep(cursor[ric], n, 1))
                    depths[pidx[-1]] = vb["t"].detach().flatten().float()
                    cursor.index_add_(0, ric, n)
                t_sorted, sort_idx = po.packed_sort(depths, total_pack_infos)
                ranks = po.inverse_permutation(sort_idx)
                dsts = [ranks[p_] for p_ in pidx]
            alphas = torch.zeros([S], **f32)
            rgbs = torch.zeros([S, 3], **f32) if with_rgb else None
            nabs = torch.zeros([S, 3], **f32) if with_normal else None
            t_grad = None
            for vb, dst in zip(live, dsts):
                vb["pidx_in_total"] = dst         # (position in the SORTED total buffer)
                alphas = alphas.index_put((dst,), vb["opacity_alpha"].flatten())
                if with_rgb:
                    rgbs = rgbs.index_put((dst,), vb["rgb"].flatten(0, -2))
                if with_normal and "nablas_in_world" in vb:
                    nabs = nabs.index_put((dst,), vb["nablas_in_world"].flatten(0, -2))
                if vb["t"].requires_grad:         # depths that carry a gradient (pose refinement) keep it
                    t_grad = (torch.zeros([S], **f32) if t_grad is None else t_grad).index_put((dst,), vb["t"].flatten())
            if t_grad is not None:                # (values of the other sources from the kernel's output)
                got = torch.zeros([S], dtype=torch.bool, device=dev)
                for vb, dst in zip(live, dsts):
                    if vb["t"].requires_grad:
                        got[dst] = True
                t_sorted = torch.where(got, t_grad, t_sorted)
            total_volume_buffer = dict(type="packed", rays_inds_hit=total_rays_inds_hit, pack_infos_hit=total_pack_infos,
                                       t=t_sorted, opacity_alpha=alphas)
            if with_rgb:
                total_volume_buffer["rgb"] = rgbs
            if with_normal:
                total_volume_buffer["nablas"] = nabs
            # ---- integrate (reference :697-718), one fused launch
            tvb = total_volume_buffer
            nab = tvb.get("nablas") if with_normal else None
            if nab is not None and not self.training:
                nab = F.normalize(nab.clamp(-1, 1), dim=-1)
            out = volume_integration(tvb["opacity_alpha"], tvb["t"], tvb.get("rgb") if with_rgb else None, nab,
                                     total_pack_infos, cfgd.get("depth_use_normalized_vw", True))
            tvb["vw"] = out["vw"]
            for k in ("mask_volume", "depth_volume", "rgb_volume", "normals_volume"):
                if k in out and k in total_rendered:
                    total_rendered[k] = total_rendered[k].index_put((total_rays_inds_hit,), out[k])
            # ---- every object's weights in the context of the whole scene (reference :720-727)
            for raw in raw_per_obj_model.values():
                vb = raw["volume_buffer"]
                if vb["type"] != "empty":
                    vb["vw_in_total"] = out["vw"][vb["pidx_in_total"]]
                    thre = float(cfgd.get("distant_bwd_trans_thre", os.environ.get("NSIM_DISTANT_BWD_THRE", 1e-3)))
                    if thre > 0 and self.training and "_bwd_holder" in raw:
                        raw["_bwd_holder"]["keep"] = (out["trans"][vb["pidx_in_total"]] >= thre).to(torch.uint8)
        norm_depth = cfgd.get("depth_use_normalized_vw", True)

        def share(vb, pack_infos):
            """Per-pack sums of one object buffer weighted by its vw_in_total -> dict of [P(,3)]."""
            vw = vb["vw_in_total"].reshape(-1)
            out_ = dict(mask_volume=po.packed_sum(vw, pack_infos), depth_volume=po.packed_sum(vw * vb["t"].flatten(), pack_infos))
            if with_rgb:
                out_["rgb_volume"] = po.packed_sum(vw[:, None] * vb["rgb"].flatten(0, -2), pack_infos)
            if with_normal and "nablas_in_world" in vb:
                out_["normals_volume"] = po.packed_sum(vw[:, None] * vb["nablas_in_world"].flatten(0, -2), pack_infos)
            return out_
        rendered_per_class, rendered_per_obj = {}, {}
        if render_per_class_in_scene:
            for dr in drawables:
                rendered_per_class.setdefault(dr.class_name, prepare_empty_rendered([N], dev, with_rgb=with_rgb,
                                                                                    with_normal=with_normal))
            if sky_model is not None:
                rendered_per_class["Sky"] = prepare_empty_rendered([N], dev, with_rgb=with_rgb, with_normal=with_normal)
            for raw in raw_per_obj_model.values():
                vb = raw["volume_buffer"]
                if vb["type"] == "empty" or "vw_in_total" not in vb:
                    continue
                tgt = rendered_per_class[raw["class_name"]]
                for k, v in share(vb, vb["pack_infos_collect"]).items():      # index_add: several objects per class
                    tgt[k] = tgt[k].index_add(0, vb["rays_inds_collect"], v)
            if norm_depth:
                for v in rendered_per_class.values():
                    v["depth_volume"] = v["depth_volume"] / (v["mask_volume"] + 1e-10)
        if render_per_obj_in_scene:
            for dr in drawables:
                rendered_per_obj[dr.id] = prepare_empty_rendered([N], dev, with_rgb=with_rgb, with_normal=with_normal)
            for raw in raw_per_obj_model.values():
                vb = raw["volume_buffer"]
                if vb["type"] == "empty" or "vw_in_total" not in vb:
                    continue
                sh = share(vb, vb["pack_infos_hit"])
                if isinstance(raw["obj_id"], list):             # batched model: one image per item of the batch
                    ids = raw["obj_id"]
                    cur = prepare_empty_rendered([len(ids), N], dev, with_rgb=with_rgb, with_normal=with_normal)
                    where = (vb["rays_full_bidx_hit"], vb["rays_inds_hit"])
                    for k, v in sh.items():
                        cur[k] = cur[k].index_put(where, v)
                    if norm_depth:
                        cur["depth_volume"] = cur["depth_volume"] / (cur["mask_volume"] + 1e-10)
                    for i, oid in enumerate(ids):
                        rendered_per_obj[oid] = {k: v[i] for k, v in cur.items()}
                else:
                    cur = rendered_per_obj[raw["obj_id"]]
                    for k, v in sh.items():
                        cur[k] = cur[k].index_put((vb["rays_inds_hit"],), v)
                    if norm_depth:
                        cur["depth_volume"] = cur["depth_volume"] / (cur["mask_volume"] + 1e-10)
        if with_rgb:
            total_rendered["rgb_volume_occupied"] = total_rendered["rgb_volume"]
            if sky_model is not None and cfgd.get("with_env", True):
                env = sky_model(v=F.normalize(rays_d, dim=-1), h_appear=rays_h_appear)
                total_rendered["rgb_sky"] = env
                total_rendered["rgb_volume_non_occupied"] = blend = (1.0 - total_rendered["mask_volume"][..., None]) * env
                total_rendered["rgb_volume"] = total_rendered["rgb_volume"] + blend
                if render_per_class_in_scene:                   # reference :821-823
                    rendered_per_class["Sky"]["rgb_volume"] = blend
                    rendered_per_class["Sky"]["mask_volume"] = 1 - total_rendered["mask_volume"]
        ret = dict(rendered=total_rendered, ray_intersections=dict(samples_cnt=ray_visible_samples))
        if render_per_class_in_scene:
            ret["rendered_per_class_in_scene"] = rendered_per_class
        if render_per_obj_in_scene:
            ret["rendered_per_obj_in_scene"] = rendered_per_obj
        if return_buffer:
            ret["volume_buffer"] = total_volume_buffer
        if return_details:
            ret["raw_per_obj_model"] = raw_per_obj_model
        return ret
