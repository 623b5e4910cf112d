"""LoTD NeuS model -- host side of the fused gfx950 field / sampling kernels.

Mirrors ``nr3d_lib.models.fields.neus.LoTDNeuSModel`` (+ ``NeusRendererMixin``) at the API the reference's
renderers and losses call:
  * ``ray_test(rays_o, rays_d, near, far, rays_ts=, rays_pix=, rays_h_appear=) -> dict``
    (app/renderers/single_volume_renderer.py:235-238, keys :289-300)
  * ``ray_query(ray_input=, ray_tested=, config=, return_buffer=, return_details=, render_per_obj_individual=)``
    (single_volume_renderer.py:244-246) -> ``volume_buffer{type,rays_inds_hit,pack_infos_hit,t,opacity_alpha,rgb,nablas}``
  * ``forward_sdf / forward_sdf_nablas / query_sdf`` (code_single/tools/inspect_rendering.py:120-128,305,417),
    ``sample_pts_uniform`` (app/loss/eikonal.py:215), ``forward_inv_s`` (code_single/tools/eval.py:218-219)
Model hyper-parameters follow code_single/configs/object_centric/lotd_neus.dtu.230814.yaml:80-173.

All arithmetic runs in the HIP kernels (csrc/field.hip, csrc/sampling.hip, csrc/pack_ops.hip) through the C ABI;
this file only allocates tensors, sequences launches on the current stream and defines autograd boundaries.
Host synchronisations per render: two (hit-ray compaction, marched-sample total) -- the reference has three
(single_volume_renderer.py:340,345,414).
"""
import ctypes
import math
import os
import time
from typing import Dict, Optional, Sequence

import torch
import torch.nn as nn

from .. import _lib
from ..grid_encodings.lotd import LoTDConfig, LoTDEncoding
from ..graphics import pack_ops as po
from ..model_base import ModelMixin
from ..spatial import AABBSpace, aabb_ray_test

def C_memmove(dst, src):
    ctypes.memmove(ctypes.byref(dst), ctypes.byref(src), ctypes.sizeof(src))


DEFAULT_LOD_RES = [16, 23, 31, 43, 59, 81, 112, 154, 213, 295, 407, 562, 777, 1073, 1483, 2048]
RAD_IN = 26


def _flat_sizes(D: int, L: int = 16, E: int = 0):
    """E: width of the embedded-position block appended to the SDF decoder's input (``extra_pos_embed_cfg``; 0 = none)."""
    n_sdf_w = 64 * (2 * L + E) + (64 * 64 if D == 2 else 0) + 64
    n_sdf_b = 64 * D + 1
    n_rad_w = 64 * RAD_IN + 64 * 64 + 3 * 64
    n_rad_b = 131
    return n_sdf_w, n_sdf_b, n_rad_w, n_rad_b


def append_extra_points(model, rays_o, rays_d, t, ridx, h_appear, extra_x):
    """Free points appended to a ray-mode query as M zero-length rays (origin = the point, direction e_z, depth 0,
    appearance code 0): ray arrays grow to [R+M], sample arrays to [S+M]."""
    dev = rays_o.device
    R, M = rays_o.shape[0], extra_x.shape[0]
    xe = extra_x.detach().float().reshape(-1, 3)
    ck = (M, str(dev))
    cache = getattr(model, "_extra_cache", None)
    if cache is None or cache[0] != ck:          # constants of the M zero-length rays: dir e_z, t 0, codes 0
        ez = torch.zeros([M, 3], dtype=torch.float32, device=dev)
        ez[:, 2] = 1.0
        cache = model._extra_cache = (ck, ez, torch.zeros([M], dtype=torch.float32, device=dev),
                                      torch.zeros([M, 4], dtype=torch.float32, device=dev))
    _, ez, tz, hz = cache
    rays_o, rays_d = torch.cat([rays_o, xe]), torch.cat([rays_d, ez])
    t = torch.cat([t, tz])
    ridx = torch.cat([ridx, torch.arange(R, R + M, device=dev)])
    if h_appear is not None:
        hz = hz if h_appear.shape[1] == 4 else hz.new_zeros([M, h_appear.shape[1]])
        h_appear = torch.cat([h_appear.detach().float(), hz])
    return rays_o, rays_d, t, ridx, h_appear


def fine_list(qp: dict):
    """``num_fine`` per up-sampling stage.  A list names the stages (``num_fine [8, 8, 32]`` with ``upsample_inv_s_factors [1, 4,
    16]``, lotd_neus.dtu.230814.yaml:150-152); the multi-object configs give ONE number next to two factors (``num_fine: 16,
    upsample_inv_s_factors: [1, 4]``, all_occ.240201.yaml:481-484): taken as the total, dealt evenly to the stages (the
    sampler lives in the absent nr3d_lib: semantics fixed here)."""
    nf = qp.get("num_fine", [8, 8, 32])
    if isinstance(nf, (list, tuple)):
        return [int(n) for n in nf]
    k = max(len(qp.get("upsample_inv_s_factors", [1, 4, 16])), 1)
    return [max(int(nf) // k, 1)] * k


def marched_only(qp: dict) -> bool:
    """``query_param.upsample_on_marched_only`` (default True; env NSIM_UPSAMPLE_ON_MARCHED_ONLY=0 flips the default): coarse
    and fine samples only on the rays whose occupancy march found occupied voxels; the other tested rays stay without
    samples.  The reference's volume buffers list the rays that produced samples -- ``rays_inds_hit`` / ``pack_infos_hit``
    are a subset [R'] of the tested rays [R], and a model whose march finds nothing returns ``type: 'empty'``
    (single_volume_renderer.py:209-220,289-300) --, which is also what an occupancy grid is for: a ray that crosses only
    empty voxels is not queried.  False restores rounds 1-4: every AABB-hit ray gets num_coarse + sum(num_fine) SDF queries
    (on the object workload ~60 % of the tested rays see no occupied voxel and were 55 % of the sampling queries)."""
    v = qp.get("upsample_on_marched_only", None)
    if v is None:
        return os.environ.get("NSIM_UPSAMPLE_ON_MARCHED_ONLY", "1") == "1"
    return bool(v)


# --------------------------------------------------------------------------------------------- autograd


_SPEC_FORWARD = os.environ.get("NSIM_SPEC_FORWARD", "1") == "1"
# evaluation (no-grad) forward through the level-major gather + the decoders on the planes instead of the fused point-major
# kernel (k_field<0,2,1,1>: 512 registers + 61 spilled, one wave per SIMD).  Round 5 A/B: see DESIGN.md sec. 4
_EVAL_PLANES = os.environ.get("NSIM_EVAL_PLANES", "0") == "1"
_SDF_FUSED_BELOW = int(os.environ.get("NSIM_SDF_FUSED_BELOW", "0"))
# up-sampling: the merge of stage k and the draws of stage k + 1 as ONE launch (nsim_merge_upsample; 0: two launches)
_FUSE_MERGE_UPSAMPLE = os.environ.get("NSIM_FUSE_MERGE_UPSAMPLE", "1") == "1"


class _FieldFn(torch.autograd.Function):
    """(grid, sdf_w, sdf_b, rad_w, rad_b, h_appear) -> (sdf [S], nablas [S,3], rgb [S,3]).
    ``nablas`` is an ordinary differentiable output: its gradient w.r.t. grid and decoder weights (the
    "double backward" of the reference, app/loss/eikonal.py:216-251) is produced analytically by nsim_field_bwd."""

    @staticmethod
    def forward(ctx, model, grid, sdf_w, sdf_b, rad_w, rad_b, h_appear, x, rays_o, rays_d, t, ridx, with_rgb,
                goff=None, extra_x=None, pre=None):
        """goff [R] int64 (batched model only): table offset of every ray's instance; needs ridx.
        extra_x [M,3] (ray mode only): additional free points evaluated by the SAME launches (appended as M
        zero-length rays); their sdf / nablas come back as two extra outputs.  The trainer uses it for the uniform
        eikonal points (code_single/tools/train.py:602-613): a separate 4096-point launch chain costs ~0.2 ms of
        fixed per-launch latency.
        pre (``ray_query``'s speculative launch): the forward kernels are ALREADY queued, at a capacity, on the arrays
        in ``pre`` (extra points appended) -- only the bookkeeping is left."""
        dev = grid.device
        M = 0
        ctx.PS = 0
        if pre is not None:
            M = pre["M"]
            rays_o, rays_d, t, ridx, ha = pre["rays_o"], pre["rays_d"], pre["t"], pre["ridx"], pre["ha"]
            S = t.shape[0]
            sdf, nablas, rgb = pre["sdf"][:S], pre["nablas"][:S], (pre["rgb"][:S] if with_rgb else None)
            h_pl, J_pl, ctx.PS = pre["h_pl"], pre["J_pl"], pre["PS"]
            if _lib.TIMER is not None:
                _lib.TIMER.note_units("nsim_field_fwd", S)
            ctx.model, ctx.S, ctx.with_rgb, ctx.M = model, S, with_rgb, M
            ctx.enc_state = pre.get("enc_state")
            ctx.grid_numel = grid.numel()
            ctx.grid16 = None
            ctx.x_shape = None
            ctx.save_for_backward(None, rays_o, rays_d, t, ridx, ha, nablas, rgb, h_pl, J_pl)
            ctx.goff = None
            ctx.ha_shape = ha.shape if ha is not None else None
            ctx.set_materialize_grads(False)
            if M == 0:
                return (sdf, nablas, rgb) if with_rgb else (sdf, nablas)
            Sm = S - M
            return (sdf[:Sm], nablas[:Sm]) + ((rgb[:Sm],) if with_rgb else ()) + (sdf[Sm:], nablas[Sm:])
        if extra_x is not None:
            assert x is None and goff is None, "extra points ride on a ray-mode query of a single-instance model"
            M = extra_x.shape[0]
            rays_o, rays_d, t, ridx, h_appear = append_extra_points(model, rays_o, rays_d, t, ridx, h_appear, extra_x)
        S = x.shape[0] if x is not None else t.shape[0]
        grid16, wpack = model._shadow()
        sdf = torch.empty([S], dtype=torch.float32, device=dev)
        nablas = torch.empty([S, 3], dtype=torch.float32, device=dev)
        rgb = torch.empty([S, 3], dtype=torch.float32, device=dev) if with_rgb else None
        ha = h_appear.detach().float().contiguous() if h_appear is not None else None
        need_bwd = any(ctx.needs_input_grad)
        # level-major planes of the gathered features / their x-derivative, saved so the backward never re-gathers
        # (pyramids with more than 16 levels exist on the level-major path only: planes in evaluation as well)
        NLP = model.plane_levels
        need_pl = need_bwd or NLP > 16 or model.planes_always or _EVAL_PLANES
        PS = _lib.plane_pitch(S)
        h_pl = torch.empty([NLP, PS, 2], dtype=torch.float32, device=dev) if need_pl else None
        J_pl = torch.empty([NLP, PS, 2, 3], dtype=_lib.jplane_dtype(model.field_meta), device=dev) if need_pl else None
        # (the encoding's hook may return per-query state its backward hooks need -- the permutohedral model's condition z --,
        # carried on ctx: the backward of a query uses what the query was MADE with)
        ctx.enc_state = model._enc_field_fwd(grid16, wpack, x, rays_o, rays_d, t, ridx, goff, ha, S, sdf, nablas, rgb, h_pl, J_pl,
                                             None, 0)
        if _lib.TIMER is not None:
            _lib.TIMER.note_units("nsim_field_fwd", S)
        ctx.model, ctx.S, ctx.with_rgb, ctx.M = model, S, with_rgb, M
        ctx.grid_numel = grid.numel()
        ctx.grid16 = grid16        # the table the query read (a grown table of a condition that may be cleaned before the backward)
        ctx.x_shape = x.shape if x is not None else None
        # save_for_backward, not a ctx attribute: without extra points nablas / rgb ARE the outputs, and output -> grad_fn
        # -> ctx -> output would be a reference cycle (the planes of a step, ~150 MB, until the cyclic collector runs)
        ctx.save_for_backward(x, rays_o, rays_d, t, ridx, ha, nablas, rgb, h_pl, J_pl)
        ctx.goff = goff
        ctx.ha_shape = ha.shape if ha is not None else None
        ctx.set_materialize_grads(False)      # unused outputs (the free points' sdf) arrive as None: backward handles it
        if M == 0:
            return (sdf, nablas, rgb) if with_rgb else (sdf, nablas)
        Sm = S - M
        outs = (sdf[:Sm], nablas[:Sm]) + ((rgb[:Sm],) if with_rgb else ()) + (sdf[Sm:], nablas[Sm:])
        return outs

    @staticmethod
    def backward(ctx, *grads):
        model = ctx.model
        x, rays_o, rays_d, t, ridx, ha, nab_fwd, rgb_fwd, h_pl, J_pl = ctx.saved_tensors
        S, M = ctx.S, ctx.M
        dev = nab_fwd.device
        g_sdf, g_nab = grads[0], grads[1]
        g_rgb = grads[2] if ctx.with_rgb else None
        if M > 0:       # re-join the gradients of the main samples and of the extra points
            Sm = S - M
            ge_s, ge_n = grads[-2], grads[-1]

            def join(a, b, tail):
                if a is None and b is None:
                    return None
                a = a.float() if a is not None else _lib.zeros([Sm, *tail], device=dev)
                b = b.float() if b is not None else _lib.zeros([M, *tail], device=dev)
                return torch.cat([a, b])
            g_sdf, g_nab = join(g_sdf, ge_s, ()), join(g_nab, ge_n, (3,))
            if g_rgb is not None:
                g_rgb = join(g_rgb, None, (3,))
        grid16 = ctx.grid16 if ctx.grid16 is not None else model._table16()
        wpack = model._weight_pack()
        n_sdf_w, n_sdf_b, n_rad_w, n_rad_b = _flat_sizes(model.sdf_D, model.encoding.cfg.num_levels, model.pos_embed_E)
        need = ctx.needs_input_grad
        # (zero-filled accumulators: views of the step's arena when the trainer opened one, _lib.zeros)
        dgrid = _lib.zeros([ctx.grid_numel], device=dev) if need[1] else None
        # one memset for the four small accumulators
        dsdf_w, dsdf_b, drad_w, drad_b = _lib.zeros([n_sdf_w + n_sdf_b + n_rad_w + n_rad_b],
                                                    device=dev).split([n_sdf_w, n_sdf_b, n_rad_w, n_rad_b])
        dha = _lib.zeros(list(ctx.ha_shape), device=dev) if (ha is not None and need[6]) else None
        gs = g_sdf.float().contiguous() if g_sdf is not None else None
        gn = g_nab.float().contiguous() if g_nab is not None else None
        gr = g_rgb.float().contiguous() if (ctx.with_rgb and g_rgb is not None) else None
        fm = model.field_meta
        gn_total = gn
        # pose refinement (the rays / points carry gradients, e.g. LearnableParams of the street config): per-sample
        # dL/dx and dL/d(view dir), reduced per ray at the end
        need_x, need_o, need_d = need[7] and x is not None, need[8] and x is None, need[9] and x is None
        need_dx = need_x or need_o or need_d
        dx = dv = None
        if need_dx:
            dx = (torch.empty if gr is not None else torch.zeros)([S, 3], dtype=torch.float32, device=dev)
            dv = torch.empty([S, 3], dtype=torch.float32, device=dev) if (gr is not None and need_d) else None
        if gr is not None:      # (1) radiance branch: weight grads + total gradient w.r.t. the normals
            gn_total = torch.empty([S, 3], dtype=torch.float32, device=dev)
            _lib.call("nsim_field_bwd_rad", fm, _lib.ptr(wpack), _lib.ptr(nab_fwd.detach()), _lib.ptr(rgb_fwd.detach()),
                      _lib.ptr(x), _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(t), _lib.ptr(ridx), _lib.ptr(ha), S,
                      _lib.ptr(gn), _lib.ptr(gr), _lib.ptr(gn_total), _lib.ptr(drad_w), _lib.ptr(drad_b), _lib.ptr(dha),
                      _lib.ptr(dx), _lib.ptr(dv))
        NLP = model.plane_levels
        need_pl = dgrid is not None or (need_dx and gn_total is not None)
        dh_pl = torch.empty([NLP, S, 2], dtype=torch.float32, device=dev) if need_pl else None
        g_pl = torch.empty([NLP, S, 2], dtype=torch.float32, device=dev) if need_pl else None
        # (2) SDF-decoder branch on the saved planes
        if model.pos_embed_E:
            if need_dx:
                raise NotImplementedError("pose refinement through a model with extra_pos_embed_cfg (nsim_wide_bwd_sdf has no "
                                          "dL/dx output)")
            _lib.call("nsim_wide_bwd_sdf", fm, _lib.ptr(wpack), _lib.ptr(x), _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(t),
                      _lib.ptr(ridx), S, _lib.ptr(h_pl), _lib.ptr(J_pl), int(ctx.PS), _lib.ptr(gs), _lib.ptr(gn_total),
                      _lib.ptr(dh_pl), _lib.ptr(g_pl), _lib.ptr(dsdf_w), _lib.ptr(dsdf_b))
        else:
            _lib.call("nsim_field_bwd_sdf", fm, _lib.ptr(wpack), _lib.ptr(h_pl), _lib.ptr(J_pl), S, _lib.ptr(gs),
                      _lib.ptr(gn_total), _lib.ptr(dh_pl), _lib.ptr(g_pl), _lib.ptr(dsdf_w), _lib.ptr(dsdf_b), _lib.ptr(dx),
                      int(ctx.PS))
        d_x = d_o = d_d = None
        if need_dx:
            if gn_total is not None:    # the normals' own dependence on x (mixed second derivatives of the interpolant)
                model._enc_hess_dx(grid16, x, rays_o, rays_d, t, ridx, ctx.goff, S, g_pl, gn_total, dx)
            if need_x:
                d_x = dx.reshape(ctx.x_shape)
            else:
                Rt = rays_o.shape[0]
                d_o = torch.zeros([Rt, 3], dtype=torch.float32, device=dev) if need_o else None
                d_d = torch.zeros([Rt, 3], dtype=torch.float32, device=dev) if need_d else None
                _lib.call("nsim_ray_grad_reduce", _lib.ptr(dx), _lib.ptr(dv), _lib.ptr(t), _lib.ptr(ridx), S,
                          _lib.ptr(d_o), _lib.ptr(d_d))
                if M > 0:       # the appended zero-length rays of the extra points are not the caller's
                    d_o = d_o[:Rt - M] if d_o is not None else None
                    d_d = d_d[:Rt - M] if d_d is not None else None
        if dgrid is not None:   # (3) scatter to the hash grid
            model._enc_scatter(x, rays_o, rays_d, t, ridx, ctx.goff, S, dh_pl, g_pl, gn_total, dgrid, enc_state=ctx.enc_state)
        if _lib.TIMER is not None:
            _lib.TIMER.note_units("nsim_field_bwd_sdf", S)
            if dgrid is not None:
                _lib.TIMER.note_units("nsim_lotd_scatter", S)
            if gr is not None:
                _lib.TIMER.note_units("nsim_field_bwd_rad", S)
        if model.sdf_scale != 1.0:      # d/d(head weights) of head / sdf_scale
            dsdf_w[-64:] /= model.sdf_scale
            dsdf_b[-1:] /= model.sdf_scale
        if dha is not None and M > 0:
            dha = dha[:dha.shape[0] - M]
        if gr is None:      # no consumer of the colour (with_rgb=False: lidar renders): the radiance network is not part of
            drad_w = drad_b = dha = None        # the graph -- None, not zeros, so that optimizers skip it as torch's do
        return (None, dgrid, dsdf_w, dsdf_b, drad_w, drad_b, dha, d_x, d_o, d_d, None, None, None, None, None, None)


class _NeusAlphaFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sdf, ln_inv_s, pack_infos, factor, forward_inv_s):
        sdf = sdf.float().contiguous()
        alpha = torch.empty_like(sdf)
        _lib.call("nsim_neus_alpha_fwd", _lib.ptr(sdf), _lib.ptr(pack_infos), pack_infos.shape[0], _lib.ptr(ln_inv_s),
                  float(factor), float(forward_inv_s), _lib.ptr(alpha))
        ctx.save_for_backward(sdf, ln_inv_s, pack_infos)
        ctx.factor, ctx.fis = float(factor), float(forward_inv_s)
        return alpha

    @staticmethod
    def backward(ctx, g):
        sdf, ln_inv_s, pack_infos = ctx.saved_tensors
        dsdf = torch.empty_like(sdf)
        dln = _lib.zeros(list(ln_inv_s.shape), device=ln_inv_s.device) if ln_inv_s.dtype == torch.float32 else torch.zeros_like(ln_inv_s)
        _lib.call("nsim_neus_alpha_bwd", _lib.ptr(sdf), _lib.ptr(g.float().contiguous()), _lib.ptr(pack_infos),
                  pack_infos.shape[0], _lib.ptr(ln_inv_s), ctx.factor, ctx.fis, _lib.ptr(dsdf), _lib.ptr(dln))
        return dsdf, dln, None, None, None


class _CompositeFn(torch.autograd.Function):
    """Fused ``SingleVolumeRenderer._volume_integration`` (single_volume_renderer.py:73-102); with ``out_idx`` / ``N`` the
    per-ray results are written straight into zero-filled all-rays images (the reference's
    ``rendered[k][rays_inds_hit] = ...``) by the same launch."""

    @staticmethod
    def forward(ctx, alpha, t, rgb, nrm, pack_infos, normalized_depth, out_idx=None, N=None):
        alpha = alpha.float().contiguous()
        t = t.float().contiguous()
        rgbc = rgb.float().contiguous() if rgb is not None else None
        nrmc = nrm.float().contiguous() if nrm is not None else None
        P = pack_infos.shape[0]
        dev = alpha.device
        vw = torch.empty_like(alpha)
        trans = torch.empty_like(alpha)
        f32 = dict(dtype=torch.float32, device=dev)
        if out_idx is None:
            mask, depth = torch.empty([P], **f32), torch.empty([P], **f32)
            # written for every pack when the corresponding input is present
            rgb_o = (torch.empty if rgbc is not None else torch.zeros)([P, 3], **f32)
            nrm_o = (torch.empty if nrmc is not None else torch.zeros)([P, 3], **f32)
        else:
            sc = _lib.zeros([2, N], device=dev)
            mask, depth = sc[0], sc[1]
            vec = _lib.zeros([2, N, 3], device=dev)
            rgb_o, nrm_o = vec[0], vec[1]
            out_idx = out_idx.contiguous()
        _lib.call("nsim_composite_fwd", _lib.ptr(alpha), _lib.ptr(t), _lib.ptr(rgbc), _lib.ptr(nrmc),
                  _lib.ptr(pack_infos), P, int(normalized_depth), _lib.ptr(vw), _lib.ptr(trans), _lib.ptr(mask),
                  _lib.ptr(depth), _lib.ptr(rgb_o), _lib.ptr(nrm_o), _lib.ptr(out_idx))
        ctx.save_for_backward(alpha, trans, vw, t, rgbc, nrmc, pack_infos, mask, depth, out_idx)
        ctx.nd = int(normalized_depth)
        ctx.mark_non_differentiable(trans)
        # an image nothing consumed (a photometric loss reads rgb alone) arrives as None in backward, not as a zero tensor the
        # engine fills first: the kernel takes NULL for an absent gradient (five fills per training step otherwise)
        ctx.set_materialize_grads(False)
        return vw, mask, depth, rgb_o, nrm_o, trans

    @staticmethod
    def backward(ctx, g_vw, g_mask, g_depth, g_rgb, g_nrm, _g_trans):
        alpha, trans, vw, t, rgbc, nrmc, pack_infos, mask, depth, out_idx = ctx.saved_tensors
        P = pack_infos.shape[0]

        def c(g):
            return g.float().contiguous() if g is not None else None
        dalpha = torch.empty_like(alpha)
        drgb = torch.empty_like(rgbc) if rgbc is not None else None
        dnrm = torch.empty_like(nrmc) if nrmc is not None else None
        _lib.call("nsim_composite_bwd", _lib.ptr(alpha), _lib.ptr(trans), _lib.ptr(vw), _lib.ptr(t), _lib.ptr(rgbc),
                  _lib.ptr(nrmc), _lib.ptr(pack_infos), P, ctx.nd, _lib.ptr(mask), _lib.ptr(depth), _lib.ptr(c(g_mask)),
                  _lib.ptr(c(g_depth)), _lib.ptr(c(g_rgb)), _lib.ptr(c(g_nrm)), _lib.ptr(c(g_vw)), _lib.ptr(dalpha),
                  _lib.ptr(drgb), _lib.ptr(dnrm), _lib.ptr(out_idx))
        return dalpha, None, drgb, dnrm, None, None, None, None


def volume_integration(alpha, t, rgb, nablas, pack_infos, depth_use_normalized_vw=False, rays_inds=None,
                       num_rays: int = None) -> Dict[str, torch.Tensor]:
    """``rays_inds`` [P] + ``num_rays``: results as zero-filled [num_rays, ...] images with pack p at row rays_inds[p]."""
    vw, mask, depth, rgb_o, nrm_o, trans = _CompositeFn.apply(alpha, t, rgb, nablas, pack_infos, depth_use_normalized_vw,
                                                             rays_inds, num_rays)
    out = dict(vw=vw, mask_volume=mask, depth_volume=depth, trans=trans)
    if rgb is not None:
        out["rgb_volume"] = rgb_o
    if nablas is not None:
        out["normals_volume"] = nrm_o
    return out


# ------------------------------------------------------------------------------------------ occupancy
class OccGridAccel(nn.Module):
    """``accel_cfg{type: occ_grid, resolution, occ_val_fn_cfg{type: sdf, inv_s}, occ_thre, ema_decay,
    init_cfg/update_from_net_cfg{num_steps,num_pts}, n_steps_between_update, n_steps_warmup}``
    (lotd_neus.dtu.230814.yaml:140-155; nr3d_lib.models.accelerations.OccGridAccel / OccGridEma)."""

    def __init__(self, aabb: torch.Tensor, resolution=(64, 64, 64), occ_thre=0.3, ema_decay=0.95, inv_s=256.0,
                 num_steps=4, num_pts=2 ** 20, n_steps_between_update=16, n_steps_warmup=256, device=None,
                 update_from_samples_cfg: Optional[dict] = None, init_cfg: Optional[dict] = None):
        super().__init__()
        self.init_cfg = dict(init_cfg or {})         # ``init_cfg{mode: from_net, num_steps, num_pts}``: sizes of ``init``
        # ``update_from_samples_cfg: {}`` (dtu yaml:158): the SDFs of a training step's sampling pass are max-folded into
        # the value grid as they are computed (``collect``); the periodic refresh then decays, adds its random queries
        # and re-thresholds.  None = off.  (Implementation in the absent nr3d_lib: semantics fixed here, oracle/render.py
        # ``occ_collect``.)  ``collect_armed`` is set by the model's ``training_before_per_step`` for ONE sampling pass.
        self.update_from_samples_cfg = None if update_from_samples_cfg is None else dict(update_from_samples_cfg)
        self.collect_armed = False
        # data parallel: ranks render different rays, so their collected values differ -- a trainer sets ``sync_values``
        # (all-reduce MAX of the 1 MB value grid) and every refresh starts from the union (the
        # reference's DDP re-broadcasts rank 0's buffers instead)
        self.sync_values = None
        self.resolution = [int(r) for r in resolution]
        self.occ_thre, self.ema_decay, self.inv_s = float(occ_thre), float(ema_decay), float(inv_s)
        self.num_steps, self.num_pts = int(num_steps), int(num_pts)
        self.n_steps_between_update, self.n_steps_warmup = n_steps_between_update, n_steps_warmup
        nvox = self.resolution[0] * self.resolution[1] * self.resolution[2]
        self.register_buffer("aabb", aabb.float().reshape(2, 3).clone())
        self.register_buffer("occ_val", torch.zeros(nvox, dtype=torch.float32))
        # the bit-packed grid the marching kernels read (uint32 words stored as int32)
        self.register_buffer("occ_bits", torch.full([(nvox + 31) // 32], -1, dtype=torch.int32))
        if device is not None:
            self.to(device)
        self.meta = self._make_meta()

    def _make_meta(self):
        m = _lib.OccMeta()
        a = self.aabb.detach().cpu()
        res = torch.tensor(self.resolution, dtype=torch.float32)
        scale = res / (a[1] - a[0])
        for i in range(3):
            m.aabb_min[i] = float(a[0, i])
            m.aabb_max[i] = float(a[1, i])
            m.scale[i] = float(scale[i])
            m.res[i] = self.resolution[i]
        return m

    @property
    def occ_grid(self) -> torch.Tensor:
        """bool [X,Y,Z] view of the occupancy (``accel.occ.occ_grid``, code_single/tools/extract_occgrid.py:108)."""
        r = self.resolution
        return (self.occ_val > self.occ_thre).view(r[2], r[1], r[0]).permute(2, 1, 0)

    def frac_occupied(self) -> float:
        return float((self.occ_val > self.occ_thre).float().mean())

    def set_all_occupied(self):
        self.occ_val.fill_(1.0)
        self.pack_bits()

    def pack_bits(self):
        _lib.call("nsim_occ_pack_bits", _lib.ptr(self.occ_val), self.occ_val.shape[0], self.occ_thre,
                  _lib.ptr(self.occ_bits))

    @torch.no_grad()
    def draw_points(self, n: int, generator=None) -> torch.Tensor:
        """The n query points of one refresh pass: point i lies in voxel (sweep + i) mod n_voxels -- voxels in storage
        order (x fastest), the sweep continuing where the previous pass stopped -- at a uniformly random offset inside
        the voxel: a STRATIFIED uniform draw (every voxel gets n / n_voxels points per pass, 4 at 2^20 points on 64^3)
        in a spatially coherent order.  A pass over 2^20 i.i.d. points costs 0.65 ms on MI355X (every hash-table read of
        every point is a cache miss), the same number of voxel-ordered points 0.385 ms (tools/refresh_probe.py).
        ``NSIM_OCC_IID=1``: i.i.d. uniform points in the box.  (The sampler of the absent nr3d_lib is not known; both
        draws are uniform over the box.)"""
        dev = self.occ_val.device
        lo, hi = self.aabb[0], self.aabb[1]
        if os.environ.get("NSIM_OCC_IID", "0") == "1":
            return lo + torch.rand([n, 3], device=dev, generator=generator) * (hi - lo)
        rx, ry, rz = self.resolution
        nvox = rx * ry * rz
        sweep = getattr(self, "_sweep", 0) % nvox
        self._sweep = (sweep + n) % nvox
        key = (n, sweep, str(dev), self.aabb._version)
        cache = getattr(self, "_corner_cache", None)
        if cache is None or cache[0] != key:
            v = (torch.arange(n, device=dev) + sweep) % nvox
            ijk = torch.stack([v % rx, (v // rx) % ry, v // (rx * ry)], dim=-1).float()
            cell = (hi - lo) / torch.tensor([rx, ry, rz], dtype=torch.float32, device=dev)
            cache = self._corner_cache = (key, lo + ijk * cell, cell)       # min corner of each point's voxel
        return torch.addcmul(cache[1], torch.rand([n, 3], device=dev, generator=generator), cache[2])

    def update_from_net(self, query_sdf, num_steps=None, num_pts=None, generator=None):
        """EMA-max refresh from random SDF queries (``init_cfg`` / ``update_from_net_cfg``)."""
        if self.update_from_samples_cfg is not None and self.sync_values is not None:
            self.sync_values(self.occ_val)
        for _ in range(num_steps or self.num_steps):
            pts = self.draw_points(num_pts or self.num_pts, generator)
            self.update_from_samples(pts, query_sdf(pts), pack=False)
        self.pack_bits()

    @torch.no_grad()
    def update_from_samples(self, pts, sdf, pack=True):
        pts = pts.detach().float().contiguous()
        sdf = sdf.detach().float().contiguous()
        nvox = self.occ_val.shape[0]
        _lib.call("nsim_occ_decay", _lib.ptr(self.occ_val), nvox, self.ema_decay)
        _lib.call("nsim_occ_update", _lib.ptr(self.occ_val), _lib.ptr(pts), _lib.ptr(sdf), pts.shape[0], self.meta,
                  self.inv_s)
        if pack:
            self.pack_bits()

    @torch.no_grad()
    def collect(self, pts: torch.Tensor, sdf: torch.Tensor, n_dev: torch.Tensor = None, n_add: int = 0):
        """Fold the (pts [n,3], sdf [n]) of a sampling-pass query into the value grid (max, no decay, bits untouched)."""
        _lib.call("nsim_occ_collect", _lib.ptr(self.occ_val), _lib.ptr(pts), _lib.ptr(sdf), sdf.shape[0], _lib.ptr(n_dev),
                  int(n_add), self.meta, self.inv_s)

    def init(self, query_sdf, logger=None, **kw):
        self.occ_val.zero_()
        kw.setdefault("num_steps", self.init_cfg.get("num_steps"))
        kw.setdefault("num_pts", self.init_cfg.get("num_pts"))
        self.update_from_net(query_sdf, **kw)

    def cur_batch__step(self, it: int, query_sdf, generator=None):
        """``training_before_per_step`` hook (app/resources/asset_bank.py:291-298).  ``generator``: a rank-shared
        generator keeps data-parallel replicas of the grid identical."""
        if getattr(self, "_last_step_it", None) == it:      # trainer and model hook may both call in: once per iteration
            return
        self._last_step_it = it
        if it >= self.n_steps_warmup and it % self.n_steps_between_update == 0:
            self.update_from_net(query_sdf, generator=generator)


# ---------------------------------------------------------------------------------------------- model
class LoTDNeuSModel(ModelMixin, nn.Module):
    is_ray_query_supported = True

    def __init__(self, lod_res: Sequence[int] = None, log2_hashmap_size: int = 19, sdf_D: int = 2, W: int = 64,
                 precision: str = "fp16", softplus_beta: float = 100.0, ln_inv_s_init: float = 0.1,
                 ln_inv_s_factor: float = 10.0, bounding_size: float = 2.0, aabb: torch.Tensor = None,
                 accel_cfg: dict = None, ray_query_cfg: dict = None, param_bound: float = 1e-4, seed: int = 42,
                 sdf_scale: float = 1.0, inside_out: bool = False, device=None, pos_embed_frequencies: int = None,
                 **reference_params):
        """``sdf_scale``: the decoder output is divided by it (street config ``sdf_scale: 25``,
        withmask_withlidar_joint.240219.yaml:158); ``inside_out``: sign of the geometric initialisation (indoor config
        ``inside_out: true``, lotd_neus.replica.230814.yaml:95).  Both live in the absent nr3d_lib -- semantics fixed
        here: sdf = head(h) / sdf_scale (folded into the packed head weights), inside_out => initial sdf = r - |x|.

        ``pos_embed_frequencies`` N (``surface_cfg.extra_pos_embed_cfg{type: sinusoidal_legacy, n_frequencies: N}``,
        no_fg_occ.221218.yaml:319-321): the SDF decoder reads [features | x_n, sin(2^k x_n), cos(2^k x_n), k < N] (x_n = the
        AABB-normalised position); the first layer of the MFMA decoders then contracts over two more 32-input chunks holding
        the block (csrc/field.hip NE = 2: ``nsim_field_fwd`` on the planes, ``nsim_wide_bwd_sdf``); the no-grad query of the
        sampling pass runs in f32 on csrc/wide_field.hip -- gather, radiance backward and scatter stay the common path.

        ``reference_params``: the reference's ``model_params`` block passed verbatim, as
        ``import_str(model_class)(**model_params, device=device)`` does (app/resources/asset_bank.py:129-138) --
        ``dtype, var_ctrl_cfg, cos_anneal_cfg, use_tcnn_backend, surface_cfg, radiance_cfg`` next to ``accel_cfg`` /
        ``ray_query_cfg`` (fields/ref_config.py).  Blocks that size themselves from the AABB (``lotd_use_cuboid``,
        ``accel_cfg.vox_size``: the street model) are built by ``populate(aabb=...)``, as in the reference
        (app/models/single/neus.py:152-196)."""
        if reference_params:
            from . import ref_config
            params = dict(reference_params, accel_cfg=accel_cfg, ray_query_cfg=ray_query_cfg)
            if aabb is None and ref_config.neus_needs_aabb(params):
                nn.Module.__init__(self)
                self._deferred_params, self._deferred_seed = params, seed
                self._pending_device = device
                return
            kw, post = ref_config.neus_native_kwargs(params, aabb=aabb)
            LoTDNeuSModel.__init__(self, seed=seed, device=device, **kw)
            self._reference_post = post
            self.geo_init_method = post.get("geo_init_method", self.geo_init_method)
            if "var_ctrl" in post:
                self.set_var_ctrl(**post["var_ctrl"])
            return
        super().__init__()
        self._deferred_params = None
        assert W == 64 and sdf_D in (1, 2), "gfx950 fused kernels: hidden width 64, 1 or 2 hidden SDF layers"
        self.sdf_scale, self.inside_out = float(sdf_scale), bool(inside_out)
        lod_res = list(lod_res) if lod_res is not None else list(DEFAULT_LOD_RES)
        assert 1 <= len(lod_res) <= 32, "gfx950 decoder kernels: up to 32 levels x 2 features (64 decoder inputs)"
        # the level-major planes hold 16 levels per feature chunk of the decoder's first layer (csrc/field.hip: NC)
        self.plane_levels = 16 if len(lod_res) <= 16 else 32
        self.sdf_D, self.ln_inv_s_factor = sdf_D, float(ln_inv_s_factor)
        self.pos_embed_n = None if pos_embed_frequencies is None else int(pos_embed_frequencies)
        self.pos_embed_E = 0 if self.pos_embed_n is None else 3 + 6 * self.pos_embed_n
        assert self.pos_embed_E <= 63 and 2 * len(lod_res) + self.pos_embed_E <= 128, "first layers up to 64 features + 63 embedded values"
        if self.pos_embed_E:
            self.planes_always = True       # the decoder with the embedded-position block reads the level-major planes in every mode
        self.encoding = LoTDEncoding(LoTDConfig(lod_res, 2, log2_hashmap_size), bound=param_bound, seed=seed)
        n_sdf_w, n_sdf_b, n_rad_w, n_rad_b = _flat_sizes(sdf_D, len(lod_res), self.pos_embed_E)
        g = torch.Generator().manual_seed(seed + 1)

        def lin(o, i, scale=1.0):
            b = 1.0 / math.sqrt(i)
            return ((torch.rand(o, i, generator=g) * 2 - 1) * b * scale, (torch.rand(o, generator=g) * 2 - 1) * b * scale)
        ws, bs = [], []
        dims = [2 * len(lod_res) + self.pos_embed_E] + [64] * sdf_D + [1]
        for li in range(len(dims) - 1):
            w, b = lin(dims[li + 1], dims[li])
            ws.append(w.reshape(-1))
            bs.append(b)
        self.sdf_w = nn.Parameter(torch.cat(ws))
        self.sdf_b = nn.Parameter(torch.cat(bs))
        ws, bs = [], []
        rdims = [RAD_IN, 64, 64, 3]
        for li in range(3):
            w, b = lin(rdims[li + 1], rdims[li])
            ws.append(w.reshape(-1))
            bs.append(b)
        self.rad_w = nn.Parameter(torch.cat(ws))
        self.rad_b = nn.Parameter(torch.cat(bs))
        self.ln_inv_s = nn.Parameter(torch.tensor([float(ln_inv_s_init)]))
        # ``implicit_surface.is_pretrained`` / ``.geo_init_method`` as the reference's asset classes read them
        # (app/models/single/neus.py:203-206, 229)
        self.register_buffer("is_pretrained", torch.tensor([False]), persistent=True)
        self.geo_init_method = "pretrain_after_zero_out"
        assert self.sdf_w.numel() == n_sdf_w and self.rad_w.numel() == n_rad_w
        if aabb is None:
            h = bounding_size / 2.0
            aabb = torch.tensor([[-h, -h, -h], [h, h, h]])
        self.encoding.cfg.set_aabb(aabb)           # the pyramid spans the AABB (per axis); must precede fm.lotd = meta below
        accel_cfg = dict(accel_cfg) if accel_cfg is not None else dict(update_from_samples_cfg={})     # dtu yaml:140-160
        self.accel = OccGridAccel(aabb, resolution=accel_cfg.get("resolution", (64, 64, 64)),
                                  occ_thre=accel_cfg.get("occ_thre", 0.3), ema_decay=accel_cfg.get("ema_decay", 0.95),
                                  inv_s=accel_cfg.get("occ_val_fn_cfg", {}).get("inv_s", 256.0),
                                  num_steps=accel_cfg.get("update_from_net_cfg", {}).get("num_steps", 4),
                                  num_pts=accel_cfg.get("update_from_net_cfg", {}).get("num_pts", 2 ** 20),
                                  n_steps_between_update=accel_cfg.get("n_steps_between_update", 16),
                                  n_steps_warmup=accel_cfg.get("n_steps_warmup", 256),
                                  update_from_samples_cfg=accel_cfg.get("update_from_samples_cfg", None),
                                  init_cfg=accel_cfg.get("init_cfg", None))
        self.ray_query_cfg = dict(ray_query_cfg or dict(
            query_mode="march_occ_multi_upsample_compressed",     # the reference's default (dtu yaml:157)
            query_param=dict(nablas_has_grad=True, num_coarse=64, num_fine=[8, 8, 32], upsample_inv_s=64.0,
                             upsample_inv_s_factors=[1, 4, 16], upsample_use_estimate_alpha=True,
                             march_cfg=dict(step_size=0.005, max_steps=4096))))
        fm = _lib.FieldMeta()
        fm.lotd = self.encoding.cfg.meta
        fm.sdf_D = sdf_D
        fm.embed_E = self.pos_embed_E       # the first layer's embedded-position block (two more MFMA input chunks): csrc/field.hip NE
        fm.precision = {"fp16": 0, "f32": 1}[precision]
        # ``decoder_cfg.activation``: softplus(beta) (the single-object / street configs) or relu (the Vehicle decoder of
        # no_fg_occ.221218.yaml:354-357) -- a non-positive beta selects relu in the kernels
        fm.softplus_beta = float(softplus_beta) if float(softplus_beta) > 0 else -1.0
        self.sdf_activation = "softplus" if float(softplus_beta) > 0 else "relu"
        self.field_meta = fm
        self._wpack = None
        self._sdf_fused = os.environ.get("NSIM_SDF_FUSED", "0") == "1" and self.plane_levels == 16
        # size the sampling buffers from the previous step's density instead of reading the marched total back
        self._speculate = os.environ.get("NSIM_SPECULATE", "1") == "1" and not self._sdf_fused
        self._wpack_versions = None
        # precision of the SAMPLING pass's no-grad SDF queries (``_sampling_ctx``): None = the field precision
        # "split" (default): the discrete decisions of a step follow f32-equivalent arithmetic whatever the field precision
        self.sampling_precision = os.environ.get("NSIM_SAMPLING_PRECISION") or "split"
        if device is not None:
            self.to(device)

    # ------------------------------------------------------------------ reference life cycle
    def populate(self, aabb: torch.Tensor = None, device=None, **unused):
        """``model.populate(...)`` of the reference's asset mixin (app/models/single/neus.py:55-59 ``populate(device=)``,
        :152-196 ``populate(aabb=, device=)`` for the street model): builds a model whose ``model_params`` needed the
        AABB, moves it to the device."""
        if getattr(self, "_deferred_params", None) is not None:
            assert aabb is not None, "this model_params block sizes itself from the AABB: populate(aabb=...)"
            params, seed = self._deferred_params, self._deferred_seed
            device = device if device is not None else self._pending_device
            acc, rq = params.pop("accel_cfg", None), params.pop("ray_query_cfg", None)
            LoTDNeuSModel.__init__(self, aabb=torch.as_tensor(aabb, dtype=torch.float32).cpu(), seed=seed, accel_cfg=acc,
                                   ray_query_cfg=rq, **params)
        elif aabb is not None:
            assert torch.allclose(torch.as_tensor(aabb, dtype=torch.float32).cpu(), self.accel.aabb.cpu()), \
                "populate(aabb=...) differs from the AABB the model was built with"
        if device is not None:
            self.to(device)
        return self

    def pretrain_sdf_fn(self, target_fn, num_iters: int = 300, lr: float = 2e-3, num_pts: int = 2 ** 14, seed: int = 0,
                        logger=None, w_eikonal: float = 0.0) -> float:
        """The reference's SDF pre-training as an optimisation (``nr3d_lib.models.fields.sdf.pretrain_sdf_*``, called with
        ``initialize_cfg{num_iters, lr, ...}`` by app/models/single/neus.py:198-236): fit the SDF to ``target_fn(x)``
        (x [N,3] object coordinates on the model's device -> [N]) at uniformly drawn points of the box with Adam over the
        encoding and the SDF decoder, through the model's own forward / backward kernels.  ``w_eikonal`` > 0 adds
        w (|grad sdf| - 1)^2 on the same points (the reference's ``initialize_cfg.w_eikonal``): a hashed encoding fitted
        by values alone ends up rough at the scale of its finest levels.  Returns the last L1 loss."""
        dev = self.encoding.flattened_params.device
        params = [self.encoding.flattened_params, self.sdf_w, self.sdf_b]
        opt = torch.optim.Adam(params, lr=lr)
        g = torch.Generator(device=dev).manual_seed(seed)
        lo, hi = self.accel.aabb[0].to(dev), self.accel.aabb[1].to(dev)
        loss = torch.zeros([])
        with torch.enable_grad():
            for it in range(int(num_iters)):
                x = lo + (hi - lo) * torch.rand([num_pts, 3], device=dev, generator=g)
                with torch.no_grad():
                    target = target_fn(x).to(dev).float()
                out = self.forward_sdf_nablas(x, nablas_has_grad=w_eikonal > 0)
                loss = (out["sdf"] - target).abs().mean()
                total = loss if w_eikonal <= 0 else loss + w_eikonal * ((out["nablas"].norm(dim=-1) - 1.0) ** 2).mean()
                opt.zero_grad(set_to_none=True)
                total.backward()
                opt.step()
                self._wpack_versions = None
                if logger is not None and it % 100 == 0:
                    logger.info(f"pretrain_sdf: it {it} loss {float(loss.detach()):.5f}")
        for q in params:
            q.grad = None
        self.is_pretrained.fill_(True)
        return float(loss.detach())

    def pretrain_sdf_sphere(self, radius: float = 0.5, num_iters: int = 300, lr: float = 2e-3, num_pts: int = 2 ** 14,
                            seed: int = 0, logger=None, w_eikonal: float = 0.0) -> float:
        """``pretrain_sdf_sphere``: target |u| - radius, u = the AABB-normalised position (r - |u| for ``inside_out``)."""
        lo, hi = self.accel.aabb[0], self.accel.aabb[1]
        sign = -1.0 if self.inside_out else 1.0

        def target(x):
            u = (x - (lo.to(x.device) + hi.to(x.device)) * 0.5) / ((hi.to(x.device) - lo.to(x.device)) * 0.5)
            return sign * (u.norm(dim=-1) - radius)
        return self.pretrain_sdf_fn(target, num_iters=num_iters, lr=lr, num_pts=num_pts, seed=seed, logger=logger,
                                    w_eikonal=w_eikonal)

    @torch.no_grad()
    def training_initialize(self, config=None, logger=None, log_prefix=None) -> bool:
        """``asset_training_initialize`` -> ``training_initialize`` (app/models/single/neus.py:61-64, :198-236): the
        geometric initialisation named by ``surface_cfg.geo_init_method`` / ``radius_init`` (here the deterministic
        sphere of ``geometric_init_sphere`` instead of ``initialize_cfg.num_iters`` pre-training steps) followed by
        ``accel.init(self.query_sdf)``.  Returns True when the weights were (re-)initialised."""
        post = getattr(self, "_reference_post", {})
        updated = False
        method = post.get("geo_init_method", "pretrain_after_zero_out")
        if ("pretrain" in method) and not bool(self.is_pretrained):
            if "zero_out" in method:
                self.encoding.flattened_params.zero_()
            ext = (self.accel.aabb[1] - self.accel.aabb[0]).cpu()
            # radius_init is in object units; the sphere is written in the [-1, 1] coordinates of the shortest axis
            r = float(post.get("radius_init", 0.5)) / (float(ext.min()) / 2.0)
            cfg_i = dict(config or {})
            if cfg_i.get("geo_init_impl", os.environ.get("NSIM_GEO_INIT", "write")) == "pretrain":
                # the reference's own procedure: ``initialize_cfg{num_iters, lr}`` optimisation steps on the zeroed table
                self.pretrain_sdf_sphere(min(r, 0.95), num_iters=int(cfg_i.get("num_iters", 500)), lr=float(cfg_i.get("lr", 2e-3)),
                                         num_pts=int(cfg_i.get("num_pts", 2 ** 14)), logger=logger)
            else:       # default: the deterministic write (exact sphere, no iterations)
                self.geometric_init_sphere(min(r, 0.95), noise_scale=0.25, level=self._geo_init_level())
            self.is_pretrained.fill_(True)
            updated = True
        if self.accel is not None:
            self.accel.init(self.query_sdf, logger=logger)
        an = post.get("anneal")
        if an is not None:
            self.anneal_levels(0, **an)
        return updated

    # ------------------------------------------------------------------ bookkeeping
    @property
    def device(self):
        return self.sdf_w.device

    @property
    def space_aabb(self):
        return self.accel.aabb

    @property
    def space(self) -> AABBSpace:
        """``model.space`` (``.aabb``, ``.get_bounding_volume()``, ``.ray_test``): app/resources/nodes.py:92-103,
        app/models/single/nerf.py:175."""
        a = self.accel.aabb
        key = (a.data_ptr(), a._version, str(a.device))       # no value comparison: that would be a device->host sync
        sp = getattr(self, "_space", None)
        if sp is None or sp[0] != key:
            from ..spatial import AABBSpace
            sp = (key, AABBSpace(aabb=a.detach().clone(), device=a.device))
            object.__setattr__(self, "_space", sp)          # a view of the model's box, not a registered sub-module
        return sp[1]

    # ------------------------------------------------------------------ optimizer (model_base.ModelMixin)
    def _param_groups(self, cfg: dict):
        """``training_cfg{lr, eps, betas, invs_betas}`` (lotd_neus.dtu.230814.yaml:178-185): the table (with its fp16
        shadow written by the same Adam pass), the two decoders, and ``ln_inv_s`` with its own betas."""
        enc = self.encoding
        enc.shadow()
        return [dict(name="implicit_surface.encoding", params=[enc.flattened_params], shadow16=lambda: enc.shadow()),
                dict(name="implicit_surface.decoder", params=[self.sdf_w, self.sdf_b]),
                dict(name="radiance_net", params=[self.rad_w, self.rad_b]),
                dict(name="ln_inv_s", params=[self.ln_inv_s], betas=tuple(cfg.get("invs_betas", (0.9, 0.999))))]

    def _after_optimizer_step(self):
        self._wpack_versions = None         # MLP weights changed in place: re-pack the MFMA fragments lazily

    def _weight_reg_tensors(self):
        return [self.sdf_w, self.rad_w]

    def set_precision(self, precision: str):
        self.field_meta.precision = {"fp16": 0, "f32": 1}[precision]
        self._wpack_versions = None

    def set_active_levels(self, n: Optional[int]):
        """Hardmask level annealing state (``encoding_cfg.anneal_cfg{type: hardmask}``, lotd_neus.dtu.230814.yaml:104-108):
        only the first n levels are read / trained; None or >= 16 = all."""
        self.encoding.cfg.set_active_levels(n)
        self.field_meta.lotd.n_active_levels = self.encoding.cfg.meta.n_active_levels

    def _geo_init_level(self) -> Optional[int]:
        """The level a deterministic geometric initialisation is written to: the finest dense level that is ACTIVE at
        iteration 0.  With ``encoding_cfg.anneal_cfg{type: hardmask, start_level}`` (lotd_neus.dtu.230814.yaml:104-108,
        withmask_withlidar_joint.240219.yaml:168-172) only levels <= start_level are read when training starts: a sphere /
        road written into a finer level would be invisible until the annealing reaches it (the reference pre-trains THROUGH
        the mask -- ``pretrain_after_zero_out`` -- so its initial geometry lives in the active levels as well).  None: no
        annealing configured, the finest dense level."""
        an = (getattr(self, "_reference_post", None) or {}).get("anneal")
        if an is None:
            return None
        cfg = self.encoding.cfg
        dense = [l for l, t in enumerate(cfg.lod_types) if t == "Dense" and l <= int(an.get("start_level", 2))]
        return max(dense) if dense else None

    def anneal_levels(self, it: int, start_it: int = 0, stop_it: int = 1000, start_level: int = 2):
        """Level l is active once it >= start_it + (l - start_level) / (L - 1 - start_level) * (stop_it - start_it)
        (levels <= start_level from the start, all L by stop_it).  Returns the number of active levels."""
        L = self.encoding.cfg.num_levels
        r = min(max((it - start_it) / max(1, stop_it - start_it), 0.0), 1.0)
        n = int(math.floor(start_level + r * (L - 1 - start_level) + 1e-9)) + 1
        self.set_active_levels(n)
        return n

    def _pack_for(self, fm, slot: str):
        """MFMA-fragment weight pack for the kernels selected by ``fm`` (its precision), cached in ``slot`` and refreshed
        lazily when a parameter changed in place."""
        vers = (self.sdf_w._version, self.sdf_b._version, self.rad_w._version, self.rad_b._version,
                fm.precision, str(self.sdf_w.device), self.sdf_scale)
        cur = getattr(self, slot, None)
        if cur is None or cur[0] != vers:
            lib = _lib.get_lib()
            nbytes = int(lib.nsim_field_wpack_bytes(fm))
            buf = cur[1] if cur is not None else None
            if buf is None or buf.numel() != nbytes or buf.device != self.sdf_w.device:
                buf = torch.zeros([nbytes], dtype=torch.uint8, device=self.sdf_w.device)
            sw, sb = self.sdf_w.detach(), self.sdf_b.detach()
            if self.sdf_scale != 1.0:       # sdf = head / sdf_scale: fold the divisor into the packed head weights
                sw, sb = sw.clone(), sb.clone()
                sw[-64:] /= self.sdf_scale
                sb[-1:] /= self.sdf_scale
            _lib.call("nsim_field_pack_weights", fm, _lib.ptr(sw), _lib.ptr(sb),
                      _lib.ptr(self.rad_w.detach()), _lib.ptr(self.rad_b.detach()), _lib.ptr(buf))
            object.__setattr__(self, slot, (vers, buf))
        return getattr(self, slot)[1]

    def _per_ray_condition(self) -> bool:
        """True when the encoding kernels read a per-RAY condition (the conditioned permutohedral model): the sampling pass
        then hands them the ray index of every sample."""
        return False

    def _table(self) -> torch.Tensor:
        """The flat f32 table the with-grad query differentiates (a model whose tables are GROWN from latents returns the
        generated tensor of the current condition: fields/batched_neus.py)."""
        return self.encoding.flattened_params

    def _table16(self) -> torch.Tensor:
        """The fp16 copy of ``_table()`` the gather kernels read."""
        return self.encoding.shadow()

    def _weight_pack(self):
        """The MFMA-fragment weight pack alone (the backward of a query made under a condition that has been cleaned since
        -- grown / conditioned tables -- needs the decoders' weights, not the current table)."""
        self._invalidate_packs()
        self._wpack = self._pack_for(self.field_meta, "_wpack_slot")
        return self._wpack

    def _invalidate_packs(self):
        """After an invalidation by hand (optimizer step, precision switch): the packs are stale, their BUFFERS stay (a fresh
        ``torch.zeros`` per step and pack was two fill launches of host pace each)."""
        if self._wpack_versions is None:
            for slot in ("_wpack_slot", "_wpack_slot_s"):
                cur = getattr(self, slot, None)
                object.__setattr__(self, slot, (None, cur[1]) if cur is not None else None)
            self._wpack_versions = True

    def _shadow(self):
        """(fp16 grid shadow, MFMA-fragment weight pack), refreshed lazily when a parameter changed in place."""
        grid16 = self._table16()
        self._invalidate_packs()
        self._pack_pair()
        self._wpack = self._pack_for(self.field_meta, "_wpack_slot")
        return grid16, self._wpack

    def _pack_pair(self):
        """Both packs of a model whose sampling pass runs at another precision than its field are stale after every optimizer
        step: re-pack them in ONE launch (nsim_field_pack_weights2) instead of one each."""
        sp, fm = self.sampling_precision, self.field_meta
        code = {"fp16": 0, "f32": 1, "split": 2}
        if sp is None or code[sp] == fm.precision or (sp == "split" and fm.precision == 1) or self.pos_embed_E or self.sdf_scale != 1.0:
            return
        a, b = getattr(self, "_wpack_slot", None), getattr(self, "_wpack_slot_s", None)
        if a is None or b is None or a[1] is None or b[1] is None:        # first use: the single-pack path allocates
            return
        dev = str(self.sdf_w.device)
        va = (self.sdf_w._version, self.sdf_b._version, self.rad_w._version, self.rad_b._version, fm.precision, dev, self.sdf_scale)
        vb = va[:4] + (code[sp], dev, self.sdf_scale)
        if a[0] == va or b[0] == vb or a[1].device != self.sdf_w.device or b[1].device != self.sdf_w.device:
            return
        fs = getattr(self, "_field_meta_s", None)
        if fs is None:
            return
        C_memmove(fs, fm)
        fs.precision = code[sp]
        lib = _lib.get_lib()
        if a[1].numel() != int(lib.nsim_field_wpack_bytes(fm)) or b[1].numel() != int(lib.nsim_field_wpack_bytes(fs)):
            return
        _lib.call("nsim_field_pack_weights2", fm, _lib.ptr(a[1]), fs, _lib.ptr(b[1]), _lib.ptr(self.sdf_w.detach()),
                  _lib.ptr(self.sdf_b.detach()), _lib.ptr(self.rad_w.detach()), _lib.ptr(self.rad_b.detach()))
        object.__setattr__(self, "_wpack_slot", (va, a[1]))
        object.__setattr__(self, "_wpack_slot_s", (vb, b[1]))

    def _sampling_ctx(self):
        """(FieldMeta, weight pack) of the SAMPLING pass's no-grad SDF queries.  ``sampling_precision = "f32"`` runs them
        on the exact-f32 kernels (f32 feature planes, v_mfma_f32_32x32x2_f32), ``"split"`` (default) on the f16 matrix
        cores with every operand carried as hi + lo (22 significant bits, three MFMAs per product: csrc/field.hip
        ``SPLIT_LO_SCALE``) -- f32-equivalent at a fraction of the f32 MFMA's cost -- while the with-grad query stays fp16: the
        up-sampler multiplies SDF differences by inv_s up to 1024 and the compressed mode thresholds visibility weights
        at 1e-4, so the DISCRETE decisions of a step (where fine samples land, which samples are kept) then follow the
        f32 arithmetic -- they are what the fp16 rounding of an SDF (half an ulp = 2.4e-4 at |sdf| in [0.5, 1)) moves.
        None / equal to the field precision: one pack, one meta."""
        sp = self.sampling_precision
        fm = self.field_meta
        code = {"fp16": 0, "f32": 1, "split": 2}
        if sp is None or code[sp] == fm.precision or (sp == "split" and fm.precision == 1):
            return fm, self._shadow()[1]
        fs = getattr(self, "_field_meta_s", None)
        if fs is None:
            fs = _lib.FieldMeta()
            object.__setattr__(self, "_field_meta_s", fs)
        C_memmove(fs, fm)
        fs.precision = code[sp]
        self._shadow()
        return fs, self._pack_for(fs, "_wpack_slot_s")

    @torch.no_grad()
    def geometric_init_sphere(self, radius: float = 0.5, noise_scale: float = 0.25, inside_out: bool = None,
                              level: int = None):
        """Deterministic stand-in for the reference's SDF pre-training (``geo_init_method: pretrain_after_zero_out``,
        ``radius_init`` -- lotd_neus.dtu.230814.yaml:125-126; app/models/single/neus.py:198-236): feature 0 of the
        finest dense level holds |x_vertex| - radius and units 0 / 1 of every hidden layer pass +-it through
        (softplus(s) - softplus(-s) == s), so the initial SDF is a (trilinearly sampled) sphere; the remaining
        weights keep a small random part so every gradient path is exercised."""
        cfg = self.encoding.cfg
        lv = max(l for l, t in enumerate(cfg.lod_types) if t == "Dense") if level is None else int(level)
        assert cfg.lod_types[lv] == "Dense", "the sphere is written to a dense level"
        Rx, Ry, Rz = cfg.lod_res3[lv]
        zz, yy, xx = torch.meshgrid(torch.linspace(-1.0, 1.0, Rz), torch.linspace(-1.0, 1.0, Ry),
                                    torch.linspace(-1.0, 1.0, Rx), indexing="ij")
        sdf = (torch.sqrt(xx ** 2 + yy ** 2 + zz ** 2) - radius).reshape(-1)
        if self.inside_out if inside_out is None else inside_out:
            sdf = -sdf                     # camera inside the surface (indoor scenes): positive inside the sphere
        sdf = sdf * self.sdf_scale         # the head divides by sdf_scale
        lvl = self.encoding.flattened_params.data[cfg.lod_offsets[lv]: cfg.lod_offsets[lv] + cfg.lod_sizes[lv] * 2].view(-1, 2)
        lvl[:, 0] = sdf.half().float().to(lvl.device)
        D = self.sdf_D
        F1 = 2 * cfg.num_levels + self.pos_embed_E      # (row width of W1: the embedded-position columns come last)
        # units 0 / 1 of every hidden layer carry +s / -s (s = the stored SDF): softplus(s) - softplus(-s) == s exactly, and
        # the activations are small near the surface, where the fp16 MFMA operands need their resolution (one unit
        # carrying s + 2 through softplus' linear region has an fp16 spacing of 2e-3: a staircase SDF)
        w1 = self.sdf_w.data[:64 * F1].view(64, F1)
        w1.mul_(noise_scale)
        w1[0].zero_()
        w1[1].zero_()
        w1[0, 2 * lv], w1[1, 2 * lv] = 1.0, -1.0
        self.sdf_b.data[:64].mul_(noise_scale)
        self.sdf_b.data[0:2] = 0.0
        if D == 2:
            w2 = self.sdf_w.data[64 * F1:64 * F1 + 4096].view(64, 64)
            w2.mul_(noise_scale)
            w2[0].zero_()
            w2[1].zero_()
            w2[0, 0], w2[0, 1], w2[1, 0], w2[1, 1] = 1.0, -1.0, -1.0, 1.0
            self.sdf_b.data[64:128].mul_(noise_scale)
            self.sdf_b.data[64:66] = 0.0
        wh = self.sdf_w.data[-64:]
        wh.mul_(noise_scale * 0.05)
        wh[0], wh[1] = 1.0, -1.0
        self.sdf_b.data[-1] = 0.0
        self.encoding.flattened_params.add_(0)      # bump versions: refresh fp16 shadow / weight pack lazily
        self.sdf_w.add_(0)
        return self

    @property
    def implicit_surface(self):
        """The reference's models keep the SDF network in ``self.implicit_surface``; here the model is its own."""
        return self

    @torch.no_grad()
    def geometric_init_fn(self, sdf_fn, noise_scale: float = 0.25, level: int = None):
        """Like ``geometric_init_sphere`` for an arbitrary target: ``sdf_fn(x [V,3] in OBJECT coordinates) -> [V]`` signed
        distance in object units, sampled at the vertices of the finest dense level (the road-surface / capsule targets
        of the street model's pre-training, app/models/single/neus.py:198-236)."""
        self.geometric_init_sphere(0.5, noise_scale=noise_scale, inside_out=False, level=level)   # decoder pass-through
        cfg = self.encoding.cfg
        lv = max(l for l, t in enumerate(cfg.lod_types) if t == "Dense") if level is None else int(level)
        Rx, Ry, Rz = cfg.lod_res3[lv]
        zz, yy, xx = torch.meshgrid(torch.linspace(-1.0, 1.0, Rz), torch.linspace(-1.0, 1.0, Ry),
                                    torch.linspace(-1.0, 1.0, Rx), indexing="ij")
        a = self.accel.aabb.detach().cpu()
        xn = torch.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], dim=-1)
        x_obj = a[0] + (xn + 1.0) * 0.5 * (a[1] - a[0])
        sdf = sdf_fn(x_obj).reshape(-1).float() * self.sdf_scale
        lvl = self.encoding.flattened_params.data[cfg.lod_offsets[lv]: cfg.lod_offsets[lv] + cfg.lod_sizes[lv] * 2].view(-1, 2)
        lvl[:, 0] = sdf.half().float().to(lvl.device)
        self.encoding.flattened_params.add_(0)
        return self

    # ------------------------------------------------------------------ inv_s control (var_ctrl_cfg)
    def set_var_ctrl(self, ctrl_type: Optional[str] = "mix_linear", start_it: int = 0, stop_it: int = 1,
                     final_inv_s: float = 2000.0):
        """``var_ctrl_cfg{ctrl_type: mix_linear, start_it, stop_it, final_inv_s}`` (lotd_neus.dtu.230814.yaml:83-89):
        from ``start_it`` to ``stop_it`` the effective inv_s moves linearly from the learned ``exp(ln_inv_s * factor)``
        to ``final_inv_s``:  inv_s(it) = (1 - a) exp(ln_inv_s factor) + a final_inv_s,  a = clip((it - start) / (stop -
        start), 0, 1).  The controller lives in the absent nr3d_lib -- this blend is the semantics fixed here (parity
        unpinned).  The learned parameter keeps receiving (1 - a) of the gradient; no kernel knows about the schedule:
        the blend is a 1-element torch expression handed to the kernels in place of ``ln_inv_s``."""
        assert ctrl_type in (None, "mix_linear")
        self._var_ctrl = None if ctrl_type is None else dict(start_it=int(start_it), stop_it=int(stop_it),
                                                             final_inv_s=float(final_inv_s))
        self._ctrl_mix = 0.0

    def training_before_per_step(self, it: int, logger=None):
        """Per-iteration hook of the reference's trainer (app/resources/asset_bank.py:291-298): inv_s control here; the
        occupancy refresh (``accel.cur_batch__step``) and the level annealing are driven by the trainer."""
        if self.accel is not None and self.accel.update_from_samples_cfg is not None:
            self.accel.collect_armed = True          # the coming step's sampling pass feeds the occupancy values
        vc = getattr(self, "_var_ctrl", None)
        if vc is not None:
            span = max(vc["stop_it"] - vc["start_it"], 1)
            self._ctrl_mix = min(max((int(it) - vc["start_it"]) / span, 0.0), 1.0)
        post = getattr(self, "_reference_post", None)
        if post is not None:        # built from the reference's model_params: the model drives its own schedules
            if post.get("anneal") is not None:
                self.anneal_levels(int(it), **post["anneal"])
            self.accel.cur_batch__step(int(it), self.query_sdf, generator=getattr(self, "refresh_generator", None))

    def training_after_per_step(self, it: int, logger=None):
        """After ``optimizer.step`` (app/resources/asset_bank.py:300-308): nothing to do -- the fp16 table shadow and
        the packed MFMA weight fragments refresh lazily from the parameter versions."""

    def rendering_before_per_view(self, renderer=None, observer=None, per_frame_info: dict = None):
        """app/resources/asset_bank.py:310-318: no per-view state."""

    def model_setup(self):
        """app/resources/asset_bank.py:269-277: make the derived buffers current once (shadow + weight pack)."""
        self._shadow()

    def _ln_inv_s_eff(self) -> torch.Tensor:
        """The 1-element tensor the alpha / compress kernels read as ln_inv_s (autograd-transparent)."""
        a = getattr(self, "_ctrl_mix", 0.0)
        if a <= 0.0 or getattr(self, "_var_ctrl", None) is None:
            return self.ln_inv_s
        inv_s = (1.0 - a) * torch.exp(self.ln_inv_s * self.ln_inv_s_factor) + a * self._var_ctrl["final_inv_s"]
        return torch.log(inv_s) / self.ln_inv_s_factor

    def forward_inv_s(self) -> torch.Tensor:
        return torch.exp(self._ln_inv_s_eff() * self.ln_inv_s_factor)

    # ------------------------------------------------------------------ encoding hooks
    # The decoder kernels work on level-major planes of features / d features / d x; what fills the planes and what turns
    # the hand-off planes back into table gradients is the encoding's business.  LoTD: the fused entry points of field.hip
    # (gather inside nsim_field_fwd).  fields/permuto_neus.py overrides the four hooks with csrc/permuto.hip.
    planes_always = False       # True: even an evaluation forward goes through the planes (no fused point-major kernel)

    def _wide_weights(self):
        """(sdf_w, sdf_b) as the wide decoder reads them: the f32 masters with the ``sdf_scale`` divisor folded into the head."""
        sw, sb = self.sdf_w.detach(), self.sdf_b.detach()
        if self.sdf_scale != 1.0:
            sw, sb = sw.clone(), sb.clone()
            sw[-64:] /= self.sdf_scale
            sb[-1:] /= self.sdf_scale
        return sw.contiguous(), sb.contiguous()

    def _enc_field_fwd(self, grid16, wpack, x, rays_o, rays_d, t, ridx, goff, ha, S, sdf, nablas, rgb, h_pl, J_pl, n_dev,
                       n_add):
        # (a model with an embedded-position block, field_meta.embed_E > 0: same entry point -- the level-major gather, then the
        # decoder whose first layer also contracts over the block's two MFMA chunks)
        _lib.call("nsim_field_fwd", self.field_meta, _lib.ptr(grid16), _lib.ptr(wpack), _lib.ptr(x), _lib.ptr(rays_o),
                  _lib.ptr(rays_d), _lib.ptr(t), _lib.ptr(ridx), _lib.ptr(goff), _lib.ptr(ha), S, _lib.ptr(sdf),
                  _lib.ptr(nablas), _lib.ptr(rgb), _lib.ptr(h_pl), _lib.ptr(J_pl), _lib.ptr(n_dev), int(n_add))

    def _enc_gather_feat(self, fm, grid16, x, rays_o, rays_d, t, ridx, goff, S, n_dev, n_add, planes):
        _lib.call("nsim_lotd_gather_lm", fm, _lib.ptr(grid16), _lib.ptr(x), _lib.ptr(rays_o), _lib.ptr(rays_d),
                  _lib.ptr(t), _lib.ptr(ridx), _lib.ptr(goff), S, _lib.ptr(n_dev), int(n_add), _lib.ptr(planes))

    def _enc_scatter(self, x, rays_o, rays_d, t, ridx, goff, S, dh_pl, g_pl, gn_total, dgrid, enc_state=None):
        _lib.call("nsim_lotd_scatter", self.encoding.cfg.meta, _lib.ptr(x), _lib.ptr(rays_o), _lib.ptr(rays_d),
                  _lib.ptr(t), _lib.ptr(ridx), _lib.ptr(goff), S, _lib.ptr(dh_pl), _lib.ptr(g_pl),
                  _lib.ptr(gn_total), _lib.ptr(dgrid), 0, 0)

    def _enc_hess_dx(self, grid16, x, rays_o, rays_d, t, ridx, goff, S, g_pl, gn_total, dx):
        _lib.call("nsim_lotd_hess_dx", self.encoding.cfg.meta, _lib.ptr(grid16), _lib.ptr(x), _lib.ptr(rays_o),
                  _lib.ptr(rays_d), _lib.ptr(t), _lib.ptr(ridx), _lib.ptr(goff), S, _lib.ptr(g_pl),
                  _lib.ptr(gn_total), _lib.ptr(dx))

    # ------------------------------------------------------------------ point queries
    def _sdf_query(self, grid16, wpack, x, rays_o, rays_d, t, ridx, S: int, dev, goff=None, n_dev=None,
                   n_add: int = 0, collect: bool = False, fm=None) -> torch.Tensor:
        """No-grad SDF of S points: level-major gather into feature planes [16][S] (f16x2 | f32x2), then the decoder
        on the planes (csrc/field.hip: k_lotd_gather_lm, k_field_sdf<.., true>).  NSIM_SDF_FUSED=1 selects the single
        fused point-major kernel instead (same values)."""
        sdf = torch.empty([S], dtype=torch.float32, device=dev)
        if S == 0:
            return sdf
        fm = self.field_meta if fm is None else fm
        if self.pos_embed_E:
            return self._sdf_query_wide(grid16, x, rays_o, rays_d, t, ridx, S, dev, goff, n_dev, n_add, collect, sdf)
        planes = None
        # small launches (the first up-sampling draws: R' x 8 points) as ONE fused point-major launch instead of the level-major
        # gather + the decoder on the planes: NSIM_SDF_FUSED_BELOW = capacity in points (0: never; measured in DESIGN sec. 4)
        small = (_SDF_FUSED_BELOW > 0 and S <= _SDF_FUSED_BELOW and x is not None and goff is None and self.plane_levels == 16
                 and type(self)._enc_gather_feat is LoTDNeuSModel._enc_gather_feat)
        if not self._sdf_fused and not small:
            # [NLP][P] x (f16x2 | f32x2), P = the 32-point pitch: every tile of a level is one aligned piece (LDS-DMA)
            planes = torch.empty([self.plane_levels * _lib.plane_pitch(S) * (1 if fm.precision == 0 else 2)],
                                 dtype=torch.float32, device=dev)
            self._enc_gather_feat(fm, grid16, x, rays_o, rays_d, t, ridx, goff, S, n_dev, n_add, planes)
        # ``collect``: the decoder launch also folds its SDFs into the occupancy values (update_from_samples_cfg)
        acc = self.accel if collect else None
        _lib.call("nsim_field_sdf", fm, _lib.ptr(grid16), _lib.ptr(wpack), _lib.ptr(x), _lib.ptr(rays_o),
                  _lib.ptr(rays_d), _lib.ptr(t), _lib.ptr(ridx), _lib.ptr(goff), S, _lib.ptr(n_dev), int(n_add),
                  _lib.ptr(sdf), _lib.ptr(planes), _lib.ptr(acc.occ_val) if acc is not None else None,
                  acc.meta if acc is not None else None, float(acc.inv_s) if acc is not None else 0.0)
        if _lib.TIMER is not None and n_dev is None:       # speculative sizes are accounted once the true size is known
            _lib.TIMER.note_units("nsim_field_sdf", S)
            if planes is not None:
                _lib.TIMER.note_units("nsim_lotd_gather_lm", S)
        return sdf

    def _sdf_query_wide(self, grid16, x, rays_o, rays_d, t, ridx, S, dev, goff, n_dev, n_add, collect, sdf):
        """The no-grad query of a model with an embedded-position block: f32 feature planes from the level-major gather, then
        csrc/wide_field.hip's decoder (one arithmetic for sampling and occupancy: f32)."""
        fm32 = getattr(self, "_field_meta_f32", None)
        if fm32 is None:
            fm32 = _lib.FieldMeta()
            object.__setattr__(self, "_field_meta_f32", fm32)
        C_memmove(fm32, self.field_meta)
        fm32.precision = 1
        planes = torch.empty([self.plane_levels * _lib.plane_pitch(S) * 2], dtype=torch.float32, device=dev)
        self._enc_gather_feat(fm32, grid16, x, rays_o, rays_d, t, ridx, goff, S, n_dev, n_add, planes)
        sw, sb = self._wide_weights()
        _lib.call("nsim_wide_sdf", fm32, int(self.pos_embed_n), _lib.ptr(sw), _lib.ptr(sb), _lib.ptr(x), _lib.ptr(rays_o),
                  _lib.ptr(rays_d), _lib.ptr(t), _lib.ptr(ridx), S, _lib.ptr(n_dev), int(n_add), _lib.ptr(planes), _lib.ptr(sdf))
        if collect:         # update_from_samples_cfg: fold the SDFs into the occupancy values (a pass of its own here)
            pts = x if x is not None else torch.addcmul(rays_o[ridx], t.unsqueeze(-1), rays_d[ridx])
            self.accel.collect(pts.contiguous(), sdf, n_dev, n_add)
        return sdf

    @torch.no_grad()
    def query_sdf(self, x: torch.Tensor) -> torch.Tensor:
        """No-grad SDF at points in [-1,1]^3 (inspect_rendering.py:120-128)."""
        shape = x.shape[:-1]
        x = x.detach().float().reshape(-1, 3).contiguous()
        grid16, wpack = self._shadow()
        sdf = self._sdf_query(grid16, wpack, x, None, None, None, None, x.shape[0], x.device)
        return sdf.reshape(shape)

    @torch.no_grad()
    def _query_sdf_rays(self, rays_o, rays_d, t, ridx, goff=None, n_dev=None, n_add=0):
        grid16, wpack = self._shadow()
        return self._sdf_query(grid16, wpack, None, rays_o, rays_d, t, ridx, t.shape[0], t.device, goff=goff,
                               n_dev=n_dev, n_add=n_add)

    def forward_sdf_nablas(self, x: torch.Tensor, nablas_has_grad: bool = True) -> Dict[str, torch.Tensor]:
        shape = x.shape[:-1]
        xf = (x if x.requires_grad else x.detach()).float().reshape(-1, 3).contiguous()   # dL/dx flows when asked for
        sdf, nablas = _FieldFn.apply(self, self._table(), self.sdf_w, self.sdf_b, self.rad_w,
                                     self.rad_b, None, xf, None, None, None, None, False)
        if not nablas_has_grad:
            nablas = nablas.detach()
        return dict(sdf=sdf.reshape(shape), nablas=nablas.reshape(*shape, 3))

    def forward_sdf(self, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        return dict(sdf=self.forward_sdf_nablas(x, nablas_has_grad=False)["sdf"])

    def sample_pts_uniform(self, num_pts: int, generator=None) -> Dict[str, torch.Tensor]:
        """Random points in the AABB -> forward_sdf_nablas (code_single/tools/train.py:602-613)."""
        lo, hi = self.accel.aabb[0], self.accel.aabb[1]
        x = lo + torch.rand([num_pts, 3], device=self.device, generator=generator) * (hi - lo)
        ret = self.forward_sdf_nablas(x)
        ret["net_x"] = x
        return ret

    def sample_pts_in_occupied(self, num_pts: int, generator=None) -> Dict[str, torch.Tensor]:
        """Random points inside occupied voxels -> forward_sdf_nablas (app/loss/eikonal.py:226-227: eikonal
        ``on_occ_ratio``).  Voxels are drawn uniformly among the occupied ones, points uniformly inside the voxel."""
        acc = self.accel
        occ_idx = (acc.occ_val > acc.occ_thre).nonzero()[:, 0]
        if occ_idx.numel() == 0:
            return self.sample_pts_uniform(num_pts, generator=generator)
        dev = self.device
        pick = occ_idx[torch.randint(0, occ_idx.numel(), [num_pts], device=dev, generator=generator)]
        rx, ry = acc.resolution[0], acc.resolution[1]
        vox = torch.stack([pick % rx, (pick // rx) % ry, pick // (rx * ry)], dim=-1).to(torch.float32)
        res = torch.tensor(acc.resolution, dtype=torch.float32, device=dev)
        lo, hi = acc.aabb[0], acc.aabb[1]
        x = lo + (vox + torch.rand([num_pts, 3], device=dev, generator=generator)) / res * (hi - lo)
        ret = self.forward_sdf_nablas(x)
        ret["net_x"] = x
        return ret

    # ------------------------------------------------------------------ rays
    @staticmethod
    def convert_rays_in_node(rays_o, rays_d, rotation: torch.Tensor, translation: torch.Tensor, scale=1.0):
        """World -> object rays, ``R^-1 (o - t) / s`` and ``R^-1 d / s`` (app/resources/scenes.py:686-708), with the
        broadcast-multiply-sum the reference insists on instead of mm/bmm (app/resources/nodes.py:79-84)."""
        Rt = rotation.transpose(-1, -2)
        o = ((rays_o - translation).unsqueeze(-2) * Rt).sum(-1) / scale
        d = (rays_d.unsqueeze(-2) * Rt).sum(-1) / scale
        return o, d

    def ray_test(self, rays_o, rays_d, near=None, far=None, **extra) -> Dict:
        """AABB slab test + compaction of the hit rays (single_volume_renderer.py:235-238); the same call as
        ``model.space.ray_test(**ray_input)`` (app/visualizer/gui_runner_single_cuboid.py:135-138)."""
        return aabb_ray_test(self.accel.aabb, self.accel.meta, rays_o, rays_d, near=near, far=far, **extra)

    def _arange_repeat(self, R: int, n: int, dev):
        """arange(R).repeat_interleave(n), cached (ray index of the batched up-sampling points)."""
        key = (R, n, str(dev))
        c = getattr(self, "_ar_cache", None)
        if c is None:
            c = self._ar_cache = {}
        if key not in c:
            if len(c) > 16:
                c.clear()
            c[key] = torch.arange(R, device=dev).repeat_interleave(n)
        return c[key]

    def _sample(self, o, d, near, far, qp: dict, jitter, jitter_c, goff=None, woff=None, cap: Optional[int] = None,
                pre_sync_hook=None, need_ridx: bool = True):
        """No-grad sampling: occupancy marching + coarse depths + multi-stage NeuS up-sampling.

        ``cap`` = None: the size M of the marched set is read back (host sync) before anything is allocated.
        ``cap`` = int: buffers are sized for M <= cap WITHOUT reading M; every consumer is per-ray (through
        ``pack_infos``) or takes the point count from device memory, rays that would overflow are emptied on the device
        (nsim_pack_infos_from_n), and the caller compares the true M (returned as a device scalar) with cap at its next
        sync -- one host round-trip less per step."""
        R = o.shape[0]
        dev = o.device
        f32 = dict(dtype=torch.float32, device=dev)
        march = qp.get("march_cfg", {})
        step, max_steps = float(march.get("step_size", 0.005)), int(march.get("max_steps", 4096))
        C = int(qp.get("num_coarse", 64))
        bits, occm = self.accel.occ_bits, self.accel.meta
        counts = torch.empty([R], dtype=torch.long, device=dev)
        _lib.call("nsim_march_count", _lib.ptr(o), _lib.ptr(d), _lib.ptr(near), _lib.ptr(far), _lib.ptr(jitter), R,
                  _lib.ptr(bits), _lib.ptr(woff), occm, step, max_steps, _lib.ptr(counts))
        nt = self._host_notify() if cap is not None else None
        self._notify_seq_m = None
        fine = fine_list(qp)
        # ``upsample_on_marched_only``: ranks of the rays whose march found something (q-th live ray <-> ray r) and the
        # device-side point counts of this pass's SDF queries -- coarse / fine samples and their queries cover R' rays
        mo = marched_only(qp)
        lr = live_idx = cnts = None
        self._live = None
        adr_m = None
        if nt is not None:      # the true marched total travels to host-mapped words (read at the compress step's wait)
            adr_m, self._notify_seq_m = nt.arm(0)
        if mo:                  # ONE single-workgroup scan: pack infos of the marched counts + the live ranks
            assert len(fine) <= 4, "upsample_on_marched_only: at most four up-sampling stages"
            lr = torch.empty([R], dtype=torch.long, device=dev)
            live_idx = torch.empty([R], dtype=torch.long, device=dev)
            cnts = torch.empty([8], dtype=torch.long, device=dev)
            pi_m = torch.empty([R, 2], dtype=torch.long, device=dev)
            total_m = torch.empty([1], dtype=torch.long, device=dev)
            adr_l, seq_l = nt.arm(2) if nt is not None else (None, 0)
            _lib.call("nsim_live_rank", _lib.ptr(counts), R, C, *((list(fine) + [0, 0, 0, 0])[:4]), _lib.ptr(lr), _lib.ptr(live_idx),
                      _lib.ptr(cnts), adr_l, seq_l, _lib.ptr(pi_m), _lib.ptr(total_m), -1 if cap is None else int(cap), adr_m,
                      self._notify_seq_m or 0)
            self._live = dict(rank=lr, idx=live_idx, cnts=cnts, seq=seq_l if nt is not None else None, n=None)
        elif nt is not None:
            pi_m, total_m = po.get_pack_infos_from_n(counts, return_total=True, cap=int(cap), notify=(adr_m, self._notify_seq_m))
        else:
            pi_m, total_m = po.get_pack_infos_from_n(counts, return_total=True, cap=-1 if cap is None else int(cap))
        Rl = R                                  # rays that get coarse / fine samples (an upper bound when sized speculatively)
        if cap is None:
            if pre_sync_hook is not None:
                pre_sync_hook()                 # independent host work queued ahead of the blocking read
            if mo:                              # host sync: size of the marched set and the number of live rays
                c_host = cnts.tolist()
                M, Rl = int(c_host[6]), int(c_host[0])
                self._live["n"] = Rl
                if Rl == 0:                     # no ray marched into an occupied voxel: no samples at all
                    self._last_S_q, self._S_per_live = 0, C + sum(int(n) for n in fine)
                    e = torch.empty([0], **f32)
                    return (e, e, torch.zeros([R, 2], dtype=torch.long, device=dev),
                            torch.empty([0], dtype=torch.long, device=dev) if need_ridx else None, counts, total_m)
            else:
                M = int(total_m.item())         # host sync: size of the marched set
            n_dev = None
        else:
            M, n_dev = int(cap), total_m

        def n_of(k):
            """device-side point count of query k (0: marched + coarse, 1 + i: the draws of stage i) when sizes are speculative"""
            if n_dev is None:
                return None
            return cnts[1 + k:2 + k] if mo else (n_dev if k == 0 else None)
        t_m = torch.empty([max(M, 1)], **f32)
        _lib.call("nsim_march_emit", _lib.ptr(o), _lib.ptr(d), _lib.ptr(near), _lib.ptr(far), _lib.ptr(jitter), R,
                  _lib.ptr(bits), _lib.ptr(woff), occm, step, max_steps, _lib.ptr(pi_m), _lib.ptr(t_m))
        t_c = torch.empty([max(Rl, 1), C], **f32)
        _lib.call("nsim_coarse_depths", _lib.ptr(near), _lib.ptr(far), _lib.ptr(jitter_c), R, C, _lib.ptr(t_c), _lib.ptr(lr))
        S = M + Rl * C
        t = torch.empty([S], **f32)
        pi = torch.empty([R, 2], dtype=torch.long, device=dev)
        # the level-major query visits every point once per XCD: hand it positions (12 B) instead of (ridx, t, o, d)
        with_x = not self._sdf_fused
        # per-sample ray indices (8 B each) are written only where somebody reads them: by the point-major / batched
        # queries on the way, and by the caller at the end (``need_ridx``: the compressed mode re-derives them)
        # per-ray state read INSIDE the encoding kernels: instance table offsets (batched LoTD) or a condition per ray (permuto)
        per_ray = goff is not None or self._per_ray_condition()
        ridx_mid = (not with_x) or per_ray
        ridx = torch.empty([S], dtype=torch.long, device=dev) if (ridx_mid or (need_ridx and not fine)) else None
        xq = torch.empty([S, 3], **f32) if with_x else None
        _lib.call("nsim_merge_sorted", _lib.ptr(t_m), None, _lib.ptr(pi_m), _lib.ptr(t_c), None, R, C, _lib.ptr(t), None,
                  _lib.ptr(pi), _lib.ptr(ridx), _lib.ptr(o), _lib.ptr(d), _lib.ptr(xq), _lib.ptr(lr))
        grid16, wpack = self._shadow()
        fm_s = None
        if with_x:
            fm_s, wpack = self._sampling_ctx()
        collect = with_x and goff is None and self.accel.collect_armed       # fused into the decoder launches below
        n0, a0 = (n_of(0), 0) if mo else (n_dev, R * C)
        if with_x:
            sdf = self._sdf_query(grid16, wpack, xq, None, None, None, ridx if per_ray else None, S, dev,
                                  goff=goff, n_dev=n0, n_add=a0, collect=collect, fm=fm_s)
        else:
            sdf = self._query_sdf_rays(o, d, t, ridx, goff, n_dev=n0, n_add=a0)
        inv_s0 = float(qp.get("upsample_inv_s", 64.0))
        use_est = 1 if qp.get("upsample_use_estimate_alpha", True) else 0
        factors = list(qp.get("upsample_inv_s_factors", [1, 4, 16]))
        stages = list(zip(fine, factors))
        fuse = _FUSE_MERGE_UPSAMPLE
        t_new = x_new = None
        for k_stage, (nf, fac) in enumerate(stages):
            nf = int(nf)
            if t_new is None:       # the stage's draws (stage 0, or every stage without the fused merge + draw launch)
                t_new = torch.empty([max(Rl, 1), nf], **f32)
                scratch = torch.empty([S], **f32)
                x_new = torch.empty([max(Rl, 1) * nf, 3], **f32) if with_x else None
                _lib.call("nsim_upsample_stage", _lib.ptr(t), _lib.ptr(sdf), _lib.ptr(pi), R, inv_s0 * float(fac), nf, use_est,
                          _lib.ptr(scratch), _lib.ptr(t_new), _lib.ptr(o), _lib.ptr(d), _lib.ptr(x_new), _lib.ptr(lr))
            if not mo:
                ridx_new = self._arange_repeat(R, nf, dev)
            elif per_ray or not with_x:     # the draws of live ray q belong to ray live_idx[q]
                ridx_new = live_idx[:Rl].repeat_interleave(nf)
            else:
                ridx_new = None
            if with_x:
                sdf_new = self._sdf_query(grid16, wpack, x_new, None, None, None, ridx_new if per_ray else None,
                                          Rl * nf, dev, goff=goff, n_dev=n_of(1 + k_stage) if mo else None, collect=collect, fm=fm_s)
            else:
                sdf_new = self._query_sdf_rays(o, d, t_new.reshape(-1)[:Rl * nf], ridx_new, goff)
            S2 = S + Rl * nf
            t2 = torch.empty([S2], **f32)
            sdf2 = torch.empty([S2], **f32)
            pi2 = torch.empty([R, 2], dtype=torch.long, device=dev)
            last = k_stage == len(stages) - 1
            ridx = torch.empty([S2], dtype=torch.long, device=dev) if (ridx_mid or (need_ridx and last)) else None
            if fuse and not last:       # merge of this stage + the draws of the next in one launch
                nf2, fac2 = int(stages[k_stage + 1][0]), stages[k_stage + 1][1]
                t_nx = torch.empty([max(Rl, 1), nf2], **f32)
                scratch = torch.empty([S2], **f32)
                x_nx = torch.empty([max(Rl, 1) * nf2, 3], **f32) if with_x else None
                _lib.call("nsim_merge_upsample", _lib.ptr(t), _lib.ptr(sdf), _lib.ptr(pi), _lib.ptr(t_new), _lib.ptr(sdf_new), R, nf,
                          _lib.ptr(t2), _lib.ptr(sdf2), _lib.ptr(pi2), _lib.ptr(ridx), inv_s0 * float(fac2), nf2, use_est,
                          _lib.ptr(scratch), _lib.ptr(t_nx), _lib.ptr(o), _lib.ptr(d), _lib.ptr(x_nx), _lib.ptr(lr))
                t_new, x_new = t_nx, x_nx
            else:
                _lib.call("nsim_merge_sorted", _lib.ptr(t), _lib.ptr(sdf), _lib.ptr(pi), _lib.ptr(t_new), _lib.ptr(sdf_new), R,
                          nf, _lib.ptr(t2), _lib.ptr(sdf2), _lib.ptr(pi2), _lib.ptr(ridx), None, None, None, _lib.ptr(lr))
                t_new = x_new = None
            t, sdf, pi, S = t2, sdf2, pi2, S2
        # SDF-only queries issued by this sampling pass (every sample is queried exactly once); with speculative sizes the
        # exact figure follows at the compress step's wait (``_note_sampling_units``)
        self._last_S_q = S
        self._S_per_live = C + sum(int(n) for n in fine)
        return t, sdf, pi, ridx, counts, total_m

    def _host_notify(self):
        """``_lib.HostNotify`` of this model (NSIM_HOST_NOTIFY=0 or a failed wait turn it off)."""
        nt = getattr(self, "_notify", None)
        if nt is None:
            nt = self._notify = _lib.HostNotify(3) if os.environ.get("NSIM_HOST_NOTIFY", "1") == "1" else False
        return nt or None

    def _compress(self, t, sdf, pi, forward_inv_s: float, thre: float, total_m=None, pre_sync_hook=None, tail: int = 0,
                  spec_launch=None):
        """``march_occ_multi_upsample_compressed``: drop the samples whose visibility weight (from the no-grad SDFs
        of the sampling pass) is negligible before the with-grad query.  Host sync (size of the kept set; the same
        round-trip also brings back the true marched total ``total_m`` of a speculatively sized sampling pass).
        -> t_k, pi_k, ridx_k, M_true (None when total_m is None); all None but M_true on overflow.  ``tail`` > 0: the
        kernel also writes ``tail`` zero-depth samples on the pseudo-rays R.. behind the kept set (the trainer's free
        eikonal points); t_k / ridx_k are then views of ``self._with_tail = (t [Sk+tail], ridx [Sk+tail])``.

        ``spec_launch(t_full, ridx_full, pi_k, total_dev, cap)`` (with host-mapped size words only): the kept set is
        emitted into buffers of a CAPACITY taken from the previous iterations and the callback queues the first
        with-grad launches (device-side point count) BEFORE the host waits for the size -- the stream then has work
        while the host reads the size and prepares the rest of the step.  ``self._spec_ok`` tells the caller whether
        those launches stand (False: the kept set outgrew the capacity; everything is redone at the exact size)."""
        R = pi.shape[0]
        dev = t.device
        counts = torch.empty([R], dtype=torch.long, device=dev)
        ln_eff = self._ln_inv_s_eff().detach()
        _lib.call("nsim_compress_count", _lib.ptr(sdf), _lib.ptr(pi), R, _lib.ptr(ln_eff),
                  self.ln_inv_s_factor, forward_inv_s, thre, _lib.ptr(counts))
        nt = self._host_notify()
        self._spec_ok = False
        cap_k = self._keep_cap(R) if (spec_launch is not None and nt is not None) else None
        seq_k = None
        if nt is not None:
            adr, seq_k = nt.arm(1)
            pi_k, total = po.get_pack_infos_from_n(counts, return_total=True, notify=(adr, seq_k),
                                                   cap=-1 if cap_k is None else cap_k)
        else:
            pi_k, total = po.get_pack_infos_from_n(counts, return_total=True)

        def emit(n_out, tail_):
            t_o = torch.empty([n_out + tail_], dtype=torch.float32, device=dev)
            r_o = torch.empty([n_out + tail_], dtype=torch.long, device=dev)
            _lib.call("nsim_compress_emit", _lib.ptr(sdf), _lib.ptr(t), _lib.ptr(pi), R, _lib.ptr(ln_eff),
                      self.ln_inv_s_factor, forward_inv_s, thre, _lib.ptr(pi_k), _lib.ptr(t_o), _lib.ptr(r_o), tail_)
            return t_o, r_o
        if cap_k is not None:       # emit + the caller's first launches at the capacity, ahead of the size read
            t_k, ridx_k = emit(cap_k, int(tail))
            spec_launch(t_k, ridx_k, pi_k, total, cap_k)
        if pre_sync_hook is not None:
            pre_sync_hook()                     # e.g. the trainer's prefetch of the next batch (its own sync lands here)
        Sk = M_true = None
        _w0 = time.perf_counter() if _lib.HOST_WAIT is not None else 0.0
        if nt is not None:          # sizes from the host-mapped words: no stream synchronisation, no copy
            Sk = nt.wait(1, seq_k)
            if Sk is not None and total_m is not None:
                M_true = nt.wait(0, self._notify_seq_m) if self._notify_seq_m is not None else None
                if M_true is None:
                    Sk = None
        if Sk is None:              # no host-mapped words, or the wait timed out: a synchronising read of the device copy
            if total_m is None:
                Sk, M_true = int(total.item()), None
            else:
                Sk, M_true = torch.cat([total, total_m]).tolist()
            if nt is not None and int(nt.view[1, 1]) != seq_k:
                # the kernel has finished and its store still is not visible: this memory is not host-coherent --
                # synchronising reads from here on.  (Visible now = the stream was merely slow, e.g. first-use code
                # loading on a cold box: the words stay in use.)
                self._notify = False
        lv = getattr(self, "_live", None)
        if lv is not None and lv["n"] is None:      # number of live rays of a speculatively sized sampling pass
            n_l = nt.wait(2, lv["seq"]) if (nt is not None and self._notify and lv["seq"] is not None) else None
            lv["n"] = int(lv["cnts"][0].item()) if n_l is None else int(n_l)
        if _lib.HOST_WAIT is not None:      # bench.py: host time spent blocked on the size of the kept set
            _lib.HOST_WAIT += time.perf_counter() - _w0
        self._keep_stat = (R, Sk)
        if cap_k is not None and Sk <= cap_k and Sk > 0:
            self._spec_ok = True
            self._with_tail = (t_k[:Sk + int(tail)], ridx_k[:Sk + int(tail)]) if tail else None
            return t_k[:Sk], pi_k, ridx_k[:Sk], M_true
        if cap_k is not None:       # outgrown (or empty): exact pack infos, exact emit
            pi_k = po.get_pack_infos_from_n(counts)
        tail = int(tail) if Sk > 0 else 0
        if Sk > 0:
            t_k, ridx_k = emit(Sk, tail)
        else:
            t_k = torch.empty([0], dtype=torch.float32, device=dev)
            ridx_k = torch.empty([0], dtype=torch.long, device=dev)
        self._with_tail = (t_k, ridx_k) if tail else None
        return t_k[:Sk], pi_k, ridx_k[:Sk], M_true

    def _keep_cap(self, R: int) -> Optional[int]:
        """Capacity for the kept set of R rays from the last observed one (1.3x + slack), or None."""
        st = getattr(self, "_keep_stat", None)
        if st is None or not self._speculate:
            return None
        R0, K0 = st
        if R0 <= 0 or K0 <= 0 or not (0.5 <= R / R0 <= 2.0):
            return None
        return ((int(1.3 * K0 * R / R0) + 4096) + 31) & ~31

    def _speculative_cap(self, R: int) -> Optional[int]:
        """Capacity for the marched set of R rays from the last observed density (1.3x + slack), or None."""
        st = getattr(self, "_march_stat", None)
        if st is None or not self._speculate:
            return None
        R0, M0 = st
        if R0 <= 0 or not (0.5 <= R / R0 <= 2.0):
            return None
        return int(1.3 * M0 * R / R0) + 8192

    def _query_samples(self, ray_tested: dict, cfg: dict, qp: dict):
        """The no-grad half of ``ray_query``: occupancy marching + coarse depths + NeuS up-sampling (+ compression) of the
        R tested rays.  -> (o, d, t, pi, ridx, sdf_ng, march_counts, goff, fis): depths t [S] packed by pi [R,2] with
        ray indices ridx [S] -- constants of the differentiable part that follows."""
        R = ray_tested["num_rays"]
        self._last_R_tested = int(R)
        o = ray_tested["rays_o"].detach().float().contiguous()
        d = ray_tested["rays_d"].detach().float().contiguous()
        near, far = ray_tested["near"].contiguous(), ray_tested["far"].contiguous()
        dev = o.device
        perturb = cfg.get("perturb", False)
        jitter = cfg.get("_jitter", None)
        jitter_c = cfg.get("_jitter_c", None)
        if jitter is None and perturb:
            jitter = torch.rand([R], device=dev)
            jitter_c = torch.rand([R, int(qp.get("num_coarse", 64))], device=dev)
        if jitter is not None:
            jitter = jitter.float().contiguous()
        if jitter_c is not None:
            jitter_c = jitter_c.float().contiguous()
        fis = cfg.get("forward_inv_s", None)
        fis = float(fis) if fis else 0.0
        mode = cfg.get("query_mode", self.ray_query_cfg.get("query_mode", "march_occ_multi_upsample"))
        goff = ray_tested.get("rays_goff", None)           # batched model: per-ray instance offsets (table / occupancy)
        woff = ray_tested.get("rays_word_off", None)
        with torch.no_grad():
            compressed = mode.endswith("_compressed")
            cap = self._speculative_cap(R) if compressed else None
            hook = cfg.get("_pre_sync_hook", None)       # called once, right before the first blocking size read
            t, sdf_ng, pi, ridx, march_counts, total_m = self._sample(o, d, near, far, qp, jitter, jitter_c, goff, woff,
                                                                      cap=cap, pre_sync_hook=hook if cap is None else None,
                                                                      need_ridx=not compressed)
            if compressed:
                thre = float(qp.get("compress_thre", 1e-4))
                tail = int(cfg.get("_tail_points", 0))
                t_k, pi_k, ridx_k, M_true = self._compress(t, sdf_ng, pi, fis, thre, total_m if cap is not None else None,
                                                           pre_sync_hook=hook if cap is not None else None, tail=tail,
                                                           spec_launch=cfg.get("_spec_launch") if cap is not None else None)
                if cap is not None and M_true > cap:          # the speculation failed: redo with the exact size
                    t, sdf_ng, pi, ridx, march_counts, total_m = self._sample(o, d, near, far, qp, jitter, jitter_c,
                                                                              goff, woff, cap=None, need_ridx=False)
                    t_k, pi_k, ridx_k, _ = self._compress(t, sdf_ng, pi, fis, thre, tail=tail)
                    M_true = int(total_m.item())
                lv = getattr(self, "_live", None)
                Rl = int(lv["n"]) if lv is not None else R          # rays that got coarse / fine samples
                per_live = int(qp.get("num_coarse", 64)) + sum(fine_list(qp))
                if M_true is None:
                    M_true = int(sdf_ng.shape[0]) - Rl * per_live
                self._march_stat = (R, M_true)
                self._last_S_q = M_true + Rl * per_live
                if _lib.TIMER is not None and cap is not None:    # the queries' true sizes, now that they are known
                    sizes = [M_true + Rl * int(qp.get("num_coarse", 64))]
                    if lv is not None:      # (the draws of a marched-only pass take their point counts from the device, too)
                        sizes += [Rl * int(n) for n in fine_list(qp)]
                    for S0 in sizes:
                        _lib.TIMER.note_units("nsim_field_sdf", S0)
                        if not self._sdf_fused:
                            _lib.TIMER.note_units("nsim_lotd_gather_lm", S0)
                t, pi, ridx = t_k, pi_k, ridx_k
        self.accel.collect_armed = False         # one sampling pass per arming (evaluation renders never collect)
        return o, d, t, pi, ridx, sdf_ng, march_counts, goff, fis

    def ray_query(self, *, ray_input: dict = None, ray_tested: dict, config, return_buffer: bool = True,
                  return_details: bool = False, render_per_obj_individual: bool = False) -> Dict:
        """``query_mode = march_occ_multi_upsample`` (single_volume_renderer.py:244-246)."""
        cfg = dict(config)
        qp = dict(cfg.get("query_param", self.ray_query_cfg.get("query_param", {})))
        with_rgb = cfg.get("with_rgb", True)
        with_normal = cfg.get("with_normal", False)
        if cfg.get("with_feature_dim", 0):          # ``n_extra_feat_from_output`` is 0 in every config of the hot path
            raise NotImplementedError("with_feature_dim > 0: the decoders of this model emit no extra feature channels")
        ret = dict()
        R = ray_tested["num_rays"]
        # ``render_per_obj_individual``: this object alone, as images over ALL the rays of ``ray_input`` (the renderers
        # index them with rays_inds / reshape them to the view: buffer_compose_renderer.py:576-577,
        # code_single/tools/train.py:1037-1040); without ray_input the hit rays only
        n_all = None
        if render_per_obj_individual and ray_input is not None and ray_input.get("rays_o") is not None:
            n_all = int(ray_input["rays_o"].shape[0])

        def empty_rendered():
            dev = ray_tested["rays_inds"].device
            z = lambda *sh: torch.zeros([n_all or 0, *sh], dtype=torch.float32, device=dev)      # noqa: E731
            r_ = dict(mask_volume=z(), depth_volume=z())
            if with_rgb:
                r_["rgb_volume"] = z(3)
            if with_normal:
                r_["normals_volume"] = z(3)
            return r_
        if R == 0:
            ret["volume_buffer"] = dict(type="empty")
            if render_per_obj_individual:
                ret["rendered"] = empty_rendered()
            if return_details:
                ret["details"] = dict()
            return ret
        h_appear = ray_tested.get("rays_h_appear", None)
        extra_x = cfg.get("_extra_pts", None)              # trainer hook: free points riding on the same launches
        o_g, d_g = ray_tested["rays_o"], ray_tested["rays_d"]
        # Speculative forward (training, compressed mode, single-instance model, constant rays): the per-ray arrays are
        # built now, and ``_compress`` calls ``spec_launch`` -- gather + decoders at a CAPACITY, point count read on the
        # device -- before the host waits for the size of the kept set (see ``_compress``; NSIM_SPEC_FORWARD=0: off)
        spec = None
        mode = cfg.get("query_mode", self.ray_query_cfg.get("query_mode", "march_occ_multi_upsample"))
        if (_SPEC_FORWARD and type(self) is LoTDNeuSModel and mode.endswith("_compressed") and torch.is_grad_enabled()
                and self.encoding.flattened_params.requires_grad and not (o_g.requires_grad or d_g.requires_grad)):
            M = int(extra_x.shape[0]) if extra_x is not None else 0
            o_a, d_a = o_g.detach().float().contiguous(), d_g.detach().float().contiguous()
            ha_a = h_appear.detach().float().contiguous() if (with_rgb and h_appear is not None) else None
            pre_od = cfg.get("_extra_pre", None)      # trainer hook: [R + M] ray arrays built with the batch (hit rays + the points)
            if M and pre_od is not None and pre_od[0].shape[0] == o_a.shape[0] + M:
                o_a, d_a = pre_od
                if ha_a is not None:                  # appearance codes: zero rows for the free points (one cat instead of five)
                    hz = getattr(self, "_extra_hz", None)
                    if hz is None or hz.shape != (M, ha_a.shape[1]) or hz.device != ha_a.device:
                        hz = self._extra_hz = torch.zeros([M, ha_a.shape[1]], dtype=torch.float32, device=ha_a.device)
                    ha_a = torch.cat([ha_a, hz])
            elif M:
                e = torch.empty([0], dtype=torch.float32, device=o_a.device)
                o_a, d_a, _, _, ha_a = append_extra_points(self, o_a, d_a, e, e.long(), ha_a, extra_x)
            spec = dict(M=M, rays_o=o_a, rays_d=d_a, ha=ha_a)

            def spec_launch(t_full, ridx_full, pi_k, total_dev, cap_k):
                dev_, NLP = t_full.device, self.plane_levels
                Sc = cap_k + M
                PSc = _lib.plane_pitch(Sc)
                f32 = dict(dtype=torch.float32, device=dev_)
                grid16, wpack = self._shadow()
                spec.update(sdf=torch.empty([Sc], **f32), nablas=torch.empty([Sc, 3], **f32),
                            rgb=torch.empty([Sc, 3], **f32) if with_rgb else None,
                            h_pl=torch.empty([NLP, PSc, 2], **f32),
                            J_pl=torch.empty([NLP, PSc, 2, 3], dtype=_lib.jplane_dtype(self.field_meta), device=dev_), PS=PSc)
                spec["enc_state"] = self._enc_field_fwd(grid16, wpack, None, o_a, d_a, t_full, ridx_full, None, ha_a, Sc, spec["sdf"],
                                                        spec["nablas"], spec["rgb"], spec["h_pl"], spec["J_pl"], total_dev, M)
            cfg["_spec_launch"], cfg["_tail_points"] = spec_launch, M
            self._with_tail = None
        o, d, t, pi, ridx, sdf_ng, march_counts, goff, fis = self._query_samples(ray_tested, cfg, qp)
        if t.shape[0] == 0:
            ret["volume_buffer"] = dict(type="empty")
            if render_per_obj_individual:
                ret["rendered"] = empty_rendered()
            if return_details:
                ret["details"] = dict(march_counts=march_counts)
            return ret
        pre = None
        if spec is not None and getattr(self, "_spec_ok", False) and "sdf" in spec:
            t_a, ridx_a = self._with_tail if spec["M"] else (t, ridx)
            pre = dict(spec, t=t_a, ridx=ridx_a)
        # rays that carry gradients (pose refinement) stay attached for the with-grad query only; the sampling above
        # is no-grad by construction (t is a constant of the differentiable step, as in the reference)
        o_in = o_g.float().contiguous() if o_g.requires_grad else o
        d_in = d_g.float().contiguous() if d_g.requires_grad else d
        outs = _FieldFn.apply(self, self._table(), self.sdf_w, self.sdf_b, self.rad_w, self.rad_b,
                              h_appear if with_rgb else None, None, o_in, d_in, t, ridx, bool(with_rgb), goff, extra_x, pre)
        sdf, nablas = outs[0], outs[1]
        rgb = outs[2] if with_rgb else None
        if extra_x is not None:
            ret["extra_pts"] = dict(sdf=outs[-2], nablas=outs[-1], net_x=extra_x)
        if not qp.get("nablas_has_grad", True):
            nablas = nablas.detach()
        alpha = _NeusAlphaFn.apply(sdf, self._ln_inv_s_eff(), pi, self.ln_inv_s_factor, fis)
        # ``upsample_on_marched_only``: the buffer lists the rays that produced samples -- the [R'] hit set of the reference's
        # buffers (single_volume_renderer.py:209-220,289-300); rays whose march found nothing are not in it.  (Per-sample
        # arrays are untouched: the dropped rows are empty packs.)  ``self._rays_sel``: the rows kept, for per-ray side arrays
        rays_inds_hit, pi_hit = ray_tested["rays_inds"], pi
        self._rays_sel = None
        lv = getattr(self, "_live", None)
        if lv is not None and lv.get("n") is not None and int(lv["n"]) < R:
            self._rays_sel = lv["idx"][:int(lv["n"])]
            rays_inds_hit, pi_hit = rays_inds_hit[self._rays_sel], pi[self._rays_sel]
        vb = dict(type="packed", rays_inds_hit=rays_inds_hit, pack_infos_hit=pi_hit, t=t, opacity_alpha=alpha,
                  nablas=nablas, sdf=sdf)
        if with_rgb:
            vb["rgb"] = rgb
        ret["volume_buffer"] = vb
        if render_per_obj_individual or cfg.get("_render", False):
            ret["rendered"] = volume_integration(alpha, t, rgb, nablas if (with_normal or cfg.get("_render", False)) else None,
                                                 pi, cfg.get("depth_use_normalized_vw", True),
                                                 rays_inds=ray_tested["rays_inds"] if n_all is not None else None,
                                                 num_rays=n_all)
            ret["rendered"].pop("vw", None)
            ret["rendered"].pop("trans", None)
        if return_details:
            ret["details"] = dict(march_counts=march_counts, sdf_nograd=sdf_ng, ridx=ridx)
        if cfg.get("with_near_sdf", False):
            # ``renderer._config_train.with_near_sdf = True`` (code_single/tools/train.py:245) -> ``details['near_sdf']``
            # read by ClearanceLoss (app/loss/clearance.py:85-87): the SDF, with gradient, where each tested ray enters
            # the model's space (x = o + near d).  (Implementation in the absent nr3d_lib: semantics fixed here.)
            x_near = (o + ray_tested["near"].float()[:, None] * d).detach()
            ret.setdefault("details", {})["near_sdf"] = self.forward_sdf_nablas(x_near, nablas_has_grad=False)["sdf"]
        return ret
