"""Oracle parity at the OTHER BASELINE configurations (the configs[1] comparison is tests/test_fullsize_parity.py):

* street  -- configs[3], code_single/configs/waymo/streetsurf/withmask_withlidar_joint.240219.yaml:146-322: 19-level cuboid
  LoTD (T = 2^20, ~33 Mi parameters), 1x64 decoder, ``sdf_scale 25``, 200 x 100 x 30 m AABB with 1 m occupancy voxels,
  ``num_coarse 128``, ``step .2``, ``upsample_use_estimate_alpha: false``, compressed query, distant model
  (``fixed_cuboid_shells``, no view directions, ``include_inf_distance: false``) + sky MLP, l1 photometric loss;
* indoor  -- configs[2], code_single/configs/indoor/lotd_neus.replica.230814.yaml:95, 236-238, 272-288: ``inside_out``
  room, 64 x 64 image patch + pixel rays, normals and depth WITH gradient through the monocular losses;
* multi   -- configs[4], code_multi/configs/exps/fg_neus=hyper_lotd/no_fg_occ.221218.yaml:307-390: street background +
  8 posed instances of one shared batched model (32^3 batched occupancy, ``num_coarse 32``, ``num_fine 8``,
  ``upsample_inv_s_factors [1, 4]``) + distant model + sky through the ``BufferComposeRenderer`` mirror.

Same structure as the configs[1] test: the product models are built by ``neuralsim_amd.scenarios`` (what bench.py times),
their weights copied verbatim into the oracle, identical rays / appearance codes / perturbation randoms on both sides;
discrete decisions (hit rays, march counts, kept-sample counts) must be bit-exact -- in f32 mode AND in the fp16 product
mode, whose sampling pass runs its SDF queries on the exact-f32 kernels (``sampling_precision``) --, images and every
gradient within the stated tolerance.

The SAME test bodies run at emulator size on the CPU (``small=True`` scenarios, ``-m "not gpu"``) and at the BASELINE
sizes on the GPU (``-m gpu``): 16 384 rays for the discrete decisions of the street / indoor sampling passes, N_GRAD rays
for the legs that need the oracle's autograd.
"""
import json
from pathlib import Path

import pytest
import torch

from oracle import distant as od, field as ofield, render as orr
from util import (distant_flat_grads, leaf, oracle_flat_grads, oracle_of_distant, oracle_of_neus, oracle_of_sky, rel_l2)

REPORT_DIR = Path(__file__).resolve().parent.parent / "gpurun_out"
N_FULL, N_GRAD = 16384, 2048

# f32 = exact-f32 MFMA validation mode, fp16 = product default (fp16 MFMA operands, f32 sampling-pass SDFs).
# Measured on MI355X at the full sizes: profiles/round3_parity/*.json.
# depth_volume is in scene units: tolerances are for depths up to 200 (street; the indoor room is 100 x smaller).
# f32 gradients: the oracle itself is only this well conditioned -- its f32 and f64 evaluations of the indoor loss differ
# by 1e-3 on the first decoder layer (tools/cond_probe.py).
TOL = dict(
    f32=dict(img=dict(mask_volume=1e-4, rgb_volume=1e-4, depth_volume=2e-2, normals_volume=2e-4), grad=3e-3, loss=2e-5),
    fp16=dict(img=dict(mask_volume=5e-3, rgb_volume=5e-3, depth_volume=1.0, normals_volume=1e-2), grad=3e-2, loss=2e-3),
)


def _report(name, rec):
    try:
        REPORT_DIR.mkdir(exist_ok=True)
        (REPORT_DIR / f"parity_configs_{name}.json").write_text(json.dumps(rec, indent=1, sort_keys=True))
    except OSError:
        pass
    print(f"[parity {name}] " + json.dumps(rec, sort_keys=True))


def _sizes(backend):
    small = backend.type == "cpu"
    return small, (96 if small else N_FULL), (48 if small else N_GRAD)


def _tol(precision, small):
    """emulator size: 48 rays -- a gradient is a sum of few cancelling terms, the f32 bound is the one of
    tests/test_ray_query.py (5e-3)"""
    t = dict(TOL[precision])
    if small:
        t["grad"] = 5e-3 if precision == "f32" else 6e-2
    return t


def _rays(intr, c2w, WH, N, seed, C, K=None):
    g = torch.Generator().manual_seed(seed)
    V = intr.shape[0]
    xy = torch.rand(N, 2, generator=g).clamp(1e-6, 1 - 1e-6)
    fidx = torch.randint(0, V, (N,), generator=g)
    o, d = orr.pinhole_rays(xy, fidx, intr.cpu(), c2w.cpu(), WH.cpu())
    out = dict(xy=xy, fidx=fidx, o=o, d=d, jit=torch.rand(N, generator=g), jit_c=torch.rand(N, C, generator=g),
               ha=torch.randn(N, 4, generator=g) * 0.1, gt=torch.rand(N, 3, generator=g))
    if K:
        out["jit_dv"] = torch.rand(N, K, generator=g)
    return out


def _qkw(m, r, compress=True):
    qp = m.ray_query_cfg["query_param"]
    from neuralsim_amd.fields.neus import fine_list
    return dict(num_coarse=qp["num_coarse"], num_fine=tuple(fine_list(qp)), upsample_inv_s=qp["upsample_inv_s"],
                upsample_inv_s_factors=tuple(qp["upsample_inv_s_factors"]), step_size=qp["march_cfg"]["step_size"],
                max_steps=qp["march_cfg"]["max_steps"], use_estimate_alpha=qp["upsample_use_estimate_alpha"],
                jitter=r["jit"], jitter_c=r["jit_c"], compress=compress, compress_thre=1e-4)


def _neus_grads(m, prefix=""):
    return {prefix + k: v for k, v in dict(grid=m.encoding.flattened_params.grad, sdf_w=m.sdf_w.grad, sdf_b=m.sdf_b.grad,
                                           rad_w=m.rad_w.grad, rad_b=m.rad_b.grad).items()}


def _zero(*mods):
    for mod in mods:
        if mod is not None:
            for q in mod.parameters():
                q.grad = None


def _discrete_leg(m, p, occ, r, near, far, rec, dev):
    """The sampling pass alone at the FULL ray count: hit rays, march counts and kept-sample counts vs the oracle."""
    a = m.accel.aabb.detach().cpu()
    with torch.no_grad():
        ret_o = orr.ray_query(p, r["o"], r["d"], r["ha"], occ, a[0], a[1], m.accel.resolution, near=near, far=far,
                              depth_use_normalized_vw=False, **_qkw(m, r))
    dv = lambda t: t.to(dev).contiguous()        # noqa: E731
    tested = m.ray_test(dv(r["o"]), dv(r["d"]), near=near, far=far)
    ri = ret_o["rays_inds"]
    assert tested["num_rays"] == ret_o["num_rays"] and torch.equal(tested["rays_inds"].cpu(), ri)
    m._march_stat = None
    cfg = dict(m.ray_query_cfg)
    cfg.update(with_rgb=False, with_normal=False, _jitter=dv(r["jit"][ri]), _jitter_c=dv(r["jit_c"][ri]))
    with torch.no_grad():
        _o, _d, t, pi, ridx, sdf_ng, mc, _g, _f = m._query_samples(tested, cfg, dict(cfg["query_param"]))
    assert torch.equal(mc.cpu(), ret_o["debug"]["march_counts"])
    n_p, n_o = pi[:, 1].cpu(), ret_o["pack_infos_tested"][:, 1]
    rec.update(rays=int(r["o"].shape[0]), hit=int(ri.shape[0]), marched=int(mc.sum()), sdf_queries=int(sdf_ng.shape[0]),
               kept=int(n_p.sum()), kept_oracle=int(n_o.sum()), rays_with_other_count=int((n_p != n_o).sum()))
    return ret_o


# ================================================================================================ street (configs[3])
@pytest.mark.parametrize("precision", ["f32", "fp16"])
def test_street_config_matches_oracle(backend, precision):
    from neuralsim_amd import scenarios as sc
    from neuralsim_amd.renderers.single_volume_renderer import SingleVolumeRenderer
    small, n_full, n_grad = _sizes(backend)
    dev = backend
    m, dm, sm = sc.street_models(dev, precision, seed=42, small=small)
    m.accel.init(m.query_sdf, generator=torch.Generator(device=dev).manual_seed(1))
    cfg_l = m.encoding.cfg
    if not small:
        assert cfg_l.num_levels == 19 and cfg_l.hashmap_size == 2 ** 20 and 30 * 2 ** 20 < cfg_l.n_params < 36 * 2 ** 20
        assert dm.cfg.n_params >= 15 * 2 ** 20 and m.accel.resolution == [200, 100, 30] and m.sdf_D == 1
    hw = 24 if small else 800
    intr, c2w, WH = sc.street_rig(n_ego=2 if small else 10, H=hw, W=hw, f=0.625 * hw)
    C, K = m.ray_query_cfg["query_param"]["num_coarse"], dm.K
    p = oracle_of_neus(m)
    occ = (m.accel.occ_val.detach().cpu() > m.accel.occ_thre)
    tol = _tol(precision, small)
    rec = dict(precision=precision, levels=cfg_l.num_levels, table_params=cfg_l.n_params, distant_params=dm.cfg.n_params,
               occupied=float(occ.float().mean()))
    # ---- (1) discrete decisions of the sampling pass at the full ray count
    r_full = _rays(intr, c2w, WH, n_full, seed=21, C=C)
    _discrete_leg(m, p, occ, r_full, 0.1, 200.0, rec, dev)
    assert rec["rays_with_other_count"] <= 2, rec              # a keep decision on the 1e-4 threshold may flip
    assert rec["kept"] > 4 * rec["hit"]
    # ---- (2) joint rendering (street + distant + sky), loss and every gradient on n_grad rays
    r = _rays(intr, c2w, WH, n_grad, seed=22, C=C, K=K)
    pd, (ws, bs) = oracle_of_distant(dm), oracle_of_sky(sm)
    for t_ in p.tensors() + pd.tensors() + ws + bs:
        t_.requires_grad_(True)
    ha_o = leaf(r["ha"])
    a = m.accel.aabb.detach().cpu()
    sc_o = orr.render_scene(p, r["o"], r["d"], ha_o, occ, a[0], a[1], m.accel.resolution,
                            query_kw=dict(near=0.1, far=200.0, **_qkw(m, r)), distant=pd,
                            distant_kw=dict(K=K, jitter=r["jit_dv"], include_inf=False), sky=(ws, bs))
    ro = sc_o["rendered"]
    w_eik = 0.01
    nab_o = sc_o["cr"]["volume_buffer"]["nablas"]
    loss_o = (ro["rgb_volume"] - r["gt"]).abs().mean() + w_eik * ((nab_o.norm(dim=-1) - 1.0) ** 2).mean()
    loss_o.backward()
    ref = oracle_flat_grads(p)
    ref.update(distant_flat_grads(pd))
    ref.update(sky_w=torch.cat([w.grad.reshape(-1) for w in ws]), sky_b=torch.cat([b.grad.reshape(-1) for b in bs]),
               h_appear=ha_o.grad)
    dv = lambda t: t.to(dev).contiguous()        # noqa: E731
    ri = sc_o["cr"]["rays_inds"]
    rend = SingleVolumeRenderer(dict(with_rgb=True, with_normal=True, near=0.1, far=200.0, depth_use_normalized_vw=False)).train()
    _zero(m, dm, sm)
    m._march_stat = None
    ha_p = leaf(r["ha"], dev)
    out = rend.render(m, rays=[dv(r["o"]), dv(r["d"])], rays_h_appear=ha_p, distant_model=dm, sky_model=sm,
                      return_buffer=True, return_details=True,
                      bypass_ray_query_cfg=dict(_jitter=dv(r["jit"][ri]), _jitter_c=dv(r["jit_c"][ri]), _jitter_dv=dv(r["jit_dv"])))
    rp = out["rendered"]
    vb = out["raw_per_obj_model"]["main"]["volume_buffer"]
    assert torch.equal(vb["rays_inds_hit"].cpu(), sc_o["cr"]["volume_buffer"]["rays_inds_hit"])
    n_p, n_o = vb["pack_infos_hit"][:, 1].cpu(), sc_o["cr"]["volume_buffer"]["pack_infos_hit"][:, 1]
    rec["grad_leg_rays_with_other_count"] = int((n_p != n_o).sum())
    assert torch.equal(out["raw_per_obj_model"]["distant"]["volume_buffer"]["valid"].cpu().bool(), sc_o["dv"]["valid"])
    for k in ("mask_volume", "depth_volume", "rgb_volume", "normals_volume"):
        rec["img_" + k] = float((rp[k].detach().cpu() - ro[k].detach()).abs().max())
    rec["sky_share"] = float((1.0 - ro["mask_volume"].detach()).mean())
    loss = (rp["rgb_volume"] - dv(r["gt"])).abs().mean() + w_eik * ((vb["nablas"].norm(dim=-1) - 1.0) ** 2).mean()
    loss.backward()
    rec["loss"], rec["loss_oracle"] = float(loss), float(loss_o)
    got = _neus_grads(m)
    got.update(ln_inv_s=m.ln_inv_s.grad, dv_grid=dm.flattened_params.grad, dv_den_w=dm.den_w.grad, dv_den_b=dm.den_b.grad,
               dv_rad_w=dm.rad_w.grad, dv_rad_b=dm.rad_b.grad, sky_w=sm.w.grad, sky_b=sm.b.grad, h_appear=ha_p.grad)
    for k, v in got.items():
        rec["grad_" + k] = rel_l2(v.cpu(), ref[k])
    _report(f"street_{precision}{'_small' if small else ''}", rec)
    assert rec["grad_leg_rays_with_other_count"] <= 1, rec
    assert 0.02 < rec["sky_share"] < 0.98                     # the distant model and the sky both carry part of the image
    for k, lim in tol["img"].items():
        assert rec["img_" + k] < lim, (k, rec["img_" + k])
    assert abs(rec["loss"] - rec["loss_oracle"]) < tol["loss"] * (1 + abs(rec["loss_oracle"]))
    if rec["grad_leg_rays_with_other_count"] == 0:
        for k in got:
            assert rec["grad_" + k] < tol["grad"], (k, rec["grad_" + k])


# ================================================================================================ indoor (configs[2])
@pytest.mark.parametrize("precision", ["f32", "fp16"])
def test_indoor_config_matches_oracle(backend, precision):
    from neuralsim_amd import scenarios as sc
    from neuralsim_amd.losses import mono_depth_loss, mono_normal_loss
    from neuralsim_amd.renderers.single_volume_renderer import SingleVolumeRenderer
    small, n_full, n_grad = _sizes(backend)
    dev = backend
    world = sc.indoor_world()
    m = sc.indoor_model(dev, precision, seed=42, small=small, world=world)
    assert m.inside_out
    m.accel.init(m.query_sdf, generator=torch.Generator(device=dev).manual_seed(1))
    hw = 24 if small else 800
    intr, c2w, WH = sc.indoor_rig(V=4 if small else 40, H=hw, W=hw, f=0.56 * hw)
    C = m.ray_query_cfg["query_param"]["num_coarse"]
    p = oracle_of_neus(m)
    occ = (m.accel.occ_val.detach().cpu() > m.accel.occ_thre)
    tol = _tol(precision, small)
    rec = dict(precision=precision, occupied=float(occ.float().mean()))
    r_full = _rays(intr, c2w, WH, n_full, seed=31, C=C)
    _discrete_leg(m, p, occ, r_full, 0.01, None, rec, dev)
    assert rec["rays_with_other_count"] <= 2 and rec["hit"] == n_full, rec           # the camera is inside the box
    # ---- image patch (rows 0 .. h*w-1) + pixel rays; photometric + eikonal + mono depth / normal losses
    ph = 6 if small else 32
    r = _rays(intr, c2w, WH, n_grad, seed=32, C=C)
    yy, xx = torch.meshgrid(torch.arange(ph), torch.arange(ph), indexing="ij")
    pxy = torch.stack([(xx.reshape(-1) + 5 + 0.5) / hw, (yy.reshape(-1) + 7 + 0.5) / hw], dim=-1)
    r["xy"][:ph * ph], r["fidx"][:ph * ph] = pxy, 1
    r["o"], r["d"] = orr.pinhole_rays(r["xy"], r["fidx"], intr, c2w, WH)
    tr_gt = world.trace(r["o"], r["d"])
    # monocular priors: depth up to scale and shift, both with a smooth error (priors that equal the model's own geometry
    # to the last bit make sign(n - n_gt) and the residuals of the scale-shift fit rounding noise: the f32 oracle then
    # differs from its own f64 evaluation by 13 % on the table gradient)
    gt_d, gt_n = sc.mono_priors(tr_gt["t"], tr_gt["normal"], r["o"] + tr_gt["t"][:, None] * r["d"])
    for t_ in p.tensors():
        t_.requires_grad_(True)
    ha_o = leaf(r["ha"])
    a = m.accel.aabb.detach().cpu()
    ret_o = orr.ray_query(p, r["o"], r["d"], ha_o, occ, a[0], a[1], m.accel.resolution, near=0.01, far=None,
                          depth_use_normalized_vw=False, **_qkw(m, r))
    assert ret_o["num_rays"] == n_grad
    ro = ret_o["rendered"]

    def total_loss(rr, nab, gt, gd, gn):
        occm = (rr["mask_volume"].detach() > 0.5).float()
        return ((rr["rgb_volume"] - gt) ** 2).mean() + 0.1 * ((nab.norm(dim=-1) - 1.0) ** 2).mean() + \
            0.05 * mono_normal_loss(rr["normals_volume"], gn, occm) + \
            0.1 * mono_depth_loss(rr["depth_volume"][:ph * ph], gd[:ph * ph], occm[:ph * ph])
    loss_o = total_loss(ro, ret_o["volume_buffer"]["nablas"], r["gt"], gt_d, gt_n)
    loss_o.backward()
    ref = oracle_flat_grads(p)
    ref["h_appear"] = ha_o.grad
    dv = lambda t: t.to(dev).contiguous()        # noqa: E731
    rend = SingleVolumeRenderer(dict(with_rgb=True, with_normal=True, near=0.01, depth_use_normalized_vw=False)).train()
    _zero(m)
    m._march_stat = None
    ha_p = leaf(r["ha"], dev)
    out = rend.render(m, rays=[dv(r["o"]), dv(r["d"])], rays_h_appear=ha_p, return_buffer=True, return_details=True,
                      bypass_ray_query_cfg=dict(_jitter=dv(r["jit"]), _jitter_c=dv(r["jit_c"])))
    rp = out["rendered"]
    vb = out["raw_per_obj_model"]["main"]["volume_buffer"]
    n_p, n_o = vb["pack_infos_hit"][:, 1].cpu(), ret_o["volume_buffer"]["pack_infos_hit"][:, 1]
    rec["grad_leg_rays_with_other_count"] = int((n_p != n_o).sum())
    for k in ("mask_volume", "depth_volume", "rgb_volume", "normals_volume"):
        rec["img_" + k] = float((rp[k].detach().cpu() - ro[k].detach()).abs().max())
    # the rendering IS the analytic room: depth to the walls, normals facing the camera
    hit_err = (ro["depth_volume"].detach() - tr_gt["t"]).abs().median()
    rec["depth_vs_world_median"] = float(hit_err)
    loss = total_loss(rp, vb["nablas"], dv(r["gt"]), dv(gt_d), dv(gt_n))
    loss.backward()
    rec["loss"], rec["loss_oracle"] = float(loss), float(loss_o)
    got = _neus_grads(m)
    got.update(ln_inv_s=m.ln_inv_s.grad, h_appear=ha_p.grad)
    for k, v in got.items():
        rec["grad_" + k] = rel_l2(v.cpu(), ref[k])
    _report(f"indoor_{precision}{'_small' if small else ''}", rec)
    assert rec["grad_leg_rays_with_other_count"] <= 1, rec
    assert rec["depth_vs_world_median"] < (0.2 if small else 0.03)
    for k, lim in tol["img"].items():
        assert rec["img_" + k] < min(lim, 2e-2 if precision == "fp16" else lim), (k, rec["img_" + k])
    assert abs(rec["loss"] - rec["loss_oracle"]) < tol["loss"] * (1 + abs(rec["loss_oracle"]))
    if rec["grad_leg_rays_with_other_count"] == 0:
        for k in got:
            assert rec["grad_" + k] < tol["grad"], (k, rec["grad_" + k])


def raws_counts(out, key):
    return out["raw_per_obj_model"][key]["volume_buffer"]["pack_infos_hit"][:, 1].cpu()


# ================================================================================================ multi (configs[4])
@pytest.mark.parametrize("precision", ["f32", "fp16"])
def test_multi_object_config_matches_oracle(backend, precision):
    """Background + 8 posed instances of the shared batched model + sky, composed per ray."""
    from neuralsim_amd import scenarios as sc
    from neuralsim_amd.renderers.buffer_compose_renderer import BufferComposeRenderer, Drawable
    small, n_full, n_grad = _sizes(backend)
    n_grad = min(n_grad, 1024)                       # the oracle composes ray by ray in Python
    dev = backend
    B = 3 if small else 8
    poses = sc.vehicle_poses(B)
    street, dm, sm = sc.street_models(dev, precision, seed=42, small=small)
    street.accel.init(street.query_sdf, generator=torch.Generator(device=dev).manual_seed(1))
    vm = sc.vehicle_model(dev, B, precision, seed=42, small=small)
    if not small:
        assert vm.accel.resolution == [32, 32, 32] and vm.num_instances == 8
    hw = 24 if small else 800
    intr, c2w, WH = sc.street_rig(n_ego=2 if small else 10, H=hw, W=hw, f=0.625 * hw)
    C = street.ray_query_cfg["query_param"]["num_coarse"]
    Cv = vm.ray_query_cfg["query_param"]["num_coarse"]
    p_s = oracle_of_neus(street)
    occ_s = (street.accel.occ_val.detach().cpu() > street.accel.occ_thre)
    # one oracle parameter set per instance: shared decoders (the SAME leaf tensors), that instance's table slice
    p_v0 = oracle_of_neus(vm)
    n_par = vm.n_params_per_instance
    full_grid = vm.encoding.flattened_params.detach().cpu().half().float()
    p_v, occ_v = [], []
    for b in range(B):
        q = ofield.FieldParams(spec=p_v0.spec, grid=full_grid[b * n_par:(b + 1) * n_par].clone(), sdf_w=p_v0.sdf_w,
                               sdf_b=p_v0.sdf_b, rad_w=p_v0.rad_w, rad_b=p_v0.rad_b, ln_inv_s=p_v0.ln_inv_s,
                               ln_inv_s_factor=p_v0.ln_inv_s_factor, sdf_scale=p_v0.sdf_scale)
        p_v.append(q)
        occ_v.append(vm.accel.occ_val.detach().cpu()[b * vm.accel.nvox:(b + 1) * vm.accel.nvox] > vm.accel.occ_thre)
    ws, bs = oracle_of_sky(sm)
    pd = oracle_of_distant(dm)
    r = _rays(intr, c2w, WH, n_grad, seed=41, C=max(C, Cv), K=dm.K)
    # aim a share of the rays at the vehicles so that every instance is in the batch
    g = torch.Generator().manual_seed(5)
    k_aim = n_grad // 2
    tgt = torch.stack([poses[i % B][1] for i in range(k_aim)]) + (torch.rand(k_aim, 3, generator=g) - 0.5) * 1.2
    r["d"][:k_aim] = torch.nn.functional.normalize(tgt - r["o"][:k_aim], dim=-1)
    leaves = p_s.tensors() + [q.grid for q in p_v] + p_v0.sdf_w + p_v0.sdf_b + p_v0.rad_w + p_v0.rad_b + [p_v0.ln_inv_s] + ws + bs \
        + pd.tensors()
    for t_ in leaves:
        t_.requires_grad_(True)
    ha_o = leaf(r["ha"])
    a_s, a_v = street.accel.aabb.detach().cpu(), vm.accel.aabb.detach().cpu()
    rs = dict(r, jit_c=r["jit_c"][:, :C])
    ret_s = orr.ray_query(p_s, r["o"], r["d"], ha_o, occ_s, a_s[0], a_s[1], street.accel.resolution, near=0.1, far=200.0,
                          depth_use_normalized_vw=False, **_qkw(street, rs))
    vbs = ret_s["volume_buffer"]
    bufs = [dict(rays_inds=vbs["rays_inds_hit"], pack_infos=vbs["pack_infos_hit"], t=vbs["t"], alpha=vbs["opacity_alpha"],
                 rgb=vbs["rgb"])]
    eik_terms = [((vbs["nablas"].norm(dim=-1) - 1.0) ** 2)]
    hit_items = 0
    rv = dict(r, jit_c=r["jit_c"][:, :Cv])
    for b, (R, t, s) in enumerate(poses):
        oo, dd = orr.convert_rays_in_node(r["o"], r["d"], R, t, s)
        rb = orr.ray_query(p_v[b], oo, dd, ha_o, occ_v[b], a_v[0], a_v[1], vm.accel.resolution, near=0.1, far=200.0,
                           depth_use_normalized_vw=False, **_qkw(vm, rv, compress=False))
        if rb["num_rays"] == 0 or rb["volume_buffer"]["type"] == "empty":
            continue
        hit_items += 1
        vbb = rb["volume_buffer"]
        bufs.append(dict(rays_inds=vbb["rays_inds_hit"], pack_infos=vbb["pack_infos_hit"], t=vbb["t"], alpha=vbb["opacity_alpha"],
                         rgb=vbb["rgb"]))
        eik_terms.append((vbb["nablas"].norm(dim=-1) - 1.0) ** 2)
    assert hit_items == B                                     # every vehicle is in view of the aimed rays
    # the distant model, queried last on ALL rays in the street's frame, from the street's ``far`` on (reference :506-531)
    near_dv = torch.full([n_grad], 0.1).index_put((ret_s["rays_inds"],), ret_s["far"])
    dvo = od.distant_ray_query(pd, r["o"], r["d"], near_dv, ha_o, a_s[0], a_s[1], K=dm.K, jitter=r["jit_dv"], include_inf=False)
    from oracle import pack_ops as opo
    bufs.append(dict(rays_inds=torch.arange(n_grad), pack_infos=opo.get_pack_infos_from_n(torch.full((n_grad,), dm.K)),
                     t=dvo["t"].flatten(), alpha=dvo["opacity_alpha"].flatten(), rgb=dvo["rgb"].flatten(0, 1)))
    mask_o, depth_o, rgb_o, cnt_o = orr.compose_buffers(bufs, n_grad, False)
    from oracle import sky as osky
    rgb_o = osky.blend_sky(rgb_o, mask_o, osky.sky_forward(torch.nn.functional.normalize(r["d"], dim=-1), ha_o, ws, bs))
    w_eik = 0.01
    loss_o = (rgb_o - r["gt"]).abs().mean() + w_eik * eik_terms[0].mean() + w_eik * torch.cat(eik_terms[1:]).mean()
    loss_o.backward()
    # ---- product
    dv = lambda t: t.to(dev).contiguous()        # noqa: E731
    drawables = [Drawable("street", "Street", street)] + \
        [Drawable(f"car{b}", "Vehicle", vm, rotation=R.to(dev), translation=t.to(dev), scale=s) for b, (R, t, s) in enumerate(poses)]
    rend = BufferComposeRenderer(dict(with_rgb=True, with_normal=True, near=0.1, far=200.0, depth_use_normalized_vw=False)).train()
    _zero(street, vm, sm, dm)
    street._march_stat = vm._march_stat = None
    ha_p = leaf(r["ha"], dev)
    ri_s = ret_s["rays_inds"]
    out = rend(dv(r["o"]), dv(r["d"]), drawables=drawables, rays_h_appear=ha_p, sky_model=sm, distant_model=dm,
               return_buffer=True, return_details=True,
               bypass_ray_query_cfg=dict(Street=dict(_jitter=dv(r["jit"][ri_s]), _jitter_c=dv(rs["jit_c"][ri_s])),
                                         Vehicle=dict(_jitter_full=dv(r["jit"]), _jitter_c_full=dv(rv["jit_c"])),
                                         Distant=dict(_jitter_dv=dv(r["jit_dv"]))))
    rp = out["rendered"]
    cnt_p = out["ray_intersections"]["samples_cnt"].cpu()
    # the object-space rays of the vehicles are computed on each side from the world rays (R^-1 (o - t) / s: three-term
    # sums whose rounding differs between the host and the device by an ulp), so a lattice sample on a voxel face may be
    # marched on one side only -- the background's counts (same rays on both sides) are bit-exact
    rec = dict(precision=precision, rays=n_grad, instances=B, rays_with_other_count=int((cnt_p != cnt_o).sum()),
               max_count_difference=int((cnt_p - cnt_o).abs().max()),
               street_counts_equal=bool(torch.equal(raws_counts(out, "street"), vbs["pack_infos_hit"][:, 1])),
               vehicle_samples=int(sum(int(b_["pack_infos"][:, 1].sum()) for b_ in bufs[1:-1])))
    rec["img_mask_volume"] = float((rp["mask_volume"].detach().cpu() - mask_o.detach()).abs().max())
    rec["img_rgb_volume"] = float((rp["rgb_volume"].detach().cpu() - rgb_o.detach()).abs().max())
    rec["img_depth_volume"] = float((rp["depth_volume"].detach().cpu() - depth_o.detach()).abs().max())
    raws = out["raw_per_obj_model"]
    eik_p = w_eik * ((raws["street"]["volume_buffer"]["nablas"].norm(dim=-1) - 1.0) ** 2).mean() + \
        w_eik * ((raws["Vehicle"]["volume_buffer"]["nablas"].norm(dim=-1) - 1.0) ** 2).mean()
    loss = (rp["rgb_volume"] - dv(r["gt"])).abs().mean() + eik_p
    loss.backward()
    rec["loss"], rec["loss_oracle"] = float(loss), float(loss_o)
    ref = {"st_" + k: v for k, v in oracle_flat_grads(p_s).items()}
    got = _neus_grads(street, "st_")
    got["st_ln_inv_s"] = street.ln_inv_s.grad
    gv = oracle_flat_grads(p_v0)
    gv["grid"] = torch.cat([q.grid.grad if q.grid.grad is not None else torch.zeros_like(q.grid) for q in p_v])
    ref.update({"veh_" + k: v for k, v in gv.items()})
    got.update(_neus_grads(vm, "veh_"))
    got["veh_ln_inv_s"] = vm.ln_inv_s.grad
    ref.update(sky_w=torch.cat([w.grad.reshape(-1) for w in ws]), sky_b=torch.cat([b.grad.reshape(-1) for b in bs]),
               h_appear=ha_o.grad)
    got.update(sky_w=sm.w.grad, sky_b=sm.b.grad, h_appear=ha_p.grad)
    ref.update(distant_flat_grads(pd))
    got.update(dv_grid=dm.flattened_params.grad, dv_den_w=dm.den_w.grad, dv_den_b=dm.den_b.grad, dv_rad_w=dm.rad_w.grad,
               dv_rad_b=dm.rad_b.grad)
    for k, v in got.items():
        rec["grad_" + k] = rel_l2(v.cpu(), ref[k])
    _report(f"multi_{precision}{'_small' if small else ''}", rec)
    tol = _tol(precision, small)
    assert rec["vehicle_samples"] > 0
    # (a keep decision of the background's compressed query sits on the 1e-4 weight threshold and may flip as well:
    # street_counts_equal is reported, the bound is on the whole ray)
    assert rec["rays_with_other_count"] <= max(2, n_grad // 100) and rec["max_count_difference"] <= 4, rec
    for k in ("mask_volume", "rgb_volume", "depth_volume"):
        assert rec["img_" + k] < tol["img"][k], (k, rec["img_" + k])
    assert abs(rec["loss"] - rec["loss_oracle"]) < tol["loss"] * (1 + abs(rec["loss_oracle"]))
    for k in got:
        # the vehicles are queried UN-compressed (their config's mode): most of their samples lie far from the surface,
        # where the eikonal residual |n| - 1 is a difference of cancelling terms -- the fp16 bound of the un-compressed
        # set (tests/test_fullsize_parity.py ``grad_full``: 1e-1; measured here 3.6e-2)
        lim = 1e-1 if (precision == "fp16" and k.startswith("veh_")) else tol["grad"]
        assert rec["grad_" + k] < lim, (k, rec["grad_" + k])
