"""NeRF++ distant-view model: shell sampling, fused 4-D LoTD + density/radiance MLPs, sigma->alpha,
forward and backward vs the oracle (oracle/distant.py)."""
import pytest
import torch

from oracle import distant as od, render as orr
from util import leaf, look_at_cameras, rel_l2

AABB = torch.tensor([[-1.0, -1, -1], [1.0, 1, 1]])


def _model_from(p, backend, precision, street=False):
    from neuralsim_amd.fields.nerf_distant import LoTDNeRFDistantModel
    m = LoTDNeRFDistantModel(aabb=AABB, precision=precision, max_steps=16, include_inf_distance=not street,
                             use_view_dirs=not street,
                             lotd_auto_compute_cfg=dict(target_num_params=2 ** 14, min_res_xyz=3, min_res_w=2,
                                                        log2_hashmap_size=10, per_level_scale=1.382))
    assert m.cfg.n_params == p.spec.n_params and m.cfg.res_xyz == p.spec.res_xyz
    with torch.no_grad():
        m.flattened_params.copy_(p.grid)
        m.den_w.copy_(torch.cat([w.reshape(-1) for w in p.den_w]))
        m.den_b.copy_(torch.cat([b.reshape(-1) for b in p.den_b]))
        m.rad_w.copy_(torch.cat([w.reshape(-1) for w in p.rad_w]))
        m.rad_b.copy_(torch.cat([b.reshape(-1) for b in p.rad_b]))
    return m.to(backend)


@pytest.mark.parametrize("precision,street", [("f32", False), ("fp16", False), ("f32", True)])
def test_distant_model_parity(backend, precision, street):
    """street = the street config's variant (withmask_withlidar_joint.240219.yaml:281-294): no view directions in the
    radiance net (first layer [64 x (F + 4)]), ``include_inf_distance: false``."""
    spec = od.make_ngp4d_spec(target_num_params=2 ** 14, min_res_xyz=3, min_res_w=2, log2_hashmap_size=10)
    assert "Hash" in spec.types and "Dense" in spec.types and spec.num_levels < 16
    p = od.make_distant_params(spec, grid_bound=0.5, use_view_dirs=not street)
    assert p.rad_w[0].shape[1] == 2 * spec.num_levels + (4 if street else 20)
    p.requires_grad_(True)
    g = torch.Generator().manual_seed(1)
    N, K = 21, 16
    intr, c2w, WH = look_at_cameras(V=3, seed=2)
    o, d = orr.pinhole_rays(torch.rand(N, 2, generator=g), torch.randint(0, 3, (N,), generator=g), intr, c2w, WH)
    o[::4] += torch.tensor([0.0, 3.0, 0.0])                     # rays that miss the close-range box
    _, far, hit = orr.aabb_ray_test(o, d, AABB[0], AABB[1], 0.01, None)
    near = torch.where(hit, far, torch.full_like(far, 0.01))
    assert 0 < int(hit.sum()) < N
    ha = torch.randn(N, 4, generator=g) * 0.3
    jit = torch.rand(N, K, generator=g)
    ha_o = leaf(ha)
    vbo = od.distant_ray_query(p, o, d, near, ha_o, AABB[0], AABB[1], K=K, jitter=jit, include_inf=not street)
    m = _model_from(p, backend, precision, street)
    if street:      # the last shell no longer absorbs everything
        assert float(vbo["opacity_alpha"][:, -1].min()) < 1.0
    dv = lambda t: t.to(backend).contiguous()
    ha_d = leaf(ha, backend)
    ret = m.ray_query(ray_tested=dict(rays_o=dv(o), rays_d=dv(d), near=dv(near), rays_h_appear=ha_d),
                      config=dict(_jitter_dv=dv(jit)), return_details=True)
    vb = ret["volume_buffer"]
    assert torch.equal(vb["valid"].cpu().bool(), vbo["valid"]) and 0.3 < float(vbo["valid"].float().mean()) < 1.0
    v = vbo["valid"]
    assert torch.allclose(vb["t"].cpu()[v], vbo["t"][v], rtol=1e-5, atol=1e-5)
    assert torch.allclose(ret["details"]["u4"].cpu().view(N, K, 4)[v], vbo["u4"][v], atol=2e-6)
    tol = dict(f32=2e-5, fp16=5e-3)[precision]
    assert (vb["sigma"].cpu() - vbo["sigma"])[v].abs().max() < tol * (1 + float(vbo["sigma"].max()))
    assert (vb["rgb"].cpu() - vbo["rgb"])[v].abs().max() < tol
    assert (vb["opacity_alpha"].cpu() - vbo["opacity_alpha"]).abs().max() < tol * 10
    wa, wr = torch.randn(N, K, generator=g), torch.randn(N, K, 3, generator=g)
    ((vbo["opacity_alpha"] * wa).sum() + (vbo["rgb"] * wr * v[..., None]).sum()).backward()
    ((vb["opacity_alpha"] * dv(wa)).sum() + (vb["rgb"] * dv(wr) * dv(v)[..., None]).sum()).backward()
    gtol = dict(f32=3e-4, fp16=3e-2)[precision]
    ref = dict(grid=p.grid.grad, den_w=torch.cat([w.grad.reshape(-1) for w in p.den_w]),
               den_b=torch.cat([b.grad.reshape(-1) for b in p.den_b]),
               rad_w=torch.cat([w.grad.reshape(-1) for w in p.rad_w]),
               rad_b=torch.cat([b.grad.reshape(-1) for b in p.rad_b]))
    got = dict(grid=m.flattened_params.grad, den_w=m.den_w.grad, den_b=m.den_b.grad, rad_w=m.rad_w.grad, rad_b=m.rad_b.grad)
    for k in ref:
        e = rel_l2(got[k].cpu(), ref[k])
        assert e < gtol, (k, e)
    assert rel_l2(ha_d.grad.cpu(), ha_o.grad) < gtol


def test_distant_cuboid_pyramid_parity(backend):
    """``lotd_use_cuboid: true`` on an elongated AABB (street config, withmask_withlidar_joint.240219.yaml:256): per-axis
    vertex counts of the 4-D pyramid -- values and table gradient against the oracle; a cubic AABB keeps cubic levels."""
    from neuralsim_amd.fields.nerf_distant import LoTDNeRFDistantModel
    aabb = torch.tensor([[-2.0, -1.0, -0.5], [2.0, 1.0, 0.5]])
    auto = dict(target_num_params=2 ** 15, min_res_xyz=3, min_res_w=2, log2_hashmap_size=11, per_level_scale=1.382)
    spec = od.make_ngp4d_spec(target_num_params=2 ** 15, min_res_xyz=3, min_res_w=2, log2_hashmap_size=11,
                              aspect=[4.0, 2.0, 1.0])
    assert spec.res3[0] == [12, 6, 3] and "Dense" in spec.types and "Hash" in spec.types
    p = od.make_distant_params(spec, grid_bound=0.5, use_view_dirs=False)
    p.requires_grad_(True)
    m = LoTDNeRFDistantModel(aabb=aabb, precision="f32", max_steps=8, include_inf_distance=False, use_view_dirs=False,
                             lotd_auto_compute_cfg=auto, lotd_use_cuboid=True)
    assert m.cfg.res3 == spec.res3 and m.cfg.n_params == spec.n_params and m.cfg.types == spec.types
    cube = LoTDNeRFDistantModel(aabb=AABB, precision="f32", max_steps=8, lotd_auto_compute_cfg=auto, lotd_use_cuboid=True)
    assert not cube.cfg.cuboid and cube.cfg.res3[0] == [3, 3, 3]
    # the reference hands the AABB over at populate time: the table is re-sized there
    late = LoTDNeRFDistantModel(precision="f32", max_steps=8, include_inf_distance=False, use_view_dirs=False,
                                lotd_auto_compute_cfg=auto, lotd_use_cuboid=True).populate(aabb=aabb)
    assert late.cfg.res3 == spec.res3 and torch.equal(late.aabb, aabb)
    with torch.no_grad():
        m.flattened_params.copy_(p.grid)
        m.den_w.copy_(torch.cat([w.reshape(-1) for w in p.den_w]))
        m.den_b.copy_(torch.cat([b.reshape(-1) for b in p.den_b]))
        m.rad_w.copy_(torch.cat([w.reshape(-1) for w in p.rad_w]))
        m.rad_b.copy_(torch.cat([b.reshape(-1) for b in p.rad_b]))
    m = m.to(backend)
    g = torch.Generator().manual_seed(4)
    N, K = 19, 8
    o = (torch.rand(N, 3, generator=g) - 0.5) * torch.tensor([3.0, 1.5, 0.8])
    d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)
    near = torch.full([N], 0.01)
    ha = torch.randn(N, 4, generator=g) * 0.3
    vbo = od.distant_ray_query(p, o, d, near, ha, aabb[0], aabb[1], K=K, include_inf=False)
    dv = lambda t: t.to(backend).contiguous()         # noqa: E731
    ret = m.ray_query(ray_tested=dict(rays_o=dv(o), rays_d=dv(d), near=dv(near), rays_h_appear=dv(ha)), config={})
    vb = ret["volume_buffer"]
    v = vbo["valid"]
    assert torch.equal(vb["valid"].cpu().bool(), v) and float(v.float().mean()) > 0.5
    assert (vb["sigma"].cpu() - vbo["sigma"])[v].abs().max() < 3e-5 * (1 + float(vbo["sigma"].detach().max()))
    assert (vb["rgb"].cpu() - vbo["rgb"])[v].abs().max() < 3e-5
    w = torch.randn(N, K, generator=g)
    (vbo["sigma"] * w * v).sum().backward()
    (vb["sigma"] * dv(w) * dv(v)).sum().backward()
    assert rel_l2(m.flattened_params.grad.cpu(), p.grid.grad) < 3e-4


def test_ngp4d_levels_match_config():
    """lotd_neus.dtu.230814.yaml:193-200: 8 Mi target -> 12 levels (5 dense + 7 hashed), 24 features."""
    from neuralsim_amd.fields.nerf_distant import LoTD4Config
    c = LoTD4Config()
    s = od.make_ngp4d_spec()
    assert (c.num_levels, c.n_params, c.res_xyz, c.res_w, c.types) == (s.num_levels, s.n_params, s.res_xyz, s.res_w, s.types)
    assert c.num_levels == 12 and c.types.count("Dense") == 5 and c.n_params >= 8 * 2 ** 20


def test_renderer_merges_close_range_and_distant(backend):
    """single_volume_renderer.py:281-442: cr (packed) + dv (batched, all rays) -> merge_two_packs_sorted -> integrate."""
    from oracle import pack_ops as opo
    from neuralsim_amd.fields.neus import OccGridAccel
    from neuralsim_amd.renderers.single_volume_renderer import SingleVolumeRenderer
    from util import make_params, model_from_params
    p = make_params(sdf_D=1, small=True, sphere=True, seed=3, ln_inv_s=0.45, grid_bound=2e-2, noise_scale=1.0)
    spec = od.make_ngp4d_spec(target_num_params=2 ** 14, min_res_xyz=3, min_res_w=2, log2_hashmap_size=10)
    pd = od.make_distant_params(spec, grid_bound=0.5)
    g = torch.Generator().manual_seed(5)
    N, K = 30, 16
    intr, c2w, WH = look_at_cameras(V=3, seed=3)
    o, d = orr.pinhole_rays(torch.rand(N, 2, generator=g) * 0.6 + 0.2, torch.randint(0, 3, (N,), generator=g), intr, c2w, WH)
    o[::6] += torch.tensor([0.0, 4.0, 0.0])
    ha = torch.randn(N, 4, generator=g) * 0.3
    res = [16, 16, 16]
    val, occ = orr.build_occ_grid(p, AABB[0], AABB[1], res, n_pts=2 ** 13, n_steps=2)
    kw = dict(near=0.01, num_coarse=8, num_fine=(4, 4), upsample_inv_s_factors=(1, 4), step_size=0.05, max_steps=128)
    with torch.no_grad():
        cr = orr.ray_query(p, o, d, ha, occ, AABB[0], AABB[1], res, depth_use_normalized_vw=False, **kw)
        _, far, hit = orr.aabb_ray_test(o, d, AABB[0], AABB[1], 0.01, None)
        near_dv = torch.where(hit, far, torch.full_like(far, 0.01))
        dvo = od.distant_ray_query(pd, o, d, near_dv, ha, AABB[0], AABB[1], K=K)
        vbc = cr["volume_buffer"]
        pi_dv = opo.get_pack_infos_from_n(torch.full((N,), K))
        pidx_dv, pidx_cr, pi_tot = opo.merge_two_packs_sorted(dvo["t"].flatten(), pi_dv, torch.arange(N), vbc["t"],
                                                               vbc["pack_infos_hit"], vbc["rays_inds_hit"])
        S = N * K + vbc["t"].shape[0]
        tt, aa, cc = torch.zeros(S), torch.zeros(S), torch.zeros(S, 3)
        tt[pidx_dv], tt[pidx_cr] = dvo["t"].flatten(), vbc["t"]
        aa[pidx_dv], aa[pidx_cr] = dvo["opacity_alpha"].flatten(), vbc["opacity_alpha"]
        cc[pidx_dv], cc[pidx_cr] = dvo["rgb"].flatten(0, 1), vbc["rgb"]
        ref = orr.volume_integration(aa, tt, cc, None, pi_tot, False)
    model = model_from_params(p, backend, precision="f32")
    model.accel = OccGridAccel(AABB, resolution=res, device=backend)
    model.accel.occ_val.copy_(val.to(backend))
    model.accel.pack_bits()
    model.ray_query_cfg = dict(query_mode="march_occ_multi_upsample",
                               query_param=dict(num_coarse=8, num_fine=[4, 4], upsample_inv_s=64.0, upsample_inv_s_factors=[1, 4],
                                                upsample_use_estimate_alpha=True, march_cfg=dict(step_size=0.05, max_steps=128)))
    dm = _model_from(pd, backend, "f32")
    rend = SingleVolumeRenderer(dict(with_rgb=True, near=0.01, depth_use_normalized_vw=False)).train()
    dv = lambda t: t.to(backend).contiguous()
    out = rend.render(model, rays=[dv(o), dv(d)], rays_h_appear=dv(ha), distant_model=dm, return_buffer=True)
    assert torch.equal(out["volume_buffer"]["pack_infos_hit"].cpu(), pi_tot)
    assert torch.equal(out["ray_intersections"]["samples_cnt"].cpu(), pi_tot[:, 1])
    for k in ("mask_volume", "depth_volume", "rgb_volume"):
        e = (out["rendered"][k].cpu() - ref[k]).abs().max()
        assert e < (2e-2 if k == "depth_volume" else 3e-4), (k, float(e))
    assert float(out["rendered"]["mask_volume"].min()) > 0.99          # include_inf_distance: every ray is opaque
    out["rendered"]["rgb_volume"].sum().backward()
    assert float(dm.flattened_params.grad.abs().sum()) > 0 and float(model.sdf_w.grad.abs().sum()) > 0


def test_backward_skips_shells_behind_an_opaque_stretch(backend):
    """The renderer hands the distant model's backward a transmittance mask (``distant_bwd_trans_thre``, default 1e-3; here 1e-4 like the
    compressed query's weight threshold): shells whose transmittance in the JOINT ray is below it get no MLP backward and
    no table scatter.  Against the unmasked backward the gradients move by the dropped weights only."""
    from neuralsim_amd.fields.neus import OccGridAccel
    from neuralsim_amd.renderers.single_volume_renderer import SingleVolumeRenderer
    from util import make_params, model_from_params
    p = make_params(sdf_D=1, small=True, sphere=True, seed=3, ln_inv_s=0.7, grid_bound=2e-2, noise_scale=1.0)   # sharp surface
    spec = od.make_ngp4d_spec(target_num_params=2 ** 14, min_res_xyz=3, min_res_w=2, log2_hashmap_size=10)
    pd = od.make_distant_params(spec, grid_bound=0.5)
    g = torch.Generator().manual_seed(5)
    N = 40
    intr, c2w, WH = look_at_cameras(V=3, seed=3)
    o, d = orr.pinhole_rays(torch.rand(N, 2, generator=g) * 0.6 + 0.2, torch.randint(0, 3, (N,), generator=g), intr, c2w, WH)
    o[::5] += torch.tensor([0.0, 4.0, 0.0])                   # rays that see the background only
    ha = torch.randn(N, 4, generator=g) * 0.3
    res = [16, 16, 16]
    val, _ = orr.build_occ_grid(p, AABB[0], AABB[1], res, n_pts=2 ** 13, n_steps=2)
    model = model_from_params(p, backend, precision="f32")
    model.accel = OccGridAccel(AABB, resolution=res, device=backend)
    model.accel.occ_val.copy_(val.to(backend))
    model.accel.pack_bits()
    model.ray_query_cfg = dict(query_mode="march_occ_multi_upsample",
                               query_param=dict(num_coarse=8, num_fine=[4, 4], upsample_inv_s=64.0, upsample_inv_s_factors=[1, 4],
                                                upsample_use_estimate_alpha=True, march_cfg=dict(step_size=0.05, max_steps=128)))
    dm = _model_from(pd, backend, "f32")
    dv = lambda t: t.to(backend).contiguous()         # noqa: E731
    wgt = torch.rand(N, 3, generator=g)
    grads, kept = {}, {}
    for thre in (0.0, 1e-4):
        rend = SingleVolumeRenderer(dict(with_rgb=True, near=0.01, depth_use_normalized_vw=False,
                                         distant_bwd_trans_thre=thre)).train()
        for q in dm.parameters():
            q.grad = None
        out = rend.render(model, rays=[dv(o), dv(d)], rays_h_appear=dv(ha), distant_model=dm, return_buffer=True,
                          return_details=True)
        holder = out["raw_per_obj_model"]["distant"]["_bwd_holder"]
        kept[thre] = holder.get("keep")
        (out["rendered"]["rgb_volume"] * dv(wgt)).sum().backward()
        grads[thre] = [q.grad.clone() for q in (dm.flattened_params, dm.den_w, dm.rad_w)]
    assert kept[0.0] is None
    k = kept[1e-4].view(N, -1).cpu()
    mask = out["rendered"]["mask_volume"].detach().cpu()
    assert 0 < int((k == 0).sum()) < k.numel()               # some shells sit behind the surface, some rays see far
    # the dropped shells carry < 1e-4 of the pixel: every gradient moves by about that much, not more
    for a, b in zip(grads[1e-4], grads[0.0]):
        assert rel_l2(a.cpu(), b.cpu()) < 2e-3
    assert float((grads[1e-4][0] != 0).float().mean()) <= float((grads[0.0][0] != 0).float().mean())
    assert float(mask.max()) > 0.99
