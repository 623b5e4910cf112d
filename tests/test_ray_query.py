"""End-to-end ``ray_test`` + ``ray_query`` (march_occ_multi_upsample) + loss + backward vs the oracle on identical
rays / weights / perturbation randoms."""
import pytest
import torch

from oracle import render as orr
from neuralsim_amd.fields.neus import volume_integration
from util import leaf, look_at_cameras, make_params, model_from_params, oracle_flat_grads, rel_l2

AABB = torch.tensor([[-1.0, -1, -1], [1.0, 1, 1]])
RES = [32, 32, 32]


def _setup(backend, precision, N=48, perturb=True, sdf_D=2, seed=3):
    # a "bumpy sphere": large enough hash noise that (|nablas| - 1) is O(0.1), which keeps the eikonal gradient
    # well conditioned in f32 (on the exact sphere init it is a 1e-3 residual of cancelling terms and even the
    # f32 oracle is 2% away from its own f64 evaluation)
    p = make_params(sdf_D=sdf_D, small=True, sphere=True, seed=seed, ln_inv_s=0.45, grid_bound=2e-2, noise_scale=1.0)
    for t in p.tensors():
        t.requires_grad_(True)
    g = torch.Generator().manual_seed(seed)
    intr, c2w, WH = look_at_cameras(V=3, seed=seed)
    xy = torch.rand(N, 2, generator=g) * 0.5 + 0.25
    fidx = torch.randint(0, 3, (N,), generator=g)
    o, d = orr.pinhole_rays(xy, fidx, intr, c2w, WH)
    o[::7] += torch.tensor([0.0, 4.0, 0.0])          # some rays miss the box
    h_appear = torch.randn(N, 4, generator=g) * 0.3
    model = model_from_params(p, backend, precision=precision)
    model.accel.resolution = RES
    from neuralsim_amd.fields.neus import OccGridAccel
    model.accel = OccGridAccel(AABB, resolution=RES, device=backend)
    val, occ = orr.build_occ_grid(p, AABB[0], AABB[1], RES, n_pts=2 ** 14, n_steps=2)
    model.accel.occ_val.copy_(val.to(backend))
    model.accel.pack_bits()
    jit = torch.rand(N, generator=g) if perturb else None
    jit_c = torch.rand(N, 16, generator=g) if perturb else None
    return p, model, o, d, h_appear, occ, jit, jit_c, g


QP = dict(nablas_has_grad=True, num_coarse=16, num_fine=[4, 4, 8], upsample_inv_s=64.0, upsample_inv_s_factors=[1, 4, 16],
          upsample_use_estimate_alpha=True, march_cfg=dict(step_size=0.02, max_steps=512))


@pytest.mark.parametrize("compressed", [False, True])
@pytest.mark.parametrize("precision", ["f32", "fp16"])
def test_ray_query_parity(backend, precision, compressed):
    p, model, o, d, h_appear, occ, jit, jit_c, g = _setup(backend, precision)
    N = o.shape[0]
    ret_o = orr.ray_query(p, o, d, h_appear, occ, AABB[0], AABB[1], RES, near=0.01, far=None, num_coarse=16,
                          num_fine=(4, 4, 8), step_size=0.02, max_steps=512, jitter=jit, jitter_c=jit_c,
                          depth_use_normalized_vw=False, compress=compressed, compress_thre=1e-3)
    dv = lambda a: a.to(backend).contiguous()
    tested = model.ray_test(dv(o), dv(d), near=0.01, far=None, rays_h_appear=dv(h_appear))
    assert tested["num_rays"] == ret_o["num_rays"] and torch.equal(tested["rays_inds"].cpu(), ret_o["rays_inds"])
    ri = ret_o["rays_inds"]
    cfg = dict(query_param=dict(QP, compress_thre=1e-3), with_rgb=True, with_normal=True, depth_use_normalized_vw=False,
               _render=True, _jitter=dv(jit[ri]), _jitter_c=dv(jit_c[ri]),
               query_mode="march_occ_multi_upsample" + ("_compressed" if compressed else ""))
    ret = model.ray_query(ray_tested=tested, config=cfg, return_details=True)
    vb, vbo = ret["volume_buffer"], ret_o["volume_buffer"]
    assert torch.equal(ret["details"]["march_counts"].cpu(), ret_o["debug"]["march_counts"])
    if compressed:
        kept, total = int(vbo["pack_infos_hit"][:, 1].sum()), int(ret_o["debug"]["pack_infos"][:, 1].sum())
        assert 0 < kept < total                       # something was dropped, something survived
    if precision == "f32":                            # fp16 SDFs may flip a keep decision right at the threshold
        assert torch.equal(vb["pack_infos_hit"].cpu(), vbo["pack_infos_hit"])
    # f32 MFMA path: per-sample parity.  fp16 MFMA path: the up-sampler amplifies the 1e-3 fp16 SDF error by
    # inv_s = 1024, so individual samples may move; the rendered per-ray values are what must agree.
    tt = dict(f32=(1e-4, 2e-4), fp16=(None, 3e-2))[precision]
    if tt[0] is not None:
        assert (vb["t"].cpu() - vbo["t"]).abs().max() < tt[0]
    if compressed:                                    # compression changes the rendering by O(thre) only
        full = orr.ray_query(p, o, d, h_appear, occ, AABB[0], AABB[1], RES, near=0.01, far=None, num_coarse=16,
                             num_fine=(4, 4, 8), step_size=0.02, max_steps=512, jitter=jit, jitter_c=jit_c,
                             depth_use_normalized_vw=False)
        assert (full["rendered"]["rgb_volume"] - ret_o["rendered"]["rgb_volume"]).abs().max() < 2e-2
    for k in ("mask_volume", "depth_volume", "rgb_volume", "normals_volume"):
        err = (ret["rendered"][k].cpu() - ret_o["rendered"][k]).abs()
        assert err.max() < tt[1] * (3 if k == "depth_volume" else 1), (k, float(err.max()))
    if precision != "f32":
        return
    # loss + backward (photometric mse + eikonal on the render samples)
    gt = torch.rand(N, 3, generator=g)
    loss_o, _ = orr.render_loss(ret_o, gt, N, w_eikonal=0.1)
    loss_o.backward()
    rgb_full = torch.zeros(N, 3, device=backend).index_put((tested["rays_inds"],), ret["rendered"]["rgb_volume"])
    loss = ((rgb_full - dv(gt)) ** 2).mean() + 0.1 * ((vb["nablas"].norm(dim=-1) - 1.0) ** 2).mean()
    assert abs(float(loss) - float(loss_o)) < 1e-5
    loss.backward()
    ref = oracle_flat_grads(p)
    got = dict(grid=model.encoding.flattened_params.grad, sdf_w=model.sdf_w.grad, sdf_b=model.sdf_b.grad,
               rad_w=model.rad_w.grad, rad_b=model.rad_b.grad, ln_inv_s=model.ln_inv_s.grad)
    for k, v in got.items():
        e = rel_l2(v.cpu(), ref[k])
        assert e < 5e-3, (k, e)


@pytest.mark.parametrize("compressed", [False, True])
@pytest.mark.parametrize("sampling", ["f32", "split"])
def test_f32_sampling_pass_fixes_the_discrete_decisions_of_an_fp16_step(backend, compressed, sampling):
    """``sampling_precision``: the no-grad SDF queries of the sampling pass run on the exact-f32 kernels ("f32") or on the
    f16 matrix cores with hi + lo operands ("split", the default: f32-equivalent), the with-grad query on the fp16 ones
    -- the sample set (counts, depths) is then the f32 oracle's, the rendered values carry only the fp16 error of the
    field ON that set."""
    p, model, o, d, h_appear, occ, jit, jit_c, g = _setup(backend, "fp16")
    model.sampling_precision = sampling
    ret_o = orr.ray_query(p, o, d, h_appear, occ, AABB[0], AABB[1], RES, near=0.01, far=None, num_coarse=16,
                          num_fine=(4, 4, 8), step_size=0.02, max_steps=512, jitter=jit, jitter_c=jit_c,
                          depth_use_normalized_vw=False, compress=compressed, compress_thre=1e-3)
    dv = lambda a: a.to(backend).contiguous()       # noqa: E731
    tested = model.ray_test(dv(o), dv(d), near=0.01, far=None, rays_h_appear=dv(h_appear))
    ri = ret_o["rays_inds"]
    cfg = dict(query_param=dict(QP, compress_thre=1e-3), with_rgb=True, with_normal=True, depth_use_normalized_vw=False,
               _render=True, _jitter=dv(jit[ri]), _jitter_c=dv(jit_c[ri]),
               query_mode="march_occ_multi_upsample" + ("_compressed" if compressed else ""))
    ret = model.ray_query(ray_tested=tested, config=cfg, return_details=True)
    vb, vbo = ret["volume_buffer"], ret_o["volume_buffer"]
    assert torch.equal(vb["pack_infos_hit"].cpu(), vbo["pack_infos_hit"])         # same kept set as the f32 oracle
    assert (vb["t"].cpu() - vbo["t"]).abs().max() < 1e-4                          # ... at the same depths
    assert (ret["details"]["sdf_nograd"].cpu() - ret_o["debug"]["sdf_nograd"]).abs().max() < 2e-5
    for k in ("mask_volume", "depth_volume", "rgb_volume", "normals_volume"):
        err = (ret["rendered"][k].cpu() - ret_o["rendered"][k]).abs()
        assert err.max() < 1e-2 * (3 if k == "depth_volume" else 1), (k, float(err.max()))
    # the with-grad values are the fp16 kernels' (they differ from the f32 ones in the last bits of an fp16 operand)
    assert 1e-7 < float((vb["sdf"].detach().cpu() - vbo["sdf"].detach()).abs().max()) < 2e-3


def test_ray_query_empty_and_no_perturb(backend):
    p, model, o, d, h_appear, occ, _, _, g = _setup(backend, "f32", N=16, perturb=False)
    dv = lambda a: a.to(backend).contiguous()
    far_away = o + torch.tensor([0.0, 50.0, 0.0])
    tested = model.ray_test(dv(far_away), dv(d), near=0.01, far=None)
    assert tested["num_rays"] == 0
    assert model.ray_query(ray_tested=tested, config=dict(query_param=QP))["volume_buffer"]["type"] == "empty"
    ret_o = orr.ray_query(p, o, d, None, occ, AABB[0], AABB[1], RES, num_coarse=16, num_fine=(4, 4, 8), step_size=0.02,
                          max_steps=512, depth_use_normalized_vw=True)
    tested = model.ray_test(dv(o), dv(d), near=0.01, far=None)
    ret = model.ray_query(ray_tested=tested, config=dict(query_param=QP, with_rgb=True, _render=True,
                                                         depth_use_normalized_vw=True,
                                                         query_mode="march_occ_multi_upsample"))
    for k in ("mask_volume", "depth_volume", "rgb_volume"):
        assert (ret["rendered"][k].cpu() - ret_o["rendered"][k]).abs().max() < 2e-4, k
    # reference-style integration from the returned volume buffer (single_volume_renderer.py:73-102)
    vb = ret["volume_buffer"]
    again = volume_integration(vb["opacity_alpha"], vb["t"], vb["rgb"], None, vb["pack_infos_hit"], True)
    # ``rendered`` is per TESTED ray; the buffer lists the rays that produced samples (upsample_on_marched_only): a subset here
    assert vb["rays_inds_hit"].shape[0] < tested["num_rays"] and vb["pack_infos_hit"].shape[0] == vb["rays_inds_hit"].shape[0]
    assert bool((vb["pack_infos_hit"][:, 1] > 0).all())
    full = torch.zeros(o.shape[0], 3, device=backend)
    assert torch.allclose(full.index_put((vb["rays_inds_hit"],), again["rgb_volume"]),
                          full.index_put((tested["rays_inds"],), ret["rendered"]["rgb_volume"]))
    assert torch.equal(vb["rays_inds_hit"].cpu(), ret_o["volume_buffer"]["rays_inds_hit"])
    assert torch.equal(vb["pack_infos_hit"].cpu(), ret_o["volume_buffer"]["pack_infos_hit"])
    # ``upsample_on_marched_only: false`` (rounds 1-4): every AABB-tested ray gets coarse + fine samples and is in the buffer
    qp_all = dict(QP, upsample_on_marched_only=False)
    ret_a = model.ray_query(ray_tested=tested, config=dict(query_param=qp_all, with_rgb=True, _render=True,
                                                           depth_use_normalized_vw=True, query_mode="march_occ_multi_upsample"))
    ret_ao = orr.ray_query(p, o, d, None, occ, AABB[0], AABB[1], RES, num_coarse=16, num_fine=(4, 4, 8), step_size=0.02,
                           max_steps=512, depth_use_normalized_vw=True, upsample_on_marched_only=False)
    vba = ret_a["volume_buffer"]
    assert vba["rays_inds_hit"].shape[0] == tested["num_rays"] and torch.equal(vba["pack_infos_hit"].cpu(), ret_ao["volume_buffer"]["pack_infos_hit"])
    assert int(vba["t"].shape[0]) > int(vb["t"].shape[0])
    for k in ("mask_volume", "depth_volume", "rgb_volume"):
        assert (ret_a["rendered"][k].cpu() - ret_ao["rendered"][k]).abs().max() < 2e-4, k


def test_extra_points_ride_on_the_render_launches(backend):
    """``_extra_pts`` (the trainer's uniform eikonal points appended as zero-length rays): identical samples, identical
    values, and the gradient equals the sum of the two separate evaluations."""
    p, model, o, d, h_appear, occ, jit, jit_c, g = _setup(backend, "f32", N=32)
    dv = lambda a: a.to(backend).contiguous()
    x = (torch.rand(100, 3, generator=g) * 1.6 - 0.8)

    def run(extra):
        for prm in (model.encoding.flattened_params, model.sdf_w, model.sdf_b, model.rad_w, model.rad_b, model.ln_inv_s):
            prm.grad = None
        ha = leaf(h_appear, backend)
        tested = model.ray_test(dv(o), dv(d), near=0.01, far=None, rays_h_appear=ha)
        ri = tested["rays_inds"].cpu()
        cfg = dict(query_param=QP, with_rgb=True, with_normal=True, depth_use_normalized_vw=False, _render=True,
                   _jitter=dv(jit[ri]), _jitter_c=dv(jit_c[ri]), query_mode="march_occ_multi_upsample")
        if extra:
            cfg["_extra_pts"] = dv(x)
        ret = model.ray_query(ray_tested=tested, config=cfg)
        uni = ret["extra_pts"] if extra else model.forward_sdf_nablas(dv(x))
        loss = ret["rendered"]["rgb_volume"].sum() + 0.1 * ((ret["volume_buffer"]["nablas"].norm(dim=-1) - 1) ** 2).mean() \
            + 0.3 * ((uni["nablas"].norm(dim=-1) - 1) ** 2).mean() + 0.05 * uni["sdf"].sum()
        loss.backward()
        grads = [prm.grad.detach().cpu().clone() for prm in (model.encoding.flattened_params, model.sdf_w, model.sdf_b,
                                                             model.rad_w, model.rad_b, model.ln_inv_s)] + [ha.grad.cpu()]
        return ret, uni, float(loss), grads

    ret_a, uni_a, loss_a, g_a = run(False)
    ret_b, uni_b, loss_b, g_b = run(True)
    assert torch.equal(ret_a["volume_buffer"]["t"], ret_b["volume_buffer"]["t"])
    assert torch.equal(ret_a["volume_buffer"]["sdf"], ret_b["volume_buffer"]["sdf"])
    assert ret_b["volume_buffer"]["rgb"].shape == ret_a["volume_buffer"]["rgb"].shape
    assert torch.allclose(uni_a["sdf"], uni_b["sdf"], atol=1e-6) and torch.allclose(uni_a["nablas"], uni_b["nablas"], atol=1e-5)
    assert abs(loss_a - loss_b) < 1e-4 * abs(loss_a)
    for ga, gb in zip(g_a, g_b):
        assert rel_l2(gb, ga) < 1e-4


def test_speculative_sample_buffers(backend):
    """Compressed mode sizes its sampling buffers from the previous call's density (no read-back of the marched
    total); the result must be identical to the exact path -- also when the guess was too small (redo) or when a
    pack was emptied on the device."""
    p, model, o, d, h_appear, occ, jit, jit_c, g = _setup(backend, "f32", N=40)
    dv = lambda a: a.to(backend).contiguous()
    tested = model.ray_test(dv(o), dv(d), near=0.01, far=None, rays_h_appear=dv(h_appear))
    ri = tested["rays_inds"].cpu()
    cfg = dict(query_param=dict(QP, compress_thre=1e-3), with_rgb=True, with_normal=True, _render=True,
               _jitter=dv(jit[ri]), _jitter_c=dv(jit_c[ri]), query_mode="march_occ_multi_upsample_compressed")
    model._speculate = True
    model._march_stat = None
    ref = model.ray_query(ray_tested=tested, config=cfg, return_details=True)          # exact (no history)
    R, M = model._march_stat
    assert R == tested["num_rays"] and M == int(ref["details"]["march_counts"].sum())
    for stat in ((R, M), (R, 3 * M), (R, max(1, M // 50)), None):       # good guess, loose guess, overflow -> redo, exact
        model._march_stat = stat
        out = model.ray_query(ray_tested=tested, config=cfg, return_details=True)
        assert model._march_stat == (R, M)
        for k in ("t", "sdf", "rgb", "opacity_alpha"):
            assert torch.equal(out["volume_buffer"][k], ref["volume_buffer"][k]), (stat, k)
        assert torch.equal(out["volume_buffer"]["pack_infos_hit"], ref["volume_buffer"]["pack_infos_hit"])
        assert torch.equal(out["rendered"]["rgb_volume"], ref["rendered"]["rgb_volume"])


def test_var_ctrl_mix_linear(backend):
    """``var_ctrl_cfg{ctrl_type: mix_linear}``: the scheduled inv_s renders exactly like a model whose ln_inv_s holds
    the blended value, the learned parameter receives (1 - a) exp(ln f) / inv_s of that model's gradient, and the
    schedule end points are the learned value (a = 0) and ``final_inv_s`` (a = 1)."""
    import math
    p, model, o, d, h_appear, occ, jit, jit_c, g = _setup(backend, "f32")
    dv = lambda a: a.to(backend).contiguous()
    tested = model.ray_test(dv(o), dv(d), near=0.01, far=None, rays_h_appear=dv(h_appear))
    ri = tested["rays_inds"].cpu()
    cfg = dict(query_param=dict(QP, compress_thre=1e-3), with_rgb=True, with_normal=True, _render=True,
               _jitter=dv(jit[ri]), _jitter_c=dv(jit_c[ri]), query_mode="march_occ_multi_upsample_compressed")
    f, ln0 = model.ln_inv_s_factor, float(model.ln_inv_s)
    model.set_var_ctrl("mix_linear", start_it=100, stop_it=300, final_inv_s=400.0)
    model.training_before_per_step(50)
    assert model._ctrl_mix == 0.0 and abs(float(model.forward_inv_s()) - math.exp(ln0 * f)) < 1e-3
    model.training_before_per_step(1000)
    assert model._ctrl_mix == 1.0 and abs(float(model.forward_inv_s()) - 400.0) < 1e-2
    model.training_before_per_step(150)
    a = 0.25
    inv_eff = (1 - a) * math.exp(ln0 * f) + a * 400.0
    assert model._ctrl_mix == a and abs(float(model.forward_inv_s()) - inv_eff) < 1e-2

    def run():
        model.zero_grad()
        ret = model.ray_query(ray_tested=tested, config=cfg)
        (ret["rendered"]["rgb_volume"].square().sum() + ret["rendered"]["mask_volume"].sum()).backward()
        return ret, float(model.ln_inv_s.grad), model.sdf_w.grad.clone()
    ret_c, g_c, gw_c = run()
    # the same model without a controller, ln_inv_s set to the blended value
    model.set_var_ctrl(None)
    with torch.no_grad():
        model.ln_inv_s.fill_(math.log(inv_eff) / f)
    ret_r, g_r, gw_r = run()
    for k in ("rgb_volume", "mask_volume", "depth_volume"):
        assert torch.allclose(ret_c["rendered"][k], ret_r["rendered"][k], atol=2e-6), k
    assert torch.equal(ret_c["volume_buffer"]["pack_infos_hit"], ret_r["volume_buffer"]["pack_infos_hit"])
    assert rel_l2(gw_c.cpu(), gw_r.cpu()) < 1e-5
    chain = (1 - a) * math.exp(ln0 * f) / inv_eff
    assert abs(g_c - chain * g_r) < 1e-4 * (1 + abs(g_r)), (g_c, chain * g_r)


def test_ray_query_pose_gradients(backend):
    """Rays that carry gradients (pose refinement) through ray_test -> ray_query -> rendered pixels: d loss / d rays_o,
    d rays_d equal the oracle's autograd result; rays that miss the box get exactly zero."""
    p, model, o, d, h_appear, occ, jit, jit_c, g = _setup(backend, "f32")
    N = o.shape[0]
    o_r, d_r = leaf(o), leaf(d)
    ret_o = orr.ray_query(p, o_r, d_r, h_appear, occ, AABB[0], AABB[1], RES, near=0.01, far=None, num_coarse=16,
                          num_fine=(4, 4, 8), step_size=0.02, max_steps=512, jitter=jit, jitter_c=jit_c,
                          depth_use_normalized_vw=False)
    gt = torch.rand(ret_o["num_rays"], 3, generator=g)
    wn = torch.randn(ret_o["num_rays"], 3, generator=g) * 0.1
    lo = ((ret_o["rendered"]["rgb_volume"] - gt) ** 2).sum() + (ret_o["rendered"]["normals_volume"] * wn).sum() \
        + ret_o["rendered"]["depth_volume"].sum()
    lo.backward()
    dv = lambda a: a.to(backend).contiguous()
    o_d, d_d = leaf(o, backend), leaf(d, backend)
    tested = model.ray_test(o_d, d_d, near=0.01, far=None, rays_h_appear=dv(h_appear))
    ri = tested["rays_inds"].cpu()
    cfg = dict(query_param=dict(QP), with_rgb=True, with_normal=True, depth_use_normalized_vw=False, _render=True,
               _jitter=dv(jit[ri]), _jitter_c=dv(jit_c[ri]), query_mode="march_occ_multi_upsample")
    ret = model.ray_query(ray_tested=tested, config=cfg)
    l = ((ret["rendered"]["rgb_volume"] - dv(gt)) ** 2).sum() + (ret["rendered"]["normals_volume"] * dv(wn)).sum() \
        + ret["rendered"]["depth_volume"].sum()
    assert abs(float(l) - float(lo)) < 1e-4 * (1 + abs(float(lo)))
    l.backward()
    assert rel_l2(o_d.grad.cpu(), o_r.grad) < 2e-3, rel_l2(o_d.grad.cpu(), o_r.grad)
    assert rel_l2(d_d.grad.cpu(), d_r.grad) < 2e-3, rel_l2(d_d.grad.cpu(), d_r.grad)
    miss = torch.ones(N, dtype=torch.bool)
    miss[ri] = False
    assert miss.any() and float(o_d.grad.cpu()[miss].abs().max()) == 0.0
