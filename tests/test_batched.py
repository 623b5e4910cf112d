"""Row a20: the batched (multi-instance) model -- ``set_condition`` -> ``batched_ray_test(compact_batch=True)`` ->
``batched_ray_query`` over per-instance tables + a batched occupancy grid -- against the single-object oracle run once
per instance on that instance's rays (shared decoders, different tables)."""
import pytest
import torch

from oracle import render as orr
from util import look_at_cameras, make_params, rel_l2

AABB = torch.tensor([[-1.0, -1, -1], [1.0, 1, 1]])
RES = [32, 32, 32]
QP = dict(nablas_has_grad=True, num_coarse=16, num_fine=[4, 4, 8], upsample_inv_s=64.0, upsample_inv_s_factors=[1, 4, 16],
          upsample_use_estimate_alpha=True, march_cfg=dict(step_size=0.02, max_steps=512), compress_thre=1e-3)


def _instances(B, backend, precision="f32"):
    from neuralsim_amd.fields.batched_neus import BatchedLoTDNeuSModel
    ps = [make_params(sdf_D=2, small=True, sphere=True, seed=3 + 5 * b, ln_inv_s=0.45, grid_bound=2e-2, noise_scale=1.0)
          for b in range(B)]
    for b in range(1, B):                       # shared decoders, per-instance tables with different sphere radii
        ps[b].sdf_w, ps[b].sdf_b, ps[b].rad_w, ps[b].rad_b = ps[0].sdf_w, ps[0].sdf_b, ps[0].rad_w, ps[0].rad_b
        ps[b].ln_inv_s = ps[0].ln_inv_s
        from oracle import lotd as olotd
        ps[b].grid = olotd.write_sphere_level(ps[b].grid.clone(), ps[b].spec, 0.35 + 0.1 * b).half().float()  # fp16-representable
    for p in ps:
        for t in p.tensors():
            t.requires_grad_(True)
    p0 = ps[0]
    import math
    m = BatchedLoTDNeuSModel(B, ins_ids=[f"car{b}" for b in range(B)], lod_res=p0.spec.lod_res,
                             log2_hashmap_size=int(math.log2(p0.spec.hashmap_size)), sdf_D=2, precision=precision,
                             ln_inv_s_init=float(p0.ln_inv_s.detach()), ln_inv_s_factor=p0.ln_inv_s_factor,
                             accel_cfg=dict(resolution=RES))
    with torch.no_grad():
        m.encoding.flattened_params.copy_(torch.cat([p.grid.detach().float() for p in ps]))
        m.sdf_w.copy_(torch.cat([w.detach().reshape(-1) for w in p0.sdf_w]))
        m.sdf_b.copy_(torch.cat([b.detach().reshape(-1) for b in p0.sdf_b]))
        m.rad_w.copy_(torch.cat([w.detach().reshape(-1) for w in p0.rad_w]))
        m.rad_b.copy_(torch.cat([b.detach().reshape(-1) for b in p0.rad_b]))
    m = m.to(backend)
    occs = []
    for b, p in enumerate(ps):
        val, occ = orr.build_occ_grid(p, AABB[0], AABB[1], RES, n_pts=2 ** 14, n_steps=2)
        m.accel.occ_val[b * m.accel.nvox:(b + 1) * m.accel.nvox] = val.to(backend)
        occs.append(occ)
    m.accel.pack_bits()
    return ps, m, occs


def _rays(Bq, N, seed=3):
    g = torch.Generator().manual_seed(seed)
    intr, c2w, WH = look_at_cameras(V=3, seed=seed)
    o, d = [], []
    for b in range(Bq):
        xy = torch.rand(N, 2, generator=g) * 0.5 + 0.25
        fidx = torch.randint(0, 3, (N,), generator=g)
        ob, db = orr.pinhole_rays(xy, fidx, intr, c2w, WH)
        ob[b::5] += torch.tensor([0.0, 4.0, 0.0])           # some (item, ray) pairs miss the box
        o.append(ob)
        d.append(db)
    return torch.stack(o), torch.stack(d), g


@pytest.mark.parametrize("compressed", [False, True])
def test_batched_query_matches_per_instance_oracle(backend, compressed):
    B, N = 3, 24
    ps, m, occs = _instances(B, backend)
    cond = [2, 0]                                            # two batch items: instances 2 and 0
    o, d, g = _rays(len(cond), N)
    ha = torch.randn(len(cond), N, 4, generator=g) * 0.3
    dv = lambda a: a.to(backend).contiguous()
    m.set_condition({"ins_id": [f"car{i}" for i in cond]})
    assert torch.equal(m.ins_inds_per_batch.cpu(), torch.tensor(cond))
    bt = m.batched_ray_test(dv(o), dv(d), near=0.01, far=None, compact_batch=True, rays_h_appear=dv(ha))
    mode = "march_occ_multi_upsample" + ("_compressed" if compressed else "")
    ret = m.batched_ray_query(batched_ray_tested=bt, config=dict(query_param=QP, with_rgb=True, with_normal=True,
                                                                 depth_use_normalized_vw=False, _render=True,
                                                                 query_mode=mode), return_details=True)
    vb = ret["volume_buffer"]
    # oracle, one instance at a time
    outs, n_hit = [], []
    for k, ins in enumerate(cond):
        r = orr.ray_query(ps[ins], o[k], d[k], ha[k], occs[ins], AABB[0], AABB[1], RES, near=0.01, far=None, num_coarse=16,
                          num_fine=(4, 4, 8), step_size=0.02, max_steps=512, depth_use_normalized_vw=False,
                          compress=compressed, compress_thre=1e-3)
        outs.append(r)
        n_hit.append(r["num_rays"])
    assert bt["num_rays"] == sum(n_hit)
    # the pairs come RAY-major (ray ascending, the items of a ray consecutive: what the compose renderer's
    # unique_consecutive regrouping needs); ``perm`` takes the oracle's item-major concatenation there
    ray_cat = torch.cat([r["rays_inds"] for r in outs])
    item_cat = torch.cat([torch.full([n], k) for k, n in enumerate(n_hit)])
    perm = torch.argsort(ray_cat * len(cond) + item_cat)
    assert torch.equal(bt["rays_inds"].cpu(), ray_cat[perm]) and torch.equal(bt["rays_full_bidx"].cpu(), item_cat[perm])
    assert bool((bt["rays_inds"][1:] >= bt["rays_inds"][:-1]).all())
    assert torch.equal(bt["full_bidx_map"].cpu(), torch.arange(len(cond))) and torch.equal(bt["rays_bidx"], bt["rays_full_bidx"])
    assert torch.equal(ret["details"]["march_counts"].cpu(), torch.cat([r["debug"]["march_counts"] for r in outs])[perm])
    # per TESTED pair sample counts of the oracle; the buffers list the pairs whose march found something
    # (``upsample_on_marched_only``): rows ``live`` of the tested pairs, on both sides
    n_o = torch.cat([(r["debug"]["compress_counts"] if compressed else r["debug"]["pack_infos"][:, 1]) for r in outs])
    live = torch.cat([r["debug"]["live"] for r in outs])[perm]
    assert 0 < int(live.sum()) < bt["num_rays"]
    assert torch.equal(vb["pack_infos_hit"][:, 1].cpu(), n_o[perm][live])
    assert torch.equal(vb["rays_inds_hit"].cpu(), bt["rays_inds"].cpu()[live])
    assert torch.equal(vb["rays_bidx_hit"].cpu(), bt["rays_bidx"].cpu()[live])
    assert torch.equal(vb["rays_full_bidx_hit"].cpu(), bt["rays_full_bidx"].cpu()[live])
    assert torch.equal(vb["rays_inds_hit"].cpu(), torch.cat([r["volume_buffer"]["rays_inds_hit"] for r in outs])[
        torch.argsort(torch.cat([r["volume_buffer"]["rays_inds_hit"] * len(cond) + k for k, r in enumerate(outs)]))])
    pi_o = opo_get(n_o)
    samp = torch.cat([torch.arange(int(pi_o[k, 0]), int(pi_o[k, 0] + pi_o[k, 1])) for k in perm.tolist()])   # sample permutation
    assert (vb["t"].cpu() - torch.cat([r["volume_buffer"]["t"] for r in outs])[samp]).abs().max() < 3e-4   # f32 up-sampler noise (1e-7 sdf x inv_s 1024)
    for key in ("mask_volume", "depth_volume", "rgb_volume", "normals_volume"):
        ref = torch.cat([r["rendered"][key] for r in outs])[perm]
        assert (ret["rendered"][key].detach().cpu() - ref.detach()).abs().max() < 6e-4, key
    # the two items really differ (different instances, different radii)
    assert (outs[0]["rendered"]["depth_volume"].mean() - outs[1]["rendered"]["depth_volume"].mean()).abs() > 1e-2
    # loss + backward: per-instance table gradients land in that instance's slice, decoder gradients add up
    wgt = torch.rand(bt["num_rays"], 3, generator=g)            # indexed in the oracle's item-major order
    loss = (ret["rendered"]["rgb_volume"] * dv(wgt[perm])).sum() + 0.1 * ((vb["nablas"].norm(dim=-1) - 1.0) ** 2).sum()
    loss.backward()
    off = 0
    for r in outs:
        n = r["num_rays"]
        lo = (r["rendered"]["rgb_volume"] * wgt[off:off + n]).sum() + \
            0.1 * ((r["volume_buffer"]["nablas"].norm(dim=-1) - 1.0) ** 2).sum()
        lo.backward()
        off += n
    n_par = m.n_params_per_instance
    gg = m.encoding.flattened_params.grad.cpu().view(B, n_par)
    for ins in range(B):
        if ins in cond:
            assert rel_l2(gg[ins], ps[ins].grid.grad) < 5e-3, ins
        else:
            assert float(gg[ins].abs().max()) == 0.0                      # instance 1 was not in the condition
    p0 = ps[0]
    assert rel_l2(m.sdf_w.grad.cpu(), torch.cat([w.grad.reshape(-1) for w in p0.sdf_w])) < 5e-3
    assert rel_l2(m.rad_w.grad.cpu(), torch.cat([w.grad.reshape(-1) for w in p0.rad_w])) < 5e-3
    m.clean_condition()
    assert m.ins_inds_per_batch is None


def opo_get(n):
    from oracle import pack_ops as opo
    return opo.get_pack_infos_from_n(n)


def test_batched_point_queries_and_occupancy(backend):
    B = 3
    ps, m, occs = _instances(B, backend)
    g = torch.Generator().manual_seed(9)
    x = torch.rand(200, 3, generator=g) * 1.6 - 0.8
    from oracle import field as ofield
    for ins in range(B):
        ref = ofield.forward_sdf(x, ps[ins]).detach()
        assert (m.query_sdf(x.to(backend), ins_ind=ins).cpu() - ref).abs().max() < 2e-5
    # per-point batch items under a condition
    m.set_condition({"ins_ind": torch.tensor([1, 2])})
    bidx = torch.randint(0, 2, (200,), generator=g)
    got = m.query_sdf(x.to(backend), bidx=bidx.to(backend)).cpu()
    ref = torch.where(bidx == 0, ofield.forward_sdf(x, ps[1]).detach(), ofield.forward_sdf(x, ps[2]).detach())
    assert (got - ref).abs().max() < 2e-5
    out = m.forward_sdf_nablas(x.to(backend), bidx=bidx.to(backend))
    s1, n1 = ofield.forward_sdf_nablas(x, ps[1])
    s2, n2 = ofield.forward_sdf_nablas(x, ps[2])
    assert (out["nablas"].detach().cpu() - torch.where(bidx[:, None] == 0, n1, n2).detach()).abs().max() < 2e-4
    # occupancy refresh from the net: every instance gets ITS sphere (radii 0.5, 0.45, 0.55)
    m.accel.occ_val.zero_()
    m.accel.num_pts, m.accel.num_steps = 2 ** 14, 2
    m.init_accel(generator=torch.Generator(device=backend).manual_seed(1))
    fr = m.accel.occ_grid.float().mean(dim=(1, 2, 3)).cpu()
    assert fr[1] < fr[0] < fr[2] and float(fr.min()) > 0.002, fr
    with pytest.raises(RuntimeError):          # free per-instance tables have no latent: z_ins needs lotd_grower_cfg
        m.set_condition({"z_ins": torch.zeros(2, 128)})


@pytest.mark.parametrize("by,embed", [("ins_id", None), ("z_ins", None), ("z_ins", 6)])
def test_latent_conditioned_tables_match_the_oracle(backend, by, embed):
    """Row a20, the latent -> table path (no_fg_occ.221218.yaml:307-352): a model built with ``lotd_grower_cfg`` grows the
    batch's dense LoTD tables from latent codes in ``set_condition({'ins_id'})`` (the auto-decoder's codes) or
    ``set_condition({'z_ins'})`` (codes given) -- rendered images, and the gradients that reach the codes, the grower's
    weights and the shared decoders, against the oracle's restatement (oracle/growers.py -> oracle.lotd -> oracle.render).
    ``embed`` 6: with ``surface_cfg.extra_pos_embed_cfg{sinusoidal_legacy, 6}`` of that block (:319-321) -- the shared decoder
    reads [grown features | embedded position] on csrc/wide_field.hip, sampling pass and occupancy included."""
    from oracle import field as ofield, growers as ogrow, lotd as olotd
    from neuralsim_amd.fields.batched_neus import BatchedLoTDNeuSModel
    B, N = 3, 20
    gcfg = dict(lod_res=[3, 5, 8], lod_n_feats=4, D=2, W=32, fmm_rank=4, n_frequencies=3, out_scale=0.5)
    m = BatchedLoTDNeuSModel(B, ins_ids=[f"car{b}" for b in range(B)], lotd_grower_cfg=gcfg, latents_cfg=dict(z=dict(dim=12)),
                             sdf_D=2, precision="f32", ln_inv_s_init=0.3, log2_hashmap_size=12,
                             accel_cfg=dict(resolution=[8, 8, 8]), pos_embed_frequencies=embed).to(backend)
    assert m.encoding.cfg.lod_res == [3, 3, 5, 5, 8, 8] and m.encoding.flattened_params.numel() == 0
    assert m.sdf_w.numel() == 64 * (12 + (0 if embed is None else 39)) + 4096 + 64
    m.accel.set_all_occupied()
    with torch.no_grad():            # a visible shape: bias the SDF head so that part of the box is inside
        m.sdf_b[-1] = -0.05
        m.latents.mul_(5.0)
        m.sdf_b.add_(0)
    cond = [2, 0]
    o, d, g = _rays(len(cond), N)
    ha = torch.randn(len(cond), N, 4, generator=g) * 0.3
    dv = lambda a: a.to(backend).contiguous()        # noqa: E731
    if by == "ins_id":
        m.set_condition({"ins_id": [f"car{i}" for i in cond]})
        z_o = m.latents.detach().cpu()[cond].clone().requires_grad_(True)
    else:
        z_in = (torch.randn(len(cond), 12, generator=g) * 0.5).to(backend).requires_grad_(True)
        m.set_condition({"z_ins": z_in, "ins_ind": cond})
        z_o = z_in.detach().cpu().clone().requires_grad_(True)
    # ---- oracle: grow the tables, then one single-object query per batch item
    layers = [{k: getattr(lay, k).detach().cpu().clone().requires_grad_(True) for k in ("weight", "bias", "u_w", "u_b", "v_w", "v_b")}
              for lay in m.grower.layers]
    tables = ogrow.grow_tables(z_o, layers, gcfg["lod_res"], 4, 3, 4, 0.5)
    assert tables.shape == (len(cond), m.n_params_per_instance)
    assert torch.allclose(m._cond_table.detach().cpu().view(len(cond), -1), tables.detach(), atol=2e-6)
    p0 = ofield.params_from_flat(m.encoding.cfg.lod_res, 12, tables[0].detach(), m.sdf_w, m.sdf_b, m.rad_w, m.rad_b, m.ln_inv_s,
                                 sdf_D=2, ln_inv_s_factor=m.ln_inv_s_factor, pos_embed_n=embed)
    shared = [*p0.sdf_w, *p0.sdf_b, *p0.rad_w, *p0.rad_b, p0.ln_inv_s]
    for t_ in shared:
        t_.requires_grad_(True)
    occ = torch.ones(8 ** 3, dtype=torch.bool)
    kw = dict(near=0.01, far=None, num_coarse=16, num_fine=(4, 4, 8), step_size=0.02, max_steps=512, depth_use_normalized_vw=False)
    wgt = torch.rand(len(cond), N, 3, generator=g)
    loss_o = 0.0
    outs = []
    for k in range(len(cond)):
        pk = ofield.FieldParams(spec=p0.spec, grid=tables[k].detach().half().float() + (tables[k] - tables[k].detach()),   # fp16 values, f32 grads
                                sdf_w=p0.sdf_w, sdf_b=p0.sdf_b, rad_w=p0.rad_w, rad_b=p0.rad_b, ln_inv_s=p0.ln_inv_s,
                                ln_inv_s_factor=p0.ln_inv_s_factor, pos_embed_n=embed)
        r = orr.ray_query(pk, o[k], d[k], ha[k], occ, AABB[0], AABB[1], [8, 8, 8], **kw)
        outs.append(r)
        if r["num_rays"] > 0:
            loss_o = loss_o + (r["rendered"]["rgb_volume"] * wgt[k][r["rays_inds"]]).sum() + \
                0.1 * ((r["volume_buffer"]["nablas"].norm(dim=-1) - 1.0) ** 2).sum()
    loss_o.backward()
    # ---- product
    m.ray_query_cfg = dict(query_mode="march_occ_multi_upsample", query_param=QP)
    bt = m.batched_ray_test(dv(o), dv(d), near=0.01, far=None, compact_batch=False, rays_h_appear=dv(ha))
    ret = m.batched_ray_query(batched_ray_tested=bt, config=dict(query_param=QP, with_rgb=True, with_normal=True,
                                                                 depth_use_normalized_vw=False, _render=True,
                                                                 query_mode="march_occ_multi_upsample"))
    assert bt["num_rays"] == sum(r["num_rays"] for r in outs) > 0
    w_pairs = dv(wgt)[bt["rays_full_bidx"], bt["rays_inds"]]
    loss = (ret["rendered"]["rgb_volume"] * w_pairs).sum() + \
        0.1 * ((ret["volume_buffer"]["nablas"].norm(dim=-1) - 1.0) ** 2).sum()
    assert abs(float(loss) - float(loss_o)) < 2e-4 * (1 + abs(float(loss_o)))
    loss.backward()
    m.clean_condition()
    if by == "ins_id":
        gz = m.latents.grad.cpu()
        assert float(gz[1].abs().max()) == 0.0                      # instance 1 is not in the condition
        assert rel_l2(gz[cond], z_o.grad) < 5e-3
    else:
        assert rel_l2(z_in.grad.cpu(), z_o.grad) < 5e-3
    for lay, ref in zip(m.grower.layers, layers):
        for k in ("weight", "bias", "u_w", "v_w"):
            assert rel_l2(getattr(lay, k).grad.cpu(), ref[k].grad) < 5e-3, k
    assert rel_l2(m.sdf_w.grad.cpu(), torch.cat([w.grad.reshape(-1) for w in p0.sdf_w])) < 5e-3
    assert rel_l2(m.rad_w.grad.cpu(), torch.cat([w.grad.reshape(-1) for w in p0.rad_w])) < 5e-3


def test_grower_at_the_config_sizes():
    """no_fg_occ.221218.yaml:320-337: z 128 -> dense levels [5, 8, 13, 21] x 4 features, D 5, W 128, rank 10, 6
    frequencies: 12 095 vertices, 48 380 table entries per instance laid out as 8 kernel levels of 2 features."""
    from neuralsim_amd.grid_encodings.lotd_growers import DenseLoTDGrowerFMM
    g = DenseLoTDGrowerFMM(z_dim=128, lod_res=[5, 8, 13, 21], lod_n_feats=4, D=5, W=128, fmm_rank=10, n_frequencies=6)
    assert sum(g.n_vertices) == 12095 and g.n_params == 48380 and g.kernel_lod_res == [5, 5, 8, 8, 13, 13, 21, 21]
    assert g.vertex_embedding.shape == (12095, 3 * 13 + 4) and len(g.layers) == 6
    z = torch.zeros(2, 128)
    z[1, 3] = 1.0
    t = g(z)
    assert t.shape == (2, 48380) and bool(torch.isfinite(t).all()) and float((t[0] - t[1]).abs().max()) > 0
    # z = 0: the modulation is exactly 1 on every weight of every layer (biases rank^-1/2: sum_r u_r v_r = 1) -- the grown
    # table of the zero code is the plain MLP's output, not sqrt(rank)^6 times it
    for lay in g.layers:
        assert torch.allclose(lay.effective_weight(torch.zeros(1, 128))[0], lay.weight, rtol=1e-5, atol=1e-7)
    assert float(t[0].abs().max()) < 1.0


def _fmm_params(layers):
    return [{k: getattr(lay, k).detach().cpu().clone() for k in ("weight", "bias", "u_w", "u_b", "v_w", "v_b")} for lay in layers]


def test_vm_split_grower_is_the_factorised_field():
    """``VMSplitLoTDGrowerFMM`` (no_fg_occ.221218.yaml:338-352) grows vector-matrix levels as DENSE vertex tables: (1) equal
    to the oracle's loop restatement; (2) the expansion is exact -- the trilinear interpolant of the expanded table (what the
    LoTD kernels compute on a dense level) equals sum_c bilinear(plane_c) x linear(line_c) evaluated directly, at random
    points; (3) the reference's config block builds it (``build_grower``) next to the dense grower (``MixedLoTDGrower``)."""
    from oracle import growers as ogrow, lotd as olotd
    from neuralsim_amd.grid_encodings.lotd_growers import MixedLoTDGrower, VMSplitLoTDGrowerFMM, build_grower
    res, F = [4, 7], 4
    g = VMSplitLoTDGrowerFMM(z_dim=6, lod_res=res, lod_n_feats=F, D=2, D_head=2, W=16, fmm_rank=3, n_frequencies=2, out_scale=0.5,
                             seed=3)
    z = torch.randn(2, 6, generator=torch.Generator().manual_seed(1)) * 0.7
    tab = g(z).detach()
    ref, factors = ogrow.grow_vm_tables(z, _fmm_params(g.trunk), _fmm_params(g.plane_head), _fmm_params(g.line_head), res, F, 2, 3,
                                        0.5, return_factors=True)
    assert tab.shape == ref.shape == (2, g.n_params) and g.n_params == sum(r ** 3 * F for r in res)
    assert torch.allclose(tab, ref, atol=2e-6), float((tab - ref).abs().max())
    assert g.kernel_lod_res == [4, 4, 7, 7]
    # (2) trilinear interpolation of the expanded level == the factorised evaluation
    spec = olotd.make_lotd_spec(lod_res=g.kernel_lod_res, log2_hashmap_size=10)
    assert all(t == "Dense" for t in spec.lod_types)
    x = torch.rand(200, 3, generator=torch.Generator().manual_seed(2)) * 2 - 1
    feat = olotd.lotd_forward(x, ref[1], spec)                              # [N, 2 per kernel level]
    for l, R in enumerate(res):
        direct = ogrow.vm_feature_at(x, factors[1][3 * l:3 * l + 3], R)     # [N, F]
        got = feat[:, 4 * l:4 * l + 4]
        assert torch.allclose(got, direct, atol=3e-6), (l, float((got - direct).abs().max()))
    # (3) the reference's block
    blk = dict(target="nr3d_lib.models.grid_encodings.lotd.lotd_batched_growers.MixedLoTDGrower", param=dict(grower_configs=[
        dict(target="nr3d_lib.models.grid_encodings.lotd.lotd_batched_growers.DenseLoTDGrowerFMM",
             param=dict(z_dim=128, lod_res=[3, 5], lod_n_feats=4, pseudo_net_type="same",
                        pseudo_net_param=dict(activation="relu", fmm_rank=4, equal_lr=False, D=2, W=16,
                                              embed_cfg=dict(type="sinusoidal_legacy", n_frequencies=2)))),
        dict(target="nr3d_lib.models.grid_encodings.lotd.lotd_batched_growers.VMSplitLoTDGrowerFMM",
             param=dict(z_dim=128, lod_res=[6, 9], lod_n_feats=4, pseudo_net_type="shared",
                        pseudo_net_param=dict(activation="relu", fmm_rank=4, equal_lr=False, D=2, D_head=2, W=16,
                                              embed_cfg=dict(type="sinusoidal_legacy", n_frequencies=3))))]))
    mg = build_grower(blk, z_dim=10, seed=1)
    assert isinstance(mg, MixedLoTDGrower) and mg.z_dim == 10 and mg.kernel_lod_res == [3, 3, 5, 5, 6, 6, 9, 9]
    t = mg(torch.zeros(1, 10))
    assert t.shape == (1, mg.n_params) and mg.n_params == (27 + 125 + 216 + 729) * 4 and bool(torch.isfinite(t).all())
    with pytest.raises(NotImplementedError):
        build_grower(dict(target="x.VMSplitLoTDGrowerFMM", param=dict(lod_res=[4], pseudo_net_type="same")), z_dim=4)


def test_mixed_grower_model_queries_match_the_oracle(backend):
    """A model grown by dense + vector-matrix levels through the kernels: SDF / normals of ``forward_sdf_nablas`` and the code
    gradients against the oracle field on the oracle-grown table."""
    from oracle import field as ofield, growers as ogrow
    from neuralsim_amd.fields.batched_neus import BatchedLoTDNeuSModel
    pn = lambda D, nf, **k: dict(activation="relu", fmm_rank=3, equal_lr=False, D=D, W=16,       # noqa: E731
                                 embed_cfg=dict(type="sinusoidal_legacy", n_frequencies=nf), **k)
    blk = dict(target="x.MixedLoTDGrower", param=dict(grower_configs=[
        dict(target="x.DenseLoTDGrowerFMM", param=dict(lod_res=[3, 5], lod_n_feats=4, pseudo_net_type="same", pseudo_net_param=pn(2, 2))),
        dict(target="x.VMSplitLoTDGrowerFMM", param=dict(lod_res=[6, 9], lod_n_feats=4, pseudo_net_type="shared",
                                                         pseudo_net_param=pn(2, 2, D_head=2)))]))
    B = 2
    m = BatchedLoTDNeuSModel(B, lotd_grower_cfg=blk, latents_cfg=dict(z=dict(dim=5)), sdf_D=1, precision="f32", log2_hashmap_size=10,
                             accel_cfg=dict(resolution=[8, 8, 8])).to(backend)
    assert m.encoding.cfg.lod_res == [3, 3, 5, 5, 6, 6, 9, 9] and all(t == "Dense" for t in m.encoding.cfg.lod_types)
    z = (torch.randn(B, 5, generator=torch.Generator().manual_seed(4)) * 0.8).to(backend).requires_grad_(True)
    m.set_condition({"z_ins": z})
    dense, vm = m.grower.growers
    z_o = z.detach().cpu().clone().requires_grad_(True)
    t_d = ogrow.grow_tables(z_o, _fmm_params(dense.layers), [3, 5], 4, 2, 3, dense.out_scale)
    t_v = ogrow.grow_vm_tables(z_o, _fmm_params(vm.trunk), _fmm_params(vm.plane_head), _fmm_params(vm.line_head), [6, 9], 4, 2, 3,
                               vm.out_scale)
    tables = torch.cat([t_d, t_v], dim=1)
    assert torch.allclose(m._cond_table.detach().cpu().view(B, -1), tables.detach(), atol=2e-6)
    x = torch.rand(60, 3, generator=torch.Generator().manual_seed(5)) * 1.6 - 0.8
    bidx = torch.randint(0, B, (60,), generator=torch.Generator().manual_seed(6))
    out = m.forward_sdf_nablas(x.to(backend), bidx=bidx.to(backend))
    w = torch.randn(60, generator=torch.Generator().manual_seed(7))
    loss_o = 0.0
    for b in range(B):
        p = ofield.params_from_flat(m.encoding.cfg.lod_res, 10, tables[b].detach(), m.sdf_w, m.sdf_b, m.rad_w, m.rad_b, m.ln_inv_s,
                                    sdf_D=1, ln_inv_s_factor=m.ln_inv_s_factor)
        p.grid = tables[b].detach().half().float() + (tables[b] - tables[b].detach())
        sel = bidx == b
        sdf_o, nab_o = ofield.forward_sdf_nablas(x[sel], p)
        assert (out["sdf"].detach().cpu()[sel] - sdf_o.detach()).abs().max() < 1e-4 * (1 + float(sdf_o.abs().max()))
        assert (out["nablas"].detach().cpu()[sel] - nab_o.detach()).abs().max() < 1e-3 * (1 + float(nab_o.abs().max()))
        loss_o = loss_o + (sdf_o * w[sel]).sum() + 0.1 * (nab_o ** 2).sum()
    loss_o.backward()
    ((out["sdf"] * w.to(backend)).sum() + 0.1 * (out["nablas"] ** 2).sum()).backward()
    assert rel_l2(z.grad.cpu(), z_o.grad) < 5e-3, rel_l2(z.grad.cpu(), z_o.grad)
    m.clean_condition()


def test_batched_permuto_model_matches_per_instance_oracle(backend):
    """Rows a20 / f4: ``BatchedPermutoNeuSModel`` (the network of AD_GenerativePermutoConcatNeuSObj, all_occ.240201.yaml:425-506)
    -- ONE permutohedral table, a latent per batch item concatenated to the position, per-instance occupancy grids --
    through ``set_condition(z=, ins_inds_per_batch=)`` -> ``batched_ray_test`` -> ``batched_ray_query``: rendered colours, the
    loss, and the gradients of the table, the decoders and the CODES (learned: d L / d z, nsim_permuto_dz) against the
    single-object oracle run once per item with that item's code.  The condition is changed between forward and backward: the
    backward uses the codes the query was made under."""
    from oracle import field as ofield, permuto as operm
    from neuralsim_amd.fields.batched_permuto_neus import BatchedPermutoNeuSModel
    B, N, ZD, L = 3, 24, 4, 6
    pcfg = dict(type="multi_res", n_levels=L, n_feats=2, log2_hashmap_size=11, coarsest_res=2.0, finest_res=16.0,
                apply_random_shifts_per_level=True, seed=5)
    m = BatchedPermutoNeuSModel(B, z_dim=ZD, ins_ids=[f"car{b}" for b in range(B)], permuto_auto_compute_cfg=pcfg, sdf_D=1,
                                precision="f32", param_bound=0.4, seed=9, ln_inv_s_init=0.3, accel_cfg=dict(resolution=[8, 8, 8]))
    with torch.no_grad():
        m.sdf_b[-1] = -0.05
        m.sdf_b.add_(0)
    m = m.to(backend)
    m.accel.set_all_occupied()
    assert m.accel.num_batches == B and m.encoding.cfg.permuto.in_dim == 3 + ZD
    spec = operm.make_permuto_spec(in_dim=3 + ZD, **{k: v for k, v in pcfg.items() if k != "type"})
    p = ofield.make_field_params(lod_res=[2] * L, log2_hashmap_size=4, sdf_D=1, seed=1, sphere_init=False)
    p.spec = spec
    p.grid = m.encoding.flattened_params.detach().cpu().clone().half().float()
    F1 = 2 * L
    sw, sb = m.sdf_w.detach().cpu().clone(), m.sdf_b.detach().cpu().clone()
    p.sdf_w, p.sdf_b = [sw[:64 * F1].view(64, F1).clone(), sw[64 * F1:].view(1, 64).clone()], [sb[:64].clone(), sb[64:].clone()]
    rw, rb = m.rad_w.detach().cpu().clone(), m.rad_b.detach().cpu().clone()
    p.rad_w = [rw[:64 * 26].view(64, 26).clone(), rw[64 * 26:64 * 26 + 4096].view(64, 64).clone(), rw[64 * 26 + 4096:].view(3, 64).clone()]
    p.rad_b = [rb[:64].clone(), rb[64:128].clone(), rb[128:].clone()]
    p.ln_inv_s = m.ln_inv_s.detach().cpu().clone()
    p.aabb = m.accel.aabb.detach().cpu().clone()
    for t_ in p.tensors():
        t_.requires_grad_(True)
    cond = [2, 0]
    o, d, g = _rays(len(cond), N)
    ha = torch.randn(len(cond), N, 4, generator=g) * 0.3
    z0 = torch.randn(len(cond), ZD, generator=g) * 0.4
    z_o = z0.clone().requires_grad_(True)
    z_d = z0.clone().to(backend).requires_grad_(True)
    dv = lambda a: a.to(backend).contiguous()        # noqa: E731
    qp = dict(QP, num_fine=16, upsample_inv_s_factors=[1, 4])            # one number, two stages (all_occ.240201.yaml:481-484)
    occ = torch.ones(8 ** 3, dtype=torch.bool)
    kw = dict(near=0.01, far=None, num_coarse=16, num_fine=(8, 8), step_size=0.02, max_steps=512, depth_use_normalized_vw=False,
              upsample_inv_s_factors=(1, 4))
    wgt = torch.rand(len(cond), N, 3, generator=g)
    loss_o, outs = 0.0, []
    for k in range(len(cond)):
        p.z = z_o[k:k + 1]
        r = orr.ray_query(p, o[k], d[k], ha[k], occ, AABB[0], AABB[1], [8, 8, 8], **kw)
        outs.append(r)
        if r["num_rays"] > 0:
            loss_o = loss_o + (r["rendered"]["rgb_volume"] * wgt[k][r["rays_inds"]]).sum() + \
                0.1 * ((r["volume_buffer"]["nablas"].norm(dim=-1) - 1.0) ** 2).sum()
    loss_o.backward()
    m.set_condition(z=z_d, ins_inds_per_batch=torch.tensor(cond))
    bt = m.batched_ray_test(dv(o), dv(d), near=0.01, far=None, compact_batch=False, rays_h_appear=dv(ha))
    ret = m.batched_ray_query(batched_ray_tested=bt, config=dict(query_param=qp, with_rgb=True, with_normal=True,
                                                                 depth_use_normalized_vw=False, _render=True,
                                                                 query_mode="march_occ_multi_upsample"))
    assert bt["num_rays"] == sum(r["num_rays"] for r in outs) > 0
    w_pairs = dv(wgt)[bt["rays_full_bidx"], bt["rays_inds"]]
    loss = (ret["rendered"]["rgb_volume"] * w_pairs).sum() + 0.1 * ((ret["volume_buffer"]["nablas"].norm(dim=-1) - 1.0) ** 2).sum()
    assert abs(float(loss.detach()) - float(loss_o.detach())) < 3e-4 * (1 + abs(float(loss_o.detach())))
    # point queries of the condition's items, then ANOTHER condition -- and only then the backward of the batched query
    x = torch.rand(30, 3, generator=g) * 1.2 - 0.6
    q1 = m.query_sdf(dv(x), ins_ind=0).cpu()                 # instance 0 = batch item 1
    p.z = z_o.detach()[1:2]
    assert (q1 - ofield.forward_sdf(x, p).detach()).abs().max() < 1e-4
    m.set_condition(z=torch.zeros(1, ZD, device=backend), ins_inds_per_batch=torch.tensor([1]))
    loss.backward()
    m.clean_condition()
    assert rel_l2(z_d.grad.cpu(), z_o.grad) < 5e-3, rel_l2(z_d.grad.cpu(), z_o.grad)
    assert rel_l2(m.encoding.flattened_params.grad.cpu(), p.grid.grad) < 5e-3
    assert rel_l2(m.sdf_w.grad.cpu(), torch.cat([w.grad.reshape(-1) for w in p.sdf_w])) < 5e-3
    assert rel_l2(m.rad_w.grad.cpu(), torch.cat([w.grad.reshape(-1) for w in p.rad_w])) < 5e-3
