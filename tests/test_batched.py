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
    n_o = torch.cat([r["volume_buffer"]["pack_infos_hit"][:, 1] for r in outs])
    assert torch.equal(vb["pack_infos_hit"][:, 1].cpu(), n_o[perm])
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
    with pytest.raises(NotImplementedError):
        m.set_condition({"z_ins": torch.zeros(2, 128)})
