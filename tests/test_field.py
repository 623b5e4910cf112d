"""LoTD encoding and the fused MFMA field kernels (forward, backward incl. the second-order normal terms)
vs the oracle (oracle/lotd.py, oracle/field.py)."""
import os
import pytest
import torch

from oracle import field as ofield, lotd as olotd
from neuralsim_amd import _lib
from neuralsim_amd.fields.neus import _FieldFn
from neuralsim_amd.grid_encodings.lotd import LoTDConfig, LoTDEncoding, gen_ngp_res
from util import SMALL_RES as SMALL_RES_T, leaf, make_params, model_from_params, oracle_flat_grads, rel_l2


def test_gen_ngp_matches_reference_comment():
    # lotd_neus.dtu.230814.yaml:97
    assert gen_ngp_res(16, 2048, 16) == [16, 23, 31, 43, 59, 81, 112, 154, 213, 295, 407, 562, 777, 1073, 1483, 2048]
    cfg = LoTDConfig(gen_ngp_res(16, 2048, 16), 2, 19)
    assert cfg.n_params == 12196216 and cfg.lod_types.count("Dense") == 5


def test_mfma_selftest(backend):
    g = torch.Generator().manual_seed(0)
    A = torch.randn(32, 16, generator=g)
    B = torch.randn(16, 32, generator=g)          # asymmetric on purpose (CDNA4 guide sec. 3)
    for use_f32 in (0, 1):
        D = torch.zeros(32, 32, device=backend)
        _lib.call("nsim_selftest_mfma", _lib.ptr(A.to(backend)), _lib.ptr(B.to(backend)), _lib.ptr(D), use_f32)
        ref = (A.half().float() @ B.half().float()) if not use_f32 else A @ B
        assert torch.allclose(D.cpu(), ref, atol=1e-4 if use_f32 else 2e-3), use_f32


def test_lotd_fwd_dydx_bwd(backend):
    p = make_params(small=True, sphere=False, grid_bound=0.5)
    spec = p.spec
    g = torch.Generator().manual_seed(1)
    S = 257
    x = torch.rand(S, 3, generator=g) * 2 - 1
    x[0] = torch.tensor([-1.0, -1.0, -1.0]); x[1] = torch.tensor([1.0, 1.0, 1.0]); x[2] = torch.tensor([0.0, 1.0, -1.0])
    enc = LoTDEncoding(LoTDConfig(spec.lod_res, 2, 12)).to(backend)
    with torch.no_grad():
        enc.flattened_params.copy_(p.grid.to(backend))
    xo = leaf(x)
    grid_o = leaf(p.grid)
    h_ref = olotd.lotd_forward(xo, grid_o, spec)
    h, dydx = enc.forward_dydx(x.to(backend))
    assert torch.allclose(h.cpu(), h_ref, atol=1e-6)
    # d h / d x via autograd on the oracle, one output feature at a time for a few features
    for f in (0, 7, 16, 31):
        gx = torch.autograd.grad(h_ref[:, f].sum(), xo, retain_graph=True)[0]
        assert torch.allclose(dydx.cpu()[3:, f], gx[3:], atol=2e-4, rtol=1e-4), f
    # backward to the grid: first-order and the dy/dx (second-order) path
    w_h = torch.randn(S, 32, generator=g)
    w_j = torch.randn(S, 32, 3, generator=g)
    xo2 = leaf(x)
    h2 = olotd.lotd_forward(xo2, grid_o, spec)
    J_rows = []
    loss_ref = (h2 * w_h).sum()
    # sum_f,d w_j[s,f,d] * d h[s,f]/d x[s,d]  == grad of (h * 1) contracted; build with create_graph
    for f in range(32):
        gx = torch.autograd.grad(h2[:, f].sum(), xo2, create_graph=True)[0]
        loss_ref = loss_ref + (gx * w_j[:, f]).sum()
    loss_ref.backward()
    h3, dydx3 = enc.forward_dydx(x.to(backend))
    ((h3 * w_h.to(backend)).sum() + (dydx3 * w_j.to(backend)).sum()).backward()
    assert rel_l2(enc.flattened_params.grad.cpu(), grid_o.grad) < 1e-5


@pytest.mark.parametrize("sdf_D", [1, 2])
@pytest.mark.parametrize("precision", ["f32", "fp16"])
def test_field_fwd_bwd(backend, sdf_D, precision):
    p = make_params(sdf_D=sdf_D, small=True, sphere=False, grid_bound=0.3, seed=5, noise_scale=1.0)
    for t in p.tensors():
        t.requires_grad_(True)
    model = model_from_params(p, backend, precision=precision)
    g = torch.Generator().manual_seed(2)
    R, S = 7, 77                                  # 2.4 tiles: exercises the ragged last tile
    rays_o = torch.randn(R, 3, generator=g) * 0.1
    rays_d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    ridx = torch.randint(0, R, (S,), generator=g).sort().values
    t = torch.rand(S, generator=g) * 0.8
    h_appear = torch.randn(R, 4, generator=g) * 0.5
    x = rays_o[ridx] + t[:, None] * rays_d[ridx]
    ha_o = leaf(h_appear)
    sdf_r, nab_r, rgb_r = ofield.forward_field(x, rays_d[ridx], ha_o[ridx], p)
    ha_d = leaf(h_appear, backend)
    dv = lambda a: a.to(backend).contiguous()
    sdf, nab, rgb = _FieldFn.apply(model, model.encoding.flattened_params, model.sdf_w, model.sdf_b, model.rad_w,
                                   model.rad_b, ha_d, None, dv(rays_o), dv(rays_d), dv(t), dv(ridx), True)
    tol = dict(f32=(2e-5, 2e-4, 2e-5), fp16=(4e-3, 5e-2, 4e-3))[precision]
    assert (sdf.cpu() - sdf_r).abs().max() < tol[0] * (1 + sdf_r.abs().max())
    assert (nab.cpu() - nab_r).abs().max() < tol[1] * (1 + nab_r.abs().max())
    assert (rgb.cpu() - rgb_r).abs().max() < tol[2]
    # no-grad SDF kernel agrees with the with-grad one
    assert torch.allclose(model._query_sdf_rays(dv(rays_o), dv(rays_d), dv(t), dv(ridx)).cpu(), sdf.cpu().detach(), atol=1e-6)
    assert torch.allclose(model.query_sdf(dv(x)).cpu(), sdf.cpu().detach(), atol=2e-5 if precision == "f32" else 4e-3)
    ws, wn, wr = torch.randn(S, generator=g), torch.randn(S, 3, generator=g) * 0.1, torch.randn(S, 3, generator=g)
    (sdf_r * ws).sum().add((nab_r * wn).sum()).add((rgb_r * wr).sum()).backward()
    (sdf * dv(ws)).sum().add((nab * dv(wn)).sum()).add((rgb * dv(wr)).sum()).backward()
    ref = oracle_flat_grads(p)
    got = dict(grid=model.encoding.flattened_params.grad, sdf_w=model.sdf_w.grad, sdf_b=model.sdf_b.grad,
               rad_w=model.rad_w.grad, rad_b=model.rad_b.grad)
    gtol = dict(f32=2e-4, fp16=3e-2)[precision]
    for k, v in got.items():
        e = rel_l2(v.cpu(), ref[k])
        assert e < gtol, (k, e)
    assert rel_l2(ha_d.grad.cpu(), ha_o.grad) < gtol


@pytest.mark.parametrize("levels", [16, 19])
def test_weight_gradient_replicas_match_the_direct_flush(backend, levels, monkeypatch):
    """The joint backward launches spread their weight-gradient flush over 16 replicas of a registered scratch and fold
    them with a second launch (include/nsim.h: nsim_set_grad_scratch): same gradients as the direct flush, twice in a row
    (the scratch must come back zeroed), accumulating into a non-zero ``.grad``."""
    lod_res = list(SMALL_RES_T) if levels == 16 else [4 + int(round(2.9 * i + 0.11 * i * i)) for i in range(levels)]
    p = ofield.make_field_params(lod_res=lod_res, log2_hashmap_size=12, sdf_D=2, seed=5, sphere_init=False, grid_bound=0.3,
                                 noise_scale=1.0)
    p.grid = p.grid.float()
    model = model_from_params(p, backend, precision="f32")
    g = torch.Generator().manual_seed(12)
    R, S = 9, 700
    dv = lambda a: a.to(backend).contiguous()
    rays_o, rays_d = dv(torch.randn(R, 3, generator=g) * 0.1), dv(torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1))
    ridx, t = dv(torch.randint(0, R, (S,), generator=g).sort().values), dv(torch.rand(S, generator=g) * 0.8)
    ha = leaf(torch.randn(R, 4, generator=g) * 0.5, backend)
    ws, wn, wr = dv(torch.randn(S, generator=g)), dv(torch.randn(S, 3, generator=g) * 0.1), dv(torch.randn(S, 3, generator=g))
    params = (model.sdf_w, model.sdf_b, model.rad_w, model.rad_b)

    def grads(min_wg):
        monkeypatch.setenv("NSIM_GRAD_REPLICAS_MIN_WG", str(min_wg))
        for q in params:
            q.grad = torch.full_like(q, 0.25)                # the fold ADDS into what is there
        sdf, nab, rgb = _FieldFn.apply(model, model.encoding.flattened_params, model.sdf_w, model.sdf_b, model.rad_w,
                                       model.rad_b, ha, None, rays_o, rays_d, t, ridx, True)
        ((sdf * ws).sum() + (nab * wn).sum() + (rgb * wr).sum()).backward()
        return [q.grad.detach().cpu().clone() for q in params]
    direct = grads(10 ** 9)
    for _ in range(2):
        rep = grads(1)
        for a_, b_ in zip(rep, direct):
            assert rel_l2(a_ - 0.25, b_ - 0.25) < 2e-5
    scratch = _lib.ensure_grad_scratch(model.device)
    assert float(scratch.abs().max()) == 0.0


@pytest.mark.parametrize("which", ["sdf_only", "nablas_only", "no_rgb"])
def test_field_partial_upstream(backend, which):
    """Each upstream gradient alone (isolates the second-order path) and the with_rgb=False variant
    (lidar batches, code_single/tools/train.py:896-902)."""
    p = make_params(sdf_D=2, small=True, sphere=False, grid_bound=0.3, seed=9, noise_scale=1.0)
    for t in p.tensors():
        t.requires_grad_(True)
    model = model_from_params(p, backend, precision="f32")
    g = torch.Generator().manual_seed(4)
    S = 40
    x = torch.rand(S, 3, generator=g) * 1.6 - 0.8
    sdf_r, nab_r = ofield.forward_sdf_nablas(x, p)
    out = model.forward_sdf_nablas(x.to(backend))
    assert torch.allclose(out["sdf"].cpu(), sdf_r, atol=2e-5) and torch.allclose(out["nablas"].cpu(), nab_r, atol=5e-4)
    ws, wn = torch.randn(S, generator=g), torch.randn(S, 3, generator=g)
    if which == "sdf_only":
        (sdf_r * ws).sum().backward(); (out["sdf"] * ws.to(backend)).sum().backward()
    elif which == "nablas_only":
        (nab_r * wn).sum().backward(); (out["nablas"] * wn.to(backend)).sum().backward()
    else:
        ((nab_r.norm(dim=-1) - 1) ** 2).mean().add((sdf_r * ws).sum()).backward()
        ((out["nablas"].norm(dim=-1) - 1) ** 2).mean().add((out["sdf"] * ws.to(backend)).sum()).backward()
    ref = oracle_flat_grads(p)
    for k, v in dict(grid=model.encoding.flattened_params.grad, sdf_w=model.sdf_w.grad, sdf_b=model.sdf_b.grad).items():
        e = rel_l2(v.cpu(), ref[k])
        assert e < 2e-4, (which, k, e)


def test_sdf_scale_and_inside_out(backend):
    """``sdf_scale`` (street config :158) divides the decoder output; ``inside_out`` (indoor config :95) flips the sign
    of the geometric initialisation.  Checked against the oracle evaluated with the head weights pre-divided."""
    from neuralsim_amd.fields.neus import LoTDNeuSModel
    scale = 25.0
    p = make_params(sdf_D=2, small=True, sphere=True, seed=5, grid_bound=2e-2, noise_scale=1.0)
    m = model_from_params(p, backend, precision="f32")
    m.sdf_scale = scale
    m._wpack_versions = None
    g = torch.Generator().manual_seed(1)
    x = (torch.rand(300, 3, generator=g) * 2 - 1) * 0.9
    # oracle with the effective head
    p.sdf_w[-1] = p.sdf_w[-1] / scale
    p.sdf_b[-1] = p.sdf_b[-1] / scale
    p.requires_grad_(True)
    sdf_o, nab_o = ofield.forward_sdf_nablas(x, p)
    loss_o = (sdf_o * 0.7).sum() + ((nab_o.norm(dim=-1) - 1.0) ** 2).mean()
    loss_o.backward()
    out = m.forward_sdf_nablas(x.to(backend))
    loss = (out["sdf"] * 0.7).sum() + ((out["nablas"].norm(dim=-1) - 1.0) ** 2).mean()
    loss.backward()
    assert float((out["sdf"].detach().cpu() - sdf_o.detach()).abs().max()) <= 2e-5
    assert float((m.query_sdf(x.to(backend)).cpu() - sdf_o.detach()).abs().max()) <= 2e-5
    gw = m.sdf_w.grad.cpu()
    head_o = p.sdf_w[-1].grad.reshape(-1) / scale          # d/dW = d/dW_eff / scale
    assert float((gw[-64:] - head_o).norm() / head_o.norm()) <= 2e-4
    w1_o = p.sdf_w[0].grad.reshape(-1)
    assert float((gw[:w1_o.numel()] - w1_o).norm() / w1_o.norm()) <= 2e-4
    gg = m.encoding.flattened_params.grad.cpu()
    assert float((gg - p.grid.grad).norm() / p.grid.grad.norm()) <= 2e-4
    # geometric initialisation: sphere of radius 0.5, either sign, for a scaled head
    for inside_out in (False, True):
        mm = LoTDNeuSModel(lod_res=p.spec.lod_res, log2_hashmap_size=12, sdf_D=2, precision="f32", sdf_scale=scale,
                           inside_out=inside_out).to(backend)
        mm.geometric_init_sphere(0.5, noise_scale=0.0)
        pts = torch.tensor([[0.25, 0.1, 0.0], [0.9, 0.0, 0.0], [0.0, -0.5, 0.0]])
        sd = mm.query_sdf(pts.to(backend)).cpu()
        want = (pts.norm(dim=-1) - 0.5) * (-1.0 if inside_out else 1.0)
        assert float((sd - want).abs().max()) < 0.08, (inside_out, sd)


@pytest.mark.parametrize("case", ["cuboid", "anneal", "cuboid+anneal", "cuboid+aabb"])
def test_field_cuboid_levels_and_hardmask(backend, case):
    """Per-axis level resolutions (``lotd_use_cuboid``, street config :160) and hardmask level annealing
    (``anneal_cfg{type: hardmask}``, dtu config :104-108): values, normals and every gradient vs the oracle; the masked
    levels must get exactly zero gradient."""
    from oracle import lotd as olotd
    cub = "cuboid" in case
    n_active = 9 if "anneal" in case else None
    lod_res = olotd.cuboid_ngp_res([2.0, 1.0, 0.5], 3, 40, 16) if cub else list(SMALL_RES_T)
    p = ofield.make_field_params(lod_res=lod_res, log2_hashmap_size=12, sdf_D=2, seed=5, sphere_init=False,
                                 grid_bound=0.3, noise_scale=1.0)
    p.grid = p.grid.float()
    p.spec.n_active = n_active
    box = None
    if "aabb" in case:
        # an elongated, off-centre AABB in object units (the street model after populate(aabb=...)): the pyramid spans
        # the box per axis, positions / normals / gradients are in object coordinates
        box = torch.tensor([[-3.0, -2.5, -0.75], [5.0, 1.5, 1.25]])
        p.spec.aabb = box
    for t in p.tensors():
        t.requires_grad_(True)
    model = model_from_params(p, backend, precision="f32")
    model.set_active_levels(n_active)
    if box is not None:
        assert torch.equal(model.accel.aabb.cpu(), box) and abs(model.encoding.cfg.meta.x_scale[0] - 0.125) < 1e-7
    assert ("Hash" in p.spec.lod_types) and ("Dense" in p.spec.lod_types)
    g = torch.Generator().manual_seed(2)
    R, S = 5, 90
    rays_o = torch.randn(R, 3, generator=g) * 0.1
    rays_d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    ridx = torch.randint(0, R, (S,), generator=g).sort().values
    t = torch.rand(S, generator=g) * 0.8
    h_appear = torch.randn(R, 4, generator=g) * 0.5
    if box is not None:
        rays_o = (box[0] + box[1]) / 2 + rays_o * (box[1] - box[0]) / 2
        t = t * float((box[1] - box[0]).min()) / 2
    x = rays_o[ridx] + t[:, None] * rays_d[ridx]
    if box is not None:
        assert bool(((x > box[0]) & (x < box[1])).all())
    ha_o = leaf(h_appear)
    sdf_r, nab_r, rgb_r = ofield.forward_field(x, rays_d[ridx], ha_o[ridx], p)
    dv = lambda a: a.to(backend).contiguous()
    ha_d = leaf(h_appear, backend)
    sdf, nab, rgb = _FieldFn.apply(model, model.encoding.flattened_params, model.sdf_w, model.sdf_b, model.rad_w,
                                   model.rad_b, ha_d, None, dv(rays_o), dv(rays_d), dv(t), dv(ridx), True)
    assert (sdf.cpu() - sdf_r).abs().max() < 2e-5 * (1 + sdf_r.abs().max())
    assert (nab.cpu() - nab_r).abs().max() < 2e-4 * (1 + nab_r.abs().max())
    assert (rgb.cpu() - rgb_r).abs().max() < 2e-5
    for fused in (False, True):                   # level-major and fused no-grad query
        model._sdf_fused = fused
        assert torch.allclose(model._query_sdf_rays(dv(rays_o), dv(rays_d), dv(t), dv(ridx)).cpu(), sdf.cpu().detach(),
                              atol=2e-6)
    # standalone encoding (forward + dy/dx) sees the same levels
    h_o = olotd.lotd_forward(x, p.grid, p.spec)
    model.encoding.cfg.set_active_levels(n_active)
    h_p, _ = model.encoding.forward_dydx(dv(x))
    assert (h_p.detach().cpu() - h_o.detach()).abs().max() < 1e-5
    ws, wn, wr = torch.randn(S, generator=g), torch.randn(S, 3, generator=g) * 0.1, torch.randn(S, 3, generator=g)
    (sdf_r * ws).sum().add((nab_r * wn).sum()).add((rgb_r * wr).sum()).backward()
    (sdf * dv(ws)).sum().add((nab * dv(wn)).sum()).add((rgb * dv(wr)).sum()).backward()
    ref = oracle_flat_grads(p)
    got = dict(grid=model.encoding.flattened_params.grad, sdf_w=model.sdf_w.grad, sdf_b=model.sdf_b.grad,
               rad_w=model.rad_w.grad, rad_b=model.rad_b.grad)
    for k, v in got.items():
        e = rel_l2(v.cpu(), ref[k])
        assert e < 2e-4, (k, e)
    if n_active is not None:
        off = p.spec.lod_offsets[n_active]
        assert float(got["grid"][off:].abs().max()) == 0.0
        assert float(got["grid"][:off].abs().max()) > 0.0


def test_anneal_schedule():
    from neuralsim_amd.fields.neus import LoTDNeuSModel
    m = LoTDNeuSModel(lod_res=SMALL_RES_T, log2_hashmap_size=12)
    assert m.anneal_levels(0, 0, 1000, 2) == 3 and m.field_meta.lotd.n_active_levels == 3
    assert m.anneal_levels(500, 0, 1000, 2) == 9
    assert m.anneal_levels(1000, 0, 1000, 2) == 16 and m.field_meta.lotd.n_active_levels == 0    # 0 = all


@pytest.mark.parametrize("precision", ["f32", "fp16"])
def test_field_fewer_than_16_levels(backend, precision, poisoned_empty):
    """Pyramids with fewer than 16 levels (decoder input 2 L < 32): values, normals, colours and all gradients."""
    from oracle import lotd as olotd
    lod_res = list(SMALL_RES_T[:11])
    p = ofield.make_field_params(lod_res=lod_res, log2_hashmap_size=12, sdf_D=2, seed=7, sphere_init=False,
                                 grid_bound=0.3, noise_scale=1.0)
    p.grid = p.grid.float()
    for t in p.tensors():
        t.requires_grad_(True)
    assert p.sdf_w[0].shape == (64, 22)
    model = model_from_params(p, backend, precision=precision)
    assert model.sdf_w.numel() == 64 * 22 + 4096 + 64
    g = torch.Generator().manual_seed(2)
    R, S = 6, 100
    rays_o = torch.randn(R, 3, generator=g) * 0.1
    rays_d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    ridx = torch.randint(0, R, (S,), generator=g).sort().values
    t = torch.rand(S, generator=g) * 0.8
    h_appear = torch.randn(R, 4, generator=g) * 0.5
    x = rays_o[ridx] + t[:, None] * rays_d[ridx]
    ha_o = leaf(h_appear)
    sdf_r, nab_r, rgb_r = ofield.forward_field(x, rays_d[ridx], ha_o[ridx], p)
    dv = lambda a: a.to(backend).contiguous()
    ha_d = leaf(h_appear, backend)
    sdf, nab, rgb = _FieldFn.apply(model, model.encoding.flattened_params, model.sdf_w, model.sdf_b, model.rad_w,
                                   model.rad_b, ha_d, None, dv(rays_o), dv(rays_d), dv(t), dv(ridx), True)
    tol = dict(f32=(2e-5, 2e-4, 2e-5, 2e-4), fp16=(4e-3, 5e-2, 4e-3, 3e-2))[precision]
    assert (sdf.cpu() - sdf_r).abs().max() < tol[0] * (1 + sdf_r.abs().max())
    assert (nab.cpu() - nab_r).abs().max() < tol[1] * (1 + nab_r.abs().max())
    assert (rgb.cpu() - rgb_r).abs().max() < tol[2]
    for fused in (False, True):
        model._sdf_fused = fused
        assert torch.allclose(model._query_sdf_rays(dv(rays_o), dv(rays_d), dv(t), dv(ridx)).cpu(), sdf.cpu().detach(),
                              atol=2e-6)
    ws, wn, wr = torch.randn(S, generator=g), torch.randn(S, 3, generator=g) * 0.1, torch.randn(S, 3, generator=g)
    (sdf_r * ws).sum().add((nab_r * wn).sum()).add((rgb_r * wr).sum()).backward()
    (sdf * dv(ws)).sum().add((nab * dv(wn)).sum()).add((rgb * dv(wr)).sum()).backward()
    ref = oracle_flat_grads(p)
    got = dict(grid=model.encoding.flattened_params.grad, sdf_w=model.sdf_w.grad, sdf_b=model.sdf_b.grad,
               rad_w=model.rad_w.grad, rad_b=model.rad_b.grad)
    for k, v in got.items():
        e = rel_l2(v.cpu(), ref[k])
        assert e < tol[3], (k, e)


@pytest.mark.parametrize("levels,sdf_D,precision", [(19, 2, "f32"), (19, 2, "fp16"), (24, 1, "f32"), (32, 2, "fp16")])
def test_field_more_than_16_levels(backend, levels, sdf_D, precision, poisoned_empty):
    """Pyramids with 17..32 levels (the street configs' auto pyramids have ~18-20): the decoder's first layer contracts
    over two 16-level feature chunks (csrc/field.hip: NC = 2) -- values, normals, colours, the no-grad SDF query and all
    gradients against the oracle."""
    lod_res = [4 + int(round(2.9 * i + 0.11 * i * i)) for i in range(levels)]
    p = ofield.make_field_params(lod_res=lod_res, log2_hashmap_size=12, sdf_D=sdf_D, seed=11, sphere_init=False,
                                 grid_bound=0.3, noise_scale=1.0)
    p.grid = p.grid.float()
    for t in p.tensors():
        t.requires_grad_(True)
    assert p.sdf_w[0].shape == (64, 2 * levels)
    model = model_from_params(p, backend, precision=precision)
    assert model.plane_levels == 32
    g = torch.Generator().manual_seed(3)
    R, S = 7, 150
    rays_o = torch.randn(R, 3, generator=g) * 0.1
    rays_d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    ridx = torch.randint(0, R, (S,), generator=g).sort().values
    t = torch.rand(S, generator=g) * 0.8
    h_appear = torch.randn(R, 4, generator=g) * 0.5
    x = rays_o[ridx] + t[:, None] * rays_d[ridx]
    ha_o = leaf(h_appear)
    sdf_r, nab_r, rgb_r = ofield.forward_field(x, rays_d[ridx], ha_o[ridx], p)
    dv = lambda a: a.to(backend).contiguous()
    ha_d = leaf(h_appear, backend)
    sdf, nab, rgb = _FieldFn.apply(model, model.encoding.flattened_params, model.sdf_w, model.sdf_b, model.rad_w,
                                   model.rad_b, ha_d, None, dv(rays_o), dv(rays_d), dv(t), dv(ridx), True)
    tol = dict(f32=(2e-5, 2e-4, 2e-5, 2e-4), fp16=(4e-3, 5e-2, 4e-3, 3e-2))[precision]
    assert (sdf.cpu() - sdf_r).abs().max() < tol[0] * (1 + sdf_r.abs().max())
    assert (nab.cpu() - nab_r).abs().max() < tol[1] * (1 + nab_r.abs().max())
    assert (rgb.cpu() - rgb_r).abs().max() < tol[2]
    # no-grad paths: the sampling query and the evaluation forward (planes are used there too)
    q = model._query_sdf_rays(dv(rays_o), dv(rays_d), dv(t), dv(ridx)).cpu()
    assert (q - sdf_r.detach()).abs().max() < tol[0] * (1 + sdf_r.abs().max())
    with torch.no_grad():
        sdf_e, nab_e = _FieldFn.apply(model, model.encoding.flattened_params, model.sdf_w, model.sdf_b, model.rad_w,
                                      model.rad_b, None, None, dv(rays_o), dv(rays_d), dv(t), dv(ridx), False)
    assert torch.allclose(sdf_e.cpu(), sdf.detach().cpu(), atol=1e-6)
    assert torch.allclose(nab_e.cpu(), nab.detach().cpu(), atol=1e-5)
    ws, wn, wr = torch.randn(S, generator=g), torch.randn(S, 3, generator=g) * 0.1, torch.randn(S, 3, generator=g)
    (sdf_r * ws).sum().add((nab_r * wn).sum()).add((rgb_r * wr).sum()).backward()
    (sdf * dv(ws)).sum().add((nab * dv(wn)).sum()).add((rgb * dv(wr)).sum()).backward()
    ref = oracle_flat_grads(p)
    got = dict(grid=model.encoding.flattened_params.grad, sdf_w=model.sdf_w.grad, sdf_b=model.sdf_b.grad,
               rad_w=model.rad_w.grad, rad_b=model.rad_b.grad)
    for k, v in got.items():
        e = rel_l2(v.cpu(), ref[k])
        assert e < tol[3], (k, e)


@pytest.mark.parametrize("levels,precision", [(16, "f32"), (16, "fp16"), (19, "f32")])
def test_pose_gradients(backend, levels, precision):
    """Pose refinement (LearnableParams, withmask_withlidar_joint.240219.yaml:338-352): gradients w.r.t. rays_o / rays_d
    of the with-grad query -- through the features (dh/dx), the normals' own position dependence (mixed second
    derivatives of the interpolant), the radiance net's position input and its view direction (SH-4) -- and
    ``forward_sdf_nablas(x)`` w.r.t. x, against torch autograd on the oracle."""
    lod_res = list(SMALL_RES_T) if levels == 16 else [4 + int(round(2.9 * i + 0.11 * i * i)) for i in range(levels)]
    p = ofield.make_field_params(lod_res=lod_res, log2_hashmap_size=12, sdf_D=2, seed=5, sphere_init=False,
                                 grid_bound=0.3, noise_scale=1.0)
    p.grid = p.grid.float()
    model = model_from_params(p, backend, precision=precision)
    g = torch.Generator().manual_seed(4)
    R, S = 9, 160
    rays_o = torch.randn(R, 3, generator=g) * 0.1
    rays_d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    ridx = torch.randint(0, R, (S,), generator=g).sort().values
    t = torch.rand(S, generator=g) * 0.8
    h_appear = torch.randn(R, 4, generator=g) * 0.5
    ws, wn, wr = torch.randn(S, generator=g), torch.randn(S, 3, generator=g) * 0.1, torch.randn(S, 3, generator=g)
    # oracle
    o_r, d_r = leaf(rays_o), leaf(rays_d)
    x = o_r[ridx] + t[:, None] * d_r[ridx]
    sdf_r, nab_r, rgb_r = ofield.forward_field(x, d_r[ridx], h_appear[ridx], p, x_has_grad=True)
    (sdf_r * ws).sum().add((nab_r * wn).sum()).add((rgb_r * wr).sum()).backward()
    # device
    dv = lambda a: a.to(backend).contiguous()
    o_d, d_d = leaf(rays_o, backend), leaf(rays_d, backend)
    sdf, nab, rgb = _FieldFn.apply(model, model.encoding.flattened_params, model.sdf_w, model.sdf_b, model.rad_w,
                                   model.rad_b, dv(h_appear), None, o_d, d_d, dv(t), dv(ridx), True)
    (sdf * dv(ws)).sum().add((nab * dv(wn)).sum()).add((rgb * dv(wr)).sum()).backward()
    tol = dict(f32=3e-4, fp16=4e-2)[precision]
    assert rel_l2(o_d.grad.cpu(), o_r.grad) < tol, rel_l2(o_d.grad.cpu(), o_r.grad)
    assert rel_l2(d_d.grad.cpu(), d_r.grad) < tol, rel_l2(d_d.grad.cpu(), d_r.grad)
    # each rgb-free piece on its own: sdf only (first order), nablas only (second order)
    for w_s, w_n in ((1.0, 0.0), (0.0, 1.0)):
        o_r.grad = d_r.grad = None
        x = o_r[ridx] + t[:, None] * d_r[ridx]
        s_r, n_r = ofield.forward_sdf_nablas(x, p, x_has_grad=True)
        ((s_r * ws).sum() * w_s + (n_r * wn).sum() * w_n).backward()
        o_d.grad = d_d.grad = None
        s_d, n_d = _FieldFn.apply(model, model.encoding.flattened_params, model.sdf_w, model.sdf_b, model.rad_w,
                                  model.rad_b, None, None, o_d, d_d, dv(t), dv(ridx), False)
        ((s_d * dv(ws)).sum() * w_s + (n_d * dv(wn)).sum() * w_n).backward()
        assert rel_l2(o_d.grad.cpu(), o_r.grad) < tol, (w_s, w_n, rel_l2(o_d.grad.cpu(), o_r.grad))
        assert rel_l2(d_d.grad.cpu(), d_r.grad) < tol, (w_s, w_n)
    # point mode: forward_sdf_nablas(x) with x.requires_grad
    xp = (torch.rand(100, 3, generator=g) * 2 - 1) * 0.9
    x_r = leaf(xp)
    s_r, n_r = ofield.forward_sdf_nablas(x_r, p, x_has_grad=True)
    ((s_r * ws[:100]).sum() + (n_r * wn[:100]).sum()).backward()
    x_d = leaf(xp, backend)
    out = model.forward_sdf_nablas(x_d)
    ((out["sdf"] * dv(ws[:100])).sum() + (out["nablas"] * dv(wn[:100])).sum()).backward()
    assert rel_l2(x_d.grad.cpu(), x_r.grad) < tol


@pytest.mark.parametrize("levels,sdf_D", [(16, 2), (16, 1), (19, 2), (24, 1)])
def test_split_precision_sdf_query(backend, levels, sdf_D):
    """Precision 2 of the no-grad SDF query (csrc/field.hip ``SPLIT_LO_SCALE``): every operand of the decoder's matrix
    products -- features, weights, hidden activations -- travels through the f16 matrix cores as hi + lo, three MFMAs per
    product.  The result must be f32-accurate (the fp16 query of the same model is ~1e-3 off on these weights)."""
    lod_res = [4 + int(round(2.9 * i + 0.11 * i * i)) for i in range(levels)]
    p = ofield.make_field_params(lod_res=lod_res, log2_hashmap_size=12, sdf_D=sdf_D, seed=11, sphere_init=False,
                                 grid_bound=0.3, noise_scale=1.0)
    p.grid = p.grid.float()
    model = model_from_params(p, backend, precision="fp16")
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(500, 3, generator=g) * 2 - 1) * 0.95
    with torch.no_grad():
        ref = ofield.forward_sdf(x, p)
    xd = x.to(backend).contiguous()
    grid16, wpack = model._shadow()
    err = {}
    for sp in ("fp16", "split", "f32"):
        model.sampling_precision = sp
        fm, wp = model._sampling_ctx()
        assert fm.precision == {"fp16": 0, "split": 2, "f32": 1}[sp]
        q = model._sdf_query(grid16, wp, xd, None, None, None, None, x.shape[0], backend, fm=fm).cpu()
        err[sp] = float((q - ref).abs().max())
    scale = 1.0 + float(ref.abs().max())
    assert err["f32"] < 2e-6 * scale and err["split"] < 4e-6 * scale, err
    assert err["fp16"] > 20 * err["split"], err


@pytest.mark.parametrize("precision,sdf_D", [("f32", 2), ("fp16", 2), ("f32", 1)])
def test_field_with_relu_sdf_decoder(backend, precision, sdf_D):
    """``decoder_cfg.activation: relu`` (the Vehicle decoder of no_fg_occ.221218.yaml:354-357): values, normals, colours,
    the no-grad query and all gradients -- the curvature terms of the normals' double backward vanish for relu."""
    p = make_params(sdf_D=sdf_D, small=True, sphere=False, grid_bound=0.3, seed=8, noise_scale=1.0)
    p.sdf_activation = "relu"
    for t in p.tensors():
        t.requires_grad_(True)
    model = model_from_params(p, backend, precision=precision)
    assert model.sdf_activation == "relu" and model.field_meta.softplus_beta < 0
    g = torch.Generator().manual_seed(6)
    R, S = 7, 150
    rays_o = torch.randn(R, 3, generator=g) * 0.1
    rays_d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    ridx = torch.randint(0, R, (S,), generator=g).sort().values
    t = torch.rand(S, generator=g) * 0.8
    h_appear = torch.randn(R, 4, generator=g) * 0.5
    x = rays_o[ridx] + t[:, None] * rays_d[ridx]
    ha_o = leaf(h_appear)
    sdf_r, nab_r, rgb_r = ofield.forward_field(x, rays_d[ridx], ha_o[ridx], p)
    ha_d = leaf(h_appear, backend)
    dv = lambda a: a.to(backend).contiguous()
    sdf, nab, rgb = _FieldFn.apply(model, model.encoding.flattened_params, model.sdf_w, model.sdf_b, model.rad_w,
                                   model.rad_b, ha_d, None, dv(rays_o), dv(rays_d), dv(t), dv(ridx), True)
    tol = dict(f32=(2e-5, 2e-4, 2e-5, 3e-4), fp16=(4e-3, 5e-2, 4e-3, 3e-2))[precision]
    assert (sdf.cpu() - sdf_r).abs().max() < tol[0] * (1 + sdf_r.abs().max())
    assert (nab.cpu() - nab_r).abs().max() < tol[1] * (1 + nab_r.abs().max())
    assert (rgb.cpu() - rgb_r).abs().max() < tol[2]
    q = model._query_sdf_rays(dv(rays_o), dv(rays_d), dv(t), dv(ridx)).cpu()
    assert (q - sdf_r.detach()).abs().max() < tol[0] * (1 + sdf_r.abs().max())
    ws, wn, wr = torch.randn(S, generator=g), torch.randn(S, 3, generator=g) * 0.1, torch.randn(S, 3, generator=g)
    (sdf_r * ws).sum().add((nab_r * wn).sum()).add((rgb_r * wr).sum()).backward()
    (sdf * dv(ws)).sum().add((nab * dv(wn)).sum()).add((rgb * dv(wr)).sum()).backward()
    ref = oracle_flat_grads(p)
    got = dict(grid=model.encoding.flattened_params.grad, sdf_w=model.sdf_w.grad, sdf_b=model.sdf_b.grad,
               rad_w=model.rad_w.grad, rad_b=model.rad_b.grad)
    for k, v in got.items():
        e = rel_l2(v.cpu(), ref[k])
        assert e < tol[3], (k, e)


def test_small_sdf_query_fused_point_major_equals_level_major(backend, monkeypatch):
    """Round 5: a small no-grad SDF launch of the sampling pass (split precision) as ONE fused point-major launch
    (``NSIM_SDF_FUSED_BELOW``) == the level-major gather + the decoder on the planes, also with a device-side point count below
    the capacity (points past it are neither read nor folded into the occupancy values)."""
    from neuralsim_amd.fields import neus as nmod
    p = make_params(sdf_D=2, small=True, sphere=True, seed=5, ln_inv_s=0.45, grid_bound=2e-2, noise_scale=1.0)
    m = model_from_params(p, backend, precision="fp16")
    assert m.sampling_precision == "split"
    g = torch.Generator().manual_seed(4)
    S = 1000
    x = (torch.rand(S, 3, generator=g) * 1.8 - 0.9).to(backend)
    x[700:] = float("nan")                                   # past the device-side count: must not be touched
    fm_s, wpack = m._sampling_ctx()
    grid16, _ = m._shadow()
    n_dev = torch.tensor([640], dtype=torch.long, device=backend)
    outs = []
    for below in (0, 4096):
        monkeypatch.setattr(nmod, "_SDF_FUSED_BELOW", below)
        m.accel.occ_val.zero_()
        full = m._sdf_query(grid16, wpack, x[:700].contiguous(), None, None, None, None, 700, backend, fm=fm_s)
        part = m._sdf_query(grid16, wpack, x, None, None, None, None, S, backend, n_dev=n_dev, n_add=60, collect=True, fm=fm_s)
        outs.append((full.cpu(), part.cpu()[:700], m.accel.occ_val.cpu().clone()))
    (f0, p0, o0), (f1, p1, o1) = outs
    assert torch.equal(f0, f1) and torch.equal(p0, p1) and torch.equal(f0, p0)
    assert torch.equal(o0, o1) and bool(torch.isfinite(o1).all()) and float(o1.max()) > 0


def _with_pos_embed(p, n_freq, seed):
    """Widen the oracle decoder's first layer by the embedded-position block (random columns)."""
    g = torch.Generator().manual_seed(seed)
    E = 3 + 6 * n_freq
    w1 = p.sdf_w[0].detach()
    extra = (torch.rand(64, E, generator=g) * 2 - 1) * (1.0 / (w1.shape[1] + E) ** 0.5)
    p.sdf_w[0] = torch.cat([w1, extra], dim=1)
    p.pos_embed_n = n_freq
    return p


@pytest.mark.parametrize("case", ["vehicle_relu", "softplus_D1_aabb", "softplus_D2_scale", "ten_frequencies", "24_levels"])
def test_field_with_extra_pos_embed(backend, case, poisoned_empty):
    """``surface_cfg.extra_pos_embed_cfg{type: sinusoidal_legacy, n_frequencies: 6}`` (the StyleLoTD Vehicle block,
    no_fg_occ.221218.yaml:319-321, with its relu 2x64 decoder :354-357): the decoder reads [features | embedded position]
    (71 inputs): with-grad forward and backward on the matrix cores with the block as two more 32-input chunks of the first
    layer (csrc/field.hip NE = 2; f32-MFMA validation mode in four cases, the fp16 product mode in ``vehicle_relu``), the no-grad
    query in f32 on csrc/wide_field.hip -- values, normals, colours and every gradient against the oracle, including the
    second-order path of the normals through the embedded position's own x-derivative."""
    sdf_D = 1 if case == "softplus_D1_aabb" else 2
    if case == "24_levels":     # a 24-level pyramid (the 32-level planes of the gather) + ten frequencies: 48 + 63 = 111 inputs
        lod_res = [4 + int(round(2.9 * i + 0.11 * i * i)) for i in range(24)]
        p = ofield.make_field_params(lod_res=lod_res, log2_hashmap_size=12, sdf_D=sdf_D, seed=11, sphere_init=False,
                                     grid_bound=0.3, noise_scale=1.0)
        p.grid = p.grid.float()
    else:
        p = make_params(sdf_D=sdf_D, small=True, sphere=False, grid_bound=0.3, seed=11, noise_scale=1.0)
    # (first-layer widths 71 / 53 / 71 / 95 / 111: three MFMA input chunks up to 16 levels, four for the 24-level pyramid; the
    # 72-, 56-, 104- and 128-wide instantiations of the no-grad k_wide_sdf)
    n_freq = {"softplus_D1_aabb": 3, "ten_frequencies": 10, "24_levels": 10}.get(case, 6)
    F1 = 2 * len(p.spec.lod_res)
    _with_pos_embed(p, n_freq, seed=3)
    if case == "vehicle_relu":
        p.sdf_activation = "relu"
    if case == "softplus_D1_aabb":          # a non-cubic box: x_n and d x_n / d x differ per axis
        p.spec.aabb = torch.tensor([[-0.7, -0.5, -0.9], [0.7, 0.6, 0.8]])
    if case == "softplus_D2_scale":
        p.sdf_scale = 4.0
    for t in p.tensors():
        t.requires_grad_(True)
    model = model_from_params(p, backend, precision="fp16" if case == "vehicle_relu" else "f32")
    assert model.pos_embed_E == 3 + 6 * n_freq and model.sdf_w.numel() == 64 * (F1 + model.pos_embed_E) + (4096 if sdf_D == 2 else 0) + 64
    g = torch.Generator().manual_seed(6)
    R, S = 9, 203
    rays_o = torch.randn(R, 3, generator=g) * 0.1
    rays_d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    ridx = torch.randint(0, R, (S,), generator=g).sort().values
    t = torch.rand(S, generator=g) * 0.45
    h_appear = torch.randn(R, 4, generator=g) * 0.5
    x = rays_o[ridx] + t[:, None] * rays_d[ridx]
    ha_o = leaf(h_appear)
    sdf_r, nab_r, rgb_r = ofield.forward_field(x, rays_d[ridx], ha_o[ridx], p)
    ha_d = leaf(h_appear, backend)
    dv = lambda a: a.to(backend).contiguous()
    sdf, nab, rgb = _FieldFn.apply(model, model.encoding.flattened_params, model.sdf_w, model.sdf_b, model.rad_w,
                                   model.rad_b, ha_d, None, dv(rays_o), dv(rays_d), dv(t), dv(ridx), True)
    # (the tolerances of the other decoder tests of this file: f32-MFMA validation mode | fp16 MFMA product mode)
    ts, tn, tr = (4e-3, 5e-2, 4e-3) if case == "vehicle_relu" else (2e-5, 2e-4, 2e-5)
    assert (sdf.cpu() - sdf_r).abs().max() < ts * (1 + sdf_r.abs().max())
    en = (nab.cpu() - nab_r).abs().amax(dim=-1)
    if case == "vehicle_relu":
        # relu has a kink: a hidden unit whose pre-activation lies within the fp16 rounding of zero takes the other branch, and
        # the normal of that point moves by the unit's whole contribution (0.19 for one point of these 203).  Such points are rare
        # and the values stay continuous: all but 2 % of the points within the fp16 tolerance, the median at fp16 rounding level
        assert float((en > tn * (1 + nab_r.abs().max())).float().mean()) < 0.02 and float(en.median()) < 2e-3
        flipped = en > tn * (1 + nab_r.abs().max())
    else:
        assert en.max() < tn * (1 + nab_r.abs().max())
        flipped = torch.zeros_like(en, dtype=torch.bool)
    assert (rgb.cpu() - rgb_r).abs()[~flipped].max() < tr
    q = model._query_sdf_rays(dv(rays_o), dv(rays_d), dv(t), dv(ridx)).cpu()
    assert (q - sdf_r.detach()).abs().max() < 2e-5 * (1 + sdf_r.abs().max())
    assert (model.query_sdf(dv(x)).cpu() - sdf_r.detach()).abs().max() < 2e-5 * (1 + sdf_r.abs().max())
    # points mode, no colour (the eikonal query of uniform points)
    out = model.forward_sdf_nablas(dv(x[:50]))
    assert (out["nablas"].detach().cpu() - nab_r[:50].detach()).abs().amax(dim=-1)[~flipped[:50]].max() < tn * (1 + nab_r.abs().max())
    ws, wn, wr = torch.randn(S, generator=g), torch.randn(S, 3, generator=g) * 0.1, torch.randn(S, 3, generator=g)
    (sdf_r * ws).sum().add((nab_r * wn).sum()).add((rgb_r * wr).sum()).backward()
    (sdf * dv(ws)).sum().add((nab * dv(wn)).sum()).add((rgb * dv(wr)).sum()).backward()
    ref = oracle_flat_grads(p)
    got = dict(grid=model.encoding.flattened_params.grad, sdf_w=model.sdf_w.grad, sdf_b=model.sdf_b.grad,
               rad_w=model.rad_w.grad, rad_b=model.rad_b.grad)
    for k, v in got.items():
        e = rel_l2(v.cpu(), ref[k])
        # (fp16 mode: the tolerance of test_field_with_relu_sdf_decoder)
        assert e < (3e-2 if case == "vehicle_relu" else 3e-4), (k, e)
    assert rel_l2(ha_d.grad.cpu(), ha_o.grad) < (3e-2 if case == "vehicle_relu" else 3e-4)
    # the embedded-position columns of W1 carry gradient (first- and second-order terms)
    FIN = F1 + model.pos_embed_E
    assert float(model.sdf_w.grad[:64 * FIN].view(64, FIN)[:, F1:].abs().max()) > 0


def test_pair_pack_equals_the_two_single_packs(backend):
    """``nsim_field_pack_weights2`` (field precision + sampling precision in one launch) writes the bytes of two
    ``nsim_field_pack_weights`` calls."""
    from neuralsim_amd import _lib
    p = make_params(sdf_D=2, small=True, sphere=False, grid_bound=0.3, seed=4, noise_scale=1.0)
    m = model_from_params(p, backend, precision="fp16")
    fa, fb = _lib.FieldMeta(), _lib.FieldMeta()
    import ctypes
    for f in (fa, fb):
        ctypes.memmove(ctypes.byref(f), ctypes.byref(m.field_meta), ctypes.sizeof(f))
    fa.precision, fb.precision = 0, 2
    lib = _lib.get_lib()
    na, nb = int(lib.nsim_field_wpack_bytes(fa)), int(lib.nsim_field_wpack_bytes(fb))
    ws = [m.sdf_w.detach(), m.sdf_b.detach(), m.rad_w.detach(), m.rad_b.detach()]
    one_a, one_b = (torch.zeros([n], dtype=torch.uint8, device=backend) for n in (na, nb))
    two_a, two_b = (torch.zeros([n], dtype=torch.uint8, device=backend) for n in (na, nb))
    _lib.call("nsim_field_pack_weights", fa, *[_lib.ptr(w) for w in ws], _lib.ptr(one_a))
    _lib.call("nsim_field_pack_weights", fb, *[_lib.ptr(w) for w in ws], _lib.ptr(one_b))
    _lib.call("nsim_field_pack_weights2", fa, _lib.ptr(two_a), fb, _lib.ptr(two_b), *[_lib.ptr(w) for w in ws])
    assert torch.equal(one_a.cpu(), two_a.cpu()) and torch.equal(one_b.cpu(), two_b.cpu())
    assert int(one_a.cpu().count_nonzero()) > na // 4 and int(one_b.cpu().count_nonzero()) > nb // 4
    # the model's own lazy refresh uses it once both packs exist: same sampling-pass SDFs before / after an in-place update
    x = (torch.rand(300, 3, generator=torch.Generator().manual_seed(1)) - 0.5).to(backend)
    fm_s, _ = m._sampling_ctx()
    with torch.no_grad():
        m.sdf_w.mul_(1.01)
    m._wpack_versions = None
    calls = []
    orig = _lib.call
    _lib.call = lambda name, *a, **k: (calls.append(name), orig(name, *a, **k))[1]
    try:
        fm_s, wp_s = m._sampling_ctx()
    finally:
        _lib.call = orig
    assert calls.count("nsim_field_pack_weights2") == 1 and "nsim_field_pack_weights" not in calls, calls
    ref = torch.zeros([nb], dtype=torch.uint8, device=backend)
    _lib.call("nsim_field_pack_weights", fb, *[_lib.ptr(w.detach()) for w in (m.sdf_w, m.sdf_b, m.rad_w, m.rad_b)], _lib.ptr(ref))
    assert torch.equal(wp_s.cpu(), ref.cpu())
