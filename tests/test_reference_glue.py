"""The reference's own ``SingleVolumeRenderer`` source (app/renderers/single_volume_renderer.py, loaded unchanged from
/root/reference by tests/ref_glue.py) driving THIS repository's models through the nr3d_lib shim, compared with the
renderer mirror (``neuralsim_amd.renderers.SingleVolumeRenderer``) on identical rays and weights.

Two uses:
  * in the authoring container (reference present) the reference renderer and the mirror run side by side on the
    emulator backend and must agree -- every output image, the merged volume buffer, ``vw_in_total``, gradients;
  * the same scenario's reference-side outputs are frozen in ``tests/golden/renderer_fixture.pt``
    (``tests/golden/make_renderer_fixture.py``); ``test_mirror_matches_reference_fixture`` replays the mirror against
    them on the GPU box, where /root/reference does not exist.
"""
from pathlib import Path

import pytest
import torch

import ref_glue
from renderer_scenario import SCENARIOS, build_scenario, run_mirror, run_reference

GOLDEN = Path(__file__).resolve().parent / "golden" / "renderer_fixture.pt"
needs_reference = pytest.mark.skipif(not ref_glue.reference_available(), reason="executes the reference's own sources from /root/reference (authoring container only; emulator backend). What it pins is replayed on the GPU box from frozen reference outputs: tests/test_reference_frozen.py, test_reference_glue.py::test_*_fixture")


def _cmp(a, b, tol, what):
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if a.dtype in (torch.long, torch.int32, torch.bool, torch.uint8):
        assert torch.equal(a, b), what
    else:
        e = float((a.float() - b.float()).abs().max()) if a.numel() else 0.0
        assert e <= tol, (what, e)


@needs_reference
@pytest.mark.parametrize("name", list(SCENARIOS))
def test_reference_renderer_runs_on_the_mirror(backend, name):
    """Same process, same models: reference glue vs mirror.  Both sides call the same kernels, so the agreement is
    tight (f32 sums in a different order only)."""
    sc = build_scenario(name, backend)
    with ref_glue.reference_renderer_modules() as mods:
        ref = run_reference(mods, sc, backward=True)
    got = run_mirror(sc, backward=True)
    for k in ref["rendered"]:
        _cmp(got["rendered"][k], ref["rendered"][k], 2e-5, f"rendered.{k}")
    assert set(got["rendered"]) >= set(ref["rendered"])
    _cmp(got["samples_cnt"], ref["samples_cnt"], 0, "samples_cnt")
    for k in ("pack_infos_hit", "t", "opacity_alpha", "rgb", "vw"):
        if k in ref["volume_buffer"]:
            _cmp(got["volume_buffer"][k], ref["volume_buffer"][k], 2e-5, f"volume_buffer.{k}")
    for k in ref["vw_in_total"]:
        _cmp(got["vw_in_total"][k], ref["vw_in_total"][k], 2e-5, f"vw_in_total.{k}")
    for k, g in ref["grads"].items():
        denom = float(g.norm()) + 1e-12
        e = float((got["grads"][k] - g).norm()) / denom
        assert e < 2e-4, (k, e)


@pytest.mark.parametrize("name", list(SCENARIOS))
def test_mirror_matches_reference_fixture(backend, name):
    """The mirror against the frozen outputs of the reference's renderer (generated on the emulator backend in f32
    MFMA mode; the device differs from it by f32 rounding of the field kernels only)."""
    assert GOLDEN.exists(), "run tests/golden/make_renderer_fixture.py in the authoring container"
    fx = torch.load(GOLDEN)[name]
    sc = build_scenario(name, backend)
    got = run_mirror(sc, backward=True)
    _cmp(got["samples_cnt"], fx["samples_cnt"], 0, "samples_cnt")
    assert set(got["volume_buffer"]) == set(fx["volume_buffer"])
    if "pack_infos_hit" in fx["volume_buffer"]:
        _cmp(got["volume_buffer"]["pack_infos_hit"], fx["volume_buffer"]["pack_infos_hit"], 0, "pack_infos_hit")
    for k in fx["rendered"]:
        _cmp(got["rendered"][k], fx["rendered"][k], 2e-2 if k == "depth_volume" else 5e-4, f"rendered.{k}")
    for k in ("t", "opacity_alpha", "rgb", "vw"):
        if k in fx["volume_buffer"]:
            _cmp(got["volume_buffer"][k], fx["volume_buffer"][k], 1e-3, f"volume_buffer.{k}")
    for k, g in fx["grads"].items():
        mine = got["grads"][k]
        if (g["norm"] if isinstance(g, dict) else float(g.norm())) < 1e-4:
            # rounding noise only (the sky behind an opaque far shell: 1 - mask = 0 up to f32 rounding)
            assert float(mine.norm()) < 1e-4, k
            continue
        if isinstance(g, dict):                 # strided sample + norm of a large gradient
            assert abs(float(mine.norm()) - g["norm"]) <= 5e-3 * g["norm"], k
            mine, g = mine[::g["stride"]], g["sample"]
        e = float((mine - g).norm()) / (float(g.norm()) + 1e-12)
        assert e < 5e-3, (k, e)


@needs_reference
def test_reference_volume_integration_pins_the_oracle():
    """``SingleVolumeRenderer._volume_integration`` of the reference (single_volume_renderer.py:73-102) fed with the
    ORACLE's pack ops (pure torch, CPU) against ``oracle.render.volume_integration`` -- pins the oracle's restatement
    of the integration (vw, mask, depth with and without normalisation, rgb, normals in train and eval mode)."""
    from oracle import pack_ops as opo, render as orr
    import sys
    g = torch.Generator().manual_seed(3)
    n = torch.tensor([5, 1, 0, 9, 3])
    pi = opo.get_pack_infos_from_n(n)
    S = int(n.sum())
    alpha, t = torch.rand(S, generator=g) * 0.7, torch.rand(S, generator=g).cumsum(0)
    rgb, nab = torch.rand(S, 3, generator=g), torch.randn(S, 3, generator=g) * 1.5
    keep = n > 0
    pi_hit, rih = pi[keep], keep.nonzero()[:, 0]
    with ref_glue.reference_renderer_modules() as mods:
        m = mods["app.renderers.single_volume_renderer"]
        saved = (m.packed_alpha_to_vw, m.packed_sum, m.packed_div)
        m.packed_alpha_to_vw = lambda a, pack_infos: opo.packed_alpha_to_vw(a, pack_infos)
        m.packed_sum = opo.packed_sum
        m.packed_div = opo.packed_div
        try:
            for training in (True, False):
                for norm_depth in (True, False):
                    r = ref_glue.make_reference_renderer(mods, dict(depth_use_normalized_vw=norm_depth), training=training)
                    rendered = mods["app.renderers.utils"].prepare_empty_rendered([5], with_rgb=True, with_normal=True)
                    vb = dict(type="packed", rays_inds_hit=rih, pack_infos_hit=pi_hit, opacity_alpha=alpha.clone(),
                              t=t, rgb=rgb, nablas_in_world=nab.clone())
                    r._volume_integration(vb, rendered)
                    nn_ = nab if training else torch.nn.functional.normalize(nab.clamp(-1, 1), dim=-1)
                    o = orr.volume_integration(alpha, t, rgb, nn_, pi_hit, norm_depth)
                    assert torch.allclose(vb["vw"], o["vw"], atol=1e-6)
                    for k in ("mask_volume", "depth_volume", "rgb_volume", "normals_volume"):
                        assert torch.allclose(rendered[k][rih], o[k], atol=1e-5), (k, training, norm_depth)
                        assert float(rendered[k][~keep].abs().sum()) == 0.0
        finally:
            m.packed_alpha_to_vw, m.packed_sum, m.packed_div = saved
    assert "app" not in sys.modules


@needs_reference
def test_reference_camera_rays_pin_the_oracle_and_the_kernel(backend):
    """Row a1: ``Camera._get_selected_rays_from_ixy`` and ``get_all_rays`` of the reference (cameras.py:281-310,
    332-360; pixel-centre snapping, clamping, normalisation are the reference's code, the pinhole lift and the
    broadcast rotation are stand-ins for the absent nr3d_lib attributes) against the oracle's ``pinhole_rays`` and
    the ray-generation kernel."""
    from oracle import render as orr
    from neuralsim_amd.eval import all_pixel_xy
    from neuralsim_amd.graphics.cameras import pinhole_selected_rays
    from util import look_at_cameras
    intr, c2w, WH = look_at_cameras(V=5, seed=9, H=40, W=56, f=61.3)
    intr[:, 0, 2] += 0.37                                 # principal point off the pixel grid
    intr[:, 1, 1] *= 1.03
    g = torch.Generator().manual_seed(2)
    N = 1000
    xy = torch.rand(N, 2, generator=g).clamp(1e-6, 1 - 1e-6)
    xy[:4] = torch.tensor([[0.0, 0.0], [1.0, 1.0], [1.0, 0.0], [0.999999, 0.5]])      # the clamp at the image border
    fidx = torch.randint(0, 5, (N,), generator=g)
    with ref_glue.reference_camera_class() as Camera:
        cam = ref_glue.FakeCamera(intr, c2w, WH.float())
        for snap in (True, False):
            o_ref, d_ref = Camera._get_selected_rays_from_ixy(cam, fidx, xy, snap_to_pixel_centers=snap)
            o_o, d_o = orr.pinhole_rays(xy, fidx, intr, c2w, WH, snap_to_pixel_centers=snap)
            assert torch.equal(o_ref, o_o) and float((d_ref - d_o).abs().max()) <= 2e-7
            if snap:
                dv = lambda t: t.to(backend).contiguous()         # noqa: E731
                o_k, d_k = pinhole_selected_rays(dv(xy), dv(fidx), dv(intr), dv(c2w), dv(WH))
                assert torch.equal(o_k.cpu(), o_ref) and float((d_k.cpu() - d_ref).abs().max()) <= 3e-7
        # camera_model: opencv (the street configs): the reference's Camera code around a distorted lift vs the kernel
        from neuralsim_amd.graphics.cameras import opencv_selected_rays
        dist = torch.tensor([[0.043, -0.36, 0.0008, -0.0006, 0.01]]).repeat(5, 1) * torch.linspace(0.5, 1.5, 5)[:, None]
        cam_d = ref_glue.FakeCamera(intr, c2w, WH.float(), distortion=dist)
        o_ref, d_ref = Camera._get_selected_rays_from_ixy(cam_d, fidx, xy, snap_to_pixel_centers=True)
        o_k, d_k = opencv_selected_rays(dv(xy), dv(fidx), dv(intr), dv(dist), dv(c2w), dv(WH))
        assert torch.equal(o_k.cpu(), o_ref) and float((d_k.cpu() - d_ref).abs().max()) <= 3e-7
        assert float((d_k.cpu() - d_o).abs().max()) > 1e-3          # (it is not the pinhole direction)
        one = ref_glue.FakeCamera(intr[2], c2w[2], WH[2].float())
        o_all, d_all = Camera.get_all_rays(one)
    xy_all = all_pixel_xy(56, 40, torch.device("cpu"))
    o_o, d_o = orr.pinhole_rays(xy_all, torch.full([56 * 40], 2), intr, c2w, WH)
    assert o_all.shape == (56 * 40, 3) and torch.equal(o_all, o_o) and float((d_all - d_o).abs().max()) <= 2e-7


@needs_reference
def test_reference_line_of_sight_loss_on_a_lidar_step(backend):
    """The street config's second batch per iteration (withmask_withlidar_joint.240219.yaml:7-8, train.py:896-902):
    rays rendered with ``with_rgb=False`` (radiance net skipped), then the reference's ``LineOfSightLoss``
    (app/loss/lidar.py:57-206, loaded unchanged) reads ``volume_buffer['vw']`` / ``['t']`` / ``pack_infos_hit`` and
    ``rendered['depth_volume']`` of the mirror's return value.  Loss values and the gradient that reaches the table
    are compared with the same formulas evaluated with the oracle's pack ops on the detached buffers."""
    from oracle import pack_ops as opo
    from neuralsim_amd.renderers.single_volume_renderer import SingleVolumeRenderer
    sc = build_scenario("main_train", backend)
    r = SingleVolumeRenderer(dict(with_rgb=False, with_normal=False, near=0.01, depth_use_normalized_vw=True,
                                  perturb=False)).train()
    ret = r.ray_query(sc["rays_o"], sc["rays_d"], model=sc["model"], return_buffer=True)
    assert "rgb_volume" not in ret["rendered"] and "rgb" not in ret["volume_buffer"]
    vb = ret["volume_buffer"]
    N = sc["N"]
    g = torch.Generator().manual_seed(8)
    ranges = (ret["rendered"]["depth_volume"].detach().cpu() + torch.randn(N, generator=g) * 0.05).clamp_min(0.1)
    mask = (torch.rand(N, generator=g) < 0.8)
    gt = dict(ranges=ranges.to(backend))
    with ref_glue.reference_lidar_loss_module() as lidar:
        losses = {}
        for fn_type, param in (("nerf", dict(sigma=0.1)), ("neus_urban", dict(sigma=0.1)), ("neus_unisim", dict(epsilon=0.1))):
            los = lidar.LineOfSightLoss(w=0.5, fn_type=fn_type, fn_param=param)
            for k, v in los(None, ret, None, gt, it=0, mask=mask.to(backend)).items():
                losses[f"{fn_type}.{k}"] = v
    assert len(losses) == 5
    total = sum(losses.values())
    sc["model"].encoding.flattened_params.grad = None
    total.backward()
    g_table = sc["model"].encoding.flattened_params.grad.detach().cpu().clone()
    rg = sc["model"].rad_w.grad
    assert float(g_table.abs().sum()) > 0 and (rg is None or float(rg.abs().sum()) == 0.0)   # no radiance query happened
    # the same formulas on the oracle's pack ops (f64), vw as the leaf
    pi, t = vb["pack_infos_hit"].cpu(), vb["t"].detach().cpu().double()
    rih = vb["rays_inds_hit"].cpu()
    vw = vb["vw"].detach().cpu().double().requires_grad_(True)
    gt_ex = torch.repeat_interleave(ranges[rih].double(), pi[:, 1])
    mh = mask[rih].double()
    sig = 0.1
    tgt = torch.exp(torch.distributions.normal.Normal(0.0, sig / 3.0).log_prob(t - gt_ex))
    nb = opo.packed_sum(((t <= gt_ex + sig) & (t >= gt_ex - sig)) * (vw - tgt) ** 2, pi)
    em = opo.packed_sum((t < gt_ex - sig) * vw ** 2, pi)
    eu = opo.packed_sum(((t - gt_ex).abs() > 0.1) * vw ** 2, pi)
    want = {"nerf.lidar_loss.los.neighbor": 0.5 * (nb * mh).mean(), "nerf.lidar_loss.los.empty": 0.5 * (em * mh).mean(),
            "neus_unisim.lidar_loss.los.empty": 0.5 * (eu * mh).mean()}
    want["neus_urban.lidar_loss.los.neighbor"], want["neus_urban.lidar_loss.los.empty"] = \
        want["nerf.lidar_loss.los.neighbor"], want["nerf.lidar_loss.los.empty"]
    for k, v in want.items():
        assert abs(float(losses[k]) - float(v)) <= 1e-5 * (1 + abs(float(v))), k


@needs_reference
def test_reference_compose_renderer_runs_on_the_mirror(backend):
    """code_multi: the reference's ``BufferComposeRenderer.ray_query`` (buffer_compose_renderer.py:110-850, loaded
    unchanged) over a fake scene -- a single-object background model, a shared batched model with posed instances (one
    of them outside the view, so the batch is compacted), a sky -- against the mirror on the same models and rays."""
    import compose_scenario as cs
    sc = cs.build(backend)
    with ref_glue.reference_compose_renderer_modules() as mods:
        ref = cs.run_reference(mods, sc)
    got = cs.run_mirror(sc)
    _compare_compose(got, ref, 3e-5, 3e-4)


def _compare_compose(got, ref, tol, gtol, btol=None):
    assert torch.equal(got["samples_cnt"], ref["samples_cnt"]) and int((ref["samples_cnt"] > 0).sum()) > 20
    assert ref["vehicle_ids"] == ["car2", "car0"]                         # car1 is never hit: compacted away
    assert ref["rays_crossing_two_items"] > 3                             # several rays own two consecutive vehicle packs
    for k in ("mask_volume", "depth_volume", "rgb_volume", "normals_volume", "rgb_volume_occupied", "rgb_sky",
              "rgb_volume_non_occupied"):
        _cmp(got["rendered"][k], ref["rendered"][k], tol, k)
    _cmp(got["volume_buffer"]["pack_infos_hit"], ref["volume_buffer"]["pack_infos_hit"], 0, "pack_infos_hit")
    for k in ("t", "opacity_alpha", "rgb", "vw"):
        _cmp(got["volume_buffer"][k], ref["volume_buffer"][k], btol or tol, f"volume_buffer.{k}")
    for key in ("main", "Vehicle"):
        _cmp(got["vw_in_total"][key], ref["vw_in_total"][key], btol or tol, f"vw_in_total.{key}")
    # every class's / object's share of the joint rendering (reference :729-806, :821-823)
    assert set(ref["per_class"]) == set(got["per_class"]) == {"Main", "Vehicle", "Sky"}
    assert set(ref["per_obj"]) <= set(got["per_obj"]) and {"main", "car0", "car1", "car2"} <= set(ref["per_obj"])
    for grp in ("per_class", "per_obj"):
        for name, imgs in ref[grp].items():
            for k, v in imgs.items():
                _cmp(got[grp][name][k], v, btol or tol, f"{grp}.{name}.{k}")
    assert float(ref["per_obj"]["car2"]["mask_volume"].max()) > 0.05 and float(ref["per_obj"]["car1"]["mask_volume"].max()) == 0.0
    assert set(got["grads"]) >= set(ref["grads"])
    for k, gr in ref["grads"].items():
        mine = got["grads"][k]
        norm = gr["norm"] if isinstance(gr, dict) else float(gr.norm())
        if norm < 1e-6:
            continue
        if isinstance(gr, dict):
            assert abs(float(mine.norm()) - norm) <= gtol * norm, k
            mine, gr = mine[::gr["stride"]], gr["sample"]
        e = float((mine - gr).norm()) / (float(gr.norm()) + 1e-12)
        assert e < gtol, (k, e)


def test_compose_mirror_matches_reference_fixture(backend):
    """The compose mirror against the frozen outputs of the reference's BufferComposeRenderer (``-m gpu`` on the box)."""
    import compose_scenario as cs
    fx = torch.load(GOLDEN)["compose"]
    got = cs.run_mirror(cs.build(backend))
    # per-sample alphas amplify the f32 rounding differences between the device and the emulator by inv_s (~90 here)
    _compare_compose(got, fx, 1e-3, 5e-3, btol=4e-3)


@needs_reference
def test_reference_eikonal_loss_on_the_mirror(backend):
    """The reference's ``EikonalLoss`` (app/loss/eikonal.py:23-251, loaded unchanged; dtu config:
    ``on_uniform_samples``, ``on_occ_ratio 1``, ``on_render_type both``, ``on_render_ratio .1``) driven by the mirror's
    return value: it looks up ``raw_per_obj_model[*]['class_name' | 'model_id' | 'volume_buffer']``, calls
    ``model.sample_pts_in_occupied`` and reads ``nablas``, ``vw_in_total``, ``pack_infos_collect``.  With the noise off
    the three terms equal the plain formulas, and the gradient reaches the table through the second-order path."""
    from oracle import pack_ops as opo
    from neuralsim_amd.renderers.single_volume_renderer import SingleVolumeRenderer
    sc = build_scenario("main_train", backend)
    model = sc["model"]
    model.id = "LoTDNeuSObj#Main"
    r = SingleVolumeRenderer(sc["common"]).train()
    ret = r.ray_query(sc["rays_o"], sc["rays_d"], model=model, rays_h_appear=sc["h_appear"], return_buffer=True,
                      return_details=True)
    raw = ret["raw_per_obj_model"]["main"]
    assert raw["class_name"] == "Main" and raw["model_id"] == model.id
    uni = model.sample_pts_uniform(64, generator=torch.Generator(device=backend).manual_seed(1))

    class Bank(dict):
        pass
    scene = type("S", (), {})()
    scene.asset_bank = Bank({model.id: model})
    with ref_glue.reference_eikonal_loss_module() as eik:
        loss_mod = eik.EikonalLoss(1.0e-3, ["Main"], on_uniform_samples=True, on_occ_ratio=1.0, on_render_type="both",
                                   on_render_ratio=0.1, with_noise=0.0, safe_mse=False)
        torch.manual_seed(0)
        losses = loss_mod(scene, ret, {"Main": uni}, None, None, it=0, mode="pixel")
    assert set(losses) == {"loss_eikonal.Main.uniform", "loss_eikonal.Main.occ", "loss_eikonal.Main.render"}
    vb = raw["volume_buffer"]
    fn = lambda n: (n.norm(dim=-1) - 1.0) ** 2          # noqa: E731
    nab = vb["nablas"].detach().cpu().double()
    want_render = fn(nab).mean() + opo.packed_sum(fn(nab) * vb["vw_in_total"].detach().cpu().double(),
                                                  vb["pack_infos_collect"].cpu()).mean()
    assert abs(float(losses["loss_eikonal.Main.render"]) - 1e-3 * 0.1 * float(want_render)) < 1e-7
    assert abs(float(losses["loss_eikonal.Main.uniform"]) - 1e-3 * float(fn(uni["nablas"].detach().cpu().double()).mean())) < 1e-7
    assert float(losses["loss_eikonal.Main.occ"]) > 0
    model.encoding.flattened_params.grad = None
    sum(losses.values()).backward()
    assert float(model.encoding.flattened_params.grad.abs().sum()) > 0


@needs_reference
def test_reference_render_chunked_through_the_shim_batchify_query(backend):
    """The reference's top-level ``SingleVolumeRenderer.render`` (:495-581) in eval mode with a ``rayschunk`` smaller
    than the batch: it splits the rays with ``nr3d_lib.models.utils.batchify_query`` -- the SHIM's implementation --
    and concatenates the nested result dicts; compared with one un-chunked query and with the mirror's chunked
    ``render`` ([H, W] prefix shape restored)."""
    from neuralsim_amd.renderers.single_volume_renderer import SingleVolumeRenderer
    sc = build_scenario("main_distant_eval", backend)
    H, W = 4, 9
    o, d = sc["rays_o"].view(H, W, 3), sc["rays_d"].view(H, W, 3)
    ts = torch.zeros(H, W, device=backend)
    with ref_glue.reference_renderer_modules() as mods:
        scene = ref_glue.FakeScene(backend, image_embeddings=ref_glue.FixedEmbeddings(None),
                                   convert_rays_in_node=mods.get("convert_rays_in_node"))
        # per-chunk appearance rows: the embedding is looked up from the chunk's rays_ts
        scene.image_embeddings = type("E", (), {"__getitem__": lambda self, k: (
            lambda rays_ts, mode="interp": sc["h_appear"][:1].expand(rays_ts.shape[0], -1))})()
        scene.add(ref_glue.FakeNode(sc["model"], "Main", "main"))
        scene.add(ref_glue.FakeNode(sc["distant_model"], "Distant", "distant"))
        r = ref_glue.make_reference_renderer(mods, sc["common"], val=dict(rayschunk=10), training=False)
        cam = mods["classes"]["Camera"]("cam0")
        cam.near = cam.far = None
        scene.id, scene.observers = "scene0", {"cam0": cam}
        scene.asset_bank = type("Bank", (), {"rendering_before_per_view": lambda self, **kw: [
            m.rendering_before_per_view(kw["renderer"], kw["observer"]) for m in (sc["model"], )]})()
        chunked = r.render(scene, rays=[o, d, ts], observer=cam, rayschunk=10)
        whole = r.render(scene, rays=[o, d, ts], observer=cam, rayschunk=0)
    for k in ("rgb_volume", "depth_volume", "mask_volume", "normals_volume"):
        assert chunked["rendered"][k].shape[:2] == (H, W)
        _cmp(chunked["rendered"][k].cpu(), whole["rendered"][k].cpu(), 1e-6, k)
    _cmp(chunked["ray_intersections"]["samples_cnt"].cpu(), whole["ray_intersections"]["samples_cnt"].cpu(), 0, "samples_cnt")
    mine = SingleVolumeRenderer(dict(sc["common"], rayschunk=10)).eval()
    ha = sc["h_appear"][:1].expand(H * W, -1).contiguous().view(H, W, -1)
    got = mine.render(sc["model"], rays=[o, d], rays_h_appear=ha, distant_model=sc["distant_model"])
    for k in ("rgb_volume", "depth_volume", "mask_volume", "normals_volume"):
        _cmp(got["rendered"][k].cpu(), chunked["rendered"][k].cpu(), 2e-5, f"mirror.{k}")
    del cam


@needs_reference
def test_reference_renderer_per_object_renderings(backend):
    """``render_per_obj_individual`` / ``render_per_obj_in_scene`` of the reference's SingleVolumeRenderer (:250-256,
    :313-320, :412-442) on the mirror's models: every model returns its own all-rays ``rendered`` dict, and the
    in-scene shares of the close-range and the distant model add up to the joint image."""
    sc = build_scenario("main_distant_sky_train", backend)
    with ref_glue.reference_renderer_modules() as mods:
        scene = ref_glue.FakeScene(backend, image_embeddings=ref_glue.FixedEmbeddings(sc["h_appear"]),
                                   convert_rays_in_node=mods.get("convert_rays_in_node"))
        scene.add(ref_glue.FakeNode(sc["model"], "Main", "main"))
        scene.add(ref_glue.FakeNode(sc["distant_model"], "Distant", "distant"))
        scene.add(ref_glue.FakeNode(sc["sky_model"], "Sky", "sky"))
        r = ref_glue.make_reference_renderer(mods, dict(sc["common"], depth_use_normalized_vw=False), training=True)
        cam = mods["classes"]["Camera"]("cam0")
        ret = r.ray_query(sc["rays_o"], sc["rays_d"], rays_ts=torch.zeros(sc["N"], device=backend), scene=scene,
                          observer=cam, return_buffer=True, return_details=True, render_per_obj_individual=True,
                          render_per_obj_in_scene=True)
    N = sc["N"]
    ind, ins = ret["rendered_per_obj"], ret["rendered_per_obj_in_scene"]
    assert set(ind) == set(ins) == {"main", "distant"}
    for oid in ind:
        assert ind[oid]["mask_volume"].shape == (N,) and ind[oid]["rgb_volume"].shape == (N, 3)
    assert ind["main"]["normals_volume_in_world"].shape == (N, 3)
    hit = ret["raw_per_obj_model"]["main"]["volume_buffer"]["rays_inds_hit"]
    miss = torch.ones(N, dtype=torch.bool, device=backend)
    miss[hit] = False
    assert float(ind["main"]["mask_volume"][miss].abs().sum()) == 0.0 and float(ind["main"]["mask_volume"][hit].min()) > 0
    assert float(ind["distant"]["mask_volume"].min()) > 0.99                  # include_inf_distance: opaque on its own
    tot = ret["rendered"]
    for k in ("mask_volume", "depth_volume", "rgb_volume_occupied"):
        kk = "rgb_volume" if k == "rgb_volume_occupied" else k
        _cmp((ins["main"][kk] + ins["distant"][kk]).detach().cpu(), tot[k].detach().cpu(), 2e-5, f"in-scene sum {k}")


@needs_reference
def test_reference_compose_renderer_individual_renderings_and_segmentation(backend):
    """``render_per_obj_individual`` of the reference's BufferComposeRenderer (:268-311, :548-577): the batched model
    hands back one all-rays image per batch item, the single model its own; the reference builds its instance / class
    segmentation z-buffer from them (two vehicles overlap on the central pixels: the nearer one wins)."""
    import compose_scenario as cs
    sc = cs.build(backend)
    with ref_glue.reference_compose_renderer_modules() as mods:
        from nr3d_lib.config import ConfigDict
        AA = mods["AssetAssignment"]
        main, mb, sky = sc["main"], sc["vehicle"], sc["sky"]
        main.assigned_to, main.is_ray_query_supported, main.is_batched_query_supported = AA.OBJECT, True, False
        mb.assigned_to, mb.is_ray_query_supported = AA.MULTI_OBJ, True
        sky.assigned_to, sky.is_ray_query_supported = AA.SCENE, False
        scene = ref_glue.FakeComposeScene(backend, mods["Scene"], image_embeddings=ref_glue.FixedEmbeddings(sc["h_appear"]))
        scene.add(ref_glue.FakeNode(main, "Main", "main", ref_glue.FakeTransform(device=backend)))
        for k, (R, t, s) in sc["poses"].items():
            scene.add(ref_glue.FakeNode(mb, "Vehicle", k, ref_glue.FakeTransform(R, t, s, device=backend)))
        scene.add(ref_glue.FakeNode(sky, "Sky", "sky", ref_glue.FakeTransform(device=backend)))
        rr = mods["compose"].BufferComposeRenderer(ConfigDict(common=ConfigDict(dict(sc["common"], segmentation_threshold=0.002)),
                                                              train=ConfigDict(), val=ConfigDict()))
        rr.image_postprocessor = None
        rr.eval()
        obs = type("Camera", (mods["classes"]["Camera"], ref_glue.FakeObserver), {})("cam0")
        with torch.no_grad():
            ret = rr.ray_query(sc["rays_o"], sc["rays_d"], rays_ts=torch.zeros(sc["N"], device=backend), scene=scene,
                               observer=obs, render_per_obj_individual=True)
    N = sc["N"]
    per = ret["rendered_per_obj"]
    assert set(per) == {"main", "car0", "car1", "car2"}
    for oid, r in per.items():
        assert r["mask_volume"].shape == (N,) and r["rgb_volume"].shape == (N, 3) and r["normals_volume_in_world"].shape == (N, 3)
    # (the 14 x 14 view is very wide: the far vehicle is grazed by the four central rays only)
    assert float(per["car1"]["mask_volume"].abs().sum()) == 0.0 and float(per["car0"]["mask_volume"].max()) > 0.002
    ins_map = scene.get_drawable_instance_ind_map()
    seg = ret["ins_seg_mask_buffer"].cpu()
    both = ((per["car0"]["mask_volume"] > 0.002) & (per["car2"]["mask_volume"] > 0.002)).cpu()
    assert int(both.sum()) > 0                                               # pixels where the two vehicles overlap
    nearer = torch.where(per["car2"]["depth_volume"] < per["car0"]["depth_volume"], ins_map["car2"], ins_map["car0"]).cpu()
    d_main = torch.where(per["main"]["mask_volume"] > 0.002, per["main"]["depth_volume"],
                         torch.full_like(per["main"]["depth_volume"], float("inf"))).cpu()
    d_veh = torch.minimum(per["car2"]["depth_volume"], per["car0"]["depth_volume"]).cpu()
    sel = both & (d_veh < d_main)                                            # ... and in front of the background object
    assert int(sel.sum()) > 0 and torch.equal(seg[sel], nearer[sel])
    assert set(ret["class_seg_mask_buffer"].unique().tolist()) <= {-1, 0, 1}


@needs_reference
def test_reference_mask_entropy_loss(backend):
    """``MaskEntropyRegLoss.forward_code_single`` (app/loss/mask_entropy.py:45-160; dtu config ``mask_entropy_mode:
    crisp_cr``) on the return value of the reference renderer running on the mirror's models: the close-range /
    distant masks it rebuilds from ``vw_in_total`` + ``pack_infos_collect`` add up to the joint mask, and its value
    equals the formula on the oracle's pack ops."""
    from oracle import pack_ops as opo
    sc = build_scenario("main_distant_sky_train", backend)
    N = sc["N"]
    with ref_glue.reference_renderer_modules() as mods:
        scene = ref_glue.FakeScene(backend, image_embeddings=ref_glue.FixedEmbeddings(sc["h_appear"]),
                                   convert_rays_in_node=mods.get("convert_rays_in_node"))
        scene.add(ref_glue.FakeNode(sc["model"], "Main", "main"))
        scene.add(ref_glue.FakeNode(sc["distant_model"], "Distant", "distant"))
        scene.add(ref_glue.FakeNode(sc["sky_model"], "Sky", "sky"))
        r = ref_glue.make_reference_renderer(mods, sc["common"], training=True)
        ret = r.ray_query(sc["rays_o"], sc["rays_d"], rays_ts=torch.zeros(N, device=backend), scene=scene,
                          observer=mods["classes"]["Camera"]("cam0"), return_buffer=True, return_details=True)
    with ref_glue.reference_loss_module("mask_entropy") as me:
        crisp = me.MaskEntropyRegLoss(3.0e-3, mode="crisp_cr").forward_code_single(scene, ret, [N], it=0)
        cross = me.MaskEntropyRegLoss(1.0, mode="cross_crdv").forward_code_single(scene, ret, [N], it=0)
    raw = ret["raw_per_obj_model"]

    def mask_of(key):
        vb = raw[key]["volume_buffer"]
        m = torch.zeros(N, dtype=torch.float64)
        m[vb["rays_inds_collect"].cpu()] = opo.packed_sum(vb["vw_in_total"].detach().cpu().double().flatten(),
                                                          vb["pack_infos_collect"].cpu())
        return m
    m_cr, m_dv = mask_of("main"), mask_of("distant")
    assert float((m_cr + m_dv - ret["rendered"]["mask_volume"].detach().cpu().double()).abs().max()) < 2e-5
    eps = 1e-5
    want = -(m_cr * torch.log(m_cr.clamp_min(eps)) + (1 - m_cr) * torch.log((1 - m_cr).clamp_min(eps))).mean()
    assert abs(float(crisp["loss_mask_entropy.crisp_cr"]) - 3.0e-3 * float(want)) < 1e-7
    want1 = (m_cr * torch.log(m_dv.clamp_min(eps))).mean()
    assert abs(float(cross["loss_mask_entropy.cross_cr_on_dv"]) - float(want1)) < 1e-5
    sc["model"].encoding.flattened_params.grad = None
    (crisp["loss_mask_entropy.crisp_cr"] + cross["loss_mask_entropy.cross_dv_on_cr"]).backward()
    assert float(sc["model"].encoding.flattened_params.grad.abs().sum()) > 0


@needs_reference
def test_trainer_lidar_losses_equal_the_reference_modules(backend):
    """``RenderTrainer.lidar_losses`` (the street iteration's lidar step) against the reference's own loss code on the
    same render: ``LineOfSightLoss(fn_type='neus_unisim')`` (app/loss/lidar.py:174-206) and the masked l1 depth term
    (``l1_loss(pred, gt, mask, reduction='mean')``, :41 / :271) with the config's weights and ``discard_toofar``."""
    from neuralsim_amd.renderers.single_volume_renderer import SingleVolumeRenderer
    from neuralsim_amd.trainer import RenderTrainer
    from nr3d_lib.models.loss.recon import l1_loss
    sc = build_scenario("main_train", backend)
    r = SingleVolumeRenderer(dict(with_rgb=False, with_normal=False, near=0.01, depth_use_normalized_vw=False, perturb=False)).train()
    ret = r.ray_query(sc["rays_o"], sc["rays_d"], model=sc["model"], return_buffer=True)
    N = sc["N"]
    g = torch.Generator().manual_seed(3)
    ranges = (ret["rendered"]["depth_volume"].detach().cpu() + torch.randn(N, generator=g) * 0.2).clamp_min(0.05)
    ranges[::5] = 0.0                                   # beams without a return
    ranges[1::7] = 3.0                                  # ... and beyond discard_toofar
    ranges = ranges.to(backend)
    tr = RenderTrainer.__new__(RenderTrainer)
    tr.lidar = dict(w_depth=0.02, w_los=0.1, epsilon=0.15, discard_toofar=2.5)
    loss, parts = tr.lidar_losses(ret, ranges)
    mask = (ranges > 0) & (ranges < 2.5)
    with ref_glue.reference_lidar_loss_module() as lidar:
        los = lidar.LineOfSightLoss(w=0.1, fn_type="neus_unisim", fn_param=dict(epsilon=0.15))
        ref_los = los(None, ret, None, dict(ranges=ranges), it=0, mask=mask)["lidar_loss.los.empty"]
    ref_depth = 0.02 * l1_loss(ret["rendered"]["depth_volume"], ranges, mask, reduction="mean")
    assert abs(float(parts["los"]) - float(ref_los)) <= 1e-6 * (1 + abs(float(ref_los)))
    assert abs(float(parts["depth"]) - float(ref_depth)) <= 1e-6 * (1 + abs(float(ref_depth)))
    assert float(ref_los) > 0 and float(ref_depth) > 0
