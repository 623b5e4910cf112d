"""The reference's OWN model wrappers and model-side losses, loaded unchanged from /root/reference
(tests/ref_glue.py::reference_model_wrapper_modules), on this package's models -- the life cycle
``AssetBank.create_asset_bank(do_training_setup=True)`` and the trainer drive on every model
(app/resources/asset_bank.py:129-151, 269-321; code_single/tools/train.py:1393, 1449, 1494-1502):

  ``import_str('app.models.single.LoTDNeuSObj')(**model_params, device=)`` -> ``asset_init_config(**asset_params)`` ->
  ``asset_populate`` -> ``training_setup(training_cfg)`` / ``.optimizer`` -> ``asset_training_initialize`` ->
  ``training_update_lr(it)`` -> render -> ``ClearanceLoss`` (``details['near_sdf']``) + ``WeightRegLoss``
  (``get_weight_reg``) -> ``GradScaler.scale(loss).backward(); unscale_(optimizer); training_clip_grad();
  scaler.step(optimizer)`` -> ``stat_param(with_grad=True)``; ``space.get_bounding_volume()`` as SceneNode.update reads it.

YAML blocks come from the reference's lotd_neus.dtu.230814.yaml (the table / occupancy sizes are shrunk for the emulator).
Authoring container only (needs /root/reference)."""
from pathlib import Path

import pytest
import torch

import ref_glue

CFG = Path("/root/reference/code_single/configs")
needs_reference = pytest.mark.skipif(not (CFG.exists() and ref_glue.reference_available()),
                                     reason="executes the reference's own sources from /root/reference (authoring container only; "
                                            "emulator backend): the model wrappers / loss modules of app/ cannot travel to the GPU box")


def _cfg():
    from nr3d_lib.config import load_config
    return load_config(str(CFG / "object_centric/lotd_neus.dtu.230814.yaml"))


def _small_main(c):
    mp = c.assetbank_cfg.Main.model_params.to_dict()
    mp["surface_cfg"]["encoding_cfg"]["lotd_auto_compute_cfg"].update(num_levels=8, log2_hashmap_size=12, max_res=64)
    mp["accel_cfg"].update(resolution=[16, 16, 16], init_cfg=dict(num_steps=2, num_pts=2 ** 12),
                           update_from_net_cfg=dict(num_steps=2, num_pts=2 ** 12), n_steps_warmup=2, n_steps_between_update=2)
    mp["ray_query_cfg"]["query_param"].update(num_coarse=8, num_fine=[4, 4], upsample_inv_s_factors=[1, 4],
                                              march_cfg=dict(step_size=0.05, max_steps=128))
    return mp


def _small_distant(c):
    dp = c.assetbank_cfg.Distant.model_params.to_dict()
    dp["encoding_cfg"]["lotd_auto_compute_cfg"].update(target_num_params=2 ** 14, log2_hashmap_size=10, min_res_xyz=3,
                                                       min_res_w=2)
    dp["ray_query_cfg"]["query_param"]["march_cfg"]["max_steps"] = 8
    return dp


class _Node(ref_glue._Named):
    class_name = "Main"
    model = None


class _Scene(ref_glue._Named):
    def __init__(self, nodes):
        super().__init__("scene0")
        self.all_nodes_by_class_name = {}
        for n in nodes:
            self.all_nodes_by_class_name.setdefault(n.class_name, []).append(n)
        self.all_nodes = {n.id: n for n in nodes}
        self.asset_bank = {}


@needs_reference
def test_reference_wrappers_life_cycle_and_model_side_losses(backend):
    import importlib
    c = _cfg()
    dev = backend
    with ref_glue.reference_model_wrapper_modules() as mods:
        single = importlib.import_module("app.models.single")           # what import_str(cfg.model_class) resolves
        AssetAssignment = mods["app.models.asset_base"].AssetAssignment
        assert c.assetbank_cfg.Main.model_class == "app.models.single.LoTDNeuSObj"
        Main, Distant = single.LoTDNeuSObj, single.LoTDNeRFDistant
        main_node, dv_node = _Node("obj0"), _Node("distant0")
        dv_node.class_name = "Distant"
        scene = _Scene([main_node, dv_node])
        # ---- asset_bank.py:129-151 for the close-range model
        model = Main(**_small_main(c), device=dev)
        assert model.assigned_to == AssetAssignment.OBJECT and model.is_ray_query_supported
        model.asset_init_config(**c.assetbank_cfg.Main.asset_params.to_dict())
        model.asset_populate(scene=scene, obj=main_node, config=model.populate_cfg, device=dev)
        model.to(dev)
        model.id = Main.asset_compute_id(scene=scene, obj=main_node, class_name="Main")
        assert model.id == "LoTDNeuSObj#Main#scene0#obj0"
        main_node.model = model
        model.training_setup(model.training_cfg)
        opt = model.optimizer
        assert isinstance(opt, torch.optim.Optimizer)
        names = [g["name"] for g in opt.param_groups]
        assert names == ["implicit_surface.encoding", "implicit_surface.decoder", "radiance_net", "ln_inv_s"]
        assert all(g["lr"] == c.fglr and g["eps"] == 1e-15 for g in opt.param_groups)
        assert opt.param_groups[0]["betas"] == (0.9, 0.99) and opt.param_groups[3]["betas"] == (0.9, 0.999)   # invs_betas
        # ---- ... and for the distant model, which takes the close-range box through model.space.aabb (nerf.py:170-177)
        dm = Distant(**_small_distant(c), device=dev)
        dm.asset_init_config(**c.assetbank_cfg.Distant.asset_params.to_dict())
        dm.asset_populate(scene=scene, obj=dv_node, config=dm.populate_cfg, device=dev)
        assert dm.cr_obj is main_node and torch.equal(dm.aabb.cpu(), model.space.aabb.cpu())
        assert dm.include_inf_distance is True
        dm.training_setup(dm.training_cfg)
        assert [g["name"] for g in dm.optimizer.param_groups] == ["encoding", "density_decoder", "radiance_decoder"]
        assert dm.optimizer.param_groups[0]["lr"] == c.bglr
        # nodes.py:92-103: bounding volume = centre + radius3d of the model's space
        bv = model.space.get_bounding_volume()
        assert bv.shape == (6,) and torch.allclose(bv.cpu(), torch.tensor([0.0, 0, 0, 1, 1, 1]))
        # ---- initialisation + schedules
        assert model.asset_training_initialize(scene, main_node, model.initialize_cfg) is True
        assert dm.asset_training_initialize(scene, dv_node, dm.initialize_cfg) is False
        sch = c.training.scheduler
        assert sch.type == "exponential" and sch.warmup_steps == 1000 and sch.min_factor == 0.06
        for m_ in (model, dm):
            m_.training_update_lr(0)
        assert abs(opt.param_groups[0]["lr"] - c.fglr * 1e-3) < 1e-12               # first warm-up step: 1 / 1000
        model.training_update_lr(999)
        assert abs(opt.param_groups[1]["lr"] - c.fglr * 0.06 ** (999 / 7500)) < 1e-9
        model.training_update_lr(7500)
        assert abs(opt.param_groups[0]["lr"] - c.fglr * 0.06) < 1e-9                # min_factor at num_iters
        model.training_update_lr(10)
        # ---- one training step as train.py:1449-1502 runs it (pixel branch), losses = the reference's modules
        model.training_before_per_step(10)
        from neuralsim_amd.graphics.cameras import look_at_cameras, pinhole_selected_rays
        from neuralsim_amd.renderers.single_volume_renderer import SingleVolumeRenderer
        intr, c2w, WH = look_at_cameras(V=3, seed=2, device=dev)
        g = torch.Generator().manual_seed(0)
        xy, fidx = torch.rand(48, 2, generator=g).to(dev), torch.randint(0, 3, (48,), generator=g).to(dev)
        rays_o, rays_d = pinhole_selected_rays(xy, fidx, intr, c2w, WH)
        renderer = SingleVolumeRenderer(dict(with_rgb=True, with_normal=True, near=0.01, perturb=True,
                                             depth_use_normalized_vw=False, with_near_sdf=True)).train()     # train.py:245
        ret = renderer.render(model, rays=[rays_o, rays_d], rays_h_appear=torch.zeros(48, 4, device=dev),
                              return_buffer=True, return_details=True)
        raw = ret["raw_per_obj_model"]["main"]
        assert raw["class_name"] == "Main" and raw["volume_buffer"]["type"] == "packed"
        near_sdf = raw["details"]["near_sdf"]
        # one value per ray that passed the model's ray_test (the buffer lists the [R'] of them whose march found something)
        assert near_sdf.shape == (48,) and near_sdf.requires_grad
        assert 0 < raw["volume_buffer"]["rays_inds_hit"].shape[0] <= 48
        assert float(near_sdf.min()) > 0.5               # the rays enter the box far outside the radius-0.5 sphere
        raw["model_id"] = model.id
        scene.asset_bank = {model.id: model}
        lc = c.training.losses
        clearance = mods["app.loss.clearance"].ClearanceLoss(lc.clearance.class_name_cfgs.to_dict(), ["Main"])
        wreg = mods["app.loss.weight_reg"].WeightRegLoss(dict(Main=lc.weight_reg.class_name_cfgs.Main.to_dict()), ["Main"])
        losses = {}
        losses.update(clearance(scene, ret, None, None, None, 10, mode="pixel"))
        losses.update(wreg(scene, ret, None, None, 10))
        assert set(losses) == {"loss_clearance.Main", "loss_weight_reg.Main"}
        wr = model.get_weight_reg(norm_type=2.0)
        assert abs(float(losses["loss_weight_reg.Main"]) - 1e-6 * float(wr.sum())) < 1e-12
        # thresh 0 (yaml :367): nothing is inside the geometry -> no clearance penalty; raise the threshold to exercise it
        assert float(losses["loss_clearance.Main"]) == 0.0
        pen = mods["app.loss.clearance"].ClearanceLoss(dict(Main=dict(w=0.2, beta=2.0, thresh=2.0)), ["Main"])
        losses.update({"loss_clearance2": pen(scene, ret, None, None, None, 10, mode="pixel")["loss_clearance.Main"]})
        losses["loss_rgb"] = (ret["rendered"]["rgb_volume"] ** 2).mean()
        total = sum(v for v in losses.values())
        for o in (opt, dm.optimizer):
            o.zero_grad()
        before = [p.detach().clone() for p in (model.encoding.flattened_params, model.sdf_w, model.rad_w)]
        scaler = torch.cuda.amp.GradScaler(init_scale=128.0, enabled=dev.type == "cuda")        # train.py:1410
        scaler.scale(total).backward()
        scaler.unscale_(opt)
        model.training_clip_grad()                       # no clip keys in this training_cfg: gradients unchanged
        assert model.sdf_w.grad is not None and float(model.sdf_w.grad.abs().max()) > 0
        gnorm = float(model.sdf_w.grad.norm())
        st = model.stat_param(with_grad=True, prefix="Main")
        assert abs(st["Main.sdf_w.grad"]["norm"] - gnorm) < 1e-4 * gnorm and "Main.encoding.flattened_params.data" in st
        scaler.step(opt)
        scaler.update()
        after = [model.encoding.flattened_params, model.sdf_w, model.rad_w]
        assert all(float((a.detach() - b).abs().max()) > 0 for a, b in zip(after, before))
        assert float((model.sdf_w.detach() - before[1]).abs().max()) <= opt.param_groups[1]["lr"] * 1.002       # |first Adam step| == lr (f32 rounding next to weights ~1)
        assert torch.equal(model.encoding.shadow().float(), model.encoding.flattened_params.detach().half().float())
        model.training_after_per_step(10)
        # clipping keys, when a config has them
        model.training_cfg = dict(model.training_cfg, clip_grad_norm=1e-6)
        model.training_clip_grad()
        tot = torch.sqrt(sum((p.grad.float() ** 2).sum() for p in model.parameters() if p.grad is not None))
        assert float(tot) <= 1.01e-6
        # with_feature_dim: 0 is what the renderers pass (single_volume_renderer.py:239-242); anything else is refused
        tested = model.ray_test(rays_o, rays_d, near=0.01)
        with pytest.raises(NotImplementedError):
            model.ray_query(ray_tested=tested, config=dict(model.ray_query_cfg, with_feature_dim=3))


def test_fused_adam_torch_optimizer_matches_torch_adam(backend):
    """``FusedAdamTorch`` (what ``training_setup`` builds) == ``torch.optim.Adam`` over a few steps, per-group betas."""
    from neuralsim_amd.model_base import FusedAdamTorch, lr_factor
    g = torch.Generator().manual_seed(3)
    p0 = [torch.randn(1000, generator=g) * 0.1, torch.randn(37, generator=g) * 0.1]
    ps = [torch.nn.Parameter(t.clone().to(backend)) for t in p0]
    qs = [torch.nn.Parameter(t.clone()) for t in p0]
    opt = FusedAdamTorch([dict(name="a", params=[ps[0]], lr=1e-2), dict(name="b", params=[ps[1]], lr=3e-3, betas=(0.9, 0.999))],
                         betas=(0.9, 0.99), eps=1e-15)
    ref = torch.optim.Adam([dict(params=[qs[0]], lr=1e-2, betas=(0.9, 0.99)), dict(params=[qs[1]], lr=3e-3, betas=(0.9, 0.999))],
                           eps=1e-15)
    for it in range(4):
        for p, q in zip(ps, qs):
            gr = torch.randn(q.shape, generator=g) * 1e-3
            p.grad, q.grad = gr.clone().to(backend), gr.clone()
        opt.step()
        ref.step()
        for p, q in zip(ps, qs):
            assert float((p.detach().cpu() - q.detach()).abs().max()) < 1e-6 * (1e-2 + float(q.abs().max()))
    assert lr_factor(0, None) == 1.0 and abs(lr_factor(50, dict(type="warmup_cosine", num_iters=100, min_factor=0.1,
                                                                warmup_steps=0)) - 0.55) < 1e-9
    assert lr_factor(25, dict(type="multistep", milestones=[10, 20, 30], gamma=0.5)) == 0.25


@needs_reference
def test_reference_monocular_losses_on_an_image_patch(backend):
    """BASELINE configs[2] (indoor, lotd_neus.replica.230814.yaml: ``inside_out``, image-patch step, ``mono_depth`` +
    ``mono_normals``): the reference's MonoDepthLoss (scale-and-shift invariant, with its multi-scale gradient term) and
    MonoNormalLoss (L1 + cosine), loaded unchanged, on a 16 x 16 patch rendered WITH gradient by the mirror renderer --
    depth and normals images carry gradients back to the table and both decoders."""
    from neuralsim_amd.fields.neus import LoTDNeuSModel
    from neuralsim_amd.renderers.single_volume_renderer import SingleVolumeRenderer
    from util import SMALL_RES
    dev = backend
    qp = dict(nablas_has_grad=True, num_coarse=8, num_fine=[4, 4], upsample_inv_s=64.0, upsample_inv_s_factors=[1, 4],
              upsample_use_estimate_alpha=True, march_cfg=dict(step_size=0.05, max_steps=128))
    m = LoTDNeuSModel(lod_res=SMALL_RES, log2_hashmap_size=10, sdf_D=2, precision="f32", ln_inv_s_init=0.4, inside_out=True,
                      accel_cfg=dict(resolution=(16, 16, 16), update_from_net_cfg=dict(num_steps=1, num_pts=2048)),
                      ray_query_cfg=dict(query_mode="march_occ_multi_upsample", query_param=qp), seed=6).to(dev)
    m.geometric_init_sphere(0.8)
    m.accel.init(m.query_sdf, num_steps=4, num_pts=2 ** 15)     # (32 queries per voxel: the wall is occupied wherever a ray ends)
    g = torch.Generator().manual_seed(3)
    H = W = 16
    d = torch.nn.functional.normalize(torch.randn(H, W, 3, generator=g) * 0.2 + torch.tensor([0.0, 0.0, 1.0]), dim=-1).to(dev)
    o = torch.zeros(H, W, 3, device=dev)                                   # a camera at the centre of the room
    rend = SingleVolumeRenderer(dict(with_rgb=True, with_normal=True, near=0.01, depth_use_normalized_vw=True,
                                     perturb=False)).train()
    ret = rend.render(m, rays=[o, d], return_buffer=True, return_details=True)
    r = ret["rendered"]
    assert r["depth_volume"].shape == (H, W) and r["normals_volume"].shape == (H, W, 3) and r["depth_volume"].requires_grad
    assert float(r["mask_volume"].detach().mean()) > 0.9                  # every ray ends on the wall of the room

    class _Rot:                                                            # cam.world_transform.detach().rotate(v, inv=True)
        def detach(self):
            return self

        def rotate(self, v, inv=False):
            return v
    cam = ref_glue._Named("cam0")
    cam.world_transform = _Rot()
    scene = ref_glue._Named("scene0")
    scene.device = dev
    depth_gt = (r["depth_volume"].detach() * 0.02 + 0.3) + 0.002 * torch.randn(H, W, generator=g).to(dev)   # up to scale / shift
    hit = o + r["depth_volume"].detach()[..., None] * d
    normal_gt = -torch.nn.functional.normalize(hit, dim=-1)               # the room's wall faces the camera
    gtruth = dict(image_mono_depth=depth_gt, image_mono_normals=normal_gt)
    with ref_glue.reference_mono_loss_module() as mono:
        depth_loss = mono.MonoDepthLoss(w=0.1, loss_type="monosdf", loss_param=dict(fn_type="mse", gt_pre_scale=50.0, gt_pre_shift=1.0,
                                                                                     alpha_grad_reg=0.01, grad_reg_scales=3),
                                        ignore_mask_list=[], mask_pred_thresh=0.5, mask_erode=0, enable_after=0)
        normal_loss = mono.MonoNormalLoss(w_l1=0.05, w_cos=0.05, ignore_mask_list=[], apply_in_pixel_train_step=True)
        losses = {}
        losses.update(depth_loss(scene, ret, None, gtruth, it=10, mode="image_patch"))
        losses.update(normal_loss(scene, cam, ret, None, gtruth, it=10, mode="image_patch"))
    assert set(losses) == {"loss_mono_depth.depth", "loss_mono_depth.reg", "loss_mono_normal.l1", "loss_mono_normal.cos"}
    assert all(bool(torch.isfinite(v)) for v in losses.values())
    assert float(losses["loss_mono_normal.cos"]) < 0.05 * 0.5             # rendered normals roughly face the camera already
    sum(losses.values()).backward()
    for p in (m.encoding.flattened_params, m.sdf_w, m.sdf_b):
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()) and float(p.grad.abs().sum()) > 0


# ================================================================================================ shared (code_multi) models
MULTI = Path("/root/reference/code_multi/configs/exps")


def _obj_nodes(mods, n):
    """n vehicle nodes that ARE the ``SceneNode`` the reference's classes imported (asset_populate asserts isinstance)."""
    class _ObjNode(mods["classes"]["SceneNode"]):
        class_name = "Vehicle"

        def __init__(self, id):
            super().__init__(id)
            self.full_unique_id = f"scene0#{id}"
            self.i_valid = True
    return [_ObjNode(f"car{i}") for i in range(n)]


def _multi_cfg(name):
    from nr3d_lib.config import load_config
    return load_config(str(MULTI / name))


def _shared_model_step(model, dev, ids, with_latents=True):
    """asset_bank.py:149-151 + one batched step as buffer_compose_renderer.py:222-265 drives the model: set_condition on the
    compacted batch, batched_ray_test, batched_ray_query, clean_condition; loss -> backward -> optimizer step."""
    from neuralsim_amd.graphics.cameras import look_at_cameras, pinhole_selected_rays
    model.training_setup(model.training_cfg)
    opt = model.optimizer
    names = [g["name"] for g in opt.param_groups]
    assert names[0] == "latents" and opt.param_groups[0]["params"][0] is model.z_ins_all.weight, names
    intr, c2w, WH = look_at_cameras(V=2, seed=2, device=dev, radius=2.5)
    g = torch.Generator().manual_seed(0)
    Bq, N = 2, 40
    xy, fidx = torch.rand(Bq * N, 2, generator=g).to(dev) * 0.4 + 0.3, torch.randint(0, 2, (Bq * N,), generator=g).to(dev)
    ro, rd = pinhole_selected_rays(xy, fidx, intr, c2w, WH)
    ro, rd = ro.view(Bq, N, 3), rd.view(Bq, N, 3)
    bt = model.batched_ray_test(ro, rd, near=0.01, far=None, compact_batch=True)
    assert bt["num_rays"] > 0 and set(bt) >= {"rays_inds", "rays_bidx", "rays_full_bidx", "full_bidx_map", "near", "far"}
    hit_ids = [ids[i] for i in [1, 0]]
    model.set_condition({"ins_id": [hit_ids[int(i)] for i in bt["full_bidx_map"].tolist()]})
    assert model.z_ins_per_batch.requires_grad and model.ins_inds_per_batch.shape[0] == bt["full_bidx_map"].shape[0]
    from nr3d_lib.config import ConfigDict
    cfg = ConfigDict(**model.ray_query_cfg, with_rgb=True, with_normal=True, perturb=False, depth_use_normalized_vw=False)
    ret = model.batched_ray_query(batched_ray_tested=bt, batched_ray_input=dict(rays_o=ro, rays_d=rd), config=cfg,
                                  return_buffer=True, return_details=False, render_per_obj_individual=True)
    model.clean_condition()
    vb = ret["volume_buffer"]
    assert vb["type"] == "packed" and set(vb) >= {"rays_inds_hit", "pack_infos_hit", "t", "opacity_alpha", "rgb", "nablas",
                                                  "rays_bidx_hit"}
    assert ret["rendered"]["rgb_volume"].shape == (Bq, N, 3)
    loss = (ret["rendered"]["rgb_volume"] ** 2).mean() + 0.1 * ((vb["nablas"].norm(dim=-1) - 1.0) ** 2).mean()
    opt.zero_grad()
    z_before = model.z_ins_all.weight.detach().clone()
    loss.backward()
    gz = model.z_ins_all.weight.grad
    assert gz is not None and bool(torch.isfinite(gz).all()) and float(gz.abs().max()) > 0      # the codes are LEARNED
    used = sorted({model._index_maps["ins_id"][i] for i in hit_ids})
    unused = [i for i in range(len(ids)) if i not in used]
    assert all(float(gz[i].abs().max()) == 0.0 for i in unused)
    opt.step()
    assert float((model.z_ins_all.weight.detach() - z_before).abs().max()) > 0
    sd = model.state_dict()
    assert any(k.startswith("_latents.z_ins") for k in sd)
    return ret


@needs_reference
def test_reference_ad_generative_permuto_model_from_its_config_block(backend):
    """BASELINE configs[4], the reference's CURRENT multi-object foreground: ``model_class:
    app.models.shared.AD_GenerativePermutoConcatNeuSObj`` built by ``import_str(model_class)(**model_params, device=)`` from
    code_multi/configs/exps/fg_neus=permuto/all_occ.240201.yaml:425-506 -- the reference's class (app/models/shared/
    batched_neus.py:295-407, loaded UNCHANGED) on ``nr3d_lib.models.fields_conditional.neus.GenerativePermutoConcatNeuSModel``
    and ``nr3d_lib.models.autodecoder.AutoDecoderMixin`` of this repository: asset_populate (autodecoder_populate + populate with
    the instance count), asset_training_initialize (pre-training + all-instance occupancy init), training_setup, one batched
    render step with gradients reaching the auto-decoder's codes.  Table / grid sizes shrunk for the emulator."""
    import importlib
    c = _multi_cfg("fg_neus=permuto/all_occ.240201.yaml")
    v = c.assetbank_cfg.Vehicle
    assert v.model_class == "app.models.shared.AD_GenerativePermutoConcatNeuSObj"
    mp = v.model_params.to_dict()
    mp["dtype"] = "float"
    mp["surface_cfg"]["encoding_cfg"]["permuto_auto_compute_cfg"].update(n_levels=6, log2_hashmap_size=11, finest_res=24.0, coarsest_res=2.0)
    mp["accel_cfg"].update(resolution=[8, 8, 8], init_cfg=dict(mode="from_net", num_steps=1, num_pts=2 ** 11),
                           update_from_net_cfg=dict(num_steps=1, num_pts=2 ** 11))
    mp["ray_query_cfg"]["query_param"].update(num_coarse=8, num_fine=8, march_cfg=dict(step_size=0.1, max_steps=64))
    ap = v.asset_params.to_dict()
    ap["initialize_cfg"].update(num_iters=30, lr=5e-3, num_points=1024, batch_size=3)
    with ref_glue.reference_model_wrapper_modules() as mods:
        shared = importlib.import_module("app.models.shared")
        Cls = shared.AD_GenerativePermutoConcatNeuSObj
        model = Cls(**mp, device=backend)
        assert model.assigned_to == mods["app.models.asset_base"].AssetAssignment.MULTI_OBJ and model.is_batched_query_supported
        assert model.latents_cfg["z_ins"]["dim"] == 4 and model.accel_cfg["type"] == "occ_grid_batched_ema"
        model.asset_init_config(**ap)
        nodes = _obj_nodes(mods, 3)
        ids = [n.full_unique_id for n in nodes]
        scene = _Scene(nodes)
        model.asset_populate(scene=[scene], obj=nodes, config=dict(model.populate_cfg, accel_use_avg_resolution=False),
                             device=backend)
        assert Cls.asset_compute_id(class_name="Vehicle") == "AD_GenerativePermutoConcatNeuSObj#Vehicle"
        assert model.num_objs == 3 and model.accel.num_batches == 3 and model.z_dim == 4
        assert tuple(model.z_ins_all.weight.shape) == (3, 4) and float(model.z_ins_all.weight.abs().max()) == 0.0   # weight_init: zero
        assert model.encoding.cfg.permuto.in_dim == 7 and model._index_maps["ins_id"][ids[2]] == 2
        assert model.asset_training_initialize(scene, nodes, model.initialize_cfg) is True and bool(model.is_pretrained)
        assert model.ins_inds_per_batch is None and 0.0 < model.accel.frac_occupied() < 1.0        # all-instance accel.init ran
        with torch.no_grad():
            model.z_ins_all.weight.add_(torch.randn(3, 4, generator=torch.Generator().manual_seed(1)).to(backend) * 0.05)
        _shared_model_step(model, backend, ids)


@needs_reference
def test_reference_ad_style_lotd_model_from_its_config_block(backend):
    """The LoTD generator of the older multi-object config: ``AD_StyleLoTDNeuSObj`` (app/models/shared/batched_neus.py:70-160) on
    ``StyleLoTDNeuSModel`` from the Vehicle block of fg_neus=hyper_lotd/no_fg_occ.221218.yaml:307-390 -- MixedLoTDGrower =
    DenseLoTDGrowerFMM + VMSplitLoTDGrowerFMM, relu decoder, occ_grid_batched.  That YAML predates the reference's code: it
    names ``AD_StyleLoTDNeuS`` (the class is ``AD_StyleLoTDNeuSObj``) and ``latents_cfg.z`` (the class reads
    ``latents_cfg['z_ins']``, :108) -- both renamed here.  ``surface_cfg.extra_pos_embed_cfg{sinusoidal_legacy, 6}`` stays in the
    block: the decoder reads [grown features | embedded position] (16 + 39 inputs at this test's four grown levels, 32 + 39 in
    the YAML) on csrc/wide_field.hip."""
    import importlib
    c = _multi_cfg("fg_neus=hyper_lotd/no_fg_occ.221218.yaml")
    v = c.assetbank_cfg.Vehicle
    assert v.model_class == "app.models.shared.AD_StyleLoTDNeuS"
    mp = v.model_params.to_dict()
    mp["latents_cfg"] = dict(z_ins=dict(dim=12, weight_init="zero"))
    gc = mp["surface_cfg"]["lotd_grower_cfg"]["param"]["grower_configs"]
    assert [g["target"].rsplit(".", 1)[-1] for g in gc] == ["DenseLoTDGrowerFMM", "VMSplitLoTDGrowerFMM"]
    gc[0]["param"].update(lod_res=[3, 5])
    gc[0]["param"]["pseudo_net_param"].update(D=2, W=16, fmm_rank=3)
    gc[0]["param"]["pseudo_net_param"]["embed_cfg"].update(n_frequencies=2)
    gc[1]["param"].update(lod_res=[6, 9])
    gc[1]["param"]["pseudo_net_param"].update(D=2, D_head=2, W=16, fmm_rank=3)
    gc[1]["param"]["pseudo_net_param"]["embed_cfg"].update(n_frequencies=2)
    mp["accel_cfg"].update(resolution=[8, 8, 8], num_steps=1, num_pts=2 ** 11)
    mp["ray_query_cfg"]["query_param"].update(num_coarse=8, num_fine=8, march_cfg=dict(step_size=0.1, max_steps=64))
    ap = dict(training_cfg=v.asset_params.training_cfg.to_dict(), initialize_cfg=dict(num_iters=40, lr=5e-3, num_pts=512))
    with ref_glue.reference_model_wrapper_modules() as mods:
        shared = importlib.import_module("app.models.shared")
        Cls = shared.AD_StyleLoTDNeuSObj
        assert mp["surface_cfg"]["extra_pos_embed_cfg"] == dict(type="sinusoidal_legacy", n_frequencies=6)
        model = Cls(**mp, device=backend)
        model.asset_init_config(**ap)
        nodes = _obj_nodes(mods, 3)
        ids = [n.full_unique_id for n in nodes]
        scene = _Scene(nodes)
        model.asset_populate(scene=[scene], obj=nodes, config=model.populate_cfg, device=backend)
        assert model.num_objs == 3 and model.accel.num_batches == 3 and model.grower.z_dim == 12
        assert model.encoding.cfg.lod_res == [3, 3, 5, 5, 6, 6, 9, 9] and model.sdf_activation == "relu"
        assert model.pos_embed_n == 6 and model.sdf_w.numel() == 64 * (16 + 39) + 4096 + 64
        assert model.asset_training_initialize(scene, nodes, model.initialize_cfg) is True
        assert model.ins_inds_per_batch is None and 0.0 < model.accel.frac_occupied() <= 1.0
        with torch.no_grad():
            model.z_ins_all.weight.add_(torch.randn(3, 12, generator=torch.Generator().manual_seed(1)).to(backend) * 0.05)
        _shared_model_step(model, backend, ids)
