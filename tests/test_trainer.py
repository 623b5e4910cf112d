"""The whole training step (ray gen .. Adam) on a tiny configuration: loss goes down, replicas stay in sync."""
import pytest
import torch

from neuralsim_amd.fields.neus import LoTDNeuSModel
from neuralsim_amd.graphics.cameras import look_at_cameras
from neuralsim_amd.trainer import RenderTrainer
from util import SMALL_RES, steps_on_a_fixed_objective


def _tiny(backend, seed=42):
    qp = dict(nablas_has_grad=True, num_coarse=8, num_fine=[4, 4], upsample_inv_s=64.0, upsample_inv_s_factors=[1, 4],
              upsample_use_estimate_alpha=True, march_cfg=dict(step_size=0.05, max_steps=128))
    m = LoTDNeuSModel(lod_res=SMALL_RES, log2_hashmap_size=10, sdf_D=2, precision="fp16", ln_inv_s_init=0.3,
                      accel_cfg=dict(resolution=(16, 16, 16), update_from_net_cfg=dict(num_steps=1, num_pts=2048),
                                     update_from_samples_cfg={}, n_steps_between_update=4, n_steps_warmup=2),
                      ray_query_cfg=dict(query_mode="march_occ_multi_upsample", query_param=qp), seed=seed).to(backend)
    m.geometric_init_sphere(0.5)
    m.accel.init(m.query_sdf, num_steps=1, num_pts=4096)
    return m


def test_train_steps_reduce_loss(backend):
    m = _tiny(backend)
    intr, c2w, WH = look_at_cameras(V=4, seed=1, device=backend)
    tr = RenderTrainer(m, intr, c2w, WH, num_rays=24, lr=2e-3, num_uniform=32, perturb=True)
    assert 0.0 < m.accel.frac_occupied() < 0.6
    xy, fidx, gt = tr.sample_batch()
    fixed = lambda: (xy, fidx, gt)
    tr.sample_batch = fixed                      # overfit one batch
    losses = steps_on_a_fixed_objective(tr, range(6))        # one batch, one set of jitter draws: a fixed objective
    assert all(l == l for l in losses)           # no NaN
    assert losses[-1] < losses[0], losses
    assert 0 < tr.stats["R_live"] <= tr.stats["R_hit"] and tr.stats["S_f"] >= tr.stats["R_live"] * 16
    assert tr.stats["S_q"] >= tr.stats["S_f"]


def test_train_steps_with_a_distorted_camera(backend):
    """``camera_model: opencv`` through the training step (street configs): rays, analytic targets and the prefetched
    batches all come from the distorted lift; the step trains as with a pinhole rig."""
    m = _tiny(backend)
    intr, c2w, WH = look_at_cameras(V=4, seed=1, device=backend)
    dist = torch.tensor([[0.05, -0.3, 0.001, -0.001, 0.01]], device=backend).repeat(4, 1)
    tr = RenderTrainer(m, intr, c2w, WH, num_rays=32, lr=2e-3, num_uniform=16, perturb=True, target_sphere_radius=0.5,
                       distortion=dist)
    tr_p = RenderTrainer(_tiny(backend), intr, c2w, WH, num_rays=32, lr=2e-3, num_uniform=16, perturb=True,
                         target_sphere_radius=0.5)
    b, bp = tr._make_batch(), tr_p._make_batch()
    assert torch.equal(b["xy"], bp["xy"]) and float((b["rays_d"] - bp["rays_d"]).abs().max()) > 1e-4
    xy, fidx, gt = tr.sample_batch()
    tr.sample_batch = lambda: (xy, fidx, gt)         # overfit one batch (fresh batches of 32 rays are too noisy to compare)
    losses = steps_on_a_fixed_objective(tr, range(6))
    assert all(l == l for l in losses) and losses[-1] < losses[0], losses


def test_pose_refinement_steps(backend):
    """Pose refinement wired through the step: before ``start_it`` the poses are constants (no gradient, pipelined
    batches), afterwards the per-frame corrections receive finite non-zero gradients through ray generation ->
    ray_test -> ray_query and move; the refined poses stay rigid (orthonormal rotation, bottom row untouched)."""
    m = _tiny(backend)
    intr, c2w_true, WH = look_at_cameras(V=4, seed=1, device=backend)
    c2w = c2w_true.clone()
    c2w[:, :3, 3] += 0.02 * torch.randn(4, 3, generator=torch.Generator().manual_seed(3)).to(backend)   # noisy start
    tr = RenderTrainer(m, intr, c2w, WH, num_rays=32, lr=1e-3, num_uniform=16, perturb=True, target_sphere_radius=0.5,
                       pose_refine=dict(lr=1e-3, start_it=2), c2w_true=c2w_true)
    losses = [float(tr.train_step(it)) for it in range(2)]
    assert tr.pose_delta.grad is None and float(tr.pose_delta.abs().max()) == 0.0
    assert torch.equal(tr.current_c2w(), c2w)
    losses += [float(tr.train_step(it)) for it in range(2, 6)]
    assert all(l == l for l in losses)
    g = tr.pose_delta.grad
    assert g is not None and bool(torch.isfinite(g).all()) and float(g.abs().max()) > 0.0
    assert float(tr.pose_delta.abs().max()) > 0.0
    cur = tr.current_c2w().detach()
    R = cur[:, :3, :3]
    eye = torch.eye(3, device=R.device).expand(4, 3, 3)
    assert torch.allclose(R @ R.transpose(1, 2), eye, atol=1e-5)
    assert torch.equal(cur[:, 3], c2w[:, 3])


@pytest.mark.parametrize("variant", ["default", "all_rays_hit_no_uniform_no_perturb"])
def test_fused_step_equals_autograd_step(backend, variant):
    """The straight launch chain of ``_train_render_fused`` is the autograd step with the engine removed: same loss,
    same parameters after a few iterations (identical kernels; only the order of float atomics may differ).
    Second variant: a long lens (every ray hits the box -> images written without the scatter index), no uniform
    eikonal points, no perturbation."""
    outs = []
    for fused in (False, True):
        torch.manual_seed(0)                    # the occupancy initialisation draws from the global generator
        m = _tiny(backend)
        if variant == "default":
            intr, c2w, WH = look_at_cameras(V=4, seed=1, device=backend)
            kw = dict(num_uniform=24, perturb=True)
        else:
            intr, c2w, WH = look_at_cameras(V=4, seed=1, device=backend, f=4000.0)
            kw = dict(num_uniform=0, perturb=False)
        tr = RenderTrainer(m, intr, c2w, WH, num_rays=40, lr=2e-3, target_sphere_radius=0.5, fused_step=fused, **kw)
        assert tr._fused_ok() == fused
        losses = [float(tr.train_step(it)) for it in range(5)]
        outs.append((losses, m.encoding.flattened_params.detach().clone(), m.sdf_w.detach().clone(),
                     m.rad_w.detach().clone(), m.ln_inv_s.detach().clone(), tr.appear.detach().clone(), dict(tr.stats)))
    (la, *pa, sa), (lb, *pb, sb) = outs
    assert sa == sb and sa["S_f"] > 0 and (variant == "default" or sa["R_hit"] == 40)
    assert all(abs(x - y) < 1e-5 * (1 + abs(x)) for x, y in zip(la, lb)), (la, lb)
    for a, b in zip(pa, pb):
        # float atomics commute only approximately and Adam normalises gradients: an entry whose gradient is rounding
        # noise may take a different +-lr step in the two runs.  All but a handful of entries agree to a few 1e-6; no
        # entry is further apart than the five Adam steps allow.
        d = (a - b).abs()
        bad = d > (5e-5 + 1e-4 * b.abs())
        assert float(bad.float().mean()) < 2e-3, (float(bad.float().mean()), float(d.max()))
        assert float(d.max()) <= 2 * 5 * 2e-3


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("mode", ["speculative", "outgrown", "no_notify"])
def test_speculative_forward_equals_exact_size_forward(backend, mode, fused, monkeypatch):
    """Compressed query mode: from the second iteration on the fused step queues the with-grad gather + decoders at a
    CAPACITY, before the host has read the size of the kept sample set (``_compress(spec_launch=)``, device-side point
    count, sizes through host-mapped words).  Same losses / parameters as the step that waits for the size first; a
    kept set that outgrows the capacity is redone at the exact size (``outgrown``: capacity forced to 32); without the
    host-mapped words (``no_notify``) nothing is speculated.  ``fused`` False: the same through ``ray_query`` and the
    autograd functions (``_FieldFn(pre=)``), the path a reference trainer drives."""
    from neuralsim_amd.fields import neus as nmod
    outs = []
    for spec in (False, True):
        monkeypatch.setattr(nmod, "_SPEC_FORWARD", spec)
        torch.manual_seed(0)
        m = _tiny(backend)
        m.ray_query_cfg["query_mode"] = "march_occ_multi_upsample_compressed"
        if spec and mode == "outgrown":
            m._keep_cap = lambda R: 32
        if spec and mode == "no_notify":
            m._notify = False
        intr, c2w, WH = look_at_cameras(V=4, seed=1, device=backend)
        tr = RenderTrainer(m, intr, c2w, WH, num_rays=40, lr=2e-3, target_sphere_radius=0.5, fused_step=fused,
                           num_uniform=24, perturb=True)
        tr.spec_forward = spec
        assert tr._fused_ok() == fused
        losses, ok = [], []
        for it in range(5):
            losses.append(float(tr.train_step(it)))
            ok.append(bool(getattr(m, "_spec_ok", False)))
        if spec and mode == "speculative":
            assert ok[0] is False and all(ok[1:]), ok          # no capacity before the first observed size
        else:
            assert not any(ok), ok
        outs.append((losses, m.encoding.flattened_params.detach().clone(), m.sdf_w.detach().clone(),
                     m.rad_w.detach().clone(), tr.appear.detach().clone(), dict(tr.stats)))
    (la, *pa, sa), (lb, *pb, sb) = outs
    assert sa == sb and sa["S_f"] > 0
    assert all(abs(x - y) < 1e-5 * (1 + abs(x)) for x, y in zip(la, lb)), (la, lb)
    for a, b in zip(pa, pb):
        d = (a - b).abs()
        bad = d > (5e-5 + 1e-4 * b.abs())
        assert float(bad.float().mean()) < 2e-3, (float(bad.float().mean()), float(d.max()))
        assert float(d.max()) <= 2 * 5 * 2e-3


def test_model_built_from_reference_model_params_trains(backend):
    """A model constructed from a reference-style ``model_params`` block (the dtu yaml's keys, a small pyramid) inside
    the trainer: the model's own ``training_before_per_step`` drives level annealing, inv_s control and the occupancy
    refresh (once per scheduled iteration, with the trainer's rank-shared generator), the fused step trains it."""
    mp = dict(
        dtype="half",
        var_ctrl_cfg=dict(ln_inv_s_init=0.3, ln_inv_s_factor=10.0, ctrl_type="mix_linear", start_it=2, stop_it=6,
                          final_inv_s=400.0),
        cos_anneal_cfg=None, use_tcnn_backend=False,
        surface_cfg=dict(bounding_size=2.0, clip_level_grad_ema_factor=0,
                         encoding_cfg=dict(lotd_auto_compute_cfg=dict(type="gen_ngp", min_res=4, n_feats=2, log2_hashmap_size=10,
                                                                      per_level_scale=1.382, num_levels=8, max_res=64),
                                           anneal_cfg=dict(type="hardmask", start_it=0, start_level=2, stop_it=4),
                                           param_init_cfg=dict(type="uniform_to_type", bound=1.0e-4)),
                         decoder_cfg=dict(type="mlp", D=1, W=64, activation=dict(type="softplus", beta=100.0)),
                         n_extra_feat_from_output=0, radius_init=0.5, geo_init_method="pretrain_after_zero_out"),
        radiance_cfg=dict(use_pos=True, use_nablas=True, use_view_dirs=True, dir_embed_cfg=dict(type="spherical", degree=4),
                          D=2, W=64, n_appear_embedding=4),
        accel_cfg=dict(type="occ_grid", resolution=[16, 16, 16], occ_val_fn_cfg=dict(type="sdf", inv_s=256.0), occ_thre=0.3,
                       ema_decay=0.95, init_cfg=dict(mode="from_net", num_steps=1, num_pts=4096),
                       update_from_net_cfg=dict(num_steps=1, num_pts=2048), update_from_samples_cfg={},
                       n_steps_between_update=2, n_steps_warmup=2),
        ray_query_cfg=dict(query_mode="march_occ_multi_upsample_compressed",
                           query_param=dict(nablas_has_grad=True, num_coarse=8, num_fine=[4, 4],
                                            coarse_step_cfg=dict(step_mode="linear"),
                                            march_cfg=dict(step_size=0.05, max_steps=128), upsample_inv_s=64.0,
                                            upsample_inv_s_factors=[1, 4], upsample_use_estimate_alpha=True)))
    m = LoTDNeuSModel(**mp, device=backend)
    m.populate(device=backend)
    assert m.training_initialize(dict(lr=1e-3, num_iters=500)) is True
    assert 0.0 < m.accel.frac_occupied() < 0.8 and m.sdf_D == 1
    refreshes = []
    real = m.accel.update_from_net
    m.accel.update_from_net = lambda *a, **k: (refreshes.append(k.get("generator") is not None), real(*a, **k))[1]
    intr, c2w, WH = look_at_cameras(V=4, seed=1, device=backend)
    tr = RenderTrainer(m, intr, c2w, WH, num_rays=24, lr=2e-3, num_uniform=32, perturb=True, learn_inv_s=False)
    xy, fidx, gt = tr.sample_batch()
    tr.sample_batch = lambda: (xy, fidx, gt)
    losses, active = [], []
    real_step = tr.train_step
    tr.train_step = lambda it: (real_step(it), active.append(m.encoding.cfg.meta.n_active_levels))[0]
    losses = steps_on_a_fixed_objective(tr, range(6))            # one batch, one set of jitter draws
    tr.train_step = real_step
    # the loss falls between two level activations (a freshly unmasked level perturbs it: hardmask annealing)
    assert all(l == l for l in losses) and losses[2] < losses[0] and losses[-1] < losses[3], losses
    assert active[0] == 3 and active[-1] == 0                    # hardmask: levels 0..2 at it 0, all from stop_it on
    assert refreshes == [True, True]                             # it = 2 and 4, each once, with the shared generator
    assert abs(m._ctrl_mix - 0.75) < 1e-6                        # var_ctrl at it 5: (5 - 2) / (6 - 2)


def test_lidar_step_touches_only_the_parameters_of_its_graph(backend):
    """The street iteration = pixel step + lidar step, each with its own backward and optimizer step
    (code_single/tools/train.py:860-960, 1540-1590).  The lidar render runs with ``with_rgb=False``: radiance, appearance and
    sky parameters get NO gradient there -- they keep their value, their Adam moments and their step count, as under
    torch's Adam (per-parameter ``state['step']``, None grads skipped); the SDF side is stepped twice per iteration.  The
    lidar loss carries the eikonal term (train.py:904 ``with_normal``; app/loss/eikonal.py:233-250) next to depth + line of
    sight, and a batch in which no beam produced a sample is a no-op instead of an exception."""
    from neuralsim_amd import scenarios as sc
    tr = sc.build_street_trainer(backend, small=True, rays_per_gpu=48, lidar_rays=48, num_uniform=16)
    tr.lidar["num_uniform"] = 16
    m, dm, sm = tr.model, tr.distant_model, tr.sky_model
    tr._train_step_pixel(0)
    grp = {id(g["p"]): g for g in tr.optim.groups}
    lidar_free = [m.rad_w, m.rad_b, tr.appear, dm.rad_w, dm.rad_b, sm.w, sm.b]
    in_graph = [m.encoding.flattened_params, m.sdf_w, m.sdf_b, dm.flattened_params, dm.den_w, dm.den_b]
    assert all(grp[id(p)]["t"] == 1 for p in lidar_free + in_graph)
    snap = [(p.detach().clone(), grp[id(p)]["m"].clone(), grp[id(p)]["v"].clone()) for p in lidar_free]
    before = [p.detach().clone() for p in in_graph]
    loss = tr.train_step_lidar(0)
    assert float(loss) == float(loss) and tr.stats["lidar_samples"] > 0
    parts = tr._lidar_parts
    assert parts["depth"] > 0 and parts["los"] >= 0 and parts["eikonal_render"] > 0 and parts["eikonal_uniform"] > 0
    assert abs(float(loss) - sum(parts.values())) < 1e-5 * (1 + abs(float(loss)))
    for p, (p0, m0, v0) in zip(lidar_free, snap):
        g = grp[id(p)]
        assert p.grad is None and g["t"] == 1
        assert torch.equal(p.detach(), p0) and torch.equal(g["m"], m0) and torch.equal(g["v"], v0)
    for p, p0 in zip(in_graph, before):
        assert grp[id(p)]["t"] == 2 and not torch.equal(p.detach(), p0)
    # a batch whose beams all point away from the scene, without the distant model: nothing to differentiate
    tr.distant_model = None
    L = tr.lidar
    L["rays_o"] = torch.full_like(L["rays_o"], 1e4)
    L["rays_d"] = torch.zeros_like(L["rays_d"])
    L["rays_d"][..., 2] = 1.0
    L["num_uniform"] = 0
    before = [p.detach().clone() for p in in_graph]
    loss = tr.train_step_lidar(1)
    assert float(loss) == float(loss) and tr.stats["lidar_samples"] == 0
    assert all(torch.equal(p.detach(), p0) for p, p0 in zip(in_graph, before))
    assert all(grp[id(p)]["t"] == 2 for p in in_graph)
