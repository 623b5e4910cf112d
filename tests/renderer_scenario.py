"""Seeded render scenarios shared by tests/test_reference_glue.py and tests/golden/make_renderer_fixture.py: the same
models, rays and upstream gradients on whichever backend, run either through the REFERENCE's renderer source
(``run_reference``; needs /root/reference) or through the mirror (``run_mirror``)."""
import torch

from oracle import distant as od, render as orr, sky as osky
from util import look_at_cameras, make_params, model_from_params

AABB = torch.tensor([[-1.0, -1, -1], [1.0, 1, 1]])
RES = [16, 16, 16]
QP = dict(num_coarse=8, num_fine=[4, 4], upsample_inv_s=64.0, upsample_inv_s_factors=[1, 4],
          upsample_use_estimate_alpha=True, march_cfg=dict(step_size=0.05, max_steps=128))

# name -> (distant, sky, training, with_normal, depth_use_normalized_vw)
SCENARIOS = {
    "main_train": dict(distant=False, sky=False, training=True, with_normal=True, norm_depth=False),
    "main_distant_sky_train": dict(distant=True, sky=True, training=True, with_normal=True, norm_depth=False),
    "main_distant_eval": dict(distant=True, sky=False, training=False, with_normal=True, norm_depth=True),
    # the main object posed in the world (rotation, translation, scale): rays converted by the REFERENCE's
    # Scene.convert_rays_in_node, normals rotated back (single_volume_renderer.py:225, 262-265)
    "posed_main_distant_sky_train": dict(distant=True, sky=True, training=True, with_normal=True, norm_depth=True, posed=True),
    "main_sky_all_miss": dict(distant=False, sky=True, training=True, with_normal=False, norm_depth=True, all_miss=True),
}


def build_scenario(name, device, precision="f32"):
    from neuralsim_amd.env import SimpleSky
    from neuralsim_amd.fields.nerf_distant import LoTDNeRFDistantModel
    from neuralsim_amd.fields.neus import OccGridAccel
    s = dict(SCENARIOS[name])
    s["name"], s["device"] = name, device
    p = make_params(sdf_D=1, small=True, sphere=True, seed=3, ln_inv_s=0.45, grid_bound=2e-2, noise_scale=1.0)
    g = torch.Generator().manual_seed(5)
    N = 36
    intr, c2w, WH = look_at_cameras(V=3, seed=3)
    o, d = orr.pinhole_rays(torch.rand(N, 2, generator=g) * 0.6 + 0.2, torch.randint(0, 3, (N,), generator=g), intr, c2w, WH)
    if s.get("all_miss"):
        o = o + torch.tensor([0.0, 6.0, 0.0])
    else:
        o[::6] += torch.tensor([0.0, 4.0, 0.0])          # rays that miss the close-range box
    ha = torch.randn(N, 4, generator=g) * 0.3
    s["world_transform"] = None
    if s.get("posed"):
        ax = torch.tensor([0.3, -0.5, 0.81])
        ax = ax / ax.norm()
        K = torch.tensor([[0.0, -ax[2], ax[1]], [ax[2], 0.0, -ax[0]], [-ax[1], ax[0], 0.0]])
        R = torch.eye(3) + 0.61 * K + (1 - (1 - 0.61 ** 2) ** 0.5) * (K @ K)          # Rodrigues, sin = 0.61
        tr, sc = torch.tensor([0.3, -0.2, 0.15]), 1.25
        # the rays were drawn in the object frame: move them into the world so that the same samples come back
        o, d = (R * (o * sc).unsqueeze(-2)).sum(-1) + tr, (R * (d * sc).unsqueeze(-2)).sum(-1)
        s["world_transform"] = (R.to(device), tr.to(device), sc)
    val, _ = orr.build_occ_grid(p, AABB[0], AABB[1], RES, n_pts=2 ** 13, n_steps=2)
    model = model_from_params(p, device, precision=precision)
    model.accel = OccGridAccel(AABB, resolution=RES, device=device)
    model.accel.occ_val.copy_(val.to(device))
    model.accel.pack_bits()
    model.ray_query_cfg = dict(query_mode="march_occ_multi_upsample", query_param=QP)
    s["model"] = model
    s["distant_model"] = s["sky_model"] = None
    if s["distant"]:
        spec = od.make_ngp4d_spec(target_num_params=2 ** 14, min_res_xyz=3, min_res_w=2, log2_hashmap_size=10)
        pd = od.make_distant_params(spec, grid_bound=0.5)
        dm = LoTDNeRFDistantModel(aabb=AABB, precision=precision, max_steps=16,
                                  lotd_auto_compute_cfg=dict(target_num_params=2 ** 14, min_res_xyz=3, min_res_w=2,
                                                             log2_hashmap_size=10, per_level_scale=1.382))
        with torch.no_grad():
            dm.flattened_params.copy_(pd.grid)
            dm.den_w.copy_(torch.cat([w.reshape(-1) for w in pd.den_w]))
            dm.den_b.copy_(torch.cat([b.reshape(-1) for b in pd.den_b]))
            dm.rad_w.copy_(torch.cat([w.reshape(-1) for w in pd.rad_w]))
            dm.rad_b.copy_(torch.cat([b.reshape(-1) for b in pd.rad_b]))
        s["distant_model"] = dm.to(device)
    if s["sky"]:
        ws, bs = osky.make_sky_params(10, 4, seed=21)
        sky = SimpleSky(n_appear_embedding=4, precision=precision).to(device)
        with torch.no_grad():
            sky.w.copy_(torch.cat([w.reshape(-1) for w in ws]).to(device))
            sky.b.copy_(torch.cat(bs).to(device))
        s["sky_model"] = sky
    dv = lambda t: t.to(device).contiguous()         # noqa: E731
    s.update(rays_o=dv(o), rays_d=dv(d), h_appear=dv(ha), N=N)
    # upstream gradients of the rendered images (fixed, so both glue implementations backpropagate the same loss)
    s["w"] = dict(rgb_volume=dv(torch.randn(N, 3, generator=g)), depth_volume=dv(torch.randn(N, generator=g) * 0.1),
                  mask_volume=dv(torch.randn(N, generator=g)), normals_volume=dv(torch.randn(N, 3, generator=g) * 0.1))
    s["common"] = dict(with_rgb=True, with_normal=s["with_normal"], near=0.01, depth_use_normalized_vw=s["norm_depth"],
                       perturb=False)
    return s


def _params(s):
    out = {}
    for tag, m in (("main", s["model"]), ("distant", s["distant_model"]), ("sky", s["sky_model"])):
        if m is not None:
            for n, p in m.named_parameters():
                out[f"{tag}.{n}"] = p
    return out


def _finish(s, ret, raw, backward):
    dev_cpu = lambda t: t.detach().cpu()          # noqa: E731
    out = dict(rendered={k: dev_cpu(v) for k, v in ret["rendered"].items()},
               samples_cnt=dev_cpu(ret["ray_intersections"]["samples_cnt"]), volume_buffer={}, vw_in_total={}, grads={})
    vb = ret["volume_buffer"]
    if vb["type"] != "empty":
        out["volume_buffer"] = {k: dev_cpu(vb[k]) for k in ("pack_infos_hit", "rays_inds_hit", "t", "opacity_alpha", "rgb", "vw")
                                if k in vb}
    for k, r in raw.items():
        b = r["volume_buffer"]
        if b["type"] != "empty" and "vw_in_total" in b:
            out["vw_in_total"][k] = dev_cpu(b["vw_in_total"]).flatten()
    if backward and s["training"]:
        loss = sum((ret["rendered"][k] * w).sum() for k, w in s["w"].items() if k in ret["rendered"])
        if loss.requires_grad:
            ps = _params(s)
            for p in ps.values():
                p.grad = None
            loss.backward()
            out["grads"] = {k: dev_cpu(p.grad) for k, p in ps.items() if p.grad is not None and float(p.grad.abs().sum()) > 0}
    return out


def run_reference(mods, s, backward=False):
    """The reference's SingleVolumeRenderer.ray_query (single_volume_renderer.py:136-492) over a FakeScene."""
    import ref_glue
    dev = s["device"]
    scene = ref_glue.FakeScene(dev, main_class_name="Main", image_embeddings=ref_glue.FixedEmbeddings(s["h_appear"]),
                               convert_rays_in_node=mods.get("convert_rays_in_node"))
    wt = ref_glue.FakeTransform(*s["world_transform"], device=dev) if s["world_transform"] is not None else None
    scene.add(ref_glue.FakeNode(s["model"], "Main", "main", wt))
    if s["distant_model"] is not None:
        scene.add(ref_glue.FakeNode(s["distant_model"], "Distant", "distant"))
    if s["sky_model"] is not None:
        scene.add(ref_glue.FakeNode(s["sky_model"], "Sky", "sky"))
    r = ref_glue.make_reference_renderer(mods, s["common"], training=s["training"])
    cam = mods["classes"]["Camera"]("cam0")
    with torch.set_grad_enabled(s["training"]):
        ret = r.ray_query(s["rays_o"], s["rays_d"], rays_ts=torch.zeros(s["N"], device=dev), scene=scene, observer=cam,
                          return_buffer=True, return_details=True)
    return _finish(s, ret, ret["raw_per_obj_model"], backward)


def run_mirror(s, backward=False):
    from neuralsim_amd.renderers.single_volume_renderer import SingleVolumeRenderer
    # distant_bwd_trans_thre 0: the mirror's one deliberate deviation from the reference's renderer (no backward for
    # distant shells behind an opaque stretch, tests/test_distant.py) is off for the one-to-one comparison
    r = SingleVolumeRenderer(dict(s["common"], distant_bwd_trans_thre=0.0)).train(s["training"])
    with torch.set_grad_enabled(s["training"]):
        ret = r.ray_query(s["rays_o"], s["rays_d"], model=s["model"], rays_h_appear=s["h_appear"],
                          distant_model=s["distant_model"], sky_model=s["sky_model"], return_buffer=True,
                          return_details=True, world_transform=s["world_transform"])
    return _finish(s, ret, ret["raw_per_obj_model"], backward)
