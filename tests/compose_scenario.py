"""The multi-object scenario of tests/test_reference_glue.py / tests/golden/make_renderer_fixture.py: a single-object
background model, a shared batched model with three posed instances (one never hit), a sky; run through the
REFERENCE's BufferComposeRenderer source (``run_reference``) or the mirror (``run_mirror``)."""
import math

import torch

from oracle import render as orr, sky as osky
from util import look_at_cameras, make_params, model_from_params


def _rot_y(a):
    c, s = math.cos(a), math.sin(a)
    return torch.tensor([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]])


def build(device):
    from neuralsim_amd.env import SimpleSky
    from neuralsim_amd.eval import all_pixel_xy
    from neuralsim_amd.fields.neus import OccGridAccel
    from test_batched import AABB, QP, RES, _instances
    ps, mb, occs = _instances(3, device)
    pm = make_params(sdf_D=2, small=True, sphere=True, seed=11, ln_inv_s=0.45, grid_bound=2e-2, noise_scale=1.0)
    main = model_from_params(pm, device, precision="f32")
    main.accel = OccGridAccel(AABB, resolution=RES, device=device)
    val, _ = orr.build_occ_grid(pm, AABB[0], AABB[1], RES, n_pts=2 ** 14, n_steps=2)
    main.accel.occ_val.copy_(val.to(device))
    main.accel.pack_bits()
    main.ray_query_cfg = dict(query_mode="march_occ_multi_upsample", query_param=QP)
    mb.ray_query_cfg = dict(query_mode="march_occ_multi_upsample", query_param=QP)
    ws, bs = osky.make_sky_params(10, 4, seed=21)
    sky = SimpleSky(n_appear_embedding=4, precision="f32").to(device)
    with torch.no_grad():
        sky.w.copy_(torch.cat([w.reshape(-1) for w in ws]).to(device))
        sky.b.copy_(torch.cat(bs).to(device))
    dv = lambda a: a.to(device).contiguous()         # noqa: E731
    intr, c2w, WH = look_at_cameras(V=2, seed=4, H=14, W=14, f=20.0)
    eye, fwd = c2w[0, :3, 3], c2w[0, :3, 2]
    # car2 and car0 sit one behind the other on the optical axis of camera 0: the central rays cross BOTH items, i.e.
    # the batched buffer holds two consecutive packs for those rays
    poses = {"car2": (_rot_y(0.6), eye + 2.0 * fwd, 0.55),
             "car1": (_rot_y(0.2), torch.tensor([0.0, 40.0, 0.0]), 0.4),          # far outside every ray: never hit
             "car0": (_rot_y(-0.9), eye + 3.4 * fwd, 1.0)}
    poses = {k: (dv(R), dv(t), s) for k, (R, t, s) in poses.items()}
    o, d = orr.pinhole_rays(all_pixel_xy(14, 14, torch.device("cpu")), torch.zeros(196, dtype=torch.long), intr, c2w, WH)
    N = o.shape[0]
    g = torch.Generator().manual_seed(3)
    return dict(device=device, main=main, vehicle=mb, sky=sky, poses=poses, rays_o=dv(o), rays_d=dv(d), N=N,
                h_appear=dv(torch.tensor([[0.1, -0.2, 0.3, 0.05]]).expand(N, -1)),
                common=dict(with_rgb=True, with_normal=True, near=0.01, depth_use_normalized_vw=True, perturb=False),
                w_rgb=dv(torch.randn(N, 3, generator=g)), w_nrm=dv(torch.randn(N, 3, generator=g) * 0.1),
                w_depth=dv(torch.randn(N, generator=g) * 0.1))


def _finish(sc, ret, vehicle_ids):
    r = ret["rendered"]
    models = (("main", sc["main"]), ("veh", sc["vehicle"]), ("sky", sc["sky"]))
    for _, m in models:
        for p in m.parameters():
            p.grad = None
    ((r["rgb_volume"] * sc["w_rgb"]).sum() + (r["normals_volume"] * sc["w_nrm"]).sum()
     + (r["depth_volume"] * sc["w_depth"]).sum()).backward()
    c = c_ = lambda t: t.detach().cpu().clone()          # noqa: E731
    vb = ret["volume_buffer"]
    return dict(rendered={k: c(v) for k, v in r.items()}, samples_cnt=c(ret["ray_intersections"]["samples_cnt"]),
                volume_buffer={k: c(vb[k]) for k in ("pack_infos_hit", "t", "opacity_alpha", "rgb", "vw")},
                vw_in_total={k: c(ret["raw_per_obj_model"][k]["volume_buffer"]["vw_in_total"]).flatten()
                             for k in ("main", "Vehicle")},
                grads={f"{n}.{k}": c(p.grad) for n, m in models for k, p in m.named_parameters() if p.grad is not None},
                per_class={c: {k: c_(v) for k, v in r_.items()} for c, r_ in ret.get("rendered_per_class_in_scene", {}).items()},
                per_obj={o: {k: c_(v) for k, v in r_.items()} for o, r_ in ret.get("rendered_per_obj_in_scene", {}).items()},
                vehicle_ids=list(vehicle_ids),
                rays_crossing_two_items=int((torch.unique_consecutive(
                    ret["raw_per_obj_model"]["Vehicle"]["volume_buffer"]["rays_inds_hit"], return_counts=True)[1] > 1).sum()))


def run_reference(mods, sc):
    import ref_glue
    from nr3d_lib.config import ConfigDict
    dev, AA = sc["device"], mods["AssetAssignment"]
    main, mb, sky = sc["main"], sc["vehicle"], sc["sky"]
    main.assigned_to, main.is_ray_query_supported, main.is_batched_query_supported = AA.OBJECT, True, False
    mb.assigned_to, mb.is_ray_query_supported = AA.MULTI_OBJ, True
    assert mb.is_batched_query_supported
    sky.assigned_to, sky.is_ray_query_supported = AA.SCENE, False
    scene = ref_glue.FakeComposeScene(dev, mods["Scene"], image_embeddings=ref_glue.FixedEmbeddings(sc["h_appear"]))
    scene.add(ref_glue.FakeNode(main, "Main", "main", ref_glue.FakeTransform(device=dev)))
    for k, (R, t, s) in sc["poses"].items():
        scene.add(ref_glue.FakeNode(mb, "Vehicle", k, ref_glue.FakeTransform(R, t, s, device=dev)))
    scene.add(ref_glue.FakeNode(sky, "Sky", "sky", ref_glue.FakeTransform(device=dev)))
    rr = mods["compose"].BufferComposeRenderer(ConfigDict(common=ConfigDict(sc["common"]), train=ConfigDict(), val=ConfigDict()))
    rr.image_postprocessor = None
    rr.train()
    obs = type("Camera", (mods["classes"]["Camera"], ref_glue.FakeObserver), {})("cam0")
    ret = rr.ray_query(sc["rays_o"], sc["rays_d"], rays_ts=torch.zeros(sc["N"], device=dev), scene=scene, observer=obs,
                       return_buffer=True, return_details=True, render_per_class_in_scene=True,
                       render_per_obj_in_scene=True)
    return _finish(sc, ret, ret["raw_per_obj_model"]["Vehicle"]["obj_id"])


def run_mirror(sc):
    from neuralsim_amd.renderers.buffer_compose_renderer import BufferComposeRenderer, Drawable
    drawables = [Drawable("main", "Main", sc["main"])] + \
        [Drawable(k, "Vehicle", sc["vehicle"], rotation=R, translation=t, scale=s) for k, (R, t, s) in sc["poses"].items()]
    mine = BufferComposeRenderer(sc["common"]).train()
    ret = mine(sc["rays_o"], sc["rays_d"], drawables=drawables, rays_h_appear=sc["h_appear"], sky_model=sc["sky"],
               return_buffer=True, return_details=True, render_per_class_in_scene=True, render_per_obj_in_scene=True)
    vb = ret["raw_per_obj_model"]["Vehicle"]["volume_buffer"]
    ids = list(sc["poses"])
    hit = sorted(set(vb["rays_full_bidx_hit"].tolist()))
    return _finish(sc, ret, [ids[i] for i in hit])
