"""No reference cycles through the autograd functions: an OUTPUT stored as a plain ``ctx`` attribute makes
output -> grad_fn -> ctx -> output, which only the cyclic collector frees -- the saved planes of a step (80-150 MB at the
BASELINE sizes) then pile up between collections (21 GB after 258 distant-model steps before the fix;
``ctx.save_for_backward`` is cycle-free).  With the collector off, every output must die with its last reference."""
import gc
import weakref

import torch

from util import look_at_cameras


def _dies(make):
    """-> (every weakly referenced output died, flags, tensors found in garbage cycles by a final collection)."""
    gc.collect()
    gc.disable()
    try:
        refs = make()
        alive = [r() is not None for r in refs]
        gc.set_debug(gc.DEBUG_SAVEALL)          # keep what the collector finds unreachable in gc.garbage
        gc.collect()
        cyc = [tuple(x.shape) for x in gc.garbage if torch.is_tensor(x)]
    finally:
        gc.set_debug(0)
        gc.garbage.clear()
        gc.enable()
    return not any(alive), alive, cyc


def test_field_fn_outputs_die_without_the_cyclic_collector(backend):
    from neuralsim_amd.fields.neus import LoTDNeuSModel
    m = LoTDNeuSModel(lod_res=[4, 8, 16, 24], log2_hashmap_size=10, precision="fp16",
                      accel_cfg=dict(resolution=(8, 8, 8), update_from_net_cfg=dict(num_steps=1, num_pts=512))).to(backend)
    m.geometric_init_sphere(0.5)
    m.accel.set_all_occupied()
    intr, c2w, WH = look_at_cameras(V=2, seed=3)
    from neuralsim_amd.graphics.cameras import pinhole_selected_rays
    xy = torch.rand(40, 2)
    o, d = pinhole_selected_rays(xy.to(backend), torch.zeros(40, dtype=torch.long, device=backend), intr.to(backend),
                                 c2w.to(backend), WH.to(backend))

    def make():
        m.train()
        tested = m.ray_test(o, d, near=0.01)
        tested["rays_h_appear"] = torch.zeros(tested["num_rays"], 4, device=backend)
        ret = m.ray_query(ray_tested=tested, config=dict(m.ray_query_cfg, with_rgb=True, with_normal=True))      # no extra points
        vb = ret["volume_buffer"]
        refs = [weakref.ref(vb["nablas"]), weakref.ref(vb["rgb"]), weakref.ref(vb["sdf"])]
        (vb["rgb"].sum() + vb["nablas"].sum() + vb["sdf"].sum()).backward()
        del ret, vb, tested
        return refs
    ok, alive, cyc = _dies(make)
    assert ok and not cyc, (alive, cyc)


def test_distant_and_sky_outputs_die_without_the_cyclic_collector(backend):
    from neuralsim_amd.env import SimpleSky
    from neuralsim_amd.fields.nerf_distant import LoTDNeRFDistantModel
    aabb = torch.tensor([[-1.0, -1, -1], [1.0, 1, 1]])
    dm = LoTDNeRFDistantModel(aabb=aabb, precision="fp16", max_steps=8,
                              lotd_auto_compute_cfg=dict(target_num_params=2 ** 12, min_res_xyz=3, min_res_w=2,
                                                         log2_hashmap_size=9, per_level_scale=1.4)).to(backend)
    sky = SimpleSky(precision="fp16").to(backend)
    g = torch.Generator().manual_seed(1)
    o = (torch.randn(30, 3, generator=g) * 0.1).to(backend)
    d = torch.nn.functional.normalize(torch.randn(30, 3, generator=g), dim=-1).to(backend)

    def make():
        tested = dict(num_rays=30, rays_inds=torch.arange(30, device=backend), rays_o=o, rays_d=d,
                      near=torch.full([30], 0.01, device=backend), far=torch.full([30], 3.0, device=backend),
                      rays_h_appear=torch.zeros(30, 4, device=backend))
        ret = dm.ray_query(ray_tested=tested, config=dict(with_rgb=True))
        vb = ret["volume_buffer"]
        keys = [k for k in ("rgb", "sigma", "opacity_alpha") if k in vb and torch.is_tensor(vb[k]) and vb[k].requires_grad]
        assert keys
        refs = [weakref.ref(vb[k]) for k in keys]
        sum(vb[k].sum() for k in keys).backward()
        c = sky(d, h_appear=torch.zeros(30, 4, device=backend))
        refs.append(weakref.ref(c))
        c.sum().backward()
        del ret, vb, tested, c
        return refs
    ok, alive, cyc = _dies(make)
    assert ok and not cyc, (alive, cyc)
