"""BASELINE-size checks (configs[1]: 8192 rays, L=16 / T=2^19 table, 2x64 decoder) through size-independent
properties (the oracle comparison at this size is tests/test_fullsize_parity.py) -- the HIP path against invariants:
sortedness / counts of the sampler, agreement of the three field kernels on the same points, weight bounds of the
compositing, linearity of the backward in its upstream gradients, independence from the scatter's de-duplication
and from the query chunking.  GPU only (the emulator would take hours at this size)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    import bench
    assert torch.cuda.is_available()
    dev = torch.device("cuda", 0)
    tr = bench.build_trainer(dev, 0, 1)
    torch.manual_seed(0)
    xy, fidx, gt = tr.sample_batch()
    return tr, xy, fidx, gt


def _render(tr, xy, fidx, mode):
    from neuralsim_amd.graphics.cameras import pinhole_selected_rays
    rays_o, rays_d = pinhole_selected_rays(xy, fidx, tr.intr, tr.c2w, tr.WH)
    tested = tr.model.ray_test(rays_o, rays_d, near=0.01, rays_h_appear=tr.appear[fidx])
    cfg = dict(tr.model.ray_query_cfg)
    cfg.update(query_mode=mode, with_rgb=True, with_normal=True, perturb=False, depth_use_normalized_vw=False, _render=True)
    return tested, tr.model.ray_query(ray_tested=tested, config=cfg, return_details=True)


def test_sampler_invariants_full_size(setup):
    tr, xy, fidx, gt = setup
    assert tr.num_rays == 8192 and tr.model.encoding.cfg.n_params == 12196216
    tested, ret = _render(tr, xy, fidx, "march_occ_multi_upsample")
    vb = ret["volume_buffer"]
    pi, t = vb["pack_infos_hit"], vb["t"]
    R = tested["num_rays"]
    assert R > 0.9 * 8192
    n = pi[:, 1]
    # the buffer lists the rays whose occupancy march found something (upsample_on_marched_only, the default): those and
    # only those got num_coarse + sum(num_fine) + marched samples; packs tile the buffer
    mc = ret["details"]["march_counts"]
    live = mc > 0
    assert torch.equal(vb["rays_inds_hit"], tested["rays_inds"][live]) and 0.3 * R < int(live.sum()) < 0.7 * R
    assert int(pi[0, 0]) == 0 and torch.equal(pi[1:, 0], torch.cumsum(n, 0)[:-1]) and int(n.sum()) == t.shape[0]
    assert torch.equal(n, mc[live] + 64 + 48)
    ridx = ret["details"]["ridx"]                 # rows of the TESTED rays
    n_all = torch.zeros(R, dtype=n.dtype, device=n.device).index_put((live.nonzero()[:, 0],), n)
    assert torch.equal(ridx, torch.repeat_interleave(torch.arange(R, device=t.device), n_all))
    # ascending depths inside every ray, inside [near, far]
    same = ridx[1:] == ridx[:-1]
    assert bool(((t[1:] >= t[:-1]) | ~same).all())
    assert bool((t >= tested["near"][ridx] - 1e-5).all()) and bool((t <= tested["far"][ridx] + 1e-5).all())
    # compositing: 0 <= vw, sum_vw <= 1, alpha in [0,1], last alpha of each ray == 0
    a = vb["opacity_alpha"]
    assert float(a.min()) >= 0 and float(a.max()) <= 1 and float(a[pi[:, 0] + n - 1].abs().max()) == 0
    m = ret["rendered"]["mask_volume"]
    assert float(m.min()) >= 0 and float(m.max()) <= 1 + 1e-5
    assert 0.3 < float(m.mean()) < 0.55            # sphere r=0.75 covers ~40 % of the views
    # the three field kernels agree on the same points
    sdf_q = tr.model._query_sdf_rays(tested["rays_o"].contiguous(), tested["rays_d"].contiguous(), t, ridx)
    assert float((sdf_q - vb["sdf"]).abs().max()) < 1e-5
    x = (tested["rays_o"][ridx] + t[:, None] * tested["rays_d"][ridx])[:50000].contiguous()
    assert float((tr.model.query_sdf(x) - vb["sdf"][:50000]).abs().max()) < 2e-3
    assert float((x.norm(dim=-1) - 0.75 - vb["sdf"][:50000]).abs().max()) < 0.05      # it is the sphere
    nn = vb["nablas"][:50000].norm(dim=-1)
    assert 0.8 < float(nn.median()) < 1.2


def test_compressed_mode_close_to_full(setup):
    tr, xy, fidx, gt = setup
    _, full = _render(tr, xy, fidx, "march_occ_multi_upsample")
    _, comp = _render(tr, xy, fidx, "march_occ_multi_upsample_compressed")
    kept, tot = comp["volume_buffer"]["t"].shape[0], full["volume_buffer"]["t"].shape[0]
    assert 0.05 * tot < kept < 0.6 * tot
    for k in ("rgb_volume", "mask_volume"):
        assert float((full["rendered"][k] - comp["rendered"][k]).abs().max()) < 2e-2, k


def _grads(tr, ret, tested, w_rgb, w_eik):
    tr.optim.zero_grad()
    vb = ret["volume_buffer"]
    loss = w_rgb * (ret["rendered"]["rgb_volume"] ** 2).mean() + w_eik * ((vb["nablas"].norm(dim=-1) - 1) ** 2).mean()
    loss.backward()
    m = tr.model
    return [p.grad.clone() for p in (m.encoding.flattened_params, m.sdf_w, m.rad_w, m.sdf_b)]


def test_backward_linearity_and_dedup_independence(setup):
    tr, xy, fidx, gt = setup
    mode = "march_occ_multi_upsample_compressed"

    def run(w_rgb, w_eik):
        tested, ret = _render(tr, xy, fidx, mode)
        return _grads(tr, ret, tested, w_rgb, w_eik)
    g_a, g_b, g_ab = run(1.0, 0.0), run(0.0, 1.0), run(1.0, 1.0)
    for a, b, ab in zip(g_a, g_b, g_ab):
        ref = a + b
        assert float((ab - ref).norm() / ref.norm().clamp_min(1e-20)) < 2e-2          # fp16 MFMA operands, f32 atomics
    os.environ["NSIM_DEDUP_MAX_RES"] = "0"          # no run reduction in the scatter: same gradient
    try:
        g_nd = run(1.0, 1.0)
    finally:
        os.environ.pop("NSIM_DEDUP_MAX_RES")
    assert float((g_nd[0] - g_ab[0]).norm() / g_ab[0].norm()) < 1e-4
    # the table gradient only touches entries (its support is a small fraction of the 12.2 M parameters)
    assert 0.0 < float((g_ab[0] != 0).float().mean()) < 0.7


def test_street_shaped_config_full_size():
    """BASELINE configs[3]-shaped model (withmask_withlidar_joint.240219.yaml:146-226): cuboid LoTD with T = 2^20 and the
    config's 32 Mi parameters = 19 levels (two 16-level feature chunks in the decoder kernels), 1x64 decoder, sdf_scale 25, elongated AABB with a per-axis occupancy grid, sky MLP, 16384 rays per GPU.  Size-independent
    properties: the query kernels agree, the masked levels stay untouched, a few training steps run finite."""
    from neuralsim_amd.fields.neus import LoTDNeuSModel
    from neuralsim_amd.grid_encodings.lotd import cuboid_ngp_res
    from neuralsim_amd.env import SimpleSky
    from neuralsim_amd.graphics.cameras import look_at_cameras
    from neuralsim_amd.trainer import RenderTrainer
    dev = torch.device("cuda", 0)
    aspect = [2.0, 1.0, 0.3]
    res = cuboid_ngp_res(aspect, 16, 2048, 19)
    aabb = torch.tensor([[-1.0, -0.5, -0.15], [1.0, 0.5, 0.15]])
    m = LoTDNeuSModel(lod_res=res, log2_hashmap_size=20, sdf_D=1, precision="fp16", ln_inv_s_init=0.3, sdf_scale=25.0,
                      aabb=aabb, accel_cfg=dict(resolution=(64, 32, 10), update_from_net_cfg=dict(num_steps=2, num_pts=2 ** 18)),
                      param_bound=2e-2, seed=4).to(dev)
    cfg = m.encoding.cfg
    assert cfg.num_levels == 19 and cfg.hashmap_size == 2 ** 20 and 30 * 2 ** 20 < cfg.n_params < 36 * 2 ** 20
    assert cfg.lod_res3[0][0] > cfg.lod_res3[0][1] > cfg.lod_res3[0][2]            # per-axis resolutions
    m.geometric_init_sphere(0.12, noise_scale=0.5)                                   # a blob inside the flat box
    m.accel.init(m.query_sdf, generator=torch.Generator(device=dev).manual_seed(1))
    assert 0.0 < m.accel.frac_occupied() < 0.9
    g = torch.Generator(device=dev).manual_seed(2)
    x = (torch.rand(200000, 3, device=dev, generator=g) * 2 - 1) * (aabb[1].to(dev) * 0.98)
    out = m.forward_sdf_nablas(x)
    s_lm = m.query_sdf(x)
    assert float((s_lm - out["sdf"].detach()).abs().max()) < 2e-3                    # == with-grad forward (fp16 paths)
    assert bool(torch.isfinite(out["nablas"]).all())
    # hardmask: masked levels get exactly no gradient
    m.set_active_levels(10)
    o2 = m.forward_sdf_nablas(x[:50000])
    ((o2["nablas"].norm(dim=-1) - 1) ** 2).mean().backward()
    gg = m.encoding.flattened_params.grad
    off = cfg.lod_offsets[10]
    assert float(gg[off:].abs().max()) == 0.0 and float(gg[:off].abs().max()) > 0.0
    m.set_active_levels(None)
    m.encoding.flattened_params.grad = None
    # a few training steps at 16384 rays with the sky model
    intr, c2w, WH = look_at_cameras(V=20, seed=3, device=dev, radius=1.6)
    sky = SimpleSky(n_appear_embedding=4, precision="fp16", seed=5).to(dev)
    tr = RenderTrainer(m, intr, c2w, WH, num_rays=16384, lr=1e-3, num_uniform=4096, sky_model=sky, learn_inv_s=False)
    losses = [float(tr.train_step(300 + i)) for i in range(6)]
    assert all(l == l and l < 1e3 for l in losses), losses
    assert tr.stats["R_hit"] > 0


def test_indoor_shaped_config_full_size():
    """BASELINE configs[2]-shaped step (lotd_neus.replica.230814.yaml): ``inside_out`` geometry seen from inside, an
    image patch of 64x64 rays next to flat pixel rays (16384 in total), normals and depth rendered WITH gradient for the
    monocular losses (scale-shift-invariant depth, normal L1 + cos -- plain torch on the outputs, as in the reference)."""
    from neuralsim_amd.fields.neus import LoTDNeuSModel
    from neuralsim_amd.renderers.single_volume_renderer import SingleVolumeRenderer
    dev = torch.device("cuda", 0)
    m = LoTDNeuSModel(sdf_D=2, precision="fp16", ln_inv_s_init=0.5, inside_out=True, seed=6).to(dev)
    m.geometric_init_sphere(0.8)
    m.accel.init(m.query_sdf, generator=torch.Generator(device=dev).manual_seed(1))
    g = torch.Generator(device=dev).manual_seed(3)
    rend = SingleVolumeRenderer(dict(with_rgb=True, with_normal=True, near=0.01, depth_use_normalized_vw=True,
                                     perturb=True)).train()

    def rays(shape):
        d = torch.nn.functional.normalize(torch.randn(*shape, 3, device=dev, generator=g), dim=-1)
        o = (torch.rand(*shape, 3, device=dev, generator=g) - 0.5) * 0.2          # cameras near the centre of the room
        return o, d
    o_p, d_p = rays((64, 64))                                                      # image patch
    o_f, d_f = rays((12288,))                                                      # + 12288 pixel rays = 16384
    out_p = rend.render(m, rays=[o_p, d_p], return_buffer=True)
    out_f = rend.render(m, rays=[o_f, d_f], return_buffer=True)
    rp, rf = out_p["rendered"], out_f["rendered"]
    assert rp["depth_volume"].shape == (64, 64) and rp["normals_volume"].shape == (64, 64, 3)
    assert rf["rgb_volume"].shape == (12288, 3)
    assert float(rf["mask_volume"].detach().mean()) > 0.95                               # every ray ends on the wall
    hitp = o_f + rf["depth_volume"].detach()[:, None] * d_f
    assert float((hitp.norm(dim=-1) - 0.8).abs().median()) < 0.05                  # ... at the sphere of radius 0.8
    nv = torch.nn.functional.normalize(rf["normals_volume"].detach(), dim=-1)
    assert float((nv * torch.nn.functional.normalize(hitp, dim=-1)).sum(-1).median()) < -0.8   # normals face the camera
    # monocular-style losses on the patch (depth up to scale/shift, normals) + photometric on the flat rays
    dp = rp["depth_volume"].reshape(-1)
    tgt = (dp.detach() * 1.7 + 0.3) + 0.01 * torch.randn_like(dp)
    A = torch.stack([dp, torch.ones_like(dp)], dim=-1)
    sol = torch.linalg.lstsq(A.detach(), tgt[:, None]).solution
    loss = ((A @ sol).squeeze(-1) - tgt).abs().mean()
    n_hat = torch.nn.functional.normalize(rp["normals_volume"].reshape(-1, 3), dim=-1)
    n_gt = torch.nn.functional.normalize(-(o_p + rp["depth_volume"].detach()[..., None] * d_p).reshape(-1, 3), dim=-1)
    loss = loss + (n_hat - n_gt).abs().sum(-1).mean() + (1 - (n_hat * n_gt).sum(-1)).mean()
    loss = loss + (rf["rgb_volume"] ** 2).mean()
    loss.backward()
    for p in (m.encoding.flattened_params, m.sdf_w, m.rad_w):
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()) and float(p.grad.abs().sum()) > 0


def test_bench_line_carries_the_contract_on_the_device():
    """``bench.timed_run`` on the device: the one JSON line's contract keys plus the objects the measurement section asks for
    -- ``roofline`` of the dominant kernel from live HIP events (bound / achieved / peak / unit / frac / traffic), the
    per-kernel table, the launch count and the host-wait figure; the value is the rays the steps processed over the time."""
    import json
    import bench
    dev = torch.device("cuda", 0)
    tr = bench.build_trainer(dev, 0, 1)
    out, it = bench.timed_run(tr, steps=6, warmup=3, rank=0, world=1, dev=dev, rays_per_gpu=tr.num_rays)
    json.dumps(out)                                            # serialisable as it stands
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "kernels", "abi_calls_per_step", "host_wait_ms_per_step"):
        assert k in out, k
    assert out["steps"] == 6 and out["warmup"] == 3 and out["n_gpus"] == 1 and out["dtype"] == "fp16"
    assert abs(out["value"] - tr.num_rays / (out["ms_per_step"] * 1e-3)) / out["value"] < 1e-2
    rf = out["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma") and rf["kernel"] in out["kernels"]
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4 and 0.0 < rf["frac"] < 1.0
    assert 20 <= out["abi_calls_per_step"] <= 60 and 0.0 <= out["host_wait_ms_per_step"] < out["ms_per_step"]
    cc = rf.get("cache_ceilings")
    if cc is not None:                                         # recorded counters (profiles/round4_l2_requests.json)
        assert 0.0 < cc["gather"]["frac"] < 1.0 and 0.0 < cc["scatter"]["frac"] < 1.0
