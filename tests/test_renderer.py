"""SingleVolumeRenderer mirror + full-image eval loop: chunked rendering equals un-chunked, image PSNR of the HIP
path against the oracle on identical weights (BASELINE metric: 'PSNR vs ref')."""
import pytest
import torch

from oracle import render as orr
from neuralsim_amd.eval import all_pixel_xy, psnr, render_image
from neuralsim_amd.fields.neus import OccGridAccel
from neuralsim_amd.renderers.single_volume_renderer import SingleVolumeRenderer
from util import look_at_cameras, make_params, model_from_params

AABB = torch.tensor([[-1.0, -1, -1], [1.0, 1, 1]])
RES = [32, 32, 32]
QP = dict(nablas_has_grad=True, num_coarse=16, num_fine=[4, 4, 8], upsample_inv_s=64.0, upsample_inv_s_factors=[1, 4, 16],
          upsample_use_estimate_alpha=True, march_cfg=dict(step_size=0.02, max_steps=512))


@pytest.mark.parametrize("precision", ["f32", "fp16"])
def test_image_psnr_vs_oracle(backend, precision):
    p = make_params(sdf_D=2, small=True, sphere=True, seed=3, ln_inv_s=0.6, grid_bound=2e-2, noise_scale=1.0)
    W = H = 20
    intr, c2w, WH = look_at_cameras(V=2, seed=5, H=H, W=W, f=30.0)
    model = model_from_params(p, backend, precision=precision)
    model.ray_query_cfg = dict(query_mode="march_occ_multi_upsample", query_param=QP)
    model.accel = OccGridAccel(AABB, resolution=RES, device=backend)
    val, occ = orr.build_occ_grid(p, AABB[0], AABB[1], RES, n_pts=2 ** 14, n_steps=2)
    model.accel.occ_val.copy_(val.to(backend))
    model.accel.pack_bits()
    ha = torch.tensor([[0.1, -0.2, 0.3, 0.05]])
    renderer = SingleVolumeRenderer(dict(with_rgb=True, with_normal=True, near=0.01, depth_use_normalized_vw=True,
                                         query_mode="march_occ_multi_upsample"))
    dv = lambda t: t.to(backend)
    img = render_image(renderer, model, dv(intr), dv(c2w), dv(WH), frame=1, rays_h_appear=dv(ha), rayschunk=128)
    whole = render_image(renderer, model, dv(intr), dv(c2w), dv(WH), frame=1, rays_h_appear=dv(ha), rayschunk=0)
    for k in ("rgb_volume", "mask_volume", "depth_volume"):
        assert torch.allclose(img[k], whole[k], atol=1e-6), k                     # chunking is transparent
    # oracle image
    xy = all_pixel_xy(W, H, torch.device("cpu"))
    fidx = torch.full([W * H], 1, dtype=torch.long)
    o, d = orr.pinhole_rays(xy, fidx, intr, c2w, WH)
    with torch.no_grad():
        ret = orr.ray_query(p, o, d, ha.expand(W * H, -1), occ, AABB[0], AABB[1], RES, near=0.01, num_coarse=16,
                            num_fine=(4, 4, 8), step_size=0.02, max_steps=512, depth_use_normalized_vw=True)
    rgb_o = torch.zeros(W * H, 3).index_put((ret["rays_inds"],), ret["rendered"]["rgb_volume"]).reshape(H, W, 3)
    mask_o = torch.zeros(W * H).index_put((ret["rays_inds"],), ret["rendered"]["mask_volume"]).reshape(H, W)
    assert float(mask_o.max()) > 0.5                                               # the object is actually visible
    db = psnr(img["rgb_volume"].cpu(), rgb_o)
    assert db > (60.0 if precision == "f32" else 35.0), db
    assert (img["mask_volume"].cpu() - mask_o).abs().max() < (5e-3 if precision == "f32" else 5e-2)
    n = img["normals_volume"].cpu().norm(dim=-1)
    assert float(n.max()) <= 1.0 + 1e-4                                            # eval normals are normalised (:99-101)


def test_sky_blend_in_renderer(backend):
    """rgb = rgb_volume_occupied + (1 - mask_volume) * sky(v, h_appear) (single_volume_renderer.py:449-457), with
    gradients reaching the sky weights."""
    from oracle import sky as osky
    from neuralsim_amd.env import SimpleSky
    p = make_params(sdf_D=2, small=True, sphere=True, seed=3, ln_inv_s=0.6, grid_bound=2e-2, noise_scale=1.0)
    intr, c2w, WH = look_at_cameras(V=2, seed=5, H=16, W=16, f=12.0)     # wide view: many rays miss the object
    model = model_from_params(p, backend, precision="f32")
    model.ray_query_cfg = dict(query_mode="march_occ_multi_upsample", query_param=QP)
    model.accel = OccGridAccel(AABB, resolution=RES, device=backend)
    val, occ = orr.build_occ_grid(p, AABB[0], AABB[1], RES, n_pts=2 ** 14, n_steps=2)
    model.accel.occ_val.copy_(val.to(backend))
    model.accel.pack_bits()
    ws, bs = osky.make_sky_params(10, 4, seed=21)
    sky = SimpleSky(n_appear_embedding=4, precision="f32").to(backend)
    with torch.no_grad():
        sky.w.copy_(torch.cat([w.reshape(-1) for w in ws]).to(backend))
        sky.b.copy_(torch.cat(bs).to(backend))
    xy = all_pixel_xy(16, 16, torch.device("cpu"))
    fidx = torch.zeros([256], dtype=torch.long)
    o, d = orr.pinhole_rays(xy, fidx, intr, c2w, WH)
    ha = torch.tensor([[0.1, -0.2, 0.3, 0.05]]).expand(256, -1).contiguous()
    renderer = SingleVolumeRenderer(dict(with_rgb=True, with_normal=False, near=0.01, depth_use_normalized_vw=True,
                                         query_mode="march_occ_multi_upsample")).train()
    out = renderer.render(model, rays=[o.to(backend), d.to(backend)], rays_h_appear=ha.to(backend), sky_model=sky,
                          bypass_ray_query_cfg=dict(perturb=False))
    r = out["rendered"]
    sky_ref = osky.sky_forward(torch.nn.functional.normalize(d, dim=-1), ha, ws, bs)
    assert float((r["rgb_sky"].detach().cpu() - sky_ref).abs().max()) <= 2e-5
    blend = osky.blend_sky(r["rgb_volume_occupied"].detach().cpu(), r["mask_volume"].detach().cpu(), sky_ref)
    assert float((r["rgb_volume"].detach().cpu() - blend).abs().max()) <= 3e-5
    assert float(r["mask_volume"].detach().min()) < 0.05 < 0.5 < float(r["mask_volume"].detach().max())   # both regimes present
    r["rgb_volume"].sum().backward()
    assert sky.w.grad is not None and float(sky.w.grad.abs().sum()) > 0
    assert model.encoding.flattened_params.grad is not None                 # (1 - mask) carries gradient to the SDF


def test_ssim_and_masked_psnr():
    """SSIM against a direct (loop-free but unfused) evaluation with scipy's Gaussian filter; PSNR mask variants."""
    import numpy as np
    from scipy.ndimage import correlate1d
    from neuralsim_amd.eval import ssim
    g = torch.Generator().manual_seed(0)
    a = torch.rand(40, 52, 3, generator=g)
    b = (a + 0.1 * torch.randn(40, 52, 3, generator=g)).clamp(0, 1)
    assert abs(ssim(a, a) - 1.0) < 1e-6 and ssim(a, b) < 0.99
    w = np.exp(-((np.arange(11) - 5) ** 2) / (2 * 1.5 ** 2))
    w /= w.sum()

    def blur(x):
        return correlate1d(correlate1d(x, w, axis=0, mode="constant"), w, axis=1, mode="constant")
    x, y = a.numpy().astype(np.float64), b.numpy().astype(np.float64)
    tot = 0.0
    for c in range(3):
        mx, my = blur(x[..., c]), blur(y[..., c])
        sxx, syy, sxy = blur(x[..., c] ** 2) - mx ** 2, blur(y[..., c] ** 2) - my ** 2, blur(x[..., c] * y[..., c]) - mx * my
        tot += (((2 * mx * my + 1e-4) * (2 * sxy + 9e-4)) / ((mx ** 2 + my ** 2 + 1e-4) * (sxx + syy + 9e-4))).mean()
    assert abs(ssim(a, b) - tot / 3) < 1e-5
    m = torch.zeros(40, 52, 1)
    m[10:30, 5:40] = 1
    full = psnr(a * m, b * m)
    assert abs(psnr(a, b, m, only_in_mask=False) - full) < 1e-4
    assert psnr(a, b, m, only_in_mask=True) < full                       # same error over fewer pixels
    assert 0.0 < ssim(a * m, b * m, m, only_in_mask=True) < ssim(a * m, b * m, m, only_in_mask=False)


def test_evaluate_views(backend):
    """eval.py:241-316 on the hot path: chunked full-image renders scored with PSNR / SSIM (full and foreground)."""
    from neuralsim_amd.eval import evaluate_views
    p = make_params(sdf_D=2, small=True, sphere=True, seed=3, ln_inv_s=0.6, grid_bound=2e-2, noise_scale=1.0)
    intr, c2w, WH = look_at_cameras(V=2, seed=5, H=16, W=16, f=14.0)
    model = model_from_params(p, backend, precision="f32")
    model.ray_query_cfg = dict(query_mode="march_occ_multi_upsample", query_param=QP)
    model.accel = OccGridAccel(AABB, resolution=RES, device=backend)
    val, _ = orr.build_occ_grid(p, AABB[0], AABB[1], RES, n_pts=2 ** 14, n_steps=2)
    model.accel.occ_val.copy_(val.to(backend))
    model.accel.pack_bits()
    renderer = SingleVolumeRenderer(dict(with_rgb=True, with_normal=False, near=0.01, depth_use_normalized_vw=True))
    dv = lambda t: t.to(backend)          # noqa: E731
    ha = torch.tensor([[0.1, -0.2, 0.3, 0.05]])
    first = evaluate_views(renderer, model, dv(intr), dv(c2w), dv(WH), [0, 1], [torch.zeros(16, 16, 3)] * 2,
                           rays_h_appear=dv(ha), rayschunk=100)
    from neuralsim_amd.eval import render_image
    imgs = [render_image(renderer, model, dv(intr), dv(c2w), dv(WH), frame=f, rays_h_appear=dv(ha)) for f in (0, 1)]
    res = evaluate_views(renderer, model, dv(intr), dv(c2w), dv(WH), [0, 1], [i["rgb_volume"].cpu() for i in imgs],
                         gt_masks=[(i["mask_volume"] > 0.5).cpu() for i in imgs], rays_h_appear=dv(ha), rayschunk=100)
    assert all(v > 60 for v in res["full_psnr"]) and all(v > 0.999 for v in res["full_ssim"])      # scored against itself
    assert all(a < 40 for a in first["full_psnr"]) and len(res["fg_ssim_only_in_mask"]) == 2


def test_eval_render_is_deterministic_with_a_training_renderer(backend):
    """``render_image`` with the TRAINER's renderer (``perturb: true``, un-normalised depth weights): evaluation applies
    the validation settings (dtu yaml:275-278) itself -- two renders are identical and equal the val-configured one."""
    from neuralsim_amd.eval import render_image
    p = make_params(sdf_D=2, small=True, sphere=True, seed=3, ln_inv_s=0.6, grid_bound=2e-2, noise_scale=1.0)
    intr, c2w, WH = look_at_cameras(V=1, seed=5, H=12, W=12, f=11.0)
    model = model_from_params(p, backend, precision="f32")
    model.ray_query_cfg = dict(query_mode="march_occ_multi_upsample", query_param=QP)
    model.accel = OccGridAccel(AABB, resolution=RES, device=backend)
    val, _ = orr.build_occ_grid(p, AABB[0], AABB[1], RES, n_pts=2 ** 14, n_steps=2)
    model.accel.occ_val.copy_(val.to(backend))
    model.accel.pack_bits()
    dv = lambda t: t.to(backend)          # noqa: E731
    train_r = SingleVolumeRenderer(dict(with_rgb=True, near=0.01, depth_use_normalized_vw=False, perturb=True)).train()
    val_r = SingleVolumeRenderer(dict(with_rgb=True, near=0.01, depth_use_normalized_vw=True, perturb=False)).eval()
    a = render_image(train_r, model, dv(intr), dv(c2w), dv(WH), frame=0)
    b = render_image(train_r, model, dv(intr), dv(c2w), dv(WH), frame=0)
    c = render_image(val_r, model, dv(intr), dv(c2w), dv(WH), frame=0)
    assert train_r.training and train_r.config["perturb"] is True and train_r.config["depth_use_normalized_vw"] is False
    for k in ("rgb_volume", "depth_volume", "mask_volume"):
        assert torch.equal(a[k], b[k]) and torch.equal(a[k], c[k]), k
