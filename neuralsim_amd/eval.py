"""Full-image rendering + PSNR / SSIM -- the eval loop of the reference reduced to the hot path
(``code_single/tools/eval.py:241-316``: ``renderer.render(scene, observer=cam)`` in ``rayschunk`` pieces, then
``PSNR``; ``Camera.get_all_rays`` app/resources/observers/cameras.py:332-380)."""
import torch

from .graphics.cameras import selected_rays


def all_pixel_xy(W: int, H: int, device):
    """Pixel-centre xy in [0,1] for a W x H image, row-major (cameras.py:346-350)."""
    i, j = torch.meshgrid(torch.linspace(0, W - 1, W, device=device), torch.linspace(0, H - 1, H, device=device),
                          indexing="xy")
    return torch.stack([(i.reshape(-1) + 0.5) / W, (j.reshape(-1) + 0.5) / H], dim=-1)


@torch.no_grad()
def render_image(renderer, model, intr, c2w, WH, frame: int, rays_h_appear=None, rayschunk: int = 65536,
                 distortion=None, **kw):
    """-> dict of [H, W(,3)] images of camera ``frame``, rendered with the reference's VALIDATION renderer settings
    whatever the renderer was built with: eval mode, ``perturb: false``, ``depth_use_normalized_vw: true``
    (lotd_neus.dtu.230814.yaml:272-278 ``renderer.train`` / ``renderer.val``: the reference keeps two renderer configs) --
    so evaluating with a trainer's own (perturbing) renderer is deterministic."""
    W, H = int(WH[frame, 0]), int(WH[frame, 1])
    xy = all_pixel_xy(W, H, intr.device)
    fidx = torch.full([xy.shape[0]], frame, dtype=torch.long, device=intr.device)
    rays_o, rays_d = selected_rays(xy, fidx, intr, c2w, WH, distortion=distortion)     # camera_model: pinhole | opencv
    was_training, cfg_saved = renderer.training, dict(renderer.config)
    renderer.eval()
    renderer.config.update(perturb=False, depth_use_normalized_vw=True)
    try:
        ha = rays_h_appear.expand(xy.shape[0], -1).contiguous() if rays_h_appear is not None else None
        ret = renderer.render(model, rays=[rays_o, rays_d], rays_h_appear=ha, rayschunk=rayschunk, **kw)
    finally:
        renderer.train(was_training)
        renderer.config.clear()
        renderer.config.update(cfg_saved)
    return {k: v.reshape(H, W, *v.shape[1:]) for k, v in ret["rendered"].items()}


def psnr(pred: torch.Tensor, target: torch.Tensor, mask: torch.Tensor = None, only_in_mask: bool = False) -> float:
    """``nr3d_lib.graphics.utils.PSNR`` as the eval tool calls it (code_single/tools/eval.py:269, 285-286):
    -10 log10(mse) for images in [0,1]; with ``mask`` [H,W,1] the squared error is masked, and averaged over the
    masked pixels only when ``only_in_mask``."""
    err = (pred.float() - target.float()) ** 2
    if mask is not None:
        m = mask.to(err.dtype).reshape(*err.shape[:-1], 1)
        err = err * m
        mse = err.sum() / (m.sum() * err.shape[-1]).clamp_min(1.0) if only_in_mask else err.mean()
    else:
        mse = err.mean()
    return float(-10.0 * torch.log10(mse.clamp_min(1e-20)))


def ssim(pred: torch.Tensor, target: torch.Tensor, mask: torch.Tensor = None, only_in_mask: bool = False,
         window: int = 11, sigma: float = 1.5) -> float:
    """``nr3d_lib.graphics.utils.SSIM`` (eval.py:270, 287-288): the structural similarity of Wang et al. 2004 -- 11 x 11
    Gaussian window (sigma 1.5), C1 = 0.01^2, C2 = 0.03^2, per channel, images [H,W,C] in [0,1]; the SSIM map is
    averaged over the image, or over the masked pixels when ``only_in_mask``."""
    import torch.nn.functional as F
    x = pred.float().permute(2, 0, 1).unsqueeze(0)
    y = target.float().permute(2, 0, 1).unsqueeze(0)
    C = x.shape[1]
    g = torch.exp(-((torch.arange(window, dtype=torch.float32, device=x.device) - window // 2) ** 2) / (2 * sigma ** 2))
    g = g / g.sum()
    k = (g[:, None] * g[None, :]).expand(C, 1, window, window).contiguous()
    pad = window // 2
    mu_x, mu_y = F.conv2d(x, k, padding=pad, groups=C), F.conv2d(y, k, padding=pad, groups=C)
    sxx = F.conv2d(x * x, k, padding=pad, groups=C) - mu_x ** 2
    syy = F.conv2d(y * y, k, padding=pad, groups=C) - mu_y ** 2
    sxy = F.conv2d(x * y, k, padding=pad, groups=C) - mu_x * mu_y
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    smap = ((2 * mu_x * mu_y + c1) * (2 * sxy + c2)) / ((mu_x ** 2 + mu_y ** 2 + c1) * (sxx + syy + c2))
    if mask is not None and only_in_mask:
        m = mask.to(smap.dtype).reshape(1, 1, *smap.shape[-2:])
        return float((smap * m).sum() / (m.sum() * C).clamp_min(1.0))
    return float(smap.mean())


@torch.no_grad()
def evaluate_views(renderer, model, intr, c2w, WH, frames, gt_images, gt_masks=None, rays_h_appear=None,
                   rayschunk: int = 65536, **kw):
    """The metric loop of code_single/tools/eval.py:241-316 on the hot path: every frame is rendered in ``rayschunk``
    pieces (``rgb_volume``, ``rgb_volume_occupied``, ``mask_volume``) and scored -- full-image PSNR / SSIM and, with
    ground-truth occupancy masks, the foreground scores the tool reports (:281-292).  -> dict of per-frame lists."""
    out = dict(full_psnr=[], full_ssim=[])
    if gt_masks is not None:
        out.update(fg_psnr=[], fg_psnr_only_in_mask=[], fg_ssim=[], fg_ssim_only_in_mask=[])
    for k, f in enumerate(frames):
        img = render_image(renderer, model, intr, c2w, WH, frame=int(f), rays_h_appear=rays_h_appear,
                           rayschunk=rayschunk, **kw)
        gt = gt_images[k].to(img["rgb_volume"])
        out["full_psnr"].append(psnr(img["rgb_volume"], gt))
        out["full_ssim"].append(ssim(img["rgb_volume"], gt))
        if gt_masks is not None:
            m = gt_masks[k].to(gt.device).reshape(*gt.shape[:-1], 1)
            fg_gt = gt * m
            fg = img["rgb_volume_occupied"]
            out["fg_psnr"].append(psnr(fg, fg_gt, m, only_in_mask=False))
            out["fg_psnr_only_in_mask"].append(psnr(fg, fg_gt, m, only_in_mask=True))
            out["fg_ssim"].append(ssim(fg, fg_gt, m, only_in_mask=False))
            out["fg_ssim_only_in_mask"].append(ssim(fg, fg_gt, m, only_in_mask=True))
    return out
