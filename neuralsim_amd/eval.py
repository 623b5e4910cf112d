"""Full-image rendering + PSNR -- the eval loop of the reference reduced to the hot path
(``code_single/tools/eval.py:241-316``: ``renderer.render(scene, observer=cam)`` in ``rayschunk`` pieces, then
``PSNR``; ``Camera.get_all_rays`` app/resources/observers/cameras.py:332-380)."""
import torch

from .graphics.cameras import pinhole_selected_rays


def all_pixel_xy(W: int, H: int, device):
    """Pixel-centre xy in [0,1] for a W x H image, row-major (cameras.py:346-350)."""
    i, j = torch.meshgrid(torch.linspace(0, W - 1, W, device=device), torch.linspace(0, H - 1, H, device=device),
                          indexing="xy")
    return torch.stack([(i.reshape(-1) + 0.5) / W, (j.reshape(-1) + 0.5) / H], dim=-1)


@torch.no_grad()
def render_image(renderer, model, intr, c2w, WH, frame: int, rays_h_appear=None, rayschunk: int = 65536, **kw):
    """-> dict of [H, W(,3)] images of camera ``frame`` (eval mode: perturb off, normalised depth weights)."""
    W, H = int(WH[frame, 0]), int(WH[frame, 1])
    xy = all_pixel_xy(W, H, intr.device)
    fidx = torch.full([xy.shape[0]], frame, dtype=torch.long, device=intr.device)
    rays_o, rays_d = pinhole_selected_rays(xy, fidx, intr, c2w, WH)
    was_training = renderer.training
    renderer.eval()
    try:
        ha = rays_h_appear.expand(xy.shape[0], -1).contiguous() if rays_h_appear is not None else None
        ret = renderer.render(model, rays=[rays_o, rays_d], rays_h_appear=ha, rayschunk=rayschunk, **kw)
    finally:
        renderer.train(was_training)
    return {k: v.reshape(H, W, *v.shape[1:]) for k, v in ret["rendered"].items()}


def psnr(pred: torch.Tensor, target: torch.Tensor) -> float:
    """``nr3d_lib.graphics.utils.PSNR``: -10 log10(mse) for images in [0,1]."""
    mse = ((pred.float() - target.float()) ** 2).mean().clamp_min(1e-20)
    return float(-10.0 * torch.log10(mse))
