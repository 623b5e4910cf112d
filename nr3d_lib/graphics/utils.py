"""``nr3d_lib.graphics.utils.PSNR`` (code_single/tools/train.py:47, eval.py)."""
import torch


def PSNR(pred: torch.Tensor, gt: torch.Tensor, mask: torch.Tensor = None) -> torch.Tensor:
    """-10 log10(mse) between images in [0, 1], optionally inside a mask (``PSNR(pred, gt).item()``, train.py:1071)."""
    err = (pred.float() - gt.float()) ** 2
    if mask is not None:
        m = mask.to(err.dtype)
        while m.dim() < err.dim():
            m = m.unsqueeze(-1)
        mse = (err * m).sum() / m.expand_as(err).sum().clamp_min(1.0)
    else:
        mse = err.mean()
    return -10.0 * torch.log10(mse.clamp_min(1e-20))


def SSIM(*a, **k):
    raise NotImplementedError("SSIM: neuralsim_amd.eval.ssim")


LPIPS = SSIM
