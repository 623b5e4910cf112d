"""``nr3d_lib.graphics.utils``: the image metrics the reference's tools import -- ``PSNR`` (code_single/tools/train.py:47,
1071), ``PSNR`` / ``SSIM`` / ``LPIPS`` (code_single/tools/eval.py:34, 269-315).  Every one returns a 0-dim tensor (the tools
call ``.item()``); arithmetic in neuralsim_amd/eval.py."""
import warnings

import torch

from neuralsim_amd import eval as _ev


def PSNR(pred: torch.Tensor, gt: torch.Tensor, mask: torch.Tensor = None, only_in_mask: bool = False) -> torch.Tensor:
    """-10 log10(mse) between images in [0, 1]; with ``mask`` the squared error is masked and -- ``only_in_mask`` -- averaged
    over the masked pixels only (eval.py:285-286)."""
    if mask is not None and not only_in_mask and mask.dim() < pred.dim():
        mask = mask.unsqueeze(-1)
    return torch.tensor(_ev.psnr(pred, gt, mask, only_in_mask=only_in_mask), device=pred.device)


def SSIM(pred: torch.Tensor, gt: torch.Tensor, mask: torch.Tensor = None, only_in_mask: bool = False) -> torch.Tensor:
    """Structural similarity (11 x 11 Gaussian window, sigma 1.5) of [H, W, C] images in [0, 1] (eval.py:270, 287-288)."""
    return torch.tensor(_ev.ssim(pred, gt, mask, only_in_mask=only_in_mask), device=pred.device)


_LPIPS_WARNED = [False]


def LPIPS(pred: torch.Tensor, gt: torch.Tensor, mask: torch.Tensor = None) -> torch.Tensor:
    """The learned perceptual metric needs the ``lpips`` package and its pre-trained AlexNet weights; neither is part of this
    package (nor installable here: no network).  With ``lpips`` importable it is used as the reference uses it; otherwise the
    score is NaN (the eval tool only averages and logs it, eval.py:271, 289, 314) and a warning says so once."""
    try:
        import lpips as _lp
        fn = getattr(LPIPS, "_fn", None)
        if fn is None:
            fn = LPIPS._fn = _lp.LPIPS(net="alex").to(pred.device)
        a = pred.float().permute(2, 0, 1).unsqueeze(0) * 2 - 1
        b = gt.float().permute(2, 0, 1).unsqueeze(0) * 2 - 1
        with torch.no_grad():
            r = fn(a, b)
        if not torch.is_tensor(r):        # (an import-time stand-in for the package, as tools/run_reference_train.py installs)
            raise ImportError("lpips")
        return r.reshape(())
    except Exception:
        if not _LPIPS_WARNED[0]:
            _LPIPS_WARNED[0] = True
            warnings.warn("nr3d_lib.graphics.utils.LPIPS: the lpips package / weights are not available -- scores are NaN")
        return torch.tensor(float("nan"), device=pred.device)
