"""``nr3d_lib.models.loss.safe`` (app/loss/eikonal.py:100, app/loss/mask.py:53-55): losses whose per-element value is
clipped at ``limit`` so that a few outliers cannot dominate a step.  Restated from the call sites
(``safe_mse_loss(x, y, reduction='none', limit=(-1.1, err_limit))``, ``safe_binary_cross_entropy(p, y, limit=, reduction=)``)."""
import torch
import torch.nn.functional as F


def _reduce(x, reduction):
    return x.mean() if reduction == "mean" else (x.sum() if reduction == "sum" else x)


def safe_mse_loss(pred, gt, reduction="mean", limit=1.0):
    """mse of the difference clamped to [-limit, limit] (or to the (lo, hi) pair): gradients vanish beyond the limit."""
    lo, hi = (-float(limit), float(limit)) if not isinstance(limit, (tuple, list)) else (float(limit[0]), float(limit[1]))
    return _reduce(torch.clamp(pred - gt, lo, hi) ** 2, reduction)


def safe_binary_cross_entropy(pred, gt, limit: float = 0.1, reduction="mean"):
    """BCE with the prediction kept ``limit`` away from {0, 1}: the log terms stay bounded."""
    eps = float(limit)
    return F.binary_cross_entropy(pred.clamp(eps, 1.0 - eps), gt, reduction=reduction)
