"""``nr3d_lib.models.loss.recon`` -- the elementwise reconstruction losses the reference's loss modules star-import
(app/loss/photometric.py:67-84, app/loss/weight_reg.py:15): plain torch formulas on renderer outputs, restated from
their names (implementation absent).  ``mask`` weights the elements (a [N] mask on [N, 3] errors is broadcast); ``reduction`` as in ``.utils.reduce``."""
import torch
import torch.nn.functional as F

from .utils import reduce

__all__ = ["l1_loss", "l2_loss", "mse_loss", "huber_loss", "smooth_l1_loss", "relative_l1_loss", "relative_l2_loss",
           "mape_loss", "smape_loss"]


def _reduce(x, mask, reduction):
    return reduce(x, mask=mask, reduction=reduction)       # per-ray masks broadcast over the channel dimension


def l1_loss(pred, gt, mask=None, reduction="mean"):
    return _reduce((pred - gt).abs(), mask, reduction)


def l2_loss(pred, gt, mask=None, reduction="mean"):
    return _reduce((pred - gt) ** 2, mask, reduction)


mse_loss = l2_loss


def huber_loss(pred, gt, mask=None, reduction="mean", alpha: float = 1.0):
    return _reduce(F.huber_loss(pred, gt, reduction="none", delta=alpha), mask, reduction)


def smooth_l1_loss(pred, gt, mask=None, reduction="mean", beta: float = 1.0):
    return _reduce(F.smooth_l1_loss(pred, gt, reduction="none", beta=beta), mask, reduction)


def relative_l1_loss(pred, gt, mask=None, reduction="mean", eps: float = 1e-2):
    return _reduce((pred - gt).abs() / (gt.abs() + eps), mask, reduction)


def relative_l2_loss(pred, gt, mask=None, reduction="mean", eps: float = 1e-2):
    return _reduce((pred - gt) ** 2 / (gt ** 2 + eps), mask, reduction)


def mape_loss(pred, gt, mask=None, reduction="mean", eps: float = 1e-2):
    return _reduce((pred - gt).abs() / (gt.abs() + eps), mask, reduction)


def smape_loss(pred, gt, mask=None, reduction="mean", eps: float = 1e-2):
    return _reduce((pred - gt).abs() / (0.5 * (pred.abs() + gt.abs()) + eps), mask, reduction)
