"""``nr3d_lib.models.loss.utils.reduce`` (reference import: app/loss/mono.py:19): masked reduction of an elementwise loss
(restated from its call sites, e.g. ``reduce(x, mask=mask, reduction='mean')`` mono.py:482-483)."""
import torch


def reduce(loss: torch.Tensor, mask: torch.Tensor = None, reduction: str = "mean"):
    if mask is not None:
        m = mask.to(loss.dtype)
        while m.dim() < loss.dim():
            m = m.unsqueeze(-1)
        loss = loss * m
        if reduction == "mean":
            return loss.sum() / m.expand_as(loss).sum().clamp_min(1.0)
    if reduction == "mean":
        return loss.mean()
    if reduction == "sum":
        return loss.sum()
    return loss
