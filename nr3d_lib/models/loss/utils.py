"""``nr3d_lib.models.loss.utils.reduce`` (reference import: app/loss/mono.py:19): masked reduction of an elementwise loss
(restated from its call sites: ``reduce(x, mask=mask, reduction='mean')`` mono.py:482-483; ``fn(pred [N,3], gt [N,3],
mask=remain [N], reduction='mean' | 'none')`` app/loss/photometric.py:108-142).

Reductions: ``'mean'`` = mean of ``loss * mask`` over ALL elements; ``'mean_in_mask'`` = sum / number of masked-in
elements -- two different scales, which is why the reference switched its calls from the latter to the former
(the commented-out lines photometric.py:108, 130, 139); ``'sum'``; ``'none'`` = the masked elementwise loss.  A mask with
one dimension less than the loss (a per-ray mask on per-channel errors) is broadcast over the trailing dimension."""
import torch


def reduce(loss: torch.Tensor, mask: torch.Tensor = None, reduction: str = "mean"):
    if mask is not None:
        m = mask.to(loss.dtype)
        while m.dim() < loss.dim():
            m = m.unsqueeze(-1)
        loss = loss * m
        if reduction == "mean_in_mask":
            return loss.sum() / m.expand_as(loss).sum().clamp_min(1e-5)
    if reduction in ("mean", "mean_in_mask"):
        return loss.mean()
    if reduction == "sum":
        return loss.sum()
    if reduction == "none":
        return loss
    raise ValueError(f"reduce: unknown reduction {reduction!r}")
