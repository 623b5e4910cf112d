"""Fused evaluations of the losses the object-centric training step puts on the path's outputs
and the appearance-embedding lookup.  The reference writes these as a few torch ops each (app/loss/eikonal.py:96-105,
app/loss/photometric.py:88-146, app/models/scene/image_embeddings.py:23-80); values and gradients are identical, the
launch count is not (one kernel per direction)."""
import torch

from . import _lib


class _EikonalFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, nablas):
        nab = nablas.float().reshape(-1, 3).contiguous()
        out = torch.zeros([], dtype=torch.float32, device=nab.device)
        _lib.call("nsim_eikonal_loss_fwd", _lib.ptr(nab), nab.shape[0], _lib.ptr(out))
        ctx.save_for_backward(nab)
        ctx.shape = nablas.shape
        return out

    @staticmethod
    def backward(ctx, g):
        nab, = ctx.saved_tensors
        d = torch.empty_like(nab)
        _lib.call("nsim_eikonal_loss_bwd", _lib.ptr(nab), nab.shape[0], _lib.ptr(g.float().contiguous()), _lib.ptr(d))
        return d.reshape(ctx.shape)


def eikonal_loss(nablas: torch.Tensor) -> torch.Tensor:
    """mean((|nablas| - 1)^2) -- ``EikonalLoss.fn(nablas).mean()`` with the defaults (no noise, plain mse)."""
    return _EikonalFn.apply(nablas)


class _MseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt):
        p = pred.float().contiguous()
        g = gt.float().contiguous()
        out = torch.zeros([], dtype=torch.float32, device=p.device)
        _lib.call("nsim_mse_loss_fwd", _lib.ptr(p), _lib.ptr(g), p.numel(), _lib.ptr(out))
        ctx.save_for_backward(p, g)
        return out

    @staticmethod
    def backward(ctx, gout):
        p, g = ctx.saved_tensors
        d = torch.empty_like(p)
        _lib.call("nsim_mse_loss_bwd", _lib.ptr(p), _lib.ptr(g), p.numel(), _lib.ptr(gout.float().contiguous()), _lib.ptr(d))
        return d, None


def mse_loss(pred: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    """mean((pred - gt)^2), gradient to ``pred`` only (the target is data)."""
    assert pred.shape == gt.shape
    return _MseFn.apply(pred, gt.detach())


class _EmbedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, idx):
        ctx.save_for_backward(idx)
        ctx.rows = table.shape[0]
        return table.detach()[idx]

    @staticmethod
    def backward(ctx, g):
        idx, = ctx.saved_tensors
        g = g.float().contiguous()
        C = g.shape[-1]
        out = torch.zeros([ctx.rows, C], dtype=torch.float32, device=g.device)
        _lib.call("nsim_rows_scatter_add", _lib.ptr(g), _lib.ptr(idx), idx.shape[0], C, ctx.rows, _lib.ptr(out))
        return out, None


def embedding_lookup(table: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """``table[idx]`` (table [rows, C] f32, idx [n] int64) whose backward is one scatter-add launch instead of the
    sort-based ``index_put_(accumulate=True)`` torch falls back to."""
    assert table.dim() == 2 and idx.dim() == 1 and idx.dtype == torch.long
    return _EmbedFn.apply(table, idx.contiguous())
