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
        out = _lib.zeros([], device=nab.device)
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
        out = _lib.zeros([], device=p.device)
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
        out = _lib.zeros([ctx.rows, C], device=g.device)
        _lib.call("nsim_rows_scatter_add", _lib.ptr(g), _lib.ptr(idx), idx.shape[0], C, ctx.rows, _lib.ptr(out))
        return out, None


def embedding_lookup(table: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """``table[idx]`` (table [rows, C] f32, idx [n] int64) whose backward is one scatter-add launch instead of the
    sort-based ``index_put_(accumulate=True)`` torch falls back to."""
    assert table.dim() == 2 and idx.dim() == 1 and idx.dtype == torch.long
    return _EmbedFn.apply(table, idx.contiguous())


def mono_depth_loss(depth_pred: torch.Tensor, depth_gt: torch.Tensor, mask: torch.Tensor = None) -> torch.Tensor:
    """Scale-and-shift-invariant depth loss on one image patch (``MonoSDFDepthLoss``, app/loss/mono.py:86-157 with its
    defaults ``scale_gt_to_pred=False, detach_scale_shift=False``): the closed-form least-squares (scale, shift) that
    takes the prediction to the target -- gradients flow through the solve, as in the reference -- then the masked mse.
    Plain torch on a renderer output, as the reference does it (row a18)."""
    p, t = depth_pred.reshape(-1).float(), depth_gt.reshape(-1).float()
    m = torch.ones_like(p) if mask is None else mask.reshape(-1).to(p.dtype)
    a00, a01, a11 = (m * p * p).sum(), (m * p).sum(), m.sum()
    b0, b1 = (m * p * t).sum(), (m * t).sum()
    det = a00 * a11 - a01 * a01
    ok = det != 0
    safe = torch.where(ok, det, torch.ones_like(det))
    scale = torch.where(ok, (a11 * b0 - a01 * b1) / safe, torch.zeros_like(det))
    shift = torch.where(ok, (-a01 * b0 + a00 * b1) / safe, torch.zeros_like(det))
    return (m * (scale * p + shift - t) ** 2).sum() / m.sum().clamp_min(1.0)


def mono_normal_loss(normals_pred: torch.Tensor, normals_gt: torch.Tensor, mask: torch.Tensor = None,
                     w_l1: float = 1.0, w_cos: float = 1.0) -> torch.Tensor:
    """``MonoNormalLoss.fn`` (app/loss/mono.py:479-484): L1 + (1 - cos) between the NORMALISED rendered normals and the
    normalised prior, averaged over all pixels with the mask as weight (``reduce(..., reduction='mean')``)."""
    import torch.nn.functional as F
    n_p, n_g = F.normalize(normals_pred.reshape(-1, 3), dim=-1), F.normalize(normals_gt.reshape(-1, 3), dim=-1)
    l1 = (n_p - n_g).abs().sum(-1)
    cos = 1.0 - (n_p * n_g).sum(-1)
    if mask is not None:
        m = mask.reshape(-1).to(l1.dtype)
        l1, cos = l1 * m, cos * m
    return w_l1 * l1.mean() + w_cos * cos.mean()
