"""``nr3d_lib.models.utils`` -- ``batchify_query`` (app/renderers/single_volume_renderer.py:565), ``calc_grad_norm``
(code_single/tools/train.py:1516), ``get_optimizer`` / ``get_scheduler`` (app/models/scene/learnable_params.py:293-294)."""
import torch

from neuralsim_amd.renderers.single_volume_renderer import batchify_query  # noqa: F401
from neuralsim_amd.model_base import lr_factor


def calc_grad_norm(norm_type: float = 2.0, **named_models) -> dict:
    """{'<name>': norm of all gradients of that model, ..., 'total': norm over everything}."""
    out, total = {}, []
    for name, m in named_models.items():
        gs = [p.grad.detach().float().norm(norm_type) for p in m.parameters() if p.grad is not None]
        if gs:
            out[name] = float(torch.stack(gs).norm(norm_type))
            total.append(out[name])
    out["total"] = float(torch.tensor(total).norm(norm_type)) if total else 0.0
    return out


def get_optimizer(param_groups, lr: float = 1e-3, betas=(0.9, 0.99), eps: float = 1e-15, weight_decay: float = 0.0, **unused):
    return torch.optim.Adam(param_groups, lr=float(lr), betas=tuple(betas), eps=float(eps), weight_decay=weight_decay)


def get_scheduler(optimizer, **cfg):
    return torch.optim.lr_scheduler.LambdaLR(optimizer, lambda it: lr_factor(it, cfg))
