"""Trainer-facing life cycle of a model -- mirrors ``nr3d_lib.models.model_base.ModelMixin`` at the calls the reference's
``AssetBank`` and trainer make on every model (app/resources/asset_bank.py:129-151, 269-321;
code_single/tools/train.py:1393, 1449, 1494-1502):

  ``training_setup(training_cfg)`` -> ``.optimizer`` (a real ``torch.optim.Optimizer``: the trainer hands it to
  ``GradScaler.unscale_`` / ``GradScaler.step`` and reads ``param_groups[*]['name' | 'lr']``),
  ``training_update_lr(it)`` (``training_cfg.scheduler``), ``training_clip_grad()``, ``stat_param(with_grad=, prefix=)``,
  ``get_weight_reg(norm_type=)`` (app/loss/weight_reg.py:66-67) and the no-op hooks.

The optimizer is the fused gfx950 Adam (csrc/optim.hip: f32 master + fp16 shadow in one pass) behind the torch
interface.  The implementation of ModelMixin lives in the absent nr3d_lib: the scheduler formulas and the clipping keys
are restated here (parity unpinned) from the reference's configs (``training_cfg{lr, eps, betas, invs_betas, scheduler{type:
exponential | warmup_cosine | multistep, num_iters, min_factor, warmup_steps}}``, lotd_neus.dtu.230814.yaml:178-185,
381-397).
"""
import math
from typing import Dict, List, Optional

import torch

from . import _lib


def lr_factor(it: int, scheduler: Optional[dict]) -> float:
    """lr(it) / lr(0) of ``training_cfg.scheduler``.  exponential: ``min_factor ** (it / num_iters)``; warmup_cosine:
    ``min_factor + (1 - min_factor) (1 + cos(pi p)) / 2`` over the post-warm-up progress p; multistep: ``gamma ** (number of
    milestones passed)``.  All three ramp linearly from 0 over ``warmup_steps`` (``(it + 1) / warmup_steps``)."""
    if not scheduler:
        return 1.0
    cfg = dict(scheduler)
    typ = cfg.get("type", "exponential")
    num_iters = max(int(cfg.get("num_iters", 1)), 1)
    min_factor = float(cfg.get("min_factor", 1.0))
    warm = int(cfg.get("warmup_steps", 0) or 0)
    w = min(1.0, (it + 1) / warm) if warm > 0 else 1.0
    if typ == "exponential":
        f = min_factor ** (min(max(it, 0), num_iters) / num_iters)
    elif typ == "warmup_cosine":
        p = min(max((it - warm) / max(num_iters - warm, 1), 0.0), 1.0)
        f = min_factor + (1.0 - min_factor) * 0.5 * (1.0 + math.cos(math.pi * p))
    elif typ == "multistep":
        f = float(cfg.get("gamma", 0.1)) ** sum(1 for m in cfg.get("milestones", []) if it >= int(m))
    elif typ in ("constant", "none", None):
        f = 1.0
    else:
        raise NotImplementedError(f"training_cfg.scheduler.type = {typ!r}")
    return w * f


class FusedAdamTorch(torch.optim.Optimizer):
    """``torch.optim.Adam`` semantics (no weight decay, no amsgrad) with every parameter updated by ``nsim_adam_step``;
    a group's optional ``shadow16`` callable returns the fp16 copy the gather kernels read (written in the same pass).
    Works under ``torch.cuda.amp.GradScaler`` (``unscale_`` scales ``p.grad`` in place, ``step`` calls ``step()``)."""

    def __init__(self, param_groups: List[dict], lr=1e-2, betas=(0.9, 0.99), eps=1e-15, on_step=None):
        super().__init__(param_groups, dict(lr=lr, betas=tuple(betas), eps=eps, shadow16=None, name=""))
        self._on_step = on_step

    def state_dict(self):
        """Checkpointable (``CheckpointIO.register_modules(optimizer_<name>=...)``, code_single/tools/train.py:1368-1371):
        the per-group ``shadow16`` callables are runtime wiring, not state."""
        sd = super().state_dict()
        sd["param_groups"] = [{k: v for k, v in g.items() if k != "shadow16"} for g in sd["param_groups"]]
        return sd

    def load_state_dict(self, state_dict):
        wiring = [g.get("shadow16") for g in self.param_groups]
        sd = dict(state_dict)
        sd["param_groups"] = [dict(g, shadow16=None) for g in sd["param_groups"]]
        super().load_state_dict(sd)
        for g, w in zip(self.param_groups, wiring):
            g["shadow16"] = w

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for g in self.param_groups:
            b1, b2 = g["betas"]
            for p in g["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, dtype=torch.float32)
                    st["exp_avg_sq"] = torch.zeros_like(p, dtype=torch.float32)
                st["step"] += 1
                t = st["step"]
                p16 = g["shadow16"]() if g.get("shadow16") is not None else None
                _lib.call("nsim_adam_step", _lib.ptr(p.data), _lib.ptr(p16), _lib.ptr(p.grad.contiguous()),
                          _lib.ptr(st["exp_avg"]), _lib.ptr(st["exp_avg_sq"]), p.numel(), float(g["lr"]), float(b1), float(b2),
                          float(g["eps"]), 1.0 - b1 ** t, 1.0 - b2 ** t, 1.0, 0)
        if self._on_step is not None:
            self._on_step()
        return loss


class ModelMixin:
    """See the module docstring.  A model lists its parameter groups in ``_param_groups(training_cfg)``."""
    training_cfg: dict = None
    optimizer: Optional[torch.optim.Optimizer] = None

    # ------------------------------------------------------------------ to be provided by the model
    def _param_groups(self, cfg: dict) -> List[dict]:
        return [dict(name=n, params=[p]) for n, p in self.named_parameters() if p.requires_grad]

    def _after_optimizer_step(self):
        """Derived buffers that follow the parameters (packed MFMA weight fragments) refresh lazily from versions."""

    # ------------------------------------------------------------------ reference life cycle
    def training_setup(self, training_cfg: dict = None, name_prefix: str = ""):
        cfg = dict(training_cfg or {})
        self.training_cfg = cfg
        lr = cfg.get("lr", 1e-2)
        groups = self._param_groups(cfg)
        for g in groups:
            g["name"] = name_prefix + g["name"]
            g.setdefault("lr", float(lr[g["name"]]) if isinstance(lr, dict) else float(lr))
            g["initial_lr"] = g["lr"]
        self.optimizer = FusedAdamTorch(groups, lr=float(lr) if not isinstance(lr, dict) else 1e-2,
                                        betas=tuple(cfg.get("betas", (0.9, 0.99))), eps=float(cfg.get("eps", 1e-15)),
                                        on_step=self._after_optimizer_step)
        return self.optimizer

    def training_update_lr(self, it: int):
        if self.optimizer is None:
            return
        f = lr_factor(int(it), (self.training_cfg or {}).get("scheduler"))
        for g in self.optimizer.param_groups:
            g["lr"] = g["initial_lr"] * f

    def training_clip_grad(self):
        """``training_cfg{clip_grad_val | clip_grad_norm}`` (absent from the object-centric configs: a no-op there)."""
        cfg = self.training_cfg or {}
        params = [p for p in self.parameters() if p.grad is not None]
        if not params:
            return
        if cfg.get("clip_grad_val") is not None:
            torch.nn.utils.clip_grad_value_(params, float(cfg["clip_grad_val"]))
        if cfg.get("clip_grad_norm") is not None:
            torch.nn.utils.clip_grad_norm_(params, float(cfg["clip_grad_norm"]))

    def training_before_per_step(self, cur_it: int, logger=None):
        pass

    def training_after_per_step(self, cur_it: int, logger=None):
        pass

    def rendering_before_per_view(self, renderer=None, observer=None, per_frame_info: dict = None):
        pass

    def model_setup(self):
        pass

    @torch.no_grad()
    def stat_param(self, with_grad: bool = False, prefix: str = "") -> Dict[str, float]:
        """Nested-dict-of-floats statistics of every parameter (``logger.add_nested_dict(..., d=model.stat_param(
        with_grad=True))``, code_single/tools/train.py:1521)."""
        pre = prefix + ("." if prefix and not prefix.endswith(".") else "")
        out = {}
        for n, p in self.named_parameters():
            items = [("data", p.data)] + ([("grad", p.grad)] if (with_grad and p.grad is not None) else [])
            for what, t in items:
                t = t.detach().float()
                out[f"{pre}{n}.{what}"] = dict(mean=float(t.mean()), std=float(t.std()) if t.numel() > 1 else 0.0,
                                              min=float(t.min()), max=float(t.max()), abs_mean=float(t.abs().mean()),
                                              norm=float(t.norm()))
        return out

    def _weight_reg_tensors(self) -> List[torch.Tensor]:
        return [p for n, p in self.named_parameters() if p.dim() >= 1 and "ln_inv_s" not in n]

    def get_weight_reg(self, norm_type: float = 2.0, **unused) -> torch.Tensor:
        """-> flat tensor with one ``norm_type``-norm per weight tensor (app/loss/weight_reg.py:27, 66-67: the loss sums
        it).  Plain torch on the parameters: differentiable, negligible cost."""
        return torch.stack([p.float().norm(p=norm_type) for p in self._weight_reg_tensors()])
