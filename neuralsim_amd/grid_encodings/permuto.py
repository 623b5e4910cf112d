"""Multi-resolution permutohedral-lattice hash encoding -- mirror of
``nr3d_lib.models.grid_encodings.permuto.PermutoEncoding`` as its call sites use it
(docs/exps/exp_permuto_3d_modulated.py:52-60: ``PermutoEncoding(input_dim + latent_dim, permuto_auto_compute_cfg=ConfigDict(
type='multi_res', n_levels=16, n_feats=2, log2_hashmap_size=19), dtype=..., device=...)``; all_occ.240201.yaml:438-446).
The kernels are csrc/permuto.hip; the algorithm is restated in oracle/permuto.py (published: Adams et al. 2010, Rosu &
Behnke 2023 -- the nr3d_lib implementation is absent: parity unpinned)."""
import math
from typing import Optional

import torch
import torch.nn as nn

from .. import _lib


class PermutoConfig:
    """``permuto_auto_compute_cfg{type: multi_res, coarsest_res, finest_res, n_levels, n_feats, log2_hashmap_size,
    apply_random_shifts_per_level}``: level l has r_l = coarsest (finest / coarsest)^(l / (L-1)) cells per unit length."""

    def __init__(self, in_dim: int = 3, n_levels: int = 16, n_feats: int = 2, log2_hashmap_size: int = 19,
                 coarsest_res: float = 16.0, finest_res: float = 2000.0, apply_random_shifts_per_level: bool = True,
                 seed: int = 0, type: str = "multi_res"):
        if type != "multi_res":
            raise ValueError(f"permuto_auto_compute_cfg.type {type!r}: only 'multi_res' is built")
        if n_feats != 2:
            raise ValueError("the permutohedral kernels hold two features per entry (n_feats: 2, as every reference config)")
        if not (2 <= in_dim <= 8) or not (1 <= n_levels <= _lib.NSIM_MAX_LEVELS):
            raise ValueError("in_dim must be 2..8 and n_levels 1..32")
        self.in_dim, self.num_levels, self.n_feats = int(in_dim), int(n_levels), 2
        self.hashmap_size = 2 ** int(log2_hashmap_size)
        self.res = [float(coarsest_res)] if n_levels == 1 else \
            [float(coarsest_res * (finest_res / coarsest_res) ** (l / (n_levels - 1))) for l in range(n_levels)]
        g = torch.Generator().manual_seed(seed)
        self.shifts = (torch.rand(n_levels, in_dim, generator=g) * 10.0).float() if apply_random_shifts_per_level \
            else torch.zeros(n_levels, in_dim)
        self.out_features = self.num_levels * 2
        self.n_params = self.num_levels * self.hashmap_size * 2
        m = _lib.PermutoMeta()
        m.in_dim, m.num_levels, m.n_feats, m.hashmap_size = self.in_dim, self.num_levels, 2, self.hashmap_size
        for l in range(self.num_levels):
            for i in range(self.in_dim):
                m.scale[l][i] = self.res[l] / math.sqrt((i + 1) * (i + 2))
                m.shift[l][i] = float(self.shifts[l, i])
        self.meta = m


class _PermutoFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, grid32, grid16, cfg: PermutoConfig, need_dydx: bool):
        x = x.detach().float().contiguous()
        S = x.shape[0]
        out = torch.empty([S, cfg.out_features], dtype=torch.float32, device=x.device)
        dydx = torch.empty([S, cfg.out_features, cfg.in_dim], dtype=torch.float32, device=x.device) if need_dydx else None
        _lib.call("nsim_permuto_fwd", cfg.meta, _lib.ptr(grid16), _lib.ptr(x), S, _lib.ptr(out), _lib.ptr(dydx))
        ctx.save_for_backward(x)
        ctx.cfg, ctx.n = cfg, grid32.shape[0]
        if need_dydx:
            ctx.mark_non_differentiable(dydx)
            return out, dydx
        return out

    @staticmethod
    def backward(ctx, g_out, g_dydx=None):
        (x,) = ctx.saved_tensors
        dgrid = torch.zeros([ctx.n], dtype=torch.float32, device=x.device)
        _lib.call("nsim_permuto_bwd", ctx.cfg.meta, _lib.ptr(x), x.shape[0], _lib.ptr(g_out.float().contiguous()),
                  _lib.ptr(dgrid))
        return None, dgrid, None, None, None


class PermutoEncoding(nn.Module):
    """f32 master parameters + fp16 shadow (what the kernels gather: (in_dim + 1) x 4 B per level and point).

    ``forward(x)`` -> features [..., L * 2], differentiable w.r.t. the table; ``forward_dydx(x)`` also returns
    d features / d x [..., L * 2, in_dim] (no gradient of its own: the NeuS field's second-order path runs through
    ``nsim_permuto_scatter``, fields/permuto_neus.py); ``backward_dydx`` contracts it with dL/dfeatures
    (docs/exps/exp_permuto_3d_modulated.py:63-76)."""

    def __init__(self, input_dim: int = 3, permuto_auto_compute_cfg: Optional[dict] = None, dtype=None, device=None,
                 bound: float = 1e-4, seed: int = 42):
        super().__init__()
        self.cfg = PermutoConfig(in_dim=input_dim, **dict(permuto_auto_compute_cfg or {}))
        g = torch.Generator().manual_seed(seed)
        p = ((torch.rand(self.cfg.n_params, generator=g) * 2 - 1) * bound).half().float()
        self.flattened_params = nn.Parameter(p.to(device) if device is not None else p)
        self.register_buffer("params16", self.flattened_params.detach().half(), persistent=False)
        self._shadow_version = self.flattened_params._version
        self.in_features, self.out_features = self.cfg.in_dim, self.cfg.out_features

    def shadow(self) -> torch.Tensor:
        p = self.flattened_params
        if self._shadow_version != p._version or self.params16.device != p.device:
            self.params16 = p.detach().half()
            self._shadow_version = p._version
        return self.params16

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        shape = x.shape[:-1]
        out = _PermutoFn.apply(x.reshape(-1, self.cfg.in_dim), self.flattened_params, self.shadow(), self.cfg, False)
        return out.reshape(*shape, self.cfg.out_features)

    def forward_dydx(self, x: torch.Tensor):
        shape = x.shape[:-1]
        out, dydx = _PermutoFn.apply(x.reshape(-1, self.cfg.in_dim), self.flattened_params, self.shadow(), self.cfg, True)
        return out.reshape(*shape, self.cfg.out_features), dydx.reshape(*shape, self.cfg.out_features, self.cfg.in_dim)

    @staticmethod
    def backward_dydx(dl_dh: torch.Tensor, dydx: torch.Tensor) -> torch.Tensor:
        return (dl_dh.unsqueeze(-1) * dydx).sum(dim=-2)
