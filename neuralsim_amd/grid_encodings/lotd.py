"""LoTD multi-resolution Dense/Hash grid encoding -- host side.

Mirrors ``nr3d_lib.models.grid_encodings.lotd`` as the reference configures / calls it
(code_single/configs/object_centric/lotd_neus.dtu.230814.yaml:96-111 ``lotd_cfg`` / ``lotd_auto_compute_cfg``;
code_single/tools/inspect_rendering.py:468-474; docs/exps/exp_permuto_3d_modulated.py:63-76
``encoding.forward`` / ``encoding.backward_dydx``).  Kernels: csrc/lotd.hip (standalone) and csrc/field.hip (fused).

Layout: ONE flat parameter tensor (state_dict friendly); level l occupies ``[offset_l, offset_l + size_l*2)``,
feature index fastest; Dense iff res^3 <= 2^log2_hashmap_size.
"""
import math
from typing import List, Optional

import torch
import torch.nn as nn

from .. import _lib


def gen_ngp_res(min_res: int, max_res: int, num_levels: int) -> List[int]:
    """``lotd_auto_compute_cfg{type: gen_ngp}``: ceil(min_res * s^l), s = (max/min)^(1/(L-1)) -- reproduces the
    list in lotd_neus.dtu.230814.yaml:97 for (16, 2048, 16)."""
    s = math.exp(math.log(max_res / min_res) / (num_levels - 1))
    return [int(math.ceil(min_res * s ** l - 1e-6)) for l in range(num_levels)]


def cuboid_ngp_res(aspect, min_res: int, max_res: int, num_levels: int) -> List[List[int]]:
    """``lotd_use_cuboid: true`` (withmask_withlidar_joint.240219.yaml:160): per-axis vertex counts following the aspect
    of the AABB -- the SHORTEST axis gets the gen_ngp list, the others are stretched by their extent ratio.
    (The generator lives in the absent nr3d_lib; this convention is fixed here and in oracle/lotd.py.)"""
    base = gen_ngp_res(min_res, max_res, num_levels)
    mn = min(aspect)
    return [[int(math.ceil(r * a / mn - 1e-6)) for a in aspect] for r in base]


class LoTDConfig:
    def __init__(self, lod_res: List, n_feats: int = 2, log2_hashmap_size: int = 19):
        """lod_res: per level either R (cubic) or [Rx, Ry, Rz] (cuboid)."""
        assert n_feats == 2, "the gfx950 kernels are specialised for 2 features per level"
        assert len(lod_res) <= _lib.NSIM_MAX_LEVELS
        self.lod_res3 = [[int(r)] * 3 if not isinstance(r, (list, tuple)) else [int(v) for v in r] for r in lod_res]
        self.lod_res = [r[0] if r[0] == r[1] == r[2] else list(r) for r in self.lod_res3]
        self.n_feats = n_feats
        self.hashmap_size = 2 ** log2_hashmap_size
        self.lod_types, self.lod_sizes, self.lod_offsets = [], [], []
        off = 0
        for R in self.lod_res3:
            nvert = R[0] * R[1] * R[2]
            dense = nvert <= self.hashmap_size
            self.lod_types.append("Dense" if dense else "Hash")
            self.lod_sizes.append(nvert if dense else self.hashmap_size)
            self.lod_offsets.append(off)
            off += self.lod_sizes[-1] * n_feats
        self.n_params = off
        self.num_levels = len(self.lod_res3)
        self.out_features = self.num_levels * n_feats
        m = _lib.LotdMeta()
        m.num_levels, m.n_feats, m.n_active_levels = self.num_levels, n_feats, 0
        for l in range(self.num_levels):
            for a in range(3):
                m.res[l][a] = self.lod_res3[l][a]
            m.type[l] = 0 if self.lod_types[l] == "Dense" else 1
            m.size[l] = self.lod_sizes[l]
            m.offset[l] = self.lod_offsets[l]
        self.meta = m

    def set_aabb(self, aabb):
        """The pyramid spans the model's AABB: position x (object coordinates) -> unit coordinate u = (x - lo) / (hi - lo)
        per axis (nr3d_lib's AABBSpace normalisation, app/models/single/neus.py:152-196 ``populate(aabb=...)``) -- with
        ``lotd_use_cuboid`` vertex counts the cells are then isotropic in object space.  The [-1,1]^3 cube gives
        (x_scale, x_shift) = (0.5, 0.5): u = x / 2 + 1 / 2, the arithmetic of the cubic configs bit for bit."""
        import torch
        a = torch.as_tensor(aabb, dtype=torch.float64).reshape(2, 3)
        self.aabb = a.float()
        for i in range(3):
            inv = 1.0 / float(a[1, i] - a[0, i])
            self.meta.x_scale[i] = inv
            self.meta.x_shift[i] = -float(a[0, i]) * inv
        return self

    def set_active_levels(self, n: int):
        """Hardmask level annealing (``anneal_cfg{type: hardmask, start_level, start_it, stop_it}``,
        lotd_neus.dtu.230814.yaml:104-108): levels >= n yield zero features and get no gradient."""
        self.meta.n_active_levels = 0 if n is None or n >= self.num_levels else max(1, int(n))

    @classmethod
    def from_cfg(cls, lotd_cfg: Optional[dict] = None, lotd_auto_compute_cfg: Optional[dict] = None):
        if lotd_cfg is not None:
            return cls(lotd_cfg["lod_res"], lotd_cfg.get("lod_n_feats", [2])[0] if isinstance(
                lotd_cfg.get("lod_n_feats", 2), (list, tuple)) else lotd_cfg.get("lod_n_feats", 2),
                int(math.log2(lotd_cfg.get("hashmap_size", 2 ** 19))))
        c = lotd_auto_compute_cfg
        assert c["type"] in ("gen_ngp", "ngp"), c["type"]
        L, mn = c["num_levels"], c["min_res"]
        mx = c.get("max_res", None)
        if mx is None:
            mx = int(round(mn * c["per_level_scale"] ** (L - 1) / 32.0)) * 32 if "per_level_scale" in c else 2048
        return cls(gen_ngp_res(mn, mx, L), c.get("n_feats", 2), c.get("log2_hashmap_size", 19))


class _LotdFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, grid32, grid16, cfg: LoTDConfig, need_dydx: bool):
        x = x.detach().float().contiguous()
        S = x.shape[0]
        out = torch.zeros([S, cfg.out_features], dtype=torch.float32, device=x.device)
        dydx = torch.zeros([S, cfg.out_features, 3], dtype=torch.float32, device=x.device) if need_dydx else None
        _lib.call("nsim_lotd_fwd", _lib.ptr(x), _lib.ptr(grid16), cfg.meta, S, _lib.ptr(out), _lib.ptr(dydx))
        ctx.save_for_backward(x)
        ctx.cfg, ctx.n = cfg, grid32.shape[0]
        if need_dydx:
            return out, dydx
        return out

    @staticmethod
    def backward(ctx, g_out, g_dydx=None):
        (x,) = ctx.saved_tensors
        cfg = ctx.cfg
        dgrid = torch.zeros([ctx.n], dtype=torch.float32, device=x.device)
        go = g_out.float().contiguous() if g_out is not None else None
        gd = g_dydx.float().contiguous() if g_dydx is not None else None
        _lib.call("nsim_lotd_bwd", _lib.ptr(x), _lib.ptr(go), _lib.ptr(gd), cfg.meta, x.shape[0], _lib.ptr(dgrid))
        return None, dgrid, None, None, None


class LoTDEncoding(nn.Module):
    """f32 master parameters + fp16 shadow (what the kernels gather: 512 B per point for L=16, F=2)."""

    def __init__(self, cfg: LoTDConfig, bound: float = 1e-4, device=None, seed: int = 42):
        super().__init__()
        self.cfg = cfg
        g = torch.Generator().manual_seed(seed)
        p = ((torch.rand(cfg.n_params, generator=g) * 2 - 1) * bound).half().float()
        self.flattened_params = nn.Parameter(p.to(device) if device is not None else p)
        self.register_buffer("params16", self.flattened_params.detach().half(), persistent=False)
        self._shadow_version = self.flattened_params._version

    def shadow(self) -> torch.Tensor:
        p = self.flattened_params
        if self._shadow_version != p._version or self.params16.device != p.device:
            self.params16 = p.detach().half()
            self._shadow_version = p._version
        return self.params16

    def forward(self, x):
        return _LotdFn.apply(x, self.flattened_params, self.shadow(), self.cfg, False)

    def forward_dydx(self, x):
        return _LotdFn.apply(x, self.flattened_params, self.shadow(), self.cfg, True)

    @staticmethod
    def backward_dydx(dl_dh: torch.Tensor, dydx: torch.Tensor) -> torch.Tensor:
        """``dl_dx = sum_f dl_dh[f] * dy_dx[f]`` (docs/exps/exp_permuto_3d_modulated.py:72-75; the caller divides
        by 2 there because its dy_dx is w.r.t. the [0,1] input -- ours is already w.r.t. x in [-1,1])."""
        return (dl_dh.unsqueeze(-1) * dydx).sum(dim=-2)
