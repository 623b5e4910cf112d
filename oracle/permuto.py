"""oracle.permuto -- CPU restatement of the multi-resolution permutohedral-lattice hash encoding.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED: the implementation the reference calls
(``nr3d_lib.models.grid_encodings.permuto.PermutoEncoding``; call sites: app/models/single/neus.py:64-76 ``PermutoNeuSObj``,
docs/exps/exp_permuto_3d_modulated.py:52-60, code_multi/configs/exps/fg_neus=permuto/all_occ.240201.yaml:438-446
``permuto_auto_compute_cfg{type: multi_res, coarsest_res, finest_res, n_levels, n_feats, log2_hashmap_size,
apply_random_shifts_per_level}``) is absent.  This restates the PUBLISHED algorithm:

* Adams, Baek, Davis, "Fast High-Dimensional Filtering Using the Permutohedral Lattice" (2010): elevate the d-dim point onto
  the hyperplane sum = 0 of R^(d+1), round to the nearest remainder-0 lattice point, rank the residuals to find the enclosing
  simplex, barycentric weights from the sorted residuals, the d+1 simplex vertices by remainder;
* Rosu, Behnke, "PermutoSDF" (2023): one such lattice per resolution level (geometric spacing between the coarsest and the
  finest), a hash table of T entries x F features per level, a per-level random shift of the input, features of the d+1
  vertices blended with the barycentric weights, levels concatenated; hash of a vertex = sum over its first d coordinates
  k <- (k + key_i) * 2531011 in uint32, modulo T.

Conventions fixed here (mirrored by csrc/permuto.hip):
* level l has resolution r_l = coarsest * (finest / coarsest)^(l / (L-1)) (cells per unit length); coordinate i
  (1-based) is scaled by r_l / sqrt(i (i+1)) before the elevation (the lattice's own anisotropy correction; no blur factor);
* the input is used as given (the NeuS field feeds positions of its [-1,1]^3 box); ``shifts`` [L,d] are added first;
* ties in the rank follow the published loop (i < j: residual_i < residual_j ranks i up, otherwise j);
* parameters are ONE flat tensor, level l occupying [l T F, (l+1) T F), feature index fastest; values accumulate in f32.
Everything is differentiable through autograd in both the table and x (the weights are piecewise linear in x), so normals
and their double backward come for free here.
"""
import math
from dataclasses import dataclass
from typing import List

import torch

HASH_MUL = 2531011


@dataclass
class PermutoSpec:
    in_dim: int
    res: List[float]              # per level: cells per unit length
    n_feats: int
    hashmap_size: int             # T (entries per level)
    shifts: torch.Tensor          # [L, d] f32 (zeros when apply_random_shifts_per_level is off)

    @property
    def num_levels(self):
        return len(self.res)

    @property
    def out_features(self):
        return self.num_levels * self.n_feats

    @property
    def n_params(self):
        return self.num_levels * self.hashmap_size * self.n_feats

    def scale_factors(self, level: int) -> torch.Tensor:
        """[d]: coordinate i (0-based) is multiplied by r / sqrt((i+1)(i+2))."""
        return torch.tensor([self.res[level] / math.sqrt((i + 1) * (i + 2)) for i in range(self.in_dim)], dtype=torch.float32)


def make_permuto_spec(in_dim: int = 3, n_levels: int = 16, n_feats: int = 2, log2_hashmap_size: int = 19,
                      coarsest_res: float = 16.0, finest_res: float = 2000.0, apply_random_shifts_per_level: bool = True,
                      seed: int = 0) -> PermutoSpec:
    """``permuto_auto_compute_cfg{type: multi_res, ...}`` (all_occ.240201.yaml:439-446)."""
    if n_levels == 1:
        res = [float(coarsest_res)]
    else:
        res = [float(coarsest_res * (finest_res / coarsest_res) ** (l / (n_levels - 1))) for l in range(n_levels)]
    g = torch.Generator().manual_seed(seed)
    shifts = torch.rand(n_levels, in_dim, generator=g) * 10.0 if apply_random_shifts_per_level \
        else torch.zeros(n_levels, in_dim)
    return PermutoSpec(in_dim, res, n_feats, 2 ** log2_hashmap_size, shifts.float())


def init_params_uniform(spec: PermutoSpec, bound: float = 1e-4, seed: int = 42) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    return ((torch.rand(spec.n_params, generator=g) * 2 - 1) * bound).half()


def elevate(xs: torch.Tensor) -> torch.Tensor:
    """xs [S,d] (already shifted and scaled: cf_i) -> elevated [S,d+1]:  E_0 = sum_j cf_j,  E_i = sum_{j>i} cf_j - i cf_i.
    Called with f64 inputs (``permuto_forward``), as the kernel computes it: the finest levels work at |E| ~ 3e4 with the
    random shifts, where an f32 ulp is 2e-3 lattice units -- in f32 the weights would carry ~1e-3 of rounding noise and two
    implementations would only agree if they rounded identically.  The running sum starts from the last coordinate."""
    S, d = xs.shape
    cols = [None] * (d + 1)
    sm = torch.zeros_like(xs[:, 0])
    for i in range(d, 0, -1):
        cols[i] = sm - float(i) * xs[:, i - 1]
        sm = sm + xs[:, i - 1]
    cols[0] = sm
    return torch.stack(cols, dim=1)


def simplex(elev: torch.Tensor):
    """elevated [S,d+1] -> (rem0 [S,d+1] long, rank [S,d+1] long): the enclosing simplex (discrete: no gradient)."""
    S, n = elev.shape
    e = elev.detach()
    v = e / n
    up, down = torch.ceil(v) * n, torch.floor(v) * n
    rem0 = torch.where(up - e < e - down, up, down)
    ssum = torch.round(rem0.sum(-1) / n).long()
    diff = e - rem0
    rank = torch.zeros(S, n, dtype=torch.long)
    for i in range(n):
        for j in range(i + 1, n):
            lt = diff[:, i] < diff[:, j]
            rank[:, i] += lt.long()
            rank[:, j] += (~lt).long()
    rem0 = rem0.long()
    pos, neg = ssum > 0, ssum < 0
    s2 = ssum[:, None].expand(S, n)
    wrap_p = pos[:, None] & (rank >= n - s2)
    wrap_n = neg[:, None] & (rank < -s2)
    rem0 = rem0 - wrap_p.long() * n + wrap_n.long() * n
    rank = rank + s2 - wrap_p.long() * n + wrap_n.long() * n
    return rem0, rank


def barycentric(elev: torch.Tensor, rem0: torch.Tensor, rank: torch.Tensor) -> torch.Tensor:
    """-> bary [S,d+1] (weights of the vertices with remainder 0..d), differentiable in ``elev``."""
    S, n = elev.shape
    d = n - 1
    delta = (elev - rem0.to(elev.dtype)) / n
    b = elev.new_zeros(S, n + 1)
    b = b.scatter_add(1, d - rank, delta)
    b = b.scatter_add(1, d + 1 - rank, -delta)
    first = b[:, 0] + 1.0 + b[:, n]
    return torch.cat([first[:, None], b[:, 1:n]], dim=1)


def vertex_index(rem0: torch.Tensor, rank: torch.Tensor, remainder: int, T: int) -> torch.Tensor:
    S, n = rem0.shape
    d = n - 1
    key = rem0[:, :d] + remainder - (rank[:, :d] > d - remainder).long() * n
    k = torch.zeros(S, dtype=torch.long)
    for i in range(d):
        k = ((k + key[:, i]) * HASH_MUL) & 0xFFFFFFFF
    return k % T


def permuto_forward(x: torch.Tensor, params: torch.Tensor, spec: PermutoSpec) -> torch.Tensor:
    """x [S,d] (may require grad), params flat [n_params] -> features [S, L F] f32."""
    d, F, T = spec.in_dim, spec.n_feats, spec.hashmap_size
    p32 = params if params.dtype in (torch.float32, torch.float64) else params.float()
    outs = []
    for l in range(spec.num_levels):
        # lattice arithmetic in f64 (as the kernel: see ``elevate``), weights back in the working precision
        xs = (x.double() + spec.shifts[l].double()) * spec.scale_factors(l).double()
        elev = elevate(xs)
        rem0, rank = simplex(elev)
        bary = barycentric(elev, rem0, rank).to(x.dtype)
        table = p32[l * T * F:(l + 1) * T * F].view(T, F)
        feat = x.new_zeros([x.shape[0], F])
        for r in range(d + 1):
            feat = feat + bary[:, r:r + 1] * table[vertex_index(rem0, rank, r, T)]
        outs.append(feat)
    return torch.cat(outs, dim=-1)
