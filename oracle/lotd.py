"""oracle.lotd -- CPU restatement of the LoTD multi-resolution Dense/Hash grid encoding.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED: the CUDA implementation
(``nr3d_lib.models.grid_encodings.lotd``) is absent; this restates the published Instant-NGP
scheme under the layout the reference's config documents
(``code_single/configs/object_centric/lotd_neus.dtu.230814.yaml:96-111``:
``lod_res [16,23,...,2048]``, ``lod_n_feats 2``, Dense for the first five levels, Hash (T=2^19) after;
12 196 216 fp16 parameters in total).

Conventions fixed here (and mirrored bit-for-bit by csrc/lotd.hip and csrc/field.hip):
* input ``x`` in [-1,1]^3 is mapped to ``u = x/2 + 0.5`` (docs/exps/exp_permuto_3d_modulated.py:63-76);
* a level of resolution R has R^3 *vertices*; ``pos = u * (R-1)``, ``c0 = clamp(floor(pos), 0, R-2)``,
  ``w = pos - c0`` (so u=1 lands on the last vertex with w=1);
* Dense vertex index = x + R*(y + R*z); Hash vertex index = (x*1 ^ y*2654435761 ^ z*805459861) mod T
  in uint32 arithmetic (Instant-NGP primes);
* parameters are ONE flat fp16 tensor (state_dict friendly), level l occupying
  ``[offset_l, offset_l + size_l * F)`` with feature index fastest;
* features are accumulated in f32 from the fp16-stored values.
"""
import math
from dataclasses import dataclass
from typing import List

import torch

PRIME_Y = 2654435761
PRIME_Z = 805459861


@dataclass
class LoTDSpec:
    lod_res: List[int]
    n_feats: int
    hashmap_size: int
    lod_types: List[str]
    lod_sizes: List[int]      # number of entries (vertices or hash slots) per level
    lod_offsets: List[int]    # offset (in scalars) of each level in the flat param tensor
    n_params: int
    aabb: object = None       # [2,3] the pyramid spans this box (u = (x - lo) / (hi - lo) per axis); None = [-1,1]^3

    @property
    def num_levels(self):
        return len(self.lod_res)

    def unit_coords(self, x: torch.Tensor) -> torch.Tensor:
        """position -> [0,1]^3 coordinate of the pyramid, as x * scale + shift (the kernels' arithmetic)."""
        if self.aabb is None:
            return x * 0.5 + 0.5
        a = torch.as_tensor(self.aabb, dtype=torch.float64).reshape(2, 3)
        inv = (1.0 / (a[1] - a[0]))
        return x * inv.to(x.dtype) + (-a[0] * inv).to(x.dtype)

    @property
    def out_features(self):
        return self.num_levels * self.n_feats


def _res3(R):
    """per-level resolution: int (cubic) or [Rx, Ry, Rz] (``lotd_use_cuboid``)."""
    return [int(R)] * 3 if not isinstance(R, (list, tuple)) else [int(v) for v in R]


def cuboid_ngp_res(aspect, min_res: int, max_res: int, num_levels: int):
    """``lotd_use_cuboid: true`` (withmask_withlidar_joint.240219.yaml:160) -- generator absent (nr3d_lib), convention
    fixed here: the shortest axis gets the gen_ngp list, the others are stretched by their extent ratio."""
    base = gen_ngp_res(min_res, max_res, num_levels)
    mn = min(aspect)
    return [[int(math.ceil(r * a / mn - 1e-6)) for a in aspect] for r in base]


def make_lotd_spec(lod_res: List, n_feats: int = 2, log2_hashmap_size: int = 19) -> LoTDSpec:
    T = 2 ** log2_hashmap_size
    types, sizes, offs = [], [], []
    off = 0
    for R in lod_res:
        rx, ry, rz = _res3(R)
        if rx * ry * rz <= T:
            types.append('Dense')
            sizes.append(rx * ry * rz)
        else:
            types.append('Hash')
            sizes.append(T)
        offs.append(off)
        off += sizes[-1] * n_feats
    return LoTDSpec(list(lod_res), n_feats, T, types, sizes, offs, off)


def gen_ngp_res(min_res: int, max_res: int, num_levels: int) -> List[int]:
    """``lotd_auto_compute_cfg{type: gen_ngp}``: res_l = ceil(min_res * s^l), s = (max/min)^(1/(L-1)).
    Reproduces the list in lotd_neus.dtu.230814.yaml:97 for (16, 2048, 16)."""
    s = math.exp(math.log(max_res / min_res) / (num_levels - 1))
    return [int(math.ceil(min_res * s ** l - 1e-6)) for l in range(num_levels)]


def _vertex_index(cx, cy, cz, R, typ: str, T: int):
    if typ == 'Dense':
        rx, ry, _ = _res3(R)
        return cx + rx * (cy + ry * cz)
    m = 0xFFFFFFFF
    hx = cx & m
    hy = (cy * PRIME_Y) & m
    hz = (cz * PRIME_Z) & m
    return (hx ^ hy ^ hz) % T


def lotd_forward(x: torch.Tensor, params: torch.Tensor, spec: LoTDSpec, n_active: int = None) -> torch.Tensor:
    """x [S,3] in [-1,1] (may require grad) , params flat fp16/f32 [n_params] -> h [S, L*F] f32.
    Differentiable in both x (piecewise-trilinear) and params via autograd, so
    ``nablas = d sdf / d x`` and its double-backward come for free in the oracle."""
    S = x.shape[0]
    F = spec.n_feats
    u = spec.unit_coords(x)
    p32 = params if params.dtype in (torch.float32, torch.float64) else params.float()   # f64: conditioning probes
    outs = []
    n_active = getattr(spec, 'n_active', None) if n_active is None else n_active
    for l, R in enumerate(spec.lod_res):
        if n_active is not None and l >= n_active:      # hardmask level annealing (dtu yaml:104-108): zero features
            outs.append(x.new_zeros([S, F]))
            continue
        R3 = torch.tensor(_res3(R), dtype=x.dtype)
        pos = u * (R3 - 1.0)
        c0 = torch.minimum(torch.floor(pos.detach()).clamp_min(0), R3 - 2.0).long()
        w = pos - c0.to(pos.dtype)
        table = p32[spec.lod_offsets[l]: spec.lod_offsets[l] + spec.lod_sizes[l] * F].view(-1, F)
        feat = x.new_zeros([S, F])
        for corner in range(8):
            dx, dy, dz = corner & 1, (corner >> 1) & 1, (corner >> 2) & 1
            wx = w[:, 0] if dx else 1.0 - w[:, 0]
            wy = w[:, 1] if dy else 1.0 - w[:, 1]
            wz = w[:, 2] if dz else 1.0 - w[:, 2]
            idx = _vertex_index(c0[:, 0] + dx, c0[:, 1] + dy, c0[:, 2] + dz, R, spec.lod_types[l],
                                spec.hashmap_size)
            feat = feat + (wx * wy * wz).unsqueeze(-1) * table[idx]
        outs.append(feat)
    return torch.cat(outs, dim=-1)


def init_params_uniform(spec: LoTDSpec, bound: float = 1e-4, seed: int = 42) -> torch.Tensor:
    """``param_init_cfg{type: uniform_to_type, bound: 1e-4}`` (lotd_neus.dtu.230814.yaml:112-114); fp16."""
    g = torch.Generator().manual_seed(seed)
    p = (torch.rand(spec.n_params, generator=g) * 2 - 1) * bound
    return p.half()


def finest_dense_level(spec: LoTDSpec) -> int:
    return max(l for l, t in enumerate(spec.lod_types) if t == 'Dense')


def write_sphere_level(params: torch.Tensor, spec: LoTDSpec, radius: float = 0.5, level: int = None):
    """Synthetic geometric init: feature 0 of a dense level (default: the finest one) := |x_vertex| - radius, so that a
    pass-through decoder yields a sphere SDF of ``radius_init`` (lotd_neus.dtu.230814.yaml:125).
    The reference reaches the same state by 500 iterations of SDF pre-training
    (app/models/single/neus.py:198-236); this is the deterministic stand-in used for synthetic weights."""
    level = finest_dense_level(spec) if level is None else level
    rx, ry, rz = _res3(spec.lod_res[level])
    assert spec.lod_types[level] == 'Dense'
    F = spec.n_feats
    zz, yy, xx = torch.meshgrid(torch.linspace(-1.0, 1.0, rz), torch.linspace(-1.0, 1.0, ry),
                                torch.linspace(-1.0, 1.0, rx), indexing='ij')  # index = x + Rx*(y + Ry*z)
    sdf = torch.sqrt(xx ** 2 + yy ** 2 + zz ** 2) - radius
    lvl = params[spec.lod_offsets[level]: spec.lod_offsets[level] + spec.lod_sizes[level] * F].view(-1, F)
    lvl[:, 0] = sdf.reshape(-1).to(params.dtype)
    return params
