"""oracle.field -- CPU restatement of the NeuS implicit field: LoTD -> SDF decoder MLP (+ analytic
normals) -> radiance MLP.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED (nr3d_lib absent).  Follows:
* decoder ``type mlp, D 1|2, W 64, softplus beta=100``, ``radius_init 0.5``
  (code_single/configs/object_centric/lotd_neus.dtu.230814.yaml:116-127; BASELINE config 2 asks 2x64);
* normals: ``nablas = d sdf / d x`` = ``(d sdf/d h) . dy_dx / 2`` with create_graph when
  ``nablas_has_grad`` (docs/exps/exp_permuto_3d_modulated.py:63-76) -- here simply autograd through
  the piecewise-trilinear encoding, which is the same function;
* radiance ``use_pos, use_nablas, use_view_dirs, dir_embed spherical degree 4, D 2, W 64,
  n_appear_embedding 4`` (lotd_neus.dtu.230814.yaml:128-139): input = [x(3), SH16(v), nablas(3),
  h_appear(4)] (26), ReLU hidden, sigmoid output;
* ``inv_s = exp(ln_inv_s * ln_inv_s_factor)`` with factor 10 (lotd_neus.dtu.230814.yaml:85-91).
"""
import math
from dataclasses import dataclass, field
from typing import List, Optional

import torch
import torch.nn.functional as F

from .lotd import (LoTDSpec, lotd_forward, init_params_uniform, write_sphere_level, make_lotd_spec,
                   finest_dense_level)

SOFTPLUS_BETA = 100.0
RAD_IN = 26          # 3 + 16 + 3 + 4
RAD_IN_PAD = 32      # padded K of the first radiance layer in the HIP kernels


def sh4(d: torch.Tensor) -> torch.Tensor:
    """Real spherical harmonics, degree 4 (16 values) of unit directions d [...,3]
    (``dir_embed_cfg{type: spherical, degree: 4}``)."""
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    xy, xz, yz = x * y, x * z, y * z
    x2, y2, z2 = x * x, y * y, z * z
    out = [
        torch.full_like(x, 0.28209479177387814),
        -0.48860251190291987 * y,
        0.48860251190291987 * z,
        -0.48860251190291987 * x,
        1.0925484305920792 * xy,
        -1.0925484305920792 * yz,
        0.94617469575755997 * z2 - 0.31539156525251999,
        -1.0925484305920792 * xz,
        0.54627421529603959 * x2 - 0.54627421529603959 * y2,
        0.59004358992664352 * y * (-3.0 * x2 + y2),
        2.8906114426405538 * xy * z,
        0.45704579946446572 * y * (1.0 - 5.0 * z2),
        0.3731763325901154 * z * (5.0 * z2 - 3.0),
        0.45704579946446572 * x * (1.0 - 5.0 * z2),
        1.4453057213202769 * z * (x2 - y2),
        0.59004358992664352 * x * (-x2 + 3.0 * y2),
    ]
    return torch.stack(out, dim=-1)


@dataclass
class FieldParams:
    """All learnable tensors of one NeuS field.  Same flat layout the product uses
    (neuralsim_amd/fields/neus.py) so weights can be exchanged verbatim."""
    spec: LoTDSpec
    grid: torch.Tensor                      # fp16 [n_params]
    sdf_w: List[torch.Tensor] = field(default_factory=list)   # [(out,in)] f32, D+1 entries
    sdf_b: List[torch.Tensor] = field(default_factory=list)
    rad_w: List[torch.Tensor] = field(default_factory=list)   # 3 entries: (64,26),(64,64),(3,64)
    rad_b: List[torch.Tensor] = field(default_factory=list)
    ln_inv_s: torch.Tensor = None           # scalar f32
    ln_inv_s_factor: float = 10.0
    sdf_scale: float = 1.0                  # sdf = head(h) / sdf_scale (street config ``sdf_scale: 25``, 240219.yaml:158)
    pos_embed_n: Optional[int] = None       # ``extra_pos_embed_cfg{type: sinusoidal_legacy, n_frequencies}`` (no_fg_occ.221218.yaml:319-321)

    def tensors(self):
        return [self.grid, *self.sdf_w, *self.sdf_b, *self.rad_w, *self.rad_b, self.ln_inv_s]

    def requires_grad_(self, flag=True):
        for t in self.tensors():
            t.requires_grad_(flag)
        return self

    def inv_s(self):
        return torch.exp(self.ln_inv_s * self.ln_inv_s_factor)


def _linear_init(out_f, in_f, gen, scale=1.0):
    bound = 1.0 / math.sqrt(in_f)
    w = (torch.rand(out_f, in_f, generator=gen) * 2 - 1) * bound * scale
    b = (torch.rand(out_f, generator=gen) * 2 - 1) * bound * scale
    return w, b


def make_field_params(lod_res=None, n_feats=2, log2_hashmap_size=19, sdf_D=2, W=64, seed=42,
                      grid_bound=1e-4, radius_init=0.5, ln_inv_s=0.3, sphere_init=True,
                      noise_scale=0.25) -> FieldParams:
    """Deterministic synthetic weights: hash tables U(-1e-4,1e-4) fp16, decoder =
    pass-through of (finest dense level, feat 0) (holding the sphere SDF) + small random remainder, radiance =
    torch.nn.Linear-style uniform init, ``ln_inv_s`` 0.3 => inv_s = e^3 ~ 20 (config starts at 0.1... the
    reference anneals to final_inv_s 2000; tests sweep both)."""
    if lod_res is None:
        lod_res = [16, 23, 31, 43, 59, 81, 112, 154, 213, 295, 407, 562, 777, 1073, 1483, 2048]
    spec = make_lotd_spec(lod_res, n_feats, log2_hashmap_size)
    gen = torch.Generator().manual_seed(seed)
    grid = init_params_uniform(spec, grid_bound, seed)
    in_f = spec.out_features
    sdf_w, sdf_b = [], []
    dims = [in_f] + [W] * sdf_D + [1]
    for li in range(len(dims) - 1):
        w, b = _linear_init(dims[li + 1], dims[li], gen, scale=noise_scale)
        sdf_w.append(w)
        sdf_b.append(b)
    if sphere_init:
        grid = write_sphere_level(grid, spec, radius_init)
        f_in = 2 * finest_dense_level(spec)
        # Units 0 / 1 of every hidden layer carry +s / -s, s = (sphere level, feat 0): softplus(s) - softplus(-s) == s for
        # any beta, so the pair passes the stored SDF through exactly, and the activations are SMALL near the surface
        # (~ln2/beta +- s/2), where a reduced-precision (fp16) evaluation needs its resolution.  (An earlier version sent
        # s + 2 through the linear region of ONE unit: activations ~2 have an fp16 spacing of 2e-3 -- the SDF became a
        # staircase and the compressed query kept 43 % fewer samples than this f32 restatement.)
        for li in range(sdf_D):
            for u, sg in ((0, 1.0), (1, -1.0)):
                sdf_w[li][u].zero_()
                sdf_b[li][u] = 0.0
                if li == 0:
                    sdf_w[li][u, f_in] = sg
                else:
                    sdf_w[li][u, 0], sdf_w[li][u, 1] = sg, -sg
        sdf_w[-1][0] *= 0.05
        sdf_w[-1][0, 0], sdf_w[-1][0, 1] = 1.0, -1.0
        sdf_b[-1][0] = 0.0
    rad_w, rad_b = [], []
    rdims = [RAD_IN, W, W, 3]
    for li in range(3):
        w, b = _linear_init(rdims[li + 1], rdims[li], gen)
        rad_w.append(w)
        rad_b.append(b)
    return FieldParams(spec=spec, grid=grid, sdf_w=sdf_w, sdf_b=sdf_b, rad_w=rad_w, rad_b=rad_b,
                       ln_inv_s=torch.tensor(float(ln_inv_s)))


def sdf_decoder(h: torch.Tensor, p: FieldParams) -> torch.Tensor:
    a = h
    n = len(p.sdf_w)
    relu = getattr(p, "sdf_activation", "softplus") == "relu"       # ``decoder_cfg.activation: relu`` (no_fg_occ.221218.yaml:357)
    for li in range(n - 1):
        z = F.linear(a, p.sdf_w[li], p.sdf_b[li])
        a = F.relu(z) if relu else F.softplus(z, beta=SOFTPLUS_BETA, threshold=20.0)
    out = F.linear(a, p.sdf_w[-1], p.sdf_b[-1]).squeeze(-1)
    return out if p.sdf_scale == 1.0 else out / p.sdf_scale


def encode(x: torch.Tensor, p: FieldParams) -> torch.Tensor:
    """features of the field's encoding: LoTD (oracle/lotd.py) or, for a ``PermutoSpec``, the permutohedral lattice
    (oracle/permuto.py) on AABB-normalised positions with the optional condition ``p.z`` [S or 1, z_dim] concatenated."""
    from .permuto import PermutoSpec, permuto_forward
    if isinstance(p.spec, PermutoSpec):
        aabb = getattr(p, "aabb", None)
        u = x
        if aabb is not None:
            a = torch.as_tensor(aabb, dtype=x.dtype).reshape(2, 3)
            u = (x - (a[0] + a[1]) * 0.5) / ((a[1] - a[0]) * 0.5)
        z = getattr(p, "z", None)
        if p.spec.in_dim > 3:
            zz = x.new_zeros([x.shape[0], p.spec.in_dim - 3]) if z is None else z.to(x.dtype).expand(x.shape[0], -1)
            u = torch.cat([u, zz], dim=-1)
        return permuto_forward(u, p.grid, p.spec)
    return lotd_forward(x, p.grid, p.spec)


def pos_embed(x: torch.Tensor, p: FieldParams) -> torch.Tensor:
    """``sinusoidal_legacy`` embedding of the AABB-normalised position x_n in [-1, 1]: [x_n | sin(2^k x_n), cos(2^k x_n), k < N]
    (3 + 6 N values; the embedder itself lives in the absent nr3d_lib -- order and the missing factor pi are this repo's
    convention, the learned first layer absorbs a permutation)."""
    aabb = getattr(p.spec, "aabb", None)
    xn = x
    if aabb is not None:
        a = torch.as_tensor(aabb, dtype=x.dtype).reshape(2, 3)
        xn = (x - (a[0] + a[1]) * 0.5) / ((a[1] - a[0]) * 0.5)
    out = [xn]
    for k in range(int(p.pos_embed_n)):
        out += [torch.sin(xn * float(2 ** k)), torch.cos(xn * float(2 ** k))]
    return torch.cat(out, dim=-1)


def forward_sdf(x: torch.Tensor, p: FieldParams) -> torch.Tensor:
    h = encode(x, p)
    if getattr(p, "pos_embed_n", None) is not None:
        h = torch.cat([h, pos_embed(x, p).to(h.dtype)], dim=-1)
    return sdf_decoder(h, p)


def forward_sdf_nablas(x: torch.Tensor, p: FieldParams, nablas_has_grad: bool = True, x_has_grad: bool = False):
    """-> (sdf [S], nablas [S,3]).  docs/exps/exp_permuto_3d_modulated.py:63-76.
    x_has_grad: keep x in the graph (pose refinement: sdf and nablas are then differentiable w.r.t. the sample
    positions, the latter through the mixed second derivatives of the piecewise-trilinear interpolant)."""
    outer_grad = torch.is_grad_enabled()
    with torch.enable_grad():
        xg = x if (x_has_grad and x.requires_grad) else x.detach().clone().requires_grad_(True)
        sdf = forward_sdf(xg, p)
        create = outer_grad and nablas_has_grad and (any(t.requires_grad for t in p.tensors()) or xg is x)
        nablas = torch.autograd.grad(sdf, xg, torch.ones_like(sdf), create_graph=create,
                                     retain_graph=True)[0]
    if not create:
        nablas = nablas.detach()
    if not outer_grad:
        sdf = sdf.detach()
    return sdf, nablas


def radiance(x, v, nablas, h_appear, p: FieldParams) -> torch.Tensor:
    inp = torch.cat([x, sh4(v), nablas, h_appear], dim=-1)
    a = F.relu(F.linear(inp, p.rad_w[0], p.rad_b[0]))
    a = F.relu(F.linear(a, p.rad_w[1], p.rad_b[1]))
    return torch.sigmoid(F.linear(a, p.rad_w[2], p.rad_b[2]))


def forward_field(x, v, h_appear, p: FieldParams, x_has_grad: bool = False):
    """The with-grad query of the render step: -> sdf [S], nablas [S,3], rgb [S,3]."""
    sdf, nablas = forward_sdf_nablas(x, p, nablas_has_grad=True, x_has_grad=x_has_grad)
    rgb = radiance(x if x_has_grad else x.detach(), v, nablas, h_appear, p)
    return sdf, nablas, rgb


def params_from_flat(lod_res, log2_hashmap_size, grid, sdf_w, sdf_b, rad_w, rad_b, ln_inv_s, sdf_D=2,
                     ln_inv_s_factor=10.0, n_feats=2, sdf_scale=1.0, aabb=None, pos_embed_n=None) -> FieldParams:
    """FieldParams from the product's FLAT parameter tensors (same layouts: neuralsim_amd/fields/neus.py ``_flat_sizes``)
    -- the weight exchange of the parity tests / the bench's CPU leg.  ``grid`` is rounded to fp16 and held in f32:
    the kernels read the fp16 shadow of the table (lotd_neus.dtu.230814.yaml:94 ``dtype: half``)."""
    spec = make_lotd_spec(list(lod_res), n_feats, log2_hashmap_size)
    if aabb is not None:
        spec.aabb = torch.as_tensor(aabb, dtype=torch.float32).detach().cpu().reshape(2, 3)
    F1 = spec.out_features + (0 if pos_embed_n is None else 3 + 6 * int(pos_embed_n))
    sw, sb, rw, rb = (t.detach().cpu().float() for t in (sdf_w, sdf_b, rad_w, rad_b))
    ws = [sw[:64 * F1].view(64, F1).clone()]
    bs = [sb[:64].clone()]
    if sdf_D == 2:
        ws.append(sw[64 * F1:64 * F1 + 4096].view(64, 64).clone())
        bs.append(sb[64:128].clone())
    ws.append(sw[-64:].view(1, 64).clone())
    bs.append(sb[-1:].clone())
    n1 = 64 * RAD_IN
    rws = [rw[:n1].view(64, RAD_IN).clone(), rw[n1:n1 + 4096].view(64, 64).clone(), rw[-192:].view(3, 64).clone()]
    rbs = [rb[:64].clone(), rb[64:128].clone(), rb[128:].clone()]
    return FieldParams(spec=spec, grid=grid.detach().cpu().half().float(), sdf_w=ws, sdf_b=bs, rad_w=rws, rad_b=rbs,
                       ln_inv_s=ln_inv_s.detach().cpu().float().reshape(()).clone(), ln_inv_s_factor=ln_inv_s_factor,
                       sdf_scale=float(sdf_scale), pos_embed_n=pos_embed_n)
