"""Latent -> LoTD table generators of the shared (multi-instance) foreground model -- SURVEY sec. 8 row a20.

The reference's ``code_multi`` LoTD foreground (``fg_neus=hyper_lotd/no_fg_occ.221218.yaml:307-352``:
``model_class: app.models.shared.AD_StyleLoTDNeuS``, ``latents_cfg{z{dim: 128}}``, ``lotd_grower_cfg{MixedLoTDGrower[
DenseLoTDGrowerFMM{z_dim, lod_res [5,8,13,21], lod_n_feats 4, pseudo_net_param{activation relu, fmm_rank 10, D 5, W 128,
embed_cfg{sinusoidal_legacy, n_frequencies 6}}}, VMSplitLoTDGrowerFMM{...}]}``) does not store a table per instance: a
hyper-network GROWS every instance's dense LoTD levels from its latent code, and ``set_condition({'ins_id' | 'z_ins'})``
(app/models/shared/batched_neus.py:380-407) selects the codes of the batch.

``nr3d_lib.models.grid_encodings.lotd.lotd_batched_growers`` is absent, so the architecture is restated from the config
keys (parity unpinned; oracle/growers.py states the same function):

* ``DenseLoTDGrowerFMM``: a "pseudo" coordinate network evaluated at EVERY vertex of every dense level -- input =
  sinusoidal embedding (``[x, sin(2^k x), cos(2^k x)]``, k < n_frequencies) of the vertex position in [-1, 1]^3 plus the
  one-hot level index (``pseudo_net_type: same`` = one network for all levels), D hidden layers of width W with relu,
  output ``lod_n_feats`` features -- whose weight matrices are modulated by the latent through a rank-``fmm_rank``
  factorised multiplicative modulation (FMM):  W_eff(z) = W o (U(z) V(z)^T / sqrt(rank)),  U(z), V(z) linear in z.
* the grown features are laid out as the per-instance table the field kernels gather from: the kernels are specialised
  for 2 features per level, so a level of resolution R with 4 features becomes TWO kernel levels of resolution R holding
  features (0, 1) and (2, 3) -- trilinear interpolation acts per feature, the decoder sees the same 4 numbers.

Where the arithmetic runs: the grower is a stack of batched dense GEMMs [B * V, W] x [W, W] (V = 12 095 vertices for
[5, 8, 13, 21]) -- plain library GEMMs (rocBLAS / hipBLASLt through torch.bmm), differentiable by autograd; its output
feeds the hand-written gather / scatter kernels through the per-instance table offsets (``ray_goff``).  The
``VMSplitLoTDGrowerFMM`` levels ([34, 55, 89, 144], vector-matrix factorised) are not built.
"""
import math
from typing import List, Sequence

import torch
import torch.nn as nn


def sinusoidal_legacy(x: torch.Tensor, n_frequencies: int) -> torch.Tensor:
    outs = [x]
    for k in range(n_frequencies):
        outs += [torch.sin(x * float(2 ** k)), torch.cos(x * float(2 ** k))]
    return torch.cat(outs, dim=-1)


def dense_level_vertices(lod_res: Sequence[int]) -> torch.Tensor:
    """[V, 3 + L] vertex positions in [-1, 1]^3 (x fastest, the table's storage order) + one-hot level, all levels."""
    rows = []
    L = len(lod_res)
    for l, R in enumerate(lod_res):
        lin = torch.linspace(-1.0, 1.0, int(R))
        zz, yy, xx = torch.meshgrid(lin, lin, lin, indexing="ij")
        oh = torch.zeros(int(R) ** 3, L)
        oh[:, l] = 1.0
        rows.append(torch.cat([xx.reshape(-1, 1), yy.reshape(-1, 1), zz.reshape(-1, 1), oh], dim=-1))
    return torch.cat(rows)


class FMMLinear(nn.Module):
    """y_b = (W o M(z_b)) x_b + bias,  M(z) = U(z) V(z)^T, U / V affine in z with biases rank^-1/2: M(0) = 1 exactly -- one
    modulated weight matrix per instance; at z = 0 the layer IS the plain linear layer ``weight``."""

    def __init__(self, in_f: int, out_f: int, z_dim: int, rank: int, gen: torch.Generator):
        super().__init__()
        bound = 1.0 / math.sqrt(in_f)
        self.weight = nn.Parameter((torch.rand(out_f, in_f, generator=gen) * 2 - 1) * bound)
        self.bias = nn.Parameter((torch.rand(out_f, generator=gen) * 2 - 1) * bound)
        zb = 1.0 / math.sqrt(z_dim)
        # U(0) V(0)^T = sum_r rank^-1/2 rank^-1/2 = 1: the biases start the modulation at exactly 1 for z = 0 (with
        # rank^-1/4 -- round 3 -- every layer started sqrt(rank) too large: ~1000 x over six layers)
        self.u_w = nn.Parameter((torch.rand(out_f * rank, z_dim, generator=gen) * 2 - 1) * zb * 0.3)
        self.v_w = nn.Parameter((torch.rand(in_f * rank, z_dim, generator=gen) * 2 - 1) * zb * 0.3)
        self.u_b = nn.Parameter(torch.full([out_f * rank], rank ** -0.5))
        self.v_b = nn.Parameter(torch.full([in_f * rank], rank ** -0.5))
        self.in_f, self.out_f, self.rank = in_f, out_f, rank

    def effective_weight(self, z: torch.Tensor) -> torch.Tensor:
        B = z.shape[0]
        U = (z @ self.u_w.t() + self.u_b).view(B, self.out_f, self.rank)
        V = (z @ self.v_w.t() + self.v_b).view(B, self.in_f, self.rank)
        return self.weight.unsqueeze(0) * torch.bmm(U, V.transpose(1, 2))          # [B, out, in]

    def forward(self, x: torch.Tensor, z: torch.Tensor) -> torch.Tensor:
        """x [B, V, in] (or [V, in], shared by the batch) -> [B, V, out]."""
        W = self.effective_weight(z)
        if x.dim() == 2:
            x = x.unsqueeze(0).expand(z.shape[0], -1, -1)
        return torch.baddbmm(self.bias.view(1, 1, -1), x, W.transpose(1, 2))


class DenseLoTDGrowerFMM(nn.Module):
    def __init__(self, z_dim: int = 128, lod_res: Sequence[int] = (5, 8, 13, 21), lod_n_feats: int = 4, D: int = 5,
                 W: int = 128, fmm_rank: int = 10, n_frequencies: int = 6, out_scale: float = 0.1, seed: int = 42):
        super().__init__()
        assert lod_n_feats % 2 == 0, "the table layout holds 2 features per kernel level"
        self.z_dim, self.lod_res, self.lod_n_feats = int(z_dim), [int(r) for r in lod_res], int(lod_n_feats)
        g = torch.Generator().manual_seed(seed)
        vert = dense_level_vertices(self.lod_res)
        emb = torch.cat([sinusoidal_legacy(vert[:, :3], n_frequencies), vert[:, 3:]], dim=-1)
        self.register_buffer("vertex_embedding", emb, persistent=False)
        dims = [emb.shape[1]] + [W] * D + [self.lod_n_feats]
        self.layers = nn.ModuleList([FMMLinear(dims[i], dims[i + 1], self.z_dim, fmm_rank, g) for i in range(len(dims) - 1)])
        self.out_scale = float(out_scale)
        # the per-instance table layout the kernels read: per grown level R, lod_n_feats / 2 kernel levels of resolution R
        self.kernel_lod_res: List[int] = [r for r in self.lod_res for _ in range(self.lod_n_feats // 2)]
        self.n_vertices = [r ** 3 for r in self.lod_res]
        self.n_params = sum(v * self.lod_n_feats for v in self.n_vertices)

    def forward(self, z: torch.Tensor) -> torch.Tensor:
        """z [B, z_dim] -> tables [B, n_params] in the kernels' layout (level-major, vertex, 2 features fastest)."""
        h = self.vertex_embedding
        for i, lay in enumerate(self.layers):
            h = lay(h, z)
            if i < len(self.layers) - 1:
                h = torch.relu(h)
        h = h * self.out_scale                                             # [B, V_total, F]
        out, v0 = [], 0
        for nv in self.n_vertices:
            f = h[:, v0:v0 + nv]                                           # [B, nv, F]
            for c in range(self.lod_n_feats // 2):                         # kernel level c of this resolution
                out.append(f[:, :, 2 * c:2 * c + 2].reshape(z.shape[0], -1))
            v0 += nv
        return torch.cat(out, dim=1)
