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

* ``VMSplitLoTDGrowerFMM`` (round 4): levels [34, 55, 89, 144] x 4 features in vector-matrix factorised form -- per level and
  per axis c a PLANE over the two other axes and a LINE along c, the level's feature field being
  sum_c plane_c(u, v) * line_c(w) (per feature; bilinear x linear interpolation).  Here the grower EXPANDS every such level
  into the dense R^3 vertex table the field kernels already read: trilinear interpolation is the tensor product of three
  1-D linear interpolations, so the trilinear interpolant of the vertex products P[j, k] L[i] IS bilinear(P) x linear(L) --
  the same function, exactly (tests/test_batched.py pins it) -- at the price of memory the MI355X has (4 levels x 4
  features: 15.6 M floats = 62 MB f32 per instance; 0.5 GB for the 8 vehicles of BASELINE configs[4] out of 288 GB), and
  no new level type in the gather / scatter / second-order kernels.  Planes and lines come from ONE shared FMM-modulated
  trunk (``pseudo_net_type: shared``, D hidden layers of width W) evaluated at the plane / line vertices (embedded position
  with the collapsed axes at 0, one-hot level, one-hot axis, one-hot kind) with a ``D_head``-layer head per kind; a line
  value is 1 + head output, so a fresh grower's product starts at the plane's value.
* ``MixedLoTDGrower{grower_configs: [...]}``: the levels of several growers behind one another (the config's dense
  [5, 8, 13, 21] followed by the VM-split [34, 55, 89, 144]).
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


class VMSplitLoTDGrowerFMM(nn.Module):
    """See the module docstring.  ``forward(z [B, z_dim]) -> [B, n_params]``: per level R the dense [R^3] expansion, laid out
    as ``lod_n_feats / 2`` kernel levels of resolution R with 2 features each (x fastest)."""

    def __init__(self, z_dim: int = 128, lod_res: Sequence[int] = (34, 55, 89, 144), lod_n_feats: int = 4, D: int = 4,
                 D_head: int = 2, W: int = 256, fmm_rank: int = 10, n_frequencies: int = 10, out_scale: float = 0.1,
                 seed: int = 42):
        super().__init__()
        assert lod_n_feats % 2 == 0
        self.z_dim, self.lod_res, self.lod_n_feats = int(z_dim), [int(r) for r in lod_res], int(lod_n_feats)
        g = torch.Generator().manual_seed(seed)
        L = len(self.lod_res)
        pv, lv = [], []
        self._plane_slices, self._line_slices = [], []
        for l, R in enumerate(self.lod_res):
            lin = torch.linspace(-1.0, 1.0, R)
            for c in range(3):
                a, b = [ax for ax in range(3) if ax != c]           # the plane's axes, ascending: storage [b][a] (a fastest)
                bb, aa = torch.meshgrid(lin, lin, indexing="ij")
                pos = torch.zeros(R * R, 3)
                pos[:, a], pos[:, b] = aa.reshape(-1), bb.reshape(-1)
                oh = torch.zeros(R * R, L + 3)
                oh[:, l], oh[:, L + c] = 1.0, 1.0
                self._plane_slices.append((sum(x.shape[0] for x in pv), R * R))
                pv.append(torch.cat([pos, oh], dim=-1))
                pos = torch.zeros(R, 3)
                pos[:, c] = lin
                oh = torch.zeros(R, L + 3)
                oh[:, l], oh[:, L + c] = 1.0, 1.0
                self._line_slices.append((sum(x.shape[0] for x in lv), R))
                lv.append(torch.cat([pos, oh], dim=-1))
        pv, lv = torch.cat(pv), torch.cat(lv)
        emb = lambda v, kind: torch.cat([sinusoidal_legacy(v[:, :3], n_frequencies), v[:, 3:],      # noqa: E731
                                         torch.tensor([[1.0, 0.0] if kind == 0 else [0.0, 1.0]]).expand(v.shape[0], 2)], dim=-1)
        self.register_buffer("plane_embedding", emb(pv, 0), persistent=False)
        self.register_buffer("line_embedding", emb(lv, 1), persistent=False)
        din = self.plane_embedding.shape[1]
        dims = [din] + [W] * D
        self.trunk = nn.ModuleList([FMMLinear(dims[i], dims[i + 1], self.z_dim, fmm_rank, g) for i in range(D)])
        hd = [W] * D_head + [self.lod_n_feats]
        self.plane_head = nn.ModuleList([FMMLinear(hd[i], hd[i + 1], self.z_dim, fmm_rank, g) for i in range(D_head)])
        self.line_head = nn.ModuleList([FMMLinear(hd[i], hd[i + 1], self.z_dim, fmm_rank, g) for i in range(D_head)])
        self.out_scale = float(out_scale)
        self.kernel_lod_res: List[int] = [r for r in self.lod_res for _ in range(self.lod_n_feats // 2)]
        self.n_params = sum(r ** 3 * self.lod_n_feats for r in self.lod_res)

    def planes_and_lines(self, z: torch.Tensor):
        """-> (planes [B, sum 3 R^2, F], lines [B, sum 3 R, F]) -- lines already as 1 + head output."""
        def run(x, head):
            h = x
            for lay in self.trunk:
                h = torch.relu(lay(h, z))
            for i, lay in enumerate(head):
                h = lay(h, z)
                if i < len(head) - 1:
                    h = torch.relu(h)
            return h * self.out_scale
        return run(self.plane_embedding, self.plane_head), 1.0 + run(self.line_embedding, self.line_head)

    def forward(self, z: torch.Tensor) -> torch.Tensor:
        B, F = z.shape[0], self.lod_n_feats
        P, Ln = self.planes_and_lines(z)
        out = []
        for l, R in enumerate(self.lod_res):
            T = None                                                        # [B, z, y, x, F]
            for c in range(3):
                p0, pn = self._plane_slices[3 * l + c]
                l0, ln = self._line_slices[3 * l + c]
                pl = P[:, p0:p0 + pn].view(B, R, R, F)                      # [B, b, a, F], (a, b) = the other axes ascending
                li = Ln[:, l0:l0 + ln]                                      # [B, R, F] along axis c
                if c == 0:      # plane over (y, z) = [z][y], line along x
                    t = pl[:, :, :, None, :] * li[:, None, None, :, :]
                elif c == 1:    # plane over (x, z) = [z][x], line along y
                    t = pl[:, :, None, :, :] * li[:, None, :, None, :]
                else:           # plane over (x, y) = [y][x], line along z
                    t = pl[:, None, :, :, :] * li[:, :, None, None, :]
                T = t if T is None else T + t
            T = T.reshape(B, R ** 3, F)
            for k in range(F // 2):
                out.append(T[:, :, 2 * k:2 * k + 2].reshape(B, -1))
        return torch.cat(out, dim=1)


class MixedLoTDGrower(nn.Module):
    """``MixedLoTDGrower{grower_configs: [...]}`` (no_fg_occ.221218.yaml:320-352): the levels of the member growers one
    after another, one table per instance."""

    def __init__(self, growers: Sequence[nn.Module]):
        super().__init__()
        self.growers = nn.ModuleList(growers)
        self.z_dim = int(growers[0].z_dim)
        assert all(int(g.z_dim) == self.z_dim for g in growers)
        self.lod_res = [r for g in growers for r in g.lod_res]
        self.kernel_lod_res: List[int] = [r for g in growers for r in g.kernel_lod_res]
        self.n_params = sum(int(g.n_params) for g in growers)

    def forward(self, z: torch.Tensor) -> torch.Tensor:
        return torch.cat([g(z) for g in self.growers], dim=1)


def build_grower(cfg: dict, z_dim: int = None, seed: int = 42) -> nn.Module:
    """A grower from its config: the native flat form (``DenseLoTDGrowerFMM`` keywords) or the reference's
    ``{target: nr3d_lib.models.grid_encodings.lotd.lotd_batched_growers.<Class>, param: {...}}`` block with
    ``pseudo_net_param{activation, fmm_rank, equal_lr, D, D_head, W, embed_cfg{type: sinusoidal_legacy, n_frequencies}}``
    (no_fg_occ.221218.yaml:319-352).  Options the restatement does not cover raise ``NotImplementedError`` naming the key."""
    cfg = dict(cfg)
    if "target" not in cfg:
        if z_dim is not None:
            cfg["z_dim"] = int(z_dim)
        return DenseLoTDGrowerFMM(seed=seed, **cfg)
    name = str(cfg["target"]).rsplit(".", 1)[-1]
    prm = dict(cfg.get("param") or {})
    if name == "MixedLoTDGrower":
        return MixedLoTDGrower([build_grower(c, z_dim, seed + 11 * i) for i, c in enumerate(prm["grower_configs"])])
    pn = dict(prm.get("pseudo_net_param") or {})
    if pn.get("activation", "relu") != "relu":
        raise NotImplementedError(f"pseudo_net_param.activation={pn.get('activation')!r}: relu")
    if pn.get("equal_lr", False):
        raise NotImplementedError("pseudo_net_param.equal_lr=True: plain (un-equalised) learning rates")
    ec = dict(pn.get("embed_cfg") or dict(type="sinusoidal_legacy", n_frequencies=6))
    if ec.get("type") != "sinusoidal_legacy":
        raise NotImplementedError(f"pseudo_net_param.embed_cfg.type={ec.get('type')!r}: sinusoidal_legacy")
    kw = dict(z_dim=int(z_dim if z_dim is not None else prm.get("z_dim", 128)), lod_res=list(prm["lod_res"]),
              lod_n_feats=int(prm.get("lod_n_feats", 4)), W=int(pn.get("W", 128)), fmm_rank=int(pn.get("fmm_rank", 10)),
              n_frequencies=int(ec.get("n_frequencies", 6)), seed=seed)
    if name == "DenseLoTDGrowerFMM":
        if prm.get("pseudo_net_type", "same") != "same":
            raise NotImplementedError(f"DenseLoTDGrowerFMM.pseudo_net_type={prm.get('pseudo_net_type')!r}: same")
        return DenseLoTDGrowerFMM(D=int(pn.get("D", 5)), **kw)
    if name == "VMSplitLoTDGrowerFMM":
        if prm.get("pseudo_net_type", "shared") != "shared":
            raise NotImplementedError(f"VMSplitLoTDGrowerFMM.pseudo_net_type={prm.get('pseudo_net_type')!r}: shared")
        return VMSplitLoTDGrowerFMM(D=int(pn.get("D", 4)), D_head=int(pn.get("D_head", 2)), **kw)
    raise NotImplementedError(f"lotd_grower_cfg.target={cfg['target']!r}")
