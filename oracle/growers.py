"""oracle.growers -- CPU restatement of the latent -> dense-LoTD-table generator of the shared foreground model.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED: ``nr3d_lib.models.grid_encodings.lotd.lotd_batched_growers``
is absent; the function is restated from the reference's config
(code_multi/configs/exps/fg_neus=hyper_lotd/no_fg_occ.221218.yaml:320-337: ``DenseLoTDGrowerFMM{z_dim 128, lod_res
[5, 8, 13, 21], lod_n_feats 4, pseudo_net_param{activation relu, fmm_rank 10, D 5, W 128, embed_cfg{sinusoidal_legacy,
n_frequencies 6}}}``) and its call site (``set_condition``, app/models/shared/batched_neus.py:380-403).

Function (conventions fixed here, mirrored by neuralsim_amd/grid_encodings/lotd_growers.py): for an instance with code z,
the feature vector of the vertex at position p (in [-1, 1]^3) of grown level l is  out_scale * N_z([emb(p), onehot(l)])
where emb(p) = [p, sin(2^k p), cos(2^k p)] (k < n_frequencies) and N_z is an MLP with relu hidden layers whose i-th
weight matrix is  W_i o (U_i(z) V_i(z)^T),  U_i(z) = reshape(Au_i z + bu_i, [out, rank]),  V_i(z) likewise [in, rank].
Table layout: per grown level, lod_n_feats / 2 consecutive kernel levels of the same resolution holding feature pairs.
"""
from typing import List, Sequence

import torch


def vertex_inputs(lod_res: Sequence[int], n_frequencies: int) -> List[torch.Tensor]:
    """per level: [R^3, 3 (1 + 2 n_frequencies) + L] network inputs, vertices in storage order (x fastest)."""
    L = len(lod_res)
    out = []
    for l, R in enumerate(lod_res):
        rows = []
        for iz in range(R):
            for iy in range(R):
                for ix in range(R):
                    rows.append([-1.0 + 2.0 * ix / (R - 1), -1.0 + 2.0 * iy / (R - 1), -1.0 + 2.0 * iz / (R - 1)])
        p = torch.tensor(rows, dtype=torch.float32)
        feats = [p]
        for k in range(n_frequencies):
            feats += [torch.sin(p * float(2 ** k)), torch.cos(p * float(2 ** k))]
        oh = torch.zeros(p.shape[0], L)
        oh[:, l] = 1.0
        out.append(torch.cat(feats + [oh], dim=-1))
    return out


def grow_tables(z: torch.Tensor, layers: List[dict], lod_res: Sequence[int], lod_n_feats: int, n_frequencies: int,
                rank: int, out_scale: float) -> torch.Tensor:
    """z [B, z_dim]; layers = [dict(weight [out,in], bias [out], u_w [out*rank, z], u_b, v_w [in*rank, z], v_b)] ->
    [B, n_params] tables in the kernels' layout."""
    inputs = vertex_inputs(lod_res, n_frequencies)
    tables = []
    for b in range(z.shape[0]):
        zb = z[b]
        weights = []
        for lay in layers:
            out_f, in_f = lay["weight"].shape
            U = (lay["u_w"] @ zb + lay["u_b"]).view(out_f, rank)
            V = (lay["v_w"] @ zb + lay["v_b"]).view(in_f, rank)
            weights.append(lay["weight"] * (U @ V.t()))
        chunks = []
        for x in inputs:
            h = x
            for i, (lay, W) in enumerate(zip(layers, weights)):
                h = h @ W.t() + lay["bias"]
                if i < len(layers) - 1:
                    h = torch.relu(h)
            h = h * out_scale
            for c in range(lod_n_feats // 2):
                chunks.append(h[:, 2 * c:2 * c + 2].reshape(-1))
        tables.append(torch.cat(chunks))
    return torch.stack(tables)


# ------------------------------------------------------------------------------------------- VM-split levels (round 4)
def _fmm_mlp(x: torch.Tensor, zb: torch.Tensor, layers: List[dict], rank: int, relu_last: bool) -> torch.Tensor:
    h = x
    for i, lay in enumerate(layers):
        out_f, in_f = lay["weight"].shape
        U = (lay["u_w"] @ zb + lay["u_b"]).view(out_f, rank)
        V = (lay["v_w"] @ zb + lay["v_b"]).view(in_f, rank)
        h = h @ (lay["weight"] * (U @ V.t())).t() + lay["bias"]
        if relu_last or i < len(layers) - 1:
            h = torch.relu(h)
    return h


def vm_vertex_inputs(lod_res: Sequence[int], n_frequencies: int):
    """-> per (level, axis c): (plane inputs [R^2, .] in storage order [b][a] over the two other axes a < b, line inputs
    [R, .] along c).  Input = [emb(position with the collapsed axes at 0), onehot(level), onehot(axis), onehot(kind)]."""
    L = len(lod_res)

    def emb(p):
        feats = [p]
        for k in range(n_frequencies):
            feats += [torch.sin(p * float(2 ** k)), torch.cos(p * float(2 ** k))]
        return torch.cat(feats, dim=-1)
    out = []
    for l, R in enumerate(lod_res):
        for c in range(3):
            a, b = [ax for ax in range(3) if ax != c]
            rows = []
            for ib in range(R):
                for ia in range(R):
                    p = [0.0, 0.0, 0.0]
                    p[a], p[b] = -1.0 + 2.0 * ia / (R - 1), -1.0 + 2.0 * ib / (R - 1)
                    rows.append(p)
            pp = torch.tensor(rows, dtype=torch.float32)
            tag = torch.zeros(pp.shape[0], L + 3 + 2)
            tag[:, l], tag[:, L + c], tag[:, L + 3] = 1.0, 1.0, 1.0
            plane_in = torch.cat([emb(pp), tag], dim=-1)
            lp = torch.zeros(R, 3)
            lp[:, c] = torch.tensor([-1.0 + 2.0 * i / (R - 1) for i in range(R)])
            tag = torch.zeros(R, L + 3 + 2)
            tag[:, l], tag[:, L + c], tag[:, L + 4] = 1.0, 1.0, 1.0
            out.append((plane_in, torch.cat([emb(lp), tag], dim=-1)))
    return out


def grow_vm_tables(z: torch.Tensor, trunk: List[dict], plane_head: List[dict], line_head: List[dict], lod_res: Sequence[int],
                   lod_n_feats: int, n_frequencies: int, rank: int, out_scale: float, return_factors: bool = False):
    """Vector-matrix levels EXPANDED to dense vertex tables: vertex (ix, iy, iz) of level l holds
    sum_c plane_{l,c}[the two other indices] * line_{l,c}[index along c]  per feature, line = 1 + head output.
    -> [B, n_params] (per level lod_n_feats / 2 kernel levels of 2 features, x fastest); with ``return_factors`` also the
    per-instance list of (plane [R, R, F] indexed [b][a], line [R, F]) per (level, axis)."""
    inputs = vm_vertex_inputs(lod_res, n_frequencies)
    F = lod_n_feats
    tables, factors = [], []
    for bi in range(z.shape[0]):
        zb = z[bi]
        fac = []
        for plane_in, line_in in inputs:
            P = _fmm_mlp(_fmm_mlp(plane_in, zb, trunk, rank, True), zb, plane_head, rank, False) * out_scale
            Ln = 1.0 + _fmm_mlp(_fmm_mlp(line_in, zb, trunk, rank, True), zb, line_head, rank, False) * out_scale
            R = line_in.shape[0]
            fac.append((P.view(R, R, F), Ln))
        factors.append(fac)
        chunks = []
        for l, R in enumerate(lod_res):
            T = torch.zeros(R, R, R, F)                              # [iz][iy][ix]
            for iz in range(R):
                for iy in range(R):
                    for ix in range(R):
                        idx = (ix, iy, iz)
                        for c in range(3):
                            P, Ln = fac[3 * l + c]
                            a, b = [ax for ax in range(3) if ax != c]
                            T[iz, iy, ix] = T[iz, iy, ix] + P[idx[b], idx[a]] * Ln[idx[c]]
            T = T.view(R ** 3, F)
            for k in range(F // 2):
                chunks.append(T[:, 2 * k:2 * k + 2].reshape(-1))
        tables.append(torch.cat(chunks))
    tab = torch.stack(tables)
    return (tab, factors) if return_factors else tab


def vm_feature_at(x: torch.Tensor, fac_level, R: int) -> torch.Tensor:
    """The factorised form evaluated directly: sum_c bilinear(plane_c)(the two other coordinates) * linear(line_c)(x_c) at
    points x [N, 3] in [-1, 1]^3 -> [N, F].  (What a vector-matrix level IS; the dense expansion must reproduce it.)"""
    u = (x + 1.0) * 0.5 * (R - 1)
    i0 = u.floor().clamp(0, R - 2).long()
    w = u - i0.float()
    out = 0.0
    for c in range(3):
        P, Ln = fac_level[c]
        a, b = [ax for ax in range(3) if ax != c]
        pa0, pb0, wa, wb = i0[:, a], i0[:, b], w[:, a:a + 1], w[:, b:b + 1]
        plane = (P[pb0, pa0] * (1 - wa) * (1 - wb) + P[pb0, pa0 + 1] * wa * (1 - wb) + P[pb0 + 1, pa0] * (1 - wa) * wb
                 + P[pb0 + 1, pa0 + 1] * wa * wb)
        line = Ln[i0[:, c]] * (1 - w[:, c:c + 1]) + Ln[i0[:, c] + 1] * w[:, c:c + 1]
        out = out + plane * line
    return out
