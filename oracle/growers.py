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
