"""oracle.pack_ops -- CPU restatement of ``nr3d_lib.graphics.pack_ops`` / ``nr3d_lib.graphics.nerf``.

TEST INFRASTRUCTURE (see oracle/__init__.py).  The implementation of these ops is absent from
/root/reference (nr3d_lib un-vendored); semantics are taken from the reference's call sites:

* ``get_pack_infos_from_n``, ``interleave_linstep``, ``packed_sort`` (global indices):
  ``app/renderers/buffer_compose_renderer.py:972-1049`` (the only golden fixture), ``:649-694``
* ``packed_alpha_to_vw``, ``packed_sum``, ``packed_div``: ``app/renderers/single_volume_renderer.py:73-102``
* ``packed_matmul``: ``app/renderers/utils.py:17-29``
* ``packed_geq/leq/lt``: ``app/loss/lidar.py:102-110``
* ``packed_mean``: ``app/loss/ray_vw_entropy.py:32``
* ``merge_two_packs_sorted``: ``app/renderers/single_volume_renderer.py:337-349``

``pack_infos`` is ``LongTensor[P, 2] = (first index, count)`` of each pack in a flat ("packed")
sample array.  All ops are written with plain autograd-transparent torch ops.
"""
from typing import Tuple

import torch


def get_pack_infos_from_n(n: torch.Tensor) -> torch.Tensor:
    """[P] counts -> [P,2] (exclusive-cumsum start, count).  buffer_compose_renderer.py:991,1004."""
    n = n.long()
    cs = torch.cumsum(n, 0)
    return torch.stack([cs - n, n], dim=-1)


def pack_ridx(pack_infos: torch.Tensor, total: int = None) -> torch.Tensor:
    """Pack index of every packed element (elements must tile [0, total) in pack order)."""
    P = pack_infos.shape[0]
    return torch.repeat_interleave(torch.arange(P, device=pack_infos.device), pack_infos[:, 1],
                                   output_size=total)


def _scatter_index(pack_infos: torch.Tensor):
    """(ridx, local index) for every element covered by pack_infos, in pack order."""
    n = pack_infos[:, 1]
    total = int(n.sum())
    ridx = pack_ridx(pack_infos, total)
    starts_dense = torch.cumsum(n, 0) - n
    local = torch.arange(total, device=n.device) - starts_dense[ridx]
    gidx = pack_infos[ridx, 0] + local
    return ridx, local, gidx


def packed_sum(x: torch.Tensor, pack_infos: torch.Tensor) -> torch.Tensor:
    """Per-pack sum over dim 0.  single_volume_renderer.py:84-101."""
    P = pack_infos.shape[0]
    ridx, _, gidx = _scatter_index(pack_infos)
    out = x.new_zeros([P, *x.shape[1:]])
    return out.index_add(0, ridx, x[gidx])


def packed_mean(x: torch.Tensor, pack_infos: torch.Tensor) -> torch.Tensor:
    n = pack_infos[:, 1].clamp_min(1).to(x.dtype)
    s = packed_sum(x, pack_infos)
    return s / n.view(-1, *[1] * (x.dim() - 1))


def _expand(per_pack: torch.Tensor, pack_infos: torch.Tensor, like: torch.Tensor):
    ridx, _, gidx = _scatter_index(pack_infos)
    return ridx, gidx


def packed_div(x: torch.Tensor, per_pack: torch.Tensor, pack_infos: torch.Tensor) -> torch.Tensor:
    """x[s] / per_pack[pack(s)].  single_volume_renderer.py:86."""
    ridx, gidx = _expand(per_pack, pack_infos, x)
    d = per_pack[ridx]
    if x.dim() > d.dim():
        d = d.view(-1, *[1] * (x.dim() - d.dim()))
    out = torch.zeros_like(x)
    return out.index_copy(0, gidx, x[gidx] / d)


def packed_mul(x: torch.Tensor, per_pack: torch.Tensor, pack_infos: torch.Tensor) -> torch.Tensor:
    ridx, gidx = _expand(per_pack, pack_infos, x)
    d = per_pack[ridx]
    if x.dim() > d.dim():
        d = d.view(-1, *[1] * (x.dim() - d.dim()))
    out = torch.zeros_like(x)
    return out.index_copy(0, gidx, x[gidx] * d)


def packed_matmul(x: torch.Tensor, rot: torch.Tensor, pack_infos: torch.Tensor) -> torch.Tensor:
    """out[s] = rot[pack(s)] @ x[s]  (x [S,3], rot [P,3,3]).  app/renderers/utils.py:25-29:
    the batched branch is ``(rotation * nablas.unsqueeze(-2)).sum(-1)``; no mm/bmm (cameras.py:355-359)."""
    ridx, gidx = _expand(None, pack_infos, x)
    y = (rot[ridx] * x[gidx].unsqueeze(-2)).sum(-1)
    out = torch.zeros_like(x)
    return out.index_copy(0, gidx, y)


def _packed_cmp(x, per_pack, pack_infos, op):
    ridx, gidx = _expand(per_pack, pack_infos, x)
    out = torch.zeros(x.shape, dtype=torch.bool, device=x.device)
    out[gidx] = op(x[gidx], per_pack[ridx])
    return out


def packed_geq(x, per_pack, pack_infos):
    return _packed_cmp(x, per_pack, pack_infos, torch.ge)


def packed_leq(x, per_pack, pack_infos):
    return _packed_cmp(x, per_pack, pack_infos, torch.le)


def packed_lt(x, per_pack, pack_infos):
    return _packed_cmp(x, per_pack, pack_infos, torch.lt)


def packed_gt(x, per_pack, pack_infos):
    return _packed_cmp(x, per_pack, pack_infos, torch.gt)


def interleave_linstep(start: torch.Tensor, n: torch.Tensor, step=1, return_idx: bool = False):
    """concat_p( start[p] + step * arange(n[p]) ).  buffer_compose_renderer.py:1036."""
    n = n.long()
    total = int(n.sum())
    P = n.shape[0]
    ridx = torch.repeat_interleave(torch.arange(P, device=n.device), n, output_size=total)
    base = torch.cumsum(n, 0) - n
    local = torch.arange(total, device=n.device) - base[ridx]
    out = start[ridx] + local.to(start.dtype) * step
    return (out, ridx) if return_idx else out


def to_padded(x: torch.Tensor, pack_infos: torch.Tensor, fill=0.0):
    """packed [S,...] -> padded [P, maxn, ...] + bool mask [P, maxn]."""
    P = pack_infos.shape[0]
    maxn = int(pack_infos[:, 1].max()) if P > 0 else 0
    ridx, local, gidx = _scatter_index(pack_infos)
    pad = x.new_full([P, maxn, *x.shape[1:]], fill)
    pad = pad.index_put((ridx, local), x[gidx])
    mask = torch.zeros([P, maxn], dtype=torch.bool, device=x.device)
    mask[ridx, local] = True
    return pad, mask, (ridx, local, gidx)


def packed_sort(x: torch.Tensor, pack_infos: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Ascending stable sort inside each pack; returns (sorted values [S], GLOBAL source indices [S])
    such that ``sorted == x[indices]`` (buffer_compose_renderer.py:1045-1047)."""
    pad, mask, (ridx, local, gidx) = to_padded(x.detach(), pack_infos, fill=float('inf'))
    order = torch.argsort(pad, dim=1, stable=True)
    src_global = (pack_infos[:, 0:1] + order)[ridx, local]
    indices = torch.arange(x.shape[0], device=x.device)
    indices = indices.index_copy(0, gidx, src_global)
    return x[indices], indices


def merge_two_packs_sorted(vals_a, pack_infos_a, nidx_a, vals_b, pack_infos_b, nidx_b):
    """Merge two packed, per-pack ascending sample sets that live on (possibly different) rays.

    ``nidx_*`` = ray index of every pack (ascending, unique).  Returns ``pidx_a, pidx_b`` (position of
    every a / b element in the merged total buffer) and ``pack_infos`` [U,2] of the union of rays in
    ascending ray order (single_volume_renderer.py:340-349: rows correspond to
    ``total_num_samples_per_ray.nonzero()``).  Ties: a-elements first (stable).
    """
    dev = vals_a.device
    rays = torch.unique(torch.cat([nidx_a, nidx_b]))  # sorted
    U = rays.shape[0]
    slot_a = torch.searchsorted(rays, nidx_a)
    slot_b = torch.searchsorted(rays, nidx_b)
    n_tot = torch.zeros(U, dtype=torch.long, device=dev)
    n_tot.index_add_(0, slot_a, pack_infos_a[:, 1]).index_add_(0, slot_b, pack_infos_b[:, 1])
    pack_infos = get_pack_infos_from_n(n_tot)
    ra, la, ga = _scatter_index(pack_infos_a)
    rb, lb, gb = _scatter_index(pack_infos_b)
    key_ray = torch.cat([slot_a[ra], slot_b[rb]])
    key_val = torch.cat([vals_a.detach()[ga], vals_b.detach()[gb]])
    # stable lexicographic sort by (ray, value); a-elements precede b-elements on ties
    o1 = torch.argsort(key_val, stable=True)
    o2 = torch.argsort(key_ray[o1], stable=True)
    order = o1[o2]
    pos = torch.empty_like(order)
    pos[order] = torch.arange(order.shape[0], device=dev)
    na = ga.shape[0]
    pidx_a = torch.empty(vals_a.shape[0], dtype=torch.long, device=dev)
    pidx_b = torch.empty(vals_b.shape[0], dtype=torch.long, device=dev)
    pidx_a[ga] = pos[:na]
    pidx_b[gb] = pos[na:]
    return pidx_a, pidx_b, pack_infos


# ----------------------------------------------------------------------------- graphics.nerf
def packed_alpha_to_vw(alpha: torch.Tensor, pack_infos: torch.Tensor) -> torch.Tensor:
    """Visibility weights vw_i = alpha_i * prod_{j<i}(1 - alpha_j + 1e-10) inside each pack.
    single_volume_renderer.py:79-83; inspect_rendering.py:222-225."""
    pad, mask, (ridx, local, gidx) = to_padded(alpha, pack_infos, fill=0.0)
    shifted = torch.cat([torch.ones_like(pad[:, :1]), 1.0 - pad[:, :-1] + 1e-10], dim=1)
    trans = torch.cumprod(shifted, dim=1)
    vw_pad = pad * trans
    out = torch.zeros_like(alpha)
    return out.index_copy(0, gidx, vw_pad[ridx, local])


def ray_alpha_to_vw(alpha: torch.Tensor) -> torch.Tensor:
    """Batched [..., n] variant (single_volume_renderer.py:76-78)."""
    shifted = torch.cat([torch.ones_like(alpha[..., :1]), 1.0 - alpha[..., :-1] + 1e-10], dim=-1)
    return alpha * torch.cumprod(shifted, dim=-1)
