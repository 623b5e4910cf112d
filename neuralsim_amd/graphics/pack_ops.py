"""Host side of the packed ("segmented") tensor ops -- mirrors ``nr3d_lib.graphics.pack_ops`` as the
reference calls it (app/renderers/single_volume_renderer.py:20,73-102,337-349;
app/renderers/buffer_compose_renderer.py:33,644-723; app/renderers/utils.py:15-29; app/loss/lidar.py:17,102-110).

Every op is a thin ``torch.autograd.Function`` over the HIP kernels of ``csrc/pack_ops.hip`` (C ABI:
include/nsim.h).  ``pack_infos`` is ``LongTensor[P,2] = (first index, count)``.
"""
import ctypes as C

import torch

from .. import _lib

__all__ = ["get_pack_infos_from_n", "packed_sum", "packed_mean", "packed_div", "packed_mul", "packed_add",
           "packed_sub", "packed_matmul", "packed_sort", "packed_geq", "packed_leq", "packed_lt", "packed_gt",
           "interleave_linstep", "merge_two_packs_sorted", "packed_alpha_to_vw"]


def _f32c(t):
    return t.float().contiguous()


def get_pack_infos_from_n(n: torch.Tensor, return_total: bool = False, cap: int = -1, notify=None):
    """[P] counts -> [P,2] (exclusive cumsum, count).  buffer_compose_renderer.py:991,1004.
    ``cap >= 0`` (internal, speculative buffer sizes): packs ending beyond cap get count 0; total is the true sum.
    ``notify`` = (address, seq) of a ``_lib.HostNotify`` slot: the kernel also stores (total, seq) there."""
    n = n.long().contiguous()
    P = n.shape[0]
    pi = torch.empty([P, 2], dtype=torch.long, device=n.device)
    total = torch.empty([1], dtype=torch.long, device=n.device) if P > 0 else \
        torch.zeros([1], dtype=torch.long, device=n.device)          # the kernel always writes total[0]
    if P > 0 and notify is not None:
        _lib.call("nsim_pack_infos_from_n_notify", _lib.ptr(n), P, _lib.ptr(pi), _lib.ptr(total), int(cap),
                  int(notify[0]), int(notify[1]))
    elif P > 0:
        _lib.call("nsim_pack_infos_from_n", _lib.ptr(n), P, _lib.ptr(pi), _lib.ptr(total), int(cap))
    elif notify is not None:
        # no pack, no launch: the armed slot gets its (0, seq) from the host, or a waiter would spin into its timeout
        import ctypes
        w = (ctypes.c_int64 * 2).from_address(int(notify[0]))
        w[0], w[1] = 0, int(notify[1])
    return (pi, total) if return_total else pi


def _as2d(x):
    S = x.shape[0]
    C = 1
    for v in x.shape[1:]:
        C *= int(v)
    return x.reshape(S, C), x.shape[1:]        # explicit C: reshape(0, -1) is ambiguous for an empty buffer


def _binary(x2, per_pack2, pack_infos, op):
    """x2 [S,C] or None, per_pack2 [P,Cp] -> [S,C]."""
    P = pack_infos.shape[0]
    if x2 is None:
        Cc = per_pack2.shape[1]
        S = int(pack_infos[-1].sum()) if P > 0 else 0
    else:
        S, Cc = x2.shape
    out = torch.empty([S, Cc], dtype=torch.float32, device=per_pack2.device)
    _lib.call("nsim_packed_binary", _lib.ptr(x2), Cc, _lib.ptr(per_pack2), per_pack2.shape[1], _lib.ptr(pack_infos), P,
              op, _lib.ptr(out))
    return out


def _sum2d(x2, pack_infos):
    P = pack_infos.shape[0]
    out = torch.zeros([P, x2.shape[1]], dtype=torch.float32, device=x2.device)
    _lib.call("nsim_packed_sum", _lib.ptr(x2), x2.shape[1], _lib.ptr(pack_infos), P, _lib.ptr(out))
    return out


class _PackedSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pack_infos):
        x2, tail = _as2d(_f32c(x))
        pack_infos = pack_infos.contiguous()
        ctx.save_for_backward(pack_infos)
        ctx.S, ctx.tail = x.shape[0], tail
        return _sum2d(x2, pack_infos).reshape(pack_infos.shape[0], *tail)

    @staticmethod
    def backward(ctx, g):
        (pack_infos,) = ctx.saved_tensors
        g2 = _f32c(g).reshape(pack_infos.shape[0], max(1, g.numel() // max(1, pack_infos.shape[0])))
        dx = torch.zeros([ctx.S, g2.shape[1]], dtype=torch.float32, device=g.device)
        _lib.call("nsim_packed_binary", None, g2.shape[1], _lib.ptr(g2), g2.shape[1], _lib.ptr(pack_infos),
                  pack_infos.shape[0], 0, _lib.ptr(dx))
        return dx.reshape(ctx.S, *ctx.tail), None


def packed_sum(x: torch.Tensor, pack_infos: torch.Tensor) -> torch.Tensor:
    """Per-pack sum over dim 0 (single_volume_renderer.py:84-101)."""
    return _PackedSum.apply(x, pack_infos)


def packed_mean(x: torch.Tensor, pack_infos: torch.Tensor) -> torch.Tensor:
    n = pack_infos[:, 1].clamp_min(1).to(torch.float32)
    s = packed_sum(x, pack_infos)
    return s / n.view(-1, *[1] * (x.dim() - 1))


class _PackedBinary(torch.autograd.Function):
    """out[s] = x[s] (op) per_pack[pack(s)]; op 0 mul, 1 div, 2 add, 3 sub."""

    @staticmethod
    def forward(ctx, x, per_pack, pack_infos, op):
        x2, tail = _as2d(_f32c(x))
        P = pack_infos.shape[0]
        pp2 = _f32c(per_pack).reshape(P, max(1, per_pack.numel() // max(1, P)))
        pack_infos = pack_infos.contiguous()
        out = torch.zeros_like(x2)
        _lib.call("nsim_packed_binary", _lib.ptr(x2), x2.shape[1], _lib.ptr(pp2), pp2.shape[1], _lib.ptr(pack_infos), P,
                  op, _lib.ptr(out))
        ctx.save_for_backward(x2, pp2, out, pack_infos)
        ctx.op, ctx.xshape, ctx.ppshape = op, x.shape, per_pack.shape
        return out.reshape(x.shape)

    @staticmethod
    def backward(ctx, g):
        x2, pp2, out, pack_infos = ctx.saved_tensors
        op = ctx.op
        g2 = _f32c(g).reshape(x2.shape)
        P = pack_infos.shape[0]
        dx = dpp = None
        if ctx.needs_input_grad[0]:
            if op in (0, 1):
                dx = torch.zeros_like(x2)
                _lib.call("nsim_packed_binary", _lib.ptr(g2), g2.shape[1], _lib.ptr(pp2), pp2.shape[1],
                          _lib.ptr(pack_infos), P, op, _lib.ptr(dx))
            else:
                dx = g2
            dx = dx.reshape(ctx.xshape)
        if ctx.needs_input_grad[1]:
            if op == 0:
                t = g2 * x2
            elif op == 1:      # d(x/p)/dp = -(x/p)/p
                t = -(g2 * out)
            elif op == 2:
                t = g2
            else:
                t = -g2
            s = _sum2d(t.contiguous(), pack_infos)
            if pp2.shape[1] == 1 and s.shape[1] != 1:
                s = s.sum(dim=1, keepdim=True)
            if op == 1:
                s = s / pp2
            dpp = s.reshape(ctx.ppshape)
        return dx, dpp, None, None


def packed_div(x, per_pack, pack_infos):
    """x[s] / per_pack[pack(s)]  (single_volume_renderer.py:86)."""
    return _PackedBinary.apply(x, per_pack, pack_infos, 1)


def packed_mul(x, per_pack, pack_infos):
    return _PackedBinary.apply(x, per_pack, pack_infos, 0)


def packed_add(x, per_pack, pack_infos):
    return _PackedBinary.apply(x, per_pack, pack_infos, 2)


def packed_sub(x, per_pack, pack_infos):
    return _PackedBinary.apply(x, per_pack, pack_infos, 3)


def _packed_cmp(x, per_pack, pack_infos, op):
    x = _f32c(x.detach())
    pp = _f32c(per_pack.detach())
    out = torch.zeros(x.shape, dtype=torch.uint8, device=x.device)
    _lib.call("nsim_packed_cmp", _lib.ptr(x), _lib.ptr(pp), _lib.ptr(pack_infos.contiguous()), pack_infos.shape[0], op,
              _lib.ptr(out))
    return out.bool()


def packed_geq(x, per_pack, pack_infos):
    """app/loss/lidar.py:104"""
    return _packed_cmp(x, per_pack, pack_infos, 0)


def packed_leq(x, per_pack, pack_infos):
    return _packed_cmp(x, per_pack, pack_infos, 1)


def packed_lt(x, per_pack, pack_infos):
    return _packed_cmp(x, per_pack, pack_infos, 2)


def packed_gt(x, per_pack, pack_infos):
    return _packed_cmp(x, per_pack, pack_infos, 3)


class _PackedMatmul3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, rot, pack_infos):
        x = _f32c(x)
        rot = _f32c(rot)
        pack_infos = pack_infos.contiguous()
        out = torch.zeros_like(x)
        _lib.call("nsim_packed_matmul3", _lib.ptr(x), _lib.ptr(rot), _lib.ptr(pack_infos), pack_infos.shape[0], 0,
                  _lib.ptr(out))
        ctx.save_for_backward(x, rot, pack_infos)
        return out

    @staticmethod
    def backward(ctx, g):
        x, rot, pack_infos = ctx.saved_tensors
        g = _f32c(g)
        dx = drot = None
        if ctx.needs_input_grad[0]:
            dx = torch.zeros_like(x)
            _lib.call("nsim_packed_matmul3", _lib.ptr(g), _lib.ptr(rot), _lib.ptr(pack_infos), pack_infos.shape[0], 1,
                      _lib.ptr(dx))
        if ctx.needs_input_grad[1]:
            outer = (g.unsqueeze(-1) * x.unsqueeze(-2)).reshape(-1, 9).contiguous()
            drot = _sum2d(outer, pack_infos).reshape(-1, 3, 3)
        return dx, drot, None


def packed_matmul(x, rot, pack_infos):
    """out[s] = rot[pack(s)] @ x[s]  (app/renderers/utils.py:25)."""
    return _PackedMatmul3.apply(x, rot, pack_infos)


class _PermuteFn(torch.autograd.Function):
    """``x[perm]`` for a PERMUTATION ``perm`` of range(len(x)) (``packed_sort``'s indices): the backward is the inverse
    permutation, a plain scatter -- torch's ``x[idx]`` backward is an ``index_put_(accumulate=True)`` that sorts the
    indices first (0.7 ms per [2 M, 3] buffer of the multi-object step)."""

    @staticmethod
    def forward(ctx, x, perm):
        ctx.save_for_backward(perm)
        return x[perm]

    @staticmethod
    def backward(ctx, g):
        (perm,) = ctx.saved_tensors
        out = torch.empty_like(g)
        out[perm] = g
        return out, None


def permute_rows(x: torch.Tensor, perm: torch.Tensor) -> torch.Tensor:
    """x[perm] where ``perm`` is a permutation (see ``_PermuteFn``)."""
    return _PermuteFn.apply(x, perm) if x.requires_grad else x[perm]


def inverse_permutation(perm: torch.Tensor) -> torch.Tensor:
    """ranks with ``ranks[perm[i]] = i`` (= ``torch.sort(perm).indices`` without the sort)."""
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(perm.shape[0], device=perm.device, dtype=perm.dtype)
    return inv


def packed_sort(x: torch.Tensor, pack_infos: torch.Tensor):
    """-> (sorted [S], GLOBAL indices [S]) with ``sorted == x[indices]`` (buffer_compose_renderer.py:1043-1047)."""
    xd = _f32c(x.detach())
    pack_infos = pack_infos.contiguous()
    sorted_ = torch.zeros_like(xd)
    idx = torch.zeros(xd.shape, dtype=torch.long, device=x.device)
    _lib.call("nsim_packed_sort", _lib.ptr(xd), _lib.ptr(pack_infos), pack_infos.shape[0], _lib.ptr(sorted_),
              _lib.ptr(idx))
    if x.requires_grad:
        sorted_ = permute_rows(x, idx)
    return sorted_, idx


def compose_collect_sort(sources, total_pack_infos: torch.Tensor, S: int):
    """The collect + sort steps of ``BufferComposeRenderer.ray_query`` (buffer_compose_renderer.py:648-695) in one launch
    (``nsim_compose_collect_sort``).  ``sources``: list of (t [S_k], rays_inds [P_k] ascending, pack_infos [P_k, 2]) in
    collect order; ``total_pack_infos`` [N, 2]: every ray's (start, count) in the merged buffer of S samples.
    -> (t_sorted [S], [dst_k [S_k]]): the depths of every ray in order (ties in collect order: the reference's stable sort of
    the concatenation) and, per source, each sample's position in that order."""
    dev = total_pack_infos.device
    t_sorted = torch.empty([S], dtype=torch.float32, device=dev)
    K = len(sources)
    arr = (_lib.ComposeSrc * max(K, 1))()
    keep, dsts = [], []
    for k, (t, ric, pi) in enumerate(sources):
        td, ric, pi = _f32c(t.detach()).reshape(-1), ric.long().contiguous(), pi.long().contiguous()
        dst = torch.empty([td.shape[0]], dtype=torch.long, device=dev)
        p_t, p_r, p_p, p_d = _lib._marshal((td, ric, pi, dst))
        arr[k].t, arr[k].rays_inds, arr[k].pack_infos, arr[k].P, arr[k].dst = p_t, p_r, p_p, int(ric.shape[0]), p_d
        keep.append((td, ric, pi))
        dsts.append(dst)
    tpi = total_pack_infos.long().contiguous()
    _lib.call("nsim_compose_collect_sort", C.cast(arr, C.c_void_p), K, _lib.ptr(tpi), int(tpi.shape[0]), _lib.ptr(t_sorted))
    return t_sorted, dsts


def interleave_linstep(start: torch.Tensor, n: torch.Tensor, step=1, return_idx: bool = False):
    """concat_p(start[p] + step*arange(n[p]))  (buffer_compose_renderer.py:668,1036)."""
    assert not torch.is_floating_point(start), "interleave_linstep: integer start expected"
    pi, total = get_pack_infos_from_n(n, return_total=True)
    S = int(total.item())
    out = torch.empty([S], dtype=torch.long, device=start.device)
    _lib.call("nsim_interleave_linstep", _lib.ptr(start.long().contiguous()), _lib.ptr(pi), pi.shape[0], int(step),
              _lib.ptr(out))
    if return_idx:
        ridx = torch.repeat_interleave(torch.arange(n.shape[0], device=n.device), n.long(), output_size=S)
        return out, ridx
    return out


def merge_two_packs_sorted(vals_a, pack_infos_a, nidx_a, vals_b, pack_infos_b, nidx_b, a_is_arange: bool = False):
    """single_volume_renderer.py:341-344 -> (pidx_a, pidx_b, pack_infos) ; a-first on ties.
    ``a_is_arange``: the caller guarantees nidx_a == arange(P_a) and nidx_b is a subset of it (the renderer's
    distant-model buffer covers every ray) -- skips the ``torch.unique`` (a host sync) of the general path."""
    dev = vals_a.device
    if a_is_arange:
        U = nidx_a.shape[0]
        slot_a, slot_b = nidx_a.contiguous(), nidx_b.contiguous()
    else:
        rays = torch.unique(torch.cat([nidx_a, nidx_b]))
        U = rays.shape[0]
        slot_a = torch.searchsorted(rays, nidx_a.contiguous()).contiguous()
        slot_b = torch.searchsorted(rays, nidx_b.contiguous()).contiguous()
    n_tot = torch.zeros(U, dtype=torch.long, device=dev)
    n_tot.index_add_(0, slot_a, pack_infos_a[:, 1]).index_add_(0, slot_b, pack_infos_b[:, 1])
    pi = get_pack_infos_from_n(n_tot)
    pidx_a = torch.empty(vals_a.shape[0], dtype=torch.long, device=dev)
    pidx_b = torch.empty(vals_b.shape[0], dtype=torch.long, device=dev)
    _lib.call("nsim_merge_two_packs", _lib.ptr(_f32c(vals_a.detach())), _lib.ptr(pack_infos_a.contiguous()),
              _lib.ptr(slot_a), pack_infos_a.shape[0], _lib.ptr(_f32c(vals_b.detach())),
              _lib.ptr(pack_infos_b.contiguous()), _lib.ptr(slot_b), pack_infos_b.shape[0], _lib.ptr(pi), U,
              _lib.ptr(pidx_a), _lib.ptr(pidx_b))
    return pidx_a, pidx_b, pi


class _AlphaToVw(torch.autograd.Function):
    @staticmethod
    def forward(ctx, alpha, pack_infos):
        alpha = _f32c(alpha)
        pack_infos = pack_infos.contiguous()
        vw = torch.zeros_like(alpha)
        trans = torch.ones_like(alpha)
        _lib.call("nsim_alpha_to_vw_fwd", _lib.ptr(alpha), _lib.ptr(pack_infos), pack_infos.shape[0], _lib.ptr(vw),
                  _lib.ptr(trans))
        ctx.save_for_backward(alpha, trans, vw, pack_infos)
        return vw

    @staticmethod
    def backward(ctx, g):
        alpha, trans, vw, pack_infos = ctx.saved_tensors
        dalpha = torch.zeros_like(alpha)
        _lib.call("nsim_alpha_to_vw_bwd", _lib.ptr(alpha), _lib.ptr(trans), _lib.ptr(vw), _lib.ptr(_f32c(g)),
                  _lib.ptr(pack_infos), pack_infos.shape[0], _lib.ptr(dalpha))
        return dalpha, None


def packed_alpha_to_vw(alpha: torch.Tensor, pack_infos: torch.Tensor) -> torch.Tensor:
    """nr3d_lib.graphics.nerf.packed_alpha_to_vw (single_volume_renderer.py:79-83)."""
    shape = alpha.shape
    return _AlphaToVw.apply(alpha.reshape(-1), pack_infos).reshape(shape)
