"""Packed ops + compositing kernels vs the oracle (oracle/pack_ops.py) and vs the reference's own fixture."""
import json
from pathlib import Path

import pytest
import torch

from oracle import pack_ops as opo
from neuralsim_amd.graphics import pack_ops as po
from neuralsim_amd.graphics.nerf import ray_alpha_to_vw

GOLD = Path(__file__).parent / "golden"


def leaf(t, dev=None, dtype=None):
    t = t.detach().clone()
    if dtype is not None:
        t = t.to(dtype)
    if dev is not None:
        t = t.to(dev)
    return t.requires_grad_(True)


def _ragged(seed, P=37, maxn=150, empty_every=5):
    g = torch.Generator().manual_seed(seed)
    n = torch.randint(1, maxn, (P,), generator=g)
    n[::empty_every] = 0
    n[3] = 64
    n[4] = 65
    n[6] = 1
    pi = opo.get_pack_infos_from_n(n)
    return n, pi, int(n.sum()), g


def test_pack_infos_from_n(backend):
    for P in (1, 7, 300, 5000):
        n = torch.randint(0, 9, (P,))
        pi = po.get_pack_infos_from_n(n.to(backend)).cpu()
        assert torch.equal(pi, opo.get_pack_infos_from_n(n))


def test_packed_sum_div_mean(backend):
    n, pi, S, g = _ragged(1)
    for C in ((), (3,)):
        x = torch.randn(S, *C, generator=g)
        xd = leaf(x, backend)
        xo = leaf(x)
        out = po.packed_sum(xd, pi.to(backend))
        ref = opo.packed_sum(xo, pi)
        assert torch.allclose(out.cpu(), ref, atol=1e-4, rtol=1e-5)
        w = torch.randn_like(ref)
        (out * w.to(backend)).sum().backward()
        (ref * w).sum().backward()
        assert torch.allclose(xd.grad.cpu(), xo.grad, atol=1e-6)
        m = po.packed_mean(x.to(backend), pi.to(backend)).cpu()
        assert torch.allclose(m, opo.packed_mean(x, pi), atol=1e-5)
    x = torch.randn(S, generator=g)
    pp = torch.rand(n.shape[0], generator=g) + 0.5
    xd, ppd = leaf(x, backend), leaf(pp, backend)
    xo, ppo = leaf(x), leaf(pp)
    out = po.packed_div(xd, ppd, pi.to(backend))
    ref = opo.packed_div(xo, ppo, pi)
    assert torch.allclose(out.cpu(), ref, atol=1e-6)
    w = torch.randn(S, generator=g)
    (out * w.to(backend)).sum().backward()
    (ref * w).sum().backward()
    assert torch.allclose(xd.grad.cpu(), xo.grad, atol=1e-6)
    assert torch.allclose(ppd.grad.cpu(), ppo.grad, atol=1e-4, rtol=1e-4)


def test_packed_cmp_matmul(backend):
    n, pi, S, g = _ragged(2)
    x = torch.randn(S, generator=g)
    pp = torch.randn(n.shape[0], generator=g)
    for name in ("packed_geq", "packed_leq", "packed_lt"):
        a = getattr(po, name)(x.to(backend), pp.to(backend), pi.to(backend)).cpu()
        assert torch.equal(a, getattr(opo, name)(x, pp, pi))
    v = torch.randn(S, 3, generator=g)
    R = torch.randn(n.shape[0], 3, 3, generator=g)
    vd, Rd = leaf(v, backend), leaf(R, backend)
    vo, Ro = leaf(v), leaf(R)
    out = po.packed_matmul(vd, Rd, pi.to(backend))
    ref = opo.packed_matmul(vo, Ro, pi)
    assert torch.allclose(out.cpu(), ref, atol=1e-6)
    w = torch.randn(S, 3, generator=g)
    (out * w.to(backend)).sum().backward()
    (ref * w).sum().backward()
    assert torch.allclose(vd.grad.cpu(), vo.grad, atol=1e-6)
    assert torch.allclose(Rd.grad.cpu(), Ro.grad, atol=1e-4)


def test_alpha_to_vw_fwd_bwd(backend):
    n, pi, S, g = _ragged(3, maxn=300)
    alpha = torch.rand(S, generator=g) * 0.3
    alpha[::17] = 0.0
    alpha[5::41] = 1.0
    ad = leaf(alpha, backend)
    ao = leaf(alpha, dtype=torch.double)
    vw = po.packed_alpha_to_vw(ad, pi.to(backend))
    ref = opo.packed_alpha_to_vw(ao, pi)
    assert torch.allclose(vw.cpu().double(), ref, atol=1e-6)
    w = torch.randn(S, generator=g)
    (vw * w.to(backend)).sum().backward()
    (ref * w.double()).sum().backward()
    # every sample, the alpha == 1 ones included (their derivative carries the 1 / (1 - alpha + 1e-10) factor)
    denom = ao.grad.abs().clamp_min(1.0)
    assert ((ad.grad.cpu().double() - ao.grad).abs() / denom).max() < 1e-4
    b = torch.rand(5, 33, generator=g)
    assert torch.allclose(ray_alpha_to_vw(b.to(backend)).cpu(), opo.ray_alpha_to_vw(b), atol=1e-6)


@pytest.mark.parametrize("fused", [False, True])
def test_alpha_to_vw_backward_behind_opaque_samples(backend, fused):
    """Opaque rays: alpha saturates to exactly 1 in the middle of a pack (a converged surface, or the distant model's
    ``include_inf_distance`` shell followed by nothing) and the later samples carry weights of 1e-10 and below.  The
    backward divides the sum over the LATER samples by 1 - alpha + 1e-10, so that sum has to be a true suffix sum
    (accumulated from the back): total-minus-prefix leaves a rounding residue of ~1e-7 that the division turns into
    gradients of ~1e3.  Checked on every sample -- the saturated ones included -- against the f64 oracle, for the
    stand-alone op and the fused compositing."""
    from neuralsim_amd.fields.neus import volume_integration
    from oracle import render as orr
    g = torch.Generator().manual_seed(11)
    n = torch.tensor([36, 90, 1, 64, 65, 200])
    pi = opo.get_pack_infos_from_n(n)
    S = int(n.sum())
    alpha = torch.rand(S, generator=g) * 0.4
    for st, k in pi.tolist():
        if k > 8:
            alpha[st + k // 2 - 2: st + k // 2 + 1] = torch.tensor([0.97, 1.0, 1.0])       # the surface
            alpha[st + k - 1] = 1.0                                                        # the far shell
    t = torch.rand(S, generator=g).cumsum(0)
    rgb = torch.rand(S, 3, generator=g)
    ad, ao = leaf(alpha, backend), leaf(alpha, dtype=torch.double)
    wm, wd, wr = torch.randn(6, generator=g), torch.randn(6, generator=g), torch.randn(6, 3, generator=g)
    if fused:
        out = volume_integration(ad, t.to(backend), rgb.to(backend), None, pi.to(backend), False)
        ref = orr.volume_integration(ao, t.double(), rgb.double(), None, pi, False)
        dv = lambda x: x.to(backend)        # noqa: E731
        (out["mask_volume"] * dv(wm) + out["depth_volume"] * dv(wd) + (out["rgb_volume"] * dv(wr)).sum(-1)).sum().backward()
        (ref["mask_volume"] * wm + ref["depth_volume"] * wd + (ref["rgb_volume"] * wr).sum(-1)).sum().backward()
    else:
        vw, vo = po.packed_alpha_to_vw(ad, pi.to(backend)), opo.packed_alpha_to_vw(ao, pi)
        w = torch.ones(S) * 0.7                 # a mask-loss gradient: the same on every sample of a ray (cancels exactly)
        w[: S // 2] = torch.randn(S // 2, generator=g)
        (vw * w.to(backend)).sum().backward()
        (vo * w.double()).sum().backward()
    err = (ad.grad.cpu().double() - ao.grad).abs() / ao.grad.abs().clamp_min(1.0)
    assert float(err.max()) < 1e-4, (float(err.max()), int(err.argmax()))
    assert float(ad.grad.abs().max()) < 10.0


def test_packed_sort_and_linstep(backend):
    n, pi, S, g = _ragged(4, maxn=90)
    x = torch.randn(S, generator=g)
    x[10:20] = x[10]  # ties
    s, idx = po.packed_sort(x.to(backend), pi.to(backend))
    s, idx = s.cpu(), idx.cpu()
    rs, ridx = opo.packed_sort(x, pi)
    assert torch.equal(s, x[idx])
    assert torch.equal(s, rs) and torch.equal(idx, ridx)
    start = torch.randint(0, 1000, (n.shape[0],))
    out = po.interleave_linstep(start.to(backend), n.to(backend), 2).cpu()
    assert torch.equal(out, opo.interleave_linstep(start, n, 2))


def test_packed_sort_of_concatenated_runs(backend):
    """The compose renderer's packs are concatenations of sorted runs (one per object buffer crossed by the ray): the
    kernel's identity path (one run), its run-merge path (<= 16 runs, ties inside and ACROSS runs) and its rank-sort paths
    (many runs; a pack beyond the LDS staging capacity) all give the oracle's stable order."""
    g = torch.Generator().manual_seed(21)
    packs = []
    packs.append(torch.sort(torch.rand(150, generator=g)).values)                                  # one run
    runs = [torch.sort(torch.rand(k, generator=g)).values for k in (64, 1, 37, 90)]
    runs[2][5:9] = runs[0][10]                                                                          # ties across runs
    runs[3][0] = runs[0][10]
    runs[2] = torch.sort(runs[2]).values
    runs[3] = torch.sort(runs[3]).values
    packs.append(torch.cat(runs))                                                                       # four runs
    packs.append(torch.cat([torch.sort(torch.rand(3, generator=g)).values for _ in range(16)]))     # exactly 16 runs
    packs.append(torch.rand(200, generator=g))                                                        # ~100 runs
    packs.append(torch.zeros(70))                                                                       # all equal
    packs.append(torch.cat([torch.sort(torch.rand(700, generator=g)).values for _ in range(2)]))    # beyond the capacity
    packs.append(torch.empty(0))
    packs.append(torch.tensor([0.5]))
    n = torch.tensor([q.shape[0] for q in packs])
    pi = opo.get_pack_infos_from_n(n)
    x = torch.cat(packs)
    s, idx = po.packed_sort(x.to(backend), pi.to(backend))
    rs, ridx = opo.packed_sort(x, pi)
    assert torch.equal(s.cpu(), rs) and torch.equal(idx.cpu(), ridx)


def test_merge_two_packs_sorted(backend):
    g = torch.Generator().manual_seed(5)
    na = torch.tensor([3, 0, 5, 70, 1])
    nb = torch.tensor([2, 4, 0, 66])
    ra = torch.tensor([0, 2, 3, 7, 9])
    rb = torch.tensor([2, 3, 5, 7])
    pia, pib = opo.get_pack_infos_from_n(na), opo.get_pack_infos_from_n(nb)
    va = opo.packed_sort(torch.rand(int(na.sum()), generator=g), pia)[0]
    vb = opo.packed_sort(torch.rand(int(nb.sum()), generator=g), pib)[0]
    vb[:2] = va[5:7]  # ties on the shared ray 2 / different packs are harmless
    vb, _ = opo.packed_sort(vb, pib)
    pa, pb, pi = po.merge_two_packs_sorted(va.to(backend), pia.to(backend), ra.to(backend), vb.to(backend),
                                           pib.to(backend), rb.to(backend))
    qa, qb, qi = opo.merge_two_packs_sorted(va, pia, ra, vb, pib, rb)
    assert torch.equal(pi.cpu(), qi) and torch.equal(pa.cpu(), qa) and torch.equal(pb.cpu(), qb)
    tot = torch.empty(int(qi[-1].sum()))
    tot[pa.cpu()], tot[pb.cpu()] = va, vb
    assert torch.equal(opo.packed_sort(tot, qi)[0], tot)


def test_reference_fixture_collect_and_merge(backend):
    """test_multi_buffer_collect_and_merge of app/renderers/buffer_compose_renderer.py:972-1049, replayed on
    get_pack_infos_from_n / interleave_linstep / packed_sort; expected values in tests/golden/pack_fixture.json
    (generated by tests/golden/make_pack_fixture.py from the reference's literal inputs)."""
    gold = json.loads((GOLD / "pack_fixture.json").read_text())
    dev = backend
    bufs = []
    for b in gold["buffers"]:
        d = dict(type=b["type"], rays_inds_hit=torch.tensor(b["rays_inds_hit"], device=dev),
                 t=torch.tensor(b["t"], device=dev, dtype=torch.float))
        if b["type"] == "batched":
            d["num_per_hit"] = b["num_per_hit"]
        else:
            d["pack_infos_hit"] = po.get_pack_infos_from_n(torch.tensor(b["n"], device=dev))
        bufs.append(d)
    total_num_rays = gold["total_num_rays"]
    ray_visible_samples = torch.zeros([total_num_rays], dtype=torch.long, device=dev)
    for vb in bufs:
        rih = vb["rays_inds_hit"]
        nph = torch.full_like(rih, vb["num_per_hit"]) if vb["type"] == "batched" else vb["pack_infos_hit"][:, 1]
        ric, cnt = torch.unique_consecutive(rih, return_counts=True)
        if (cnt > 1).any():
            nphc = torch.zeros([total_num_rays], device=dev, dtype=torch.long).index_add_(0, rih, nph)[ric]
            pic = po.get_pack_infos_from_n(nphc)
        else:
            nphc = nph
            pic = vb["pack_infos_hit"] if "pack_infos_hit" in vb else po.get_pack_infos_from_n(nph)
        ray_visible_samples.index_add_(0, ric, nphc)
        vb.update(rays_inds_collect=ric, pack_infos_collect=pic)
    assert ray_visible_samples.cpu().tolist() == gold["ray_visible_samples"]
    sparse = po.get_pack_infos_from_n(ray_visible_samples)
    hit = ray_visible_samples.nonzero().long()[..., 0]
    assert hit.cpu().tolist() == gold["total_rays_inds_hit"]
    total_pack_infos = sparse[hit]
    total = int(sparse[-1, :].sum().item())
    assert total == gold["total_num_samples"]
    depths = torch.zeros([total], dtype=torch.float, device=dev)
    cur = sparse[:, 0].clone()
    for vb in bufs:
        ric, pic = vb["rays_inds_collect"], vb["pack_infos_collect"]
        pidx = po.interleave_linstep(cur[ric], pic[:, 1], 1, False)
        depths[pidx] = vb["t"].flatten()
        cur.index_add_(0, ric, pic[:, 1])
    srt, indices = po.packed_sort(depths, total_pack_infos)
    assert torch.equal(srt, depths[indices])
    assert torch.allclose(srt.cpu(), torch.tensor(gold["sorted_depths"]), atol=0, rtol=0)
    # the same collect + sort as ONE launch (nsim_compose_collect_sort): the reference's sorted depths, and every buffer
    # sample's final position = ranks[pidx_in_total] of the three-step bookkeeping above
    t_sorted, dsts = po.compose_collect_sort([(vb["t"].flatten(), vb["rays_inds_collect"], vb["pack_infos_collect"]) for vb in bufs],
                                             sparse, total)
    assert torch.allclose(t_sorted.cpu(), torch.tensor(gold["sorted_depths"]), atol=0, rtol=0)
    ranks = po.inverse_permutation(indices)
    cur = sparse[:, 0].clone()
    for vb, dst in zip(bufs, dsts):
        ric, pic = vb["rays_inds_collect"], vb["pack_infos_collect"]
        pidx = po.interleave_linstep(cur[ric], pic[:, 1], 1, False)
        assert torch.equal(dst, ranks[pidx])
        cur.index_add_(0, ric, pic[:, 1])


@pytest.mark.parametrize("K", [1, 3, 9])
def test_compose_collect_sort_equals_the_three_step_bookkeeping(backend, K, poisoned_empty):
    """``nsim_compose_collect_sort`` against interleave_linstep + packed_sort + inverse permutation (the reference's collect and sort,
    buffer_compose_renderer.py:648-695) on random object buffers: sources missing on some rays, rays without samples, ties inside
    and across sources, sources that are concatenations of sorted runs (a batched model's regrouped packs), a ray beyond the
    staging capacity of the sort, more sources than fit one hand."""
    g = torch.Generator().manual_seed(100 + K)
    N = 41
    srcs, counts = [], torch.zeros([N], dtype=torch.long)
    for k in range(K):
        on = torch.rand(N, generator=g) < (0.7 if k else 0.9)
        on[5] = False                                   # a ray no source hits
        on[7] = True
        ric = on.nonzero()[:, 0]
        n = torch.randint(1, 40, (ric.shape[0],), generator=g)
        if k == 0:
            n[(ric == 7).nonzero()[0, 0]] = 1100          # ray 7: beyond PSORT_CAP
        pi = opo.get_pack_infos_from_n(n)
        t = torch.rand(int(n.sum()), generator=g)
        t = (t * 50).round() / 50 if k % 2 == 0 else t  # a coarse lattice: ties inside and across sources
        if k == 1:                                      # packs that are two sorted runs each
            half = opo.get_pack_infos_from_n(torch.stack([n // 2, n - n // 2], 1).flatten())
            t = opo.packed_sort(t, half)[0]
        else:
            t = opo.packed_sort(t, pi)[0]
        srcs.append((t, ric, pi))
        counts.index_add_(0, ric, n)
    dev = backend
    sparse, tot = po.get_pack_infos_from_n(counts.to(dev), return_total=True)
    S = int(tot.item())
    hit = counts.to(dev).nonzero()[:, 0]
    depths = torch.zeros([S], dtype=torch.float32, device=dev)
    cur = sparse[:, 0].clone()
    pidx = []
    for t, ric, pi in srcs:
        p_ = po.interleave_linstep(cur[ric.to(dev)], pi[:, 1].to(dev), 1)
        depths[p_] = t.to(dev)
        cur.index_add_(0, ric.to(dev), pi[:, 1].to(dev))
        pidx.append(p_)
    srt, indices = po.packed_sort(depths, sparse[hit])
    ranks = po.inverse_permutation(indices)
    t_sorted, dsts = po.compose_collect_sort([(t.to(dev), ric.to(dev), pi.to(dev)) for t, ric, pi in srcs], sparse, S)
    assert torch.equal(t_sorted, srt)
    for p_, dst in zip(pidx, dsts):
        assert torch.equal(dst, ranks[p_])


def test_pack_infos_large_and_capped(backend):
    """Block scan over several 1024-wide rounds and several 8192-wide super-chunks; speculative capacity semantics."""
    from neuralsim_amd.graphics import pack_ops as po
    g = torch.Generator().manual_seed(5)
    for P in (1, 63, 1024, 1025, 8192, 20011):
        n = torch.randint(0, 200, (P,), generator=g)
        pi, tot = po.get_pack_infos_from_n(n.to(backend), return_total=True)
        ref_start = torch.cumsum(n, 0) - n
        assert torch.equal(pi.cpu()[:, 0], ref_start) and torch.equal(pi.cpu()[:, 1], n) and int(tot) == int(n.sum())
        cap = int(n.sum()) // 2
        pic, totc = po.get_pack_infos_from_n(n.to(backend), return_total=True, cap=cap)
        pic = pic.cpu()
        end = ref_start + n
        assert int(totc) == int(n.sum())                                   # the true total, for the caller's check
        assert torch.equal(pic[:, 1], torch.where(end <= cap, n, torch.zeros_like(n)))
        assert torch.equal(pic[:, 0], ref_start.clamp_max(cap)) and int((pic[:, 0] + pic[:, 1]).max()) <= cap


def test_neus_composite_fwd_is_alpha_then_composite(backend):
    """``nsim_neus_composite_fwd`` (one launch: sdf -> alpha -> vw -> images) == nsim_neus_alpha_fwd + nsim_composite_fwd,
    bit for bit, on ragged packs with empty / one-sample / 64 / 65-sample rays."""
    from neuralsim_amd import _lib
    n, pi, S, g = _ragged(11)
    P = pi.shape[0]
    dev = backend
    f32 = dict(dtype=torch.float32, device=dev)
    sdf = (torch.rand(S, generator=g) * 0.2 - 0.1).to(dev)
    t = torch.rand(S, generator=g).to(dev)
    rgb, nrm = torch.rand(S, 3, generator=g).to(dev), torch.randn(S, 3, generator=g).to(dev)
    ln = torch.tensor([1.3], **f32)
    pid = pi.to(dev)
    out_idx = torch.randperm(P + 5, generator=g)[:P].to(dev)
    for fis, nd, oi in ((0.0, 1, None), (40.0, 0, out_idx)):
        res = []
        for fused in (False, True):
            alpha, vw, tr = (torch.full([S], -7.0, **f32) for _ in range(3))
            rows = P + 5
            m, dp = torch.zeros(rows, **f32), torch.zeros(rows, **f32)
            ro, no = torch.zeros(rows, 3, **f32), torch.zeros(rows, 3, **f32)
            if fused:
                _lib.call("nsim_neus_composite_fwd", _lib.ptr(sdf), _lib.ptr(ln), 2.0, fis, _lib.ptr(t), _lib.ptr(rgb),
                          _lib.ptr(nrm), _lib.ptr(pid), P, nd, _lib.ptr(alpha), _lib.ptr(vw), _lib.ptr(tr), _lib.ptr(m),
                          _lib.ptr(dp), _lib.ptr(ro), _lib.ptr(no), _lib.ptr(oi))
            else:
                _lib.call("nsim_neus_alpha_fwd", _lib.ptr(sdf), _lib.ptr(pid), P, _lib.ptr(ln), 2.0, fis, _lib.ptr(alpha))
                _lib.call("nsim_composite_fwd", _lib.ptr(alpha), _lib.ptr(t), _lib.ptr(rgb), _lib.ptr(nrm), _lib.ptr(pid), P,
                          nd, _lib.ptr(vw), _lib.ptr(tr), _lib.ptr(m), _lib.ptr(dp), _lib.ptr(ro), _lib.ptr(no), _lib.ptr(oi))
            res.append([x.cpu() for x in (alpha, vw, tr, m, dp, ro, no)])
        for a, b in zip(*res):
            assert torch.equal(a, b)


@pytest.mark.parametrize("every_ray", [False, True])
def test_render_head_equals_the_four_launches_it_replaces(backend, every_ray):
    """``nsim_render_head`` (one launch) vs nsim_neus_composite_fwd -> nsim_train_loss_head -> nsim_composite_bwd ->
    nsim_neus_alpha_bwd on random packs (empty packs, packs longer than a wave, rays outside the packs): same images, same
    three loss terms, same d alpha / d sdf / d rgb / d nablas / d ln_inv_s."""
    from neuralsim_amd import _lib
    from neuralsim_amd.graphics import pack_ops as po
    g = torch.Generator().manual_seed(5)
    N = 37
    R = N if every_ray else 23
    n = torch.randint(0, 150, (R,), generator=g)
    n[3], n[7] = 0, 1
    S, M = int(n.sum()), 11
    St = S + M
    dev = backend
    f32 = dict(dtype=torch.float32, device=dev)
    pi = po.get_pack_infos_from_n(n.to(dev))
    sdf = (torch.randn(S, generator=g) * 0.05).to(dev)
    t = torch.rand(S, generator=g).to(dev)
    rgb, nab = torch.rand(S, 3, generator=g).to(dev), (torch.randn(St, 3, generator=g) * 0.7).to(dev)
    gt = torch.rand(N, 3, generator=g).to(dev)
    ln = torch.tensor([0.35], **f32)
    out_idx = None if every_ray else torch.randperm(N, generator=g)[:R].sort().values.to(dev)
    w_eik, factor, nd = 0.1, 10.0, 0
    call, ptr = _lib.call, _lib.ptr

    def bufs():
        z = lambda *sh: torch.zeros(list(sh), **f32)          # noqa: E731
        e = lambda *sh: torch.empty(list(sh), **f32)          # noqa: E731
        return dict(alpha=e(S), vw=e(S), trans=e(S), mask=z(N), depth=z(N), img=z(N, 3), nimg=z(N, 3), acc=z(8), dalpha=e(S),
                    dsdf=z(St), drgb=z(St, 3), dnab=e(St, 3), dln=z(1))
    a = bufs()
    call("nsim_neus_composite_fwd", ptr(sdf), ptr(ln), factor, 0.0, ptr(t), ptr(rgb), ptr(nab), ptr(pi), R, nd, ptr(a["alpha"]),
         ptr(a["vw"]), ptr(a["trans"]), ptr(a["mask"]), ptr(a["depth"]), ptr(a["img"]), ptr(a["nimg"]), ptr(out_idx))
    d_img = torch.empty([N, 3], **f32)
    call("nsim_train_loss_head", ptr(a["img"]), ptr(gt), N * 3, ptr(nab), S, M, w_eik, ptr(a["acc"]), ptr(d_img), ptr(a["dnab"]))
    call("nsim_composite_bwd", ptr(a["alpha"]), ptr(a["trans"]), ptr(a["vw"]), ptr(t), ptr(rgb), ptr(nab), ptr(pi), R, nd,
         ptr(a["mask"]), ptr(a["depth"]), None, None, ptr(d_img), None, None, ptr(a["dalpha"]), ptr(a["drgb"]), None, ptr(out_idx))
    call("nsim_neus_alpha_bwd", ptr(sdf), ptr(a["dalpha"]), ptr(pi), R, ptr(ln), factor, 0.0, ptr(a["dsdf"]), ptr(a["dln"]))
    b = bufs()
    call("nsim_render_head", ptr(sdf), ptr(ln), factor, 0.0, ptr(t), ptr(rgb), ptr(nab), ptr(pi), R, nd, ptr(gt), N, S, M, w_eik,
         ptr(out_idx), ptr(b["alpha"]), ptr(b["vw"]), ptr(b["trans"]), ptr(b["mask"]), ptr(b["depth"]), ptr(b["img"]), ptr(b["nimg"]),
         ptr(b["acc"]), ptr(b["dalpha"]), ptr(b["dsdf"]), ptr(b["drgb"]), ptr(b["dnab"]), ptr(b["dln"]))
    for k in ("alpha", "vw", "trans", "mask", "depth", "img", "nimg", "dalpha", "dsdf", "drgb", "dnab"):
        x, y = a[k].cpu(), b[k].cpu()
        assert torch.allclose(x, y, rtol=2e-5, atol=1e-7), (k, float((x - y).abs().max()))
    assert torch.allclose(a["acc"].cpu()[:3], b["acc"].cpu()[:3], rtol=2e-5, atol=1e-7), (a["acc"], b["acc"])
    assert torch.allclose(a["dln"].cpu(), b["dln"].cpu(), rtol=1e-4, atol=1e-8), (a["dln"], b["dln"])
    assert float(b["acc"][0]) > 0 and float(b["acc"][1]) > 0 and float(b["acc"][2]) > 0


@pytest.mark.parametrize("every_ray", [False, True])
def test_render_head_against_the_oracle(backend, every_ray):
    """``nsim_render_head`` against the ORACLE (the one-launch head had only been compared with the four
    launches it replaces): oracle.render.neus_alpha_packed -> volume_integration (single_volume_renderer.py:73-102) -> photometric
    mse over ALL N rays (app/loss/photometric.py:88-146) + w (eikonal on the S render samples + eikonal on the M free points,
    app/loss/eikonal.py:216-251) evaluated in f64 with torch autograd: images, the three loss terms and every gradient the
    kernel emits (d sdf, d rgb, d nablas, d ln_inv_s).  The predictions are close to the targets on most rays (a trained
    state), where the mse is a difference of cancelling sums inside the kernel (f64 accumulator)."""
    from neuralsim_amd import _lib
    from oracle import pack_ops as opo, render as orr
    g = torch.Generator().manual_seed(7)
    N = 41
    R = N if every_ray else 29
    n = torch.randint(0, 140, (R,), generator=g)
    n[2], n[5], n[6] = 0, 1, 2
    S, M = int(n.sum()), 9
    St = S + M
    pi = opo.get_pack_infos_from_n(n)
    ridx = opo.pack_ridx(pi, S)
    # a surface crossing on every ray: sdf falls through zero along the pack
    u = torch.cat([torch.linspace(0, 1, int(k)) if k > 0 else torch.zeros(0) for k in n])
    t64 = (0.5 + u + 0.01 * torch.rand(S, generator=g)).double()
    sdf64 = ((0.55 + 0.3 * torch.rand(R, generator=g)[ridx] - u) * 0.4 + 0.01 * torch.randn(S, generator=g)).double().requires_grad_(True)
    rgb64 = torch.rand(S, 3, generator=g).double().requires_grad_(True)
    nab64 = (torch.nn.functional.normalize(torch.randn(St, 3, generator=g), dim=-1) * (1 + 0.1 * torch.randn(St, 1, generator=g))).double().requires_grad_(True)
    ln64 = torch.tensor([0.3], dtype=torch.float64, requires_grad=True)
    factor, w_eik = 10.0, 0.1
    out_idx = None if every_ray else torch.randperm(N, generator=g)[:R].sort().values
    rows = torch.arange(N) if every_ray else out_idx
    for nd in (0, 1):
        for x in (sdf64, rgb64, nab64, ln64):
            x.grad = None
        inv_s = torch.exp(ln64 * factor)
        alpha = orr.neus_alpha_packed(sdf64, pi, inv_s)
        out = orr.volume_integration(alpha, t64, rgb64, nab64[:S], pi, depth_use_normalized_vw=bool(nd))
        img = torch.zeros(N, 3, dtype=torch.float64).index_put((rows,), out["rgb_volume"])
        # targets: the rendering itself + a small error on the hit rays (trained state), arbitrary colours elsewhere
        gt = torch.rand(N, 3, generator=g).double()
        gt[rows] = (out["rgb_volume"].detach() + 3e-3 * torch.randn(R, 3, generator=g).double()).clamp(0, 1)
        mse = ((img - gt) ** 2).mean()
        eik_r = ((nab64[:S].norm(dim=-1) - 1.0) ** 2).mean()
        eik_f = ((nab64[S:].norm(dim=-1) - 1.0) ** 2).mean()
        (mse + w_eik * (eik_r + eik_f)).backward()
        dev = backend
        f32 = dict(dtype=torch.float32, device=dev)
        dv = lambda a: a.detach().float().to(dev).contiguous()          # noqa: E731
        z = lambda *sh: torch.zeros(list(sh), **f32)          # noqa: E731
        b = dict(alpha=z(S), vw=z(S), trans=z(S), mask=z(N), depth=z(N), img=z(N, 3), nimg=z(N, 3), acc=z(8), dalpha=z(S),
                 dsdf=z(St), drgb=z(St, 3), dnab=z(St, 3), dln=z(1))
        call, ptr = _lib.call, _lib.ptr
        call("nsim_render_head", ptr(dv(sdf64)), ptr(dv(ln64)), factor, 0.0, ptr(dv(t64)), ptr(dv(rgb64)), ptr(dv(nab64)), ptr(pi.to(dev)),
             R, nd, ptr(dv(gt)), N, S, M, w_eik, ptr(out_idx.to(dev) if out_idx is not None else None), ptr(b["alpha"]), ptr(b["vw"]),
             ptr(b["trans"]), ptr(b["mask"]), ptr(b["depth"]), ptr(b["img"]), ptr(b["nimg"]), ptr(b["acc"]), ptr(b["dalpha"]),
             ptr(b["dsdf"]), ptr(b["drgb"]), ptr(b["dnab"]), ptr(b["dln"]))
        full = lambda v, *sh: torch.zeros([N, *sh], dtype=torch.float64).index_put((rows,), v.detach())          # noqa: E731
        assert (b["alpha"].cpu().double() - alpha.detach()).abs().max() < 2e-6
        assert (b["vw"].cpu().double() - out["vw"].detach()).abs().max() < 2e-6
        assert (b["mask"].cpu().double() - full(out["mask_volume"])).abs().max() < 5e-6
        assert (b["depth"].cpu().double() - full(out["depth_volume"])).abs().max() < 2e-5
        assert (b["img"].cpu().double() - full(out["rgb_volume"], 3)).abs().max() < 5e-6
        assert (b["nimg"].cpu().double() - full(out["normals_volume"], 3)).abs().max() < 5e-6
        acc = b["acc"].cpu().double()
        # the mse of a near-converged state: relative to ITS OWN size (1e-5-ish next to sum(gt^2)/3N ~ 0.3)
        assert float(mse) < 0.3 and abs(float(acc[0]) - float(mse)) < 2e-4 * float(mse) + 1e-9, (float(acc[0]), float(mse))
        assert abs(float(acc[1]) - float(eik_r)) < 1e-5 * (1 + float(eik_r)) and abs(float(acc[2]) - float(eik_f)) < 1e-5 * (1 + float(eik_f))

        def rel(a_, b_):
            return float((a_.cpu().double() - b_).norm() / b_.norm().clamp_min(1e-12))
        # (d rgb = vw x 2 e / 3N with e = pred - gt ~ 3e-3 formed in f32 from O(1) values: relative error ~ 1e-7 / 3e-3)
        assert rel(b["drgb"][:S], rgb64.grad) < 2e-4 and rel(b["dnab"], nab64.grad) < 2e-5
        assert rel(b["dsdf"][:S], sdf64.grad) < 2e-4, rel(b["dsdf"][:S], sdf64.grad)
        assert abs(float(b["dln"].cpu()) - float(ln64.grad)) < 2e-3 * abs(float(ln64.grad)) + 1e-9


def test_rows_gather_with_zero_tail(backend, poisoned_empty):
    """``nsim_rows_gather``: out[:n] = table[idx], then ``tail`` zero rows; out-of-range indices give zero rows."""
    from neuralsim_amd import _lib
    g = torch.Generator().manual_seed(0)
    table = torch.randn(7, 4, generator=g)
    idx = torch.tensor([3, 0, 6, 6, 2, -1, 7, 1], dtype=torch.long)
    out = torch.empty([8 + 5, 4], dtype=torch.float32, device=backend)
    _lib.call("nsim_rows_gather", _lib.ptr(table.to(backend)), _lib.ptr(idx.to(backend)), 8, 4, 7, 5, _lib.ptr(out))
    ref = torch.zeros(13, 4)
    ok = (idx >= 0) & (idx < 7)
    ref[:8][ok] = table[idx[ok]]
    assert torch.equal(out.cpu(), ref)
