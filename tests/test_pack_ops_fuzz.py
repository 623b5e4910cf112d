"""Property-based fuzzing of the pack-op bookkeeping (the one part of the path the reference pins with a test of its
own, buffer_compose_renderer.py:972-1049): random ragged packs -- empty packs, single elements, ties, packs longer than
a wave -- against the oracle.  Integer / index results must be bit-exact."""
import torch
from hypothesis import HealthCheck, given, settings, strategies as st

from oracle import pack_ops as opo
from neuralsim_amd.graphics import pack_ops as po

import os  # noqa: E402
# NSIM_FUZZ_EXAMPLES=N: an exploratory sweep with fresh random examples; the default run is derandomized so that the
# suite the driver executes is reproducible (failures found by sweeps become explicit regression cases)
_N = int(os.environ.get("NSIM_FUZZ_EXAMPLES", "0"))
SET = dict(max_examples=_N or 25, derandomize=_N == 0, deadline=None,
           suppress_health_check=[HealthCheck.function_scoped_fixture])
counts = st.lists(st.sampled_from([0, 0, 1, 2, 3, 7, 63, 64, 65, 130]), min_size=1, max_size=24)


@settings(**SET)
@given(n=counts, seed=st.integers(0, 10 ** 6))
def test_fuzz_sort_linstep_sum(backend, n, seed):
    g = torch.Generator().manual_seed(seed)
    n = torch.tensor(n)
    S = int(n.sum())
    pi = opo.get_pack_infos_from_n(n)
    assert torch.equal(po.get_pack_infos_from_n(n.to(backend)).cpu(), pi)
    x = torch.randint(0, 12, (S,), generator=g).float() * 0.25            # many ties
    srt, idx = po.packed_sort(x.to(backend), pi.to(backend))
    srt_o, idx_o = opo.packed_sort(x, pi)
    assert torch.equal(srt.cpu(), srt_o)
    assert torch.equal(x[idx.cpu()], srt_o)                               # global indices, sorted == x[indices]
    for p in range(n.shape[0]):                                           # stable inside ties, confined to the pack
        a, b = int(pi[p, 0]), int(pi[p, 0] + pi[p, 1])
        ii = idx.cpu()[a:b]
        assert ((ii >= a) & (ii < b)).all()
        same = srt_o[a:b][1:] == srt_o[a:b][:-1]
        assert bool((ii[1:][same] > ii[:-1][same]).all())
    start = torch.randint(0, 1000, (n.shape[0],), generator=g)
    out, ridx = po.interleave_linstep(start.to(backend), n.to(backend), 3, return_idx=True)
    ref, ridx_o = opo.interleave_linstep(start, n, 3, return_idx=True)
    assert torch.equal(out.cpu(), ref) and torch.equal(ridx.cpu(), ridx_o)
    v = torch.randn(S, 2, generator=g)
    assert torch.allclose(po.packed_sum(v.to(backend), pi.to(backend)).cpu(), opo.packed_sum(v, pi), atol=1e-4)


run_lens = st.lists(st.sampled_from([0, 1, 2, 5, 31, 64, 100, 300, 600]), min_size=1, max_size=20)


@settings(**SET)
@given(packs=st.lists(run_lens, min_size=1, max_size=6), seed=st.integers(0, 10 ** 6))
def test_fuzz_sort_of_run_structured_packs(backend, packs, seed):
    """Packs that are concatenations of sorted runs (what the compose renderer collects): 1 run (identity path), 2..16 runs
    (merge path), > 16 runs (rank sort), packs beyond the LDS staging capacity (direct path) -- values drawn from a small
    set so that ties occur inside and across runs.  The oracle's stable order, bit for bit."""
    g = torch.Generator().manual_seed(seed)
    xs, n = [], []
    for runs in packs:
        parts = [torch.sort(torch.randint(0, 40, (k,), generator=g).float() * 0.125).values for k in runs]
        x = torch.cat(parts) if parts else torch.empty(0)
        xs.append(x)
        n.append(x.shape[0])
    x, n = torch.cat(xs), torch.tensor(n)
    pi = opo.get_pack_infos_from_n(n)
    srt, idx = po.packed_sort(x.to(backend), pi.to(backend))
    srt_o, idx_o = opo.packed_sort(x, pi)
    assert torch.equal(srt.cpu(), srt_o) and torch.equal(idx.cpu(), idx_o)


@settings(**SET)
@given(na=counts, nb=counts, seed=st.integers(0, 10 ** 6))
def test_fuzz_merge_two_packs(backend, na, nb, seed):
    """Packs of buffer a live on rays nidx_a, packs of b on nidx_b (sorted, unique, overlapping or not)."""
    g = torch.Generator().manual_seed(seed)
    U = 40

    def side(n):
        n = torch.tensor([c for c in n if c > 0] or [1])
        rays = torch.randperm(U, generator=g)[:n.shape[0]].sort().values
        pi = opo.get_pack_infos_from_n(n)
        vals = torch.cat([torch.sort(torch.randint(0, 20, (int(c),), generator=g).float() * 0.5).values for c in n])
        return vals, pi, rays
    va, pia, ra = side(na)
    vb, pib, rb = side(nb)
    pa, pb, pit = po.merge_two_packs_sorted(va.to(backend), pia.to(backend), ra.to(backend), vb.to(backend),
                                            pib.to(backend), rb.to(backend))
    pa_o, pb_o, pit_o = opo.merge_two_packs_sorted(va, pia, ra, vb, pib, rb)
    assert torch.equal(pit.cpu(), pit_o) and torch.equal(pa.cpu(), pa_o) and torch.equal(pb.cpu(), pb_o)
    tot = torch.full([int(pit_o[:, 1].sum())], float("nan"))
    tot[pa_o] = va
    tot[pb_o] = vb
    assert not bool(torch.isnan(tot).any())                               # a bijection onto the merged buffer
    for p in range(pit_o.shape[0]):                                       # ascending inside every merged pack
        a, b = int(pit_o[p, 0]), int(pit_o[p, 0] + pit_o[p, 1])
        assert bool((tot[a:b][1:] >= tot[a:b][:-1]).all())
