"""Property-based fuzzing of the per-ray sampling kernels: sorted merge, NeuS up-sampling and the
keep-set of the compressed query mode on random ragged packs -- single-sample rays, packs shorter / longer than a wave
(64), exact depth ties -- against the oracle.  Index / membership results must be bit-exact."""
import torch
from hypothesis import HealthCheck, example, given, settings, strategies as st

from oracle import pack_ops as opo, render as orr
from neuralsim_amd import _lib
from neuralsim_amd.graphics import pack_ops as po

import os  # noqa: E402
# NSIM_FUZZ_EXAMPLES=N: an exploratory sweep with fresh random examples; the default run is derandomized so that the
# suite the driver executes is reproducible (failures found by sweeps become explicit regression cases)
_N = int(os.environ.get("NSIM_FUZZ_EXAMPLES", "0"))
SET = dict(max_examples=_N or 20, derandomize=_N == 0, deadline=None,
           suppress_health_check=[HealthCheck.function_scoped_fixture])
# 448 + 64 = 512 is the per-wave LDS window of the merge / up-sampling kernels; 449, 513, 700: their global-memory path
counts = st.lists(st.sampled_from([1, 2, 3, 5, 17, 63, 64, 65, 129, 200, 448, 449, 513, 700]), min_size=1, max_size=14)


def _packs(n, g):
    n = torch.tensor(n)
    pi = opo.get_pack_infos_from_n(n)
    S = int(n.sum())
    ridx = opo.pack_ridx(pi, S)
    near = torch.rand(n.shape[0], generator=g) + 0.5
    t = near[ridx] + torch.randint(0, 400, (S,), generator=g).float() * (1.0 / 256)      # exact ties are common
    t, _ = opo.packed_sort(t, pi)
    return n, pi, S, ridx, t, near


@settings(**SET)
@given(n=counts, nf=st.sampled_from([1, 4, 8, 32, 70]), seed=st.integers(0, 10 ** 6))
@example(n=[448, 449, 3, 700, 64], nf=64, seed=7)        # both sides of the LDS window (na + nb <= 512)
@example(n=[480, 481, 513, 1], nf=32, seed=8)
def test_fuzz_merge_sorted(backend, n, nf, seed):
    g = torch.Generator().manual_seed(seed)
    n, pi, S, ridx, t, near = _packs(n, g)
    R = n.shape[0]
    sdf = torch.randn(S, generator=g)
    t_b = (near[:, None] + torch.randint(0, 400, (R, nf), generator=g).float() * (1.0 / 256)).sort(dim=1).values
    v_b = torch.randn(R, nf, generator=g)
    t_ref, pi_ref, pa, pb = orr.merge_sorted(t, pi, t_b)
    v_ref = torch.empty_like(t_ref)
    v_ref[pa] = sdf
    v_ref[pb.reshape(-1)] = v_b.reshape(-1)
    dv = lambda a: a.to(backend).contiguous()
    ro, rd = torch.randn(R, 3, generator=g), torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    T = S + R * nf
    t_out, v_out = torch.zeros(T, device=backend), torch.zeros(T, device=backend)
    pi_out = torch.zeros(R, 2, dtype=torch.long, device=backend)
    ridx_out = torch.zeros(T, dtype=torch.long, device=backend)
    x_out = torch.zeros(T, 3, device=backend)
    _lib.call("nsim_merge_sorted", _lib.ptr(dv(t)), _lib.ptr(dv(sdf)), _lib.ptr(dv(pi)), _lib.ptr(dv(t_b)), _lib.ptr(dv(v_b)),
              R, nf, _lib.ptr(t_out), _lib.ptr(v_out), _lib.ptr(pi_out), _lib.ptr(ridx_out), _lib.ptr(dv(ro)),
              _lib.ptr(dv(rd)), _lib.ptr(x_out), None)
    assert torch.equal(pi_out.cpu(), pi_ref) and torch.equal(t_out.cpu(), t_ref) and torch.equal(v_out.cpu(), v_ref)
    assert torch.equal(ridx_out.cpu(), opo.pack_ridx(pi_ref, T))
    assert torch.equal(x_out.cpu(), ro[ridx_out.cpu()] + t_out.cpu()[:, None] * rd[ridx_out.cpu()])


@settings(**SET)
@given(n=st.lists(st.sampled_from([2, 3, 5, 17, 63, 64, 65, 129, 200]), min_size=1, max_size=14),
       nf=st.sampled_from([1, 8, 32, 70]), inv_s=st.sampled_from([16.0, 64.0, 1024.0]), use_est=st.booleans(),
       seed=st.integers(0, 10 ** 6))
@example(n=[512, 513, 514, 5, 700], nf=32, inv_s=64.0, use_est=True, seed=9)     # 512 intervals = the LDS window
@example(n=[513, 514, 2, 900], nf=70, inv_s=16.0, use_est=False, seed=10)
def test_fuzz_upsample_stage(backend, n, nf, inv_s, use_est, seed):
    g = torch.Generator().manual_seed(seed)
    n, pi, S, ridx, t, near = _packs(n, g)
    # strictly increasing depths (the up-sampler divides by interval lengths), an SDF crossing zero somewhere
    t = t + torch.arange(S).float() * 1e-4
    R = n.shape[0]
    sdf = (near[ridx] + 0.3 + 0.5 * torch.rand(R, generator=g)[ridx] - t) * 0.6 + 0.01 * torch.randn(S, generator=g)
    # Where the CDF of a ray is nearly flat the inverse-CDF draw is ill-conditioned: rounding decides which side of a
    # (near) zero-weight interval the new depth lands on, and f32 / f64 evaluations of the ORACLE itself differ by
    # ~1e-3 there (found by NSIM_FUZZ_EXAMPLES sweeps: n=[..,129,..] nf=70 seed=23046 -- kernel = f64 oracle to 2e-5, f32
    # oracle 1.3e-3 off; n=[200,200,200,2,2] nf=1 seed=320767 -- kernel = f32 oracle to 2e-5, f64 oracle 1.6e-3 off).
    # Every new depth must agree with one of the two evaluations.
    ref = orr.upsample_stage(t, sdf, pi, inv_s, nf, use_est)
    ref64 = orr.upsample_stage(t.double(), sdf.double(), pi, inv_s, nf, use_est).float()
    dv = lambda a: a.to(backend).contiguous()
    ro, rd = torch.randn(R, 3, generator=g), torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    t_new, scratch = torch.zeros(R, nf, device=backend), torch.zeros(S, device=backend)
    x_new = torch.zeros(R, nf, 3, device=backend)
    _lib.call("nsim_upsample_stage", _lib.ptr(dv(t)), _lib.ptr(dv(sdf)), _lib.ptr(dv(pi)), R, inv_s, nf, int(use_est),
              _lib.ptr(scratch), _lib.ptr(t_new), _lib.ptr(dv(ro)), _lib.ptr(dv(rd)), _lib.ptr(x_new), None)
    tn = t_new.cpu()
    assert torch.isfinite(tn).all() and (tn[:, 1:] >= tn[:, :-1]).all()
    lo = t[pi[:, 0]][:, None]
    hi = t[pi[:, 0] + pi[:, 1] - 1][:, None]
    assert ((tn >= lo - 1e-5) & (tn <= hi + 1e-5)).all()                        # new depths stay inside the ray's span
    err = torch.minimum((tn - ref).abs(), (tn - ref64).abs())
    assert float(err.max()) <= 1e-4, float(err.max())
    assert torch.equal(x_new.cpu(), ro[:, None, :] + tn[..., None] * rd[:, None, :])


@settings(**SET)
@given(n=counts, inv_s=st.sampled_from([8.0, 64.0, 400.0]), thre=st.sampled_from([1e-4, 1e-2, 0.2]),
       seed=st.integers(0, 10 ** 6))
def test_fuzz_compress(backend, n, inv_s, thre, seed):
    """Keep-set of ``march_occ_multi_upsample_compressed``: samples bounding an interval whose visibility weight exceeds
    ``thre`` (oracle/render.py ray_query, compress branch)."""
    g = torch.Generator().manual_seed(seed)
    n, pi, S, ridx, t, near = _packs(n, g)
    R = n.shape[0]
    sdf = (near[ridx] + 0.2 + torch.rand(R, generator=g)[ridx] - t) * 0.5 + 0.02 * torch.randn(S, generator=g)
    vw = opo.packed_alpha_to_vw(orr.neus_alpha_packed(sdf, pi, torch.tensor(inv_s)), pi)
    margin = (vw - thre).abs().min() if S else torch.tensor(1.0)
    if float(margin) < 1e-6:        # a weight within float noise of the threshold: the keep decision is not defined
        return
    sig = vw > thre
    first = torch.zeros_like(sig)
    first[pi[:, 0][pi[:, 1] > 0]] = True
    prev_sig = torch.cat([sig[:1] & False, sig[:-1]]) & ~first
    keep = sig | prev_sig
    cnt_ref = torch.zeros(R, dtype=torch.long).index_add_(0, ridx[keep], torch.ones(int(keep.sum()), dtype=torch.long))
    dv = lambda a: a.to(backend).contiguous()
    ln = torch.zeros(1, device=backend)
    counts_d = torch.zeros(R, dtype=torch.long, device=backend)
    _lib.call("nsim_compress_count", _lib.ptr(dv(sdf)), _lib.ptr(dv(pi)), R, _lib.ptr(ln), 1.0, float(inv_s), float(thre),
              _lib.ptr(counts_d))
    assert torch.equal(counts_d.cpu(), cnt_ref)
    pi_k = po.get_pack_infos_from_n(counts_d)
    K = int(cnt_ref.sum())
    tail = (seed % 3) * 37          # 0, 37 or 74 zero-depth samples on the pseudo-rays R.. behind the kept set
    t_k = torch.full([K + tail], -1.0, device=backend)
    ridx_k = torch.full([K + tail], -1, dtype=torch.long, device=backend)
    _lib.call("nsim_compress_emit", _lib.ptr(dv(sdf)), _lib.ptr(dv(t)), _lib.ptr(dv(pi)), R, _lib.ptr(ln), 1.0, float(inv_s),
              float(thre), _lib.ptr(pi_k), _lib.ptr(t_k), _lib.ptr(ridx_k), tail)
    assert torch.equal(t_k[:K].cpu(), t[keep]) and torch.equal(ridx_k[:K].cpu(), ridx[keep])
    assert torch.equal(t_k[K:].cpu(), torch.zeros(tail)) and torch.equal(ridx_k[K:].cpu(), torch.arange(R, R + tail))


@settings(**SET)
@given(n=counts, p_one=st.sampled_from([0.0, 0.02, 0.2]), p_zero=st.sampled_from([0.0, 0.1]), scale=st.sampled_from([0.05, 0.5, 1.0]),
       norm_depth=st.booleans(), seed=st.integers(0, 10 ** 6))
def test_fuzz_compositing_forward_backward(backend, n, p_one, p_zero, scale, norm_depth, seed):
    """Fused compositing and the stand-alone alpha -> vw op on random ragged packs with exactly-opaque (alpha = 1) and
    exactly-empty (alpha = 0) samples sprinkled in: values and EVERY per-sample gradient against the f64 oracle (the
    backward's 1 / (1 - alpha + 1e-10) factor makes the sums over the later samples cancellation-critical)."""
    from neuralsim_amd.fields.neus import volume_integration
    g = torch.Generator().manual_seed(seed)
    n, pi, S, ridx, t, near = _packs(n, g)
    R = n.shape[0]
    alpha = torch.rand(S, generator=g) * scale
    u = torch.rand(S, generator=g)
    alpha[u < p_one] = 1.0
    alpha[(u >= p_one) & (u < p_one + p_zero)] = 0.0
    rgb, nrm = torch.rand(S, 3, generator=g), torch.randn(S, 3, generator=g)
    wm, wd = torch.randn(R, generator=g), torch.randn(R, generator=g) * 0.3
    wr, wn = torch.randn(R, 3, generator=g), torch.randn(R, 3, generator=g)
    dv = lambda a: a.to(backend).contiguous()         # noqa: E731
    a_d = alpha.clone().to(backend).requires_grad_(True)
    a_o = alpha.double().requires_grad_(True)
    r_d, r_o = rgb.clone().to(backend).requires_grad_(True), rgb.double().requires_grad_(True)
    out = volume_integration(a_d, dv(t), r_d, dv(nrm), dv(pi), norm_depth)
    ref = orr.volume_integration(a_o, t.double(), r_o, nrm.double(), pi, norm_depth)
    for k in ("mask_volume", "depth_volume", "rgb_volume", "normals_volume"):
        assert torch.allclose(out[k].detach().cpu().double(), ref[k].detach(), atol=2e-5, rtol=1e-5), k
    (out["mask_volume"] * dv(wm) + out["depth_volume"] * dv(wd) + (out["rgb_volume"] * dv(wr)).sum(-1)
     + (out["normals_volume"] * dv(wn)).sum(-1)).sum().backward()
    (ref["mask_volume"] * wm + ref["depth_volume"] * wd + (ref["rgb_volume"] * wr).sum(-1)
     + (ref["normals_volume"] * wn).sum(-1)).sum().backward()
    err = (a_d.grad.cpu().double() - a_o.grad).abs() / a_o.grad.abs().clamp_min(1.0)
    assert float(err.max()) < 2e-4, (float(err.max()), int(err.argmax()), float(alpha[int(err.argmax())]))
    assert torch.allclose(r_d.grad.cpu().double(), r_o.grad, atol=1e-5)
    vw = po.packed_alpha_to_vw(dv(alpha), dv(pi))
    assert torch.allclose(vw.cpu().double(), opo.packed_alpha_to_vw(alpha.double(), pi), atol=1e-6)
