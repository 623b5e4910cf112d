"""Row a16: the sky MLP (csrc/sky.hip) vs oracle/sky.py -- forward, weight / bias / appearance gradients, ragged
sizes, and the renderer blend."""
import pytest
import torch

from oracle import sky as osky
from util import leaf


def _setup(backend, precision, N, n_appear=4, F=10, seed=11):
    from neuralsim_amd.env import SimpleSky
    ws, bs = osky.make_sky_params(F, n_appear, seed=seed)
    m = SimpleSky(dict(type="sinusoidal", n_frequencies=F), n_appear_embedding=n_appear, precision=precision).to(backend)
    with torch.no_grad():
        m.w.copy_(torch.cat([w.reshape(-1) for w in ws]).to(backend))
        m.b.copy_(torch.cat(bs).to(backend))
    g = torch.Generator().manual_seed(seed + 1)
    v = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)
    ha = torch.randn(N, n_appear, generator=g) * 0.3 if n_appear > 0 else None
    wgt = torch.randn(N, 3, generator=g)
    return m, ws, bs, v, ha, wgt


def _rel(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-20))


@pytest.mark.parametrize("precision,N", [("f32", 200), ("fp16", 333), ("f32", 1)])
def test_sky_forward_backward(backend, precision, N):
    m, ws, bs, v, ha, wgt = _setup(backend, precision, N)
    ws_o = [leaf(w) for w in ws]
    bs_o = [leaf(b) for b in bs]
    ha_o = leaf(ha)
    ref = osky.sky_forward(v, ha_o, ws_o, bs_o)
    (ref * wgt).sum().backward()
    ha_p = leaf(ha, backend)
    out = m(v.to(backend), h_appear=ha_p)
    (out * wgt.to(backend)).sum().backward()
    tol_v, tol_g = (2e-5, 2e-4) if precision == "f32" else (4e-3, 3e-2)
    assert float((out.detach().cpu() - ref.detach()).abs().max()) <= tol_v
    gw_ref = torch.cat([w.grad.reshape(-1) for w in ws_o])
    gb_ref = torch.cat([b.grad for b in bs_o])
    n1 = ws[0].numel()
    n2 = n1 + ws[1].numel()
    gw = m.w.grad.cpu()
    for name, a, b in (("W1", gw[:n1], gw_ref[:n1]), ("W2", gw[n1:n2], gw_ref[n1:n2]), ("W3", gw[n2:], gw_ref[n2:]),
                       ("b", m.b.grad.cpu(), gb_ref), ("h_appear", ha_p.grad.cpu(), ha_o.grad)):
        assert _rel(a, b) <= tol_g, (name, _rel(a, b))


def test_sky_no_appearance_and_shapes(backend):
    m, ws, bs, v, _, _ = _setup(backend, "f32", 96, n_appear=0, F=4)
    ref = osky.sky_forward(v, None, ws, bs, n_frequencies=4)
    out = m(v.view(8, 12, 3).to(backend))
    assert out.shape == (8, 12, 3)
    assert float((out.detach().cpu().view(-1, 3) - ref).abs().max()) <= 2e-5
