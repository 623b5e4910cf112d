"""Fused loss reductions + embedding lookup vs their torch statements (row a18; the reference's own formulas:
app/loss/eikonal.py:96-105, app/loss/photometric.py:88-146, app/models/scene/image_embeddings.py:23-80)."""
import torch

from neuralsim_amd import losses
from util import leaf


def test_eikonal_loss(backend):
    g = torch.Generator().manual_seed(3)
    nab = torch.randn(1500, 3, generator=g) * 1.3
    nab[7] = 0.0                                  # sub-gradient 0 at the origin
    ref_in = leaf(nab)
    ref = ((ref_in.norm(dim=-1) - 1.0) ** 2).mean()
    (ref * 1.7).backward()
    x = leaf(nab, backend)
    out = losses.eikonal_loss(x)
    (out * 1.7).backward()
    assert abs(float(out) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref)))
    assert torch.allclose(x.grad.cpu(), ref_in.grad, atol=1e-7, rtol=1e-5)
    # [.., 3] shapes keep their shape in the gradient
    x2 = leaf(nab.view(30, 50, 3), backend)
    losses.eikonal_loss(x2).backward()
    assert x2.grad.shape == (30, 50, 3)


def test_mse_loss(backend):
    g = torch.Generator().manual_seed(4)
    a, b = torch.rand(777, 3, generator=g), torch.rand(777, 3, generator=g)
    ra = leaf(a)
    ref = ((ra - b) ** 2).mean()
    ref.backward()
    x = leaf(a, backend)
    out = losses.mse_loss(x, b.to(backend))
    out.backward()
    assert abs(float(out) - float(ref)) <= 1e-6
    assert torch.allclose(x.grad.cpu(), ra.grad, atol=1e-8, rtol=1e-5)


def test_embedding_lookup(backend):
    g = torch.Generator().manual_seed(5)
    for rows, C, n in ((100, 4, 5000), (3000, 4, 4000), (1, 4, 10)):      # LDS histogram / global atomics / one row
        tab = torch.randn(rows, C, generator=g)
        idx = torch.randint(0, rows, (n,), generator=g)
        w = torch.randn(n, C, generator=g)
        rt = leaf(tab)
        (rt[idx] * w).sum().backward()
        x = leaf(tab, backend)
        out = losses.embedding_lookup(x, idx.to(backend))
        assert torch.equal(out.detach().cpu(), tab[idx])
        (out * w.to(backend)).sum().backward()
        assert torch.allclose(x.grad.cpu(), rt.grad, atol=2e-4, rtol=1e-4)
