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


def test_train_loss_head_equals_the_separate_losses(backend):
    """``nsim_train_loss_head`` (one launch) = mse + w (eikonal(render) + eikonal(uniform)) of the separate ops: the three
    loss values and both gradients."""
    from neuralsim_amd import _lib
    g = torch.Generator().manual_seed(6)
    for N, S, M in ((300, 1500, 200), (50, 40, 0), (2000, 70, 4096)):
        pred, gt = torch.rand(N, 3, generator=g), torch.rand(N, 3, generator=g)
        nab = torch.randn(S + M, 3, generator=g) * 1.2
        nab[3] = 0.0
        w = 0.1
        rp, rn = leaf(pred), leaf(nab)
        parts = [((rp - gt) ** 2).mean(), ((rn[:S].norm(dim=-1) - 1.0) ** 2).mean()]
        if M:
            parts.append(((rn[S:].norm(dim=-1) - 1.0) ** 2).mean())
        (parts[0] + w * sum(parts[1:])).backward()
        dev = backend
        f32 = dict(dtype=torch.float32, device=dev)
        acc, d_img, dnab = torch.zeros(3, **f32), torch.empty(N, 3, **f32), torch.empty(S + M, 3, **f32)
        _lib.call("nsim_train_loss_head", _lib.ptr(pred.to(dev)), _lib.ptr(gt.to(dev)), N * 3, _lib.ptr(nab.to(dev)), S, M, w,
                  _lib.ptr(acc), _lib.ptr(d_img), _lib.ptr(dnab))
        for k, ref in enumerate(parts):
            assert abs(float(acc[k]) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref))), (k, float(acc[k]), float(ref))
        if not M:
            assert float(acc[2]) == 0.0
        assert torch.allclose(d_img.cpu(), rp.grad, atol=1e-9, rtol=1e-5)
        assert torch.allclose(dnab.cpu(), rn.grad, atol=1e-9, rtol=1e-5)
