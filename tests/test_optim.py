"""``nsim_adam_step`` (csrc/optim.hip, ``neuralsim_amd.optim.FusedAdam``) against ``torch.optim.Adam`` configured as the
reference's ``training_cfg`` does: ``eps 1e-15, betas [0.9, 0.99]`` (lotd_neus.dtu.230814.yaml:178-184), step site
code_single/tools/train.py:1494-1502."""
import pytest
import torch

from neuralsim_amd import _lib


def _torch_adam_trace(p0, grads, lr, betas, eps):
    p = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p], lr=lr, betas=betas, eps=eps)
    out = []
    for g in grads:
        p.grad = g.clone()
        opt.step()
        out.append(p.detach().clone())
    return out


@pytest.mark.parametrize("grad_scale", [1.0, 0.5])
def test_adam_step_matches_torch_adam(backend, grad_scale):
    dev = backend
    g = torch.Generator().manual_seed(5)
    n = 70001                                  # not a multiple of the block / wave size
    p0 = (torch.rand(n, generator=g) * 2 - 1) * 1e-2
    # gradients over many decades, some exactly zero (untouched hash entries), some tiny (eps 1e-15 matters there)
    mag = 10.0 ** (torch.rand(5, n, generator=g) * 12 - 10)
    grads = torch.randn(5, n, generator=g) * mag
    grads[:, ::7] = 0.0
    lr, betas, eps = 1e-2, (0.9, 0.99), 1e-15
    ref = _torch_adam_trace(p0, [gr * grad_scale for gr in grads], lr, betas, eps)
    p, m, v = p0.clone().to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    p16 = torch.zeros(n, dtype=torch.float16, device=dev)
    for t in range(1, 6):
        gd = grads[t - 1].to(dev).contiguous()
        _lib.call("nsim_adam_step", _lib.ptr(p), _lib.ptr(p16), _lib.ptr(gd), _lib.ptr(m), _lib.ptr(v), n, lr, betas[0],
                  betas[1], eps, 1.0 - betas[0] ** t, 1.0 - betas[1] ** t, float(grad_scale), 0)
        got, want = p.cpu(), ref[t - 1]
        # <= 1e-6 relative (a few f32 ulps of the parameter: the kernel rounds lr / bias1 * (m / denom) in another order)
        assert float((got - want).abs().max()) < 1e-6 * (lr + float(want.abs().max())), t
        assert torch.equal(p16.cpu(), p.cpu().half())            # the fp16 shadow the gather kernels read
        assert torch.equal(gd.cpu(), grads[t - 1])               # zero_grad = 0 leaves the gradient alone
    # entries that never saw a gradient do not move (untouched hash entries stay put under dense Adam)
    assert torch.equal(p.cpu()[::7], p0[::7])
    # zero_grad = 1 clears the gradient in the same pass
    gd = grads[0].to(dev).contiguous()
    _lib.call("nsim_adam_step", _lib.ptr(p), None, _lib.ptr(gd), _lib.ptr(m), _lib.ptr(v), n, lr, betas[0], betas[1], eps,
              1.0 - betas[0] ** 6, 1.0 - betas[1] ** 6, 1.0, 1)
    assert float(gd.abs().max()) == 0.0


def test_fused_adam_step_range_equals_full_step(backend):
    """``FusedAdam.step_range`` on two halves == one ``step`` (the overlapped data-parallel schedule relies on it)."""
    from neuralsim_amd.optim import FusedAdam
    from test_trainer import _tiny
    outs = []
    for split in (False, True):
        torch.manual_seed(0)
        m = _tiny(backend)
        opt = FusedAdam(m, lr=1e-2)
        gp = m.encoding.flattened_params
        g = torch.Generator().manual_seed(1)
        for it in range(3):
            for q in opt.params():
                q.grad = (torch.randn(q.shape, generator=g) * 1e-3).to(backend)
            if split:
                k = gp.numel() // 3
                full = gp.grad
                opt.step(grad_scale=0.5, skip=(gp,))
                opt.step_range(gp, 0, k, full[:k].contiguous(), grad_scale=0.5)
                opt.step_range(gp, k, gp.numel(), full[k:].contiguous(), grad_scale=0.5)
            else:
                opt.step(grad_scale=0.5)
        outs.append([q.detach().clone().cpu() for q in opt.params()] + [m.encoding.shadow().clone().cpu()])
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_adam_multi_equals_single_tensor_steps(backend):
    """``nsim_adam_multi`` (all small tensors of a step in one launch) == one ``nsim_adam_step`` per tensor, bit for bit."""
    g = torch.Generator().manual_seed(9)
    sizes = [6208, 129, 5952, 131, 400, 1]
    betas = [(0.9, 0.99)] * 5 + [(0.9, 0.999)]
    P = [torch.randn(n, generator=g) * 0.1 for n in sizes]
    G = [torch.randn(n, generator=g) * 1e-3 for n in sizes]
    lr, eps, t, gs = 3e-3, 1e-15, 4, 0.5

    def state():
        return ([p.clone().to(backend) for p in P], [x.clone().to(backend) for x in G],
                [torch.full((n,), 1e-4, device=backend) for n in sizes], [torch.full((n,), 1e-7, device=backend) for n in sizes])
    p1, g1, m1, v1 = state()
    for k in range(len(sizes)):
        b1, b2 = betas[k]
        _lib.call("nsim_adam_step", _lib.ptr(p1[k]), None, _lib.ptr(g1[k]), _lib.ptr(m1[k]), _lib.ptr(v1[k]), sizes[k], lr, b1, b2,
                  eps, 1.0 - b1 ** t, 1.0 - b2 ** t, gs, 0)
    p2, g2, m2, v2 = state()
    arr = (_lib.AdamTensor * len(sizes))()
    for k in range(len(sizes)):
        b1, b2 = betas[k]
        arr[k] = _lib.AdamTensor(p2[k].data_ptr(), None, g2[k].data_ptr(), m2[k].data_ptr(), v2[k].data_ptr(), sizes[k], b1, b2,
                                 1.0 - b1 ** t, 1.0 - b2 ** t, 1.0)
    _lib.call("nsim_adam_multi", arr, len(sizes), lr, eps, gs, 0)
    for a, b in zip(p1 + m1 + v1, p2 + m2 + v2):
        assert torch.equal(a.cpu(), b.cpu())


def test_group_without_gradient_is_skipped_and_keeps_its_own_step_count(backend):
    """every group keeps its OWN step count, advanced only when the group is updated (``torch.optim.Adam``:
    per-parameter ``state['step']``, parameters with ``grad is None`` skipped -- the lidar step of the street iteration
    leaves the radiance / appearance parameters without a gradient, code_single/tools/train.py:1540-1590).  Five steps in
    which the radiance weights have no gradient on steps 1 and 3: equal to ``torch.optim.Adam`` on the same gradients."""
    from neuralsim_amd.optim import FusedAdam
    from test_trainer import _tiny
    torch.manual_seed(0)
    m = _tiny(backend)
    opt = FusedAdam(m, lr=1e-2, eps=1e-15)
    ps = opt.params()
    ref = [torch.nn.Parameter(p.detach().cpu().clone()) for p in ps]
    betas = [g["betas"] for g in opt.groups]
    ropt = torch.optim.Adam([dict(params=[r], betas=b) for r, b in zip(ref, betas)], lr=1e-2, eps=1e-15)
    rad = next(i for i, p in enumerate(ps) if p is m.rad_w)
    g = torch.Generator().manual_seed(3)
    for it in range(5):
        for i, (p, r) in enumerate(zip(ps, ref)):
            if i == rad and it in (1, 3):
                p.grad, r.grad = None, None
                continue
            gr = torch.randn(p.shape, generator=g) * 1e-3
            p.grad, r.grad = gr.to(backend), gr.clone()
        before = m.rad_w.detach().clone()
        opt.step()
        ropt.step()
        if it in (1, 3):
            assert torch.equal(m.rad_w.detach(), before)          # no momentum-only update of a skipped group
    t = [g_.get("t", 0) for g_ in opt.groups]
    assert t[rad] == 3 and all(x == 5 for i, x in enumerate(t) if i != rad) and opt.t == 5
    for p, r in zip(ps, ref):
        assert float((p.detach().cpu() - r.detach()).abs().max()) <= 2e-6 * (1e-2 + float(r.detach().abs().max()))


def test_lazy_table_adam_updates_touched_entries_only(backend):
    """SURVEY sec. 8f-3 (opt-in, ``FusedAdam(lazy_tables=True)`` / NSIM_LAZY_ADAM=1): hash-table entries whose gradient is zero
    in a step keep value AND moments; touched entries follow Adam with the group's step count (torch.optim.SparseAdam's rule,
    restated here entry by entry); the decoder tensors stay dense; the default (False) is the reference's dense Adam."""
    from neuralsim_amd.optim import FusedAdam
    from test_trainer import _tiny
    torch.manual_seed(0)
    m = _tiny(backend)
    opt = FusedAdam(m, lr=1e-2, eps=1e-15, lazy_tables=True)
    assert FusedAdam(_tiny(backend), lr=1e-2).lazy_tables is False
    gp = m.encoding.flattened_params
    n = gp.numel()
    p_ref, m_ref, v_ref = gp.detach().cpu().double().clone(), torch.zeros(n, dtype=torch.float64), torch.zeros(n, dtype=torch.float64)
    w0 = m.sdf_w.detach().clone()
    g = torch.Generator().manual_seed(2)
    b1, b2 = 0.9, 0.99
    for it in range(1, 5):
        grad = torch.randn(n, generator=g) * 1e-3
        grad[torch.rand(n, generator=g) < 0.6] = 0.0                      # 60 % of the entries untouched this step
        gp.grad = grad.to(backend)
        m.sdf_w.grad = torch.zeros_like(m.sdf_w)                           # a dense tensor with a zero gradient still decays / moves
        m.sdf_w.grad[0] = 1e-3 if it == 1 else 0.0
        opt.step()
        t_ = grad != 0
        gd = grad.double()
        m_ref[t_] = b1 * m_ref[t_] + (1 - b1) * gd[t_]
        v_ref[t_] = b2 * v_ref[t_] + (1 - b2) * gd[t_] ** 2
        p_ref[t_] -= 1e-2 / (1 - b1 ** it) * m_ref[t_] / (v_ref[t_].sqrt() / (1 - b2 ** it) ** 0.5 + 1e-15)
    assert float((gp.detach().cpu().double() - p_ref).abs().max()) < 1e-6
    assert torch.equal(m.encoding.shadow().cpu(), gp.detach().half().cpu())           # the fp16 shadow follows the touched entries
    # dense tensor: the entry touched only in step 1 kept moving in steps 2-4 (momentum), as dense Adam does
    assert float((m.sdf_w.detach()[0] - w0[0]).abs()) > 1.5e-2
