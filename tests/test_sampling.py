"""Ray generation, AABB test, occupancy marching, up-sampling, sdf->alpha and fused compositing vs the oracle."""
import torch

from oracle import pack_ops as opo, render as orr
from neuralsim_amd import _lib
from neuralsim_amd.fields.neus import OccGridAccel, _NeusAlphaFn, volume_integration
from neuralsim_amd.graphics import pack_ops as po
from util import leaf, look_at_cameras

AABB = torch.tensor([[-1.0, -1, -1], [1.0, 1, 1]])


def _rays(N=300, seed=1):
    g = torch.Generator().manual_seed(seed)
    intr, c2w, WH = look_at_cameras(V=5, seed=seed)
    xy = torch.rand(N, 2, generator=g)
    fidx = torch.randint(0, 5, (N,), generator=g)
    return xy, fidx, intr, c2w, WH


def test_raygen_and_aabb(backend):
    xy, fidx, intr, c2w, WH = _rays()
    o_ref, d_ref = orr.pinhole_rays(xy, fidx, intr, c2w, WH)
    N = xy.shape[0]
    o = torch.zeros(N, 3, device=backend)
    d = torch.zeros(N, 3, device=backend)
    _lib.call("nsim_raygen_pinhole", _lib.ptr(xy.to(backend)), _lib.ptr(fidx.to(backend)), _lib.ptr(intr.to(backend)),
              _lib.ptr(c2w.to(backend)), _lib.ptr(WH.to(backend)), N, 1, _lib.ptr(o), _lib.ptr(d))
    assert torch.allclose(o.cpu(), o_ref, atol=0) and torch.allclose(d.cpu(), d_ref, atol=2e-7)
    # zoom some rays out of the box
    o_ref[::3] += torch.tensor([0.0, 5.0, 0.0])
    acc = OccGridAccel(AABB, device=backend)
    n_ref, f_ref, hit_ref = orr.aabb_ray_test(o_ref, d_ref, AABB[0], AABB[1], 0.01, None)
    nt = torch.zeros(N, device=backend); ft = torch.zeros(N, device=backend)
    hit = torch.zeros(N, dtype=torch.uint8, device=backend)
    _lib.call("nsim_aabb_ray_test", _lib.ptr(o_ref.to(backend)), _lib.ptr(d_ref.to(backend)), N, acc.meta, 0.01, -1.0,
              _lib.ptr(nt), _lib.ptr(ft), _lib.ptr(hit))
    assert torch.equal(hit.cpu().bool(), hit_ref)
    assert 0 < int(hit_ref.sum()) < N
    assert torch.allclose(nt.cpu()[hit_ref], n_ref[hit_ref], atol=1e-6) and torch.allclose(ft.cpu()[hit_ref], f_ref[hit_ref], atol=1e-6)


def test_raygen_pose_gradient(backend):
    """Pose refinement: d loss / d c2w through the ray generation (rays_o = T, rays_d = R l / |R l|) vs autograd on
    the oracle; the bottom row of every pose gets no gradient, untouched frames none at all."""
    from neuralsim_amd.graphics.cameras import pinhole_selected_rays
    xy, fidx, intr, c2w, WH = _rays(N=257)
    fidx[fidx == 3] = 2                                   # frame 3 is never sampled
    g = torch.Generator().manual_seed(9)
    wo, wd = torch.randn(257, 3, generator=g), torch.randn(257, 3, generator=g)
    c_r = leaf(c2w)
    o_r, d_r = orr.pinhole_rays(xy, fidx, intr, c_r, WH)
    ((o_r * wo).sum() + (d_r * wd).sum()).backward()
    c_d = leaf(c2w, backend)
    o, d = pinhole_selected_rays(xy.to(backend), fidx.to(backend), intr.to(backend), c_d, WH.to(backend))
    assert torch.allclose(o.detach().cpu(), o_r.detach(), atol=0) and torch.allclose(d.detach().cpu(), d_r.detach(), atol=2e-7)
    ((o * wo.to(backend)).sum() + (d * wd.to(backend)).sum()).backward()
    assert torch.allclose(c_d.grad.cpu(), c_r.grad, rtol=2e-4, atol=2e-5)
    assert float(c_d.grad[:, 3].abs().max()) == 0.0 and float(c_d.grad[3].abs().max()) == 0.0


def test_raygen_opencv_distortion(backend):
    """``camera_model: opencv`` (cameras.py:84-87; the street configs' ``consider_distortion: true``): the lift undistorts
    the pixel with the fixed-point iteration of cv::undistortPoints.  (1) kernel == oracle restatement; (2) the lifted
    direction projects back onto the pixel through the FORWARD distortion model (Waymo-sized coefficients, 50-degree
    lens: 5 rounds reach 2e-3 px, 10 rounds 1e-4 px); (3) zero coefficients == the pinhole kernel bit for bit; (4) the
    pose gradient goes through the same lift."""
    from neuralsim_amd.graphics.cameras import opencv_selected_rays, pinhole_selected_rays
    g = torch.Generator().manual_seed(5)
    V, N = 5, 400
    intr, c2w, WH = look_at_cameras(V=V, seed=2, H=1280, W=1920, f=2055.0)        # Waymo front camera
    dist = torch.tensor([[0.043, -0.36, 0.0008, -0.0006, 0.0]]).repeat(V, 1)
    dist = dist * (1.0 + 0.1 * torch.randn(V, 5, generator=g))
    dist[:, 4] = 0.02 * torch.randn(V, generator=g)
    xy = torch.rand(N, 2, generator=g)
    fidx = torch.randint(0, V, (N,), generator=g)
    dv = lambda t: t.to(backend)
    for n_iters in (5, 10):
        o_ref, d_ref = orr.pinhole_rays(xy, fidx, intr, c2w, WH, distortion=dist, n_iters=n_iters)
        o, d = opencv_selected_rays(dv(xy), dv(fidx), dv(intr), dv(dist), dv(c2w), dv(WH), n_iters=n_iters)
        assert torch.equal(o.cpu(), o_ref) and torch.allclose(d.cpu(), d_ref, atol=3e-7)
        # back through the forward model: direction in the camera frame -> distorted pixel
        Rm = c2w[fidx, :3, :3]
        l = (Rm.transpose(1, 2) * d.cpu().unsqueeze(-2)).sum(-1)
        xu, yu = l[:, 0] / l[:, 2], l[:, 1] / l[:, 2]
        xd, yd = orr.opencv_distort(xu, yu, dist[fidx])
        K = intr[fidx]
        u, v = xd * K[:, 0, 0] + K[:, 0, 2], yd * K[:, 1, 1] + K[:, 1, 2]
        wh = (xy * WH[fidx]).long().clamp(torch.zeros_like(WH[fidx]), WH[fidx] - 1).float() + 0.5
        err = torch.maximum((u - wh[:, 0]).abs(), (v - wh[:, 1]).abs())
        assert float(err.max()) < (2e-2 if n_iters == 5 else 2e-3), (n_iters, float(err.max()))
    # a distorted lens is not a pinhole ...
    o_p, d_p = pinhole_selected_rays(dv(xy), dv(fidx), dv(intr), dv(c2w), dv(WH))
    assert float((d_p - d).abs().max()) > 1e-3
    # ... and zero coefficients are exactly one
    o_z, d_z = opencv_selected_rays(dv(xy), dv(fidx), dv(intr), dv(torch.zeros(V, 5)), dv(c2w), dv(WH))
    assert torch.equal(o_z, o_p) and torch.equal(d_z, d_p)
    # pose gradient
    wo, wd = torch.randn(N, 3, generator=g), torch.randn(N, 3, generator=g)
    c_r = leaf(c2w)
    o_r, d_r = orr.pinhole_rays(xy, fidx, intr, c_r, WH, distortion=dist, n_iters=5)
    ((o_r * wo).sum() + (d_r * wd).sum()).backward()
    c_d = leaf(c2w, backend)
    o, d = opencv_selected_rays(dv(xy), dv(fidx), dv(intr), dv(dist), c_d, dv(WH))
    ((o * dv(wo)).sum() + (d * dv(wd)).sum()).backward()
    assert torch.allclose(c_d.grad.cpu(), c_r.grad, rtol=2e-4, atol=2e-5)


def test_raygen_fisheye(backend):
    """``camera_model: fisheye`` (cameras.py:88-92): the lift inverts the OpenCV fisheye polynomial the reference applies in
    app/resources/observers/fisheye.py:36-42 (its calibration constants are the test lens).  (1) kernel == oracle
    restatement; (2) the lifted direction goes back onto the pixel through the reference's FORWARD formulas (10 Newton
    rounds: < 1e-3 px over a 150-degree lens); (3) the shim's ``FisheyeCameraMatHW.lift`` / ``proj`` are the same
    arithmetic (what the reference's ``Camera`` calls); (4) zero coefficients = the equidistant lens; (5) pose gradient."""
    from neuralsim_amd.graphics.cameras import fisheye_selected_rays, selected_rays
    g = torch.Generator().manual_seed(8)
    V, N = 4, 500
    W_, H_ = 1280, 960
    K = torch.tensor([[318.44998905930794, 0.0, 636.2089399611955], [0.0, 317.8314899911656, 481.71423781914115],
                      [0.0, 0.0, 1.0]])                                            # fisheye.py:14-17
    D = torch.tensor([0.18198802503702904, -0.04198598106075817, 0.010013633995507613, -0.0025294664427881705])   # :21
    _, c2w, _ = look_at_cameras(V=V, seed=3)
    intr = K.repeat(V, 1, 1)
    WH = torch.tensor([[W_, H_]], dtype=torch.long).repeat(V, 1)
    dist = D.repeat(V, 1) * (1.0 + 0.05 * torch.randn(V, 4, generator=g))
    # pixels inside the lens's image circle: theta_d <= 1.8 (~83 degrees off axis; the polynomial of this lens peaks at 2.19,
    # the corners of the 1280 x 960 frame lie beyond it -- black in a real fisheye image, no theta solves them)
    cand = torch.rand(6 * N, 2, generator=g)
    td_c = torch.sqrt(((cand[:, 0] * W_ - K[0, 2]) / K[0, 0]) ** 2 + ((cand[:, 1] * H_ - K[1, 2]) / K[1, 1]) ** 2)
    xy = cand[td_c <= 1.8][:N].contiguous()
    assert xy.shape[0] == N
    fidx = torch.randint(0, V, (N,), generator=g)
    dv = lambda t: t.to(backend)
    o_ref, d_ref = orr.pinhole_rays(xy, fidx, intr, c2w, WH, distortion=dist, n_iters=10, camera_model="fisheye")
    o, d = fisheye_selected_rays(dv(xy), dv(fidx), dv(intr), dv(dist), dv(c2w), dv(WH))
    assert torch.equal(o.cpu(), o_ref) and torch.allclose(d.cpu(), d_ref, atol=5e-7)
    o2, d2 = selected_rays(dv(xy), dv(fidx), dv(intr), dv(c2w), dv(WH), distortion=dv(dist))      # [V,4] selects the model
    assert torch.equal(d2, d)
    # back through the reference's forward formulas: direction in the camera frame -> distorted pixel
    Rm = c2w[fidx, :3, :3]
    l = (Rm.transpose(1, 2) * d.cpu().unsqueeze(-2)).sum(-1)
    assert float(l[:, 2].min()) > 0.0                                   # this lens stays inside 90 degrees off axis
    xd, yd = orr.fisheye_distort(l[:, 0] / l[:, 2], l[:, 1] / l[:, 2], dist[fidx])
    u, v = xd * K[0, 0] + K[0, 2], yd * K[1, 1] + K[1, 2]
    wh = (xy * WH[fidx]).long().clamp(torch.zeros_like(WH[fidx]), WH[fidx] - 1).float() + 0.5
    err = torch.maximum((u - wh[:, 0]).abs(), (v - wh[:, 1]).abs())
    assert float(err.max()) < 1e-3 * 5, float(err.max())
    off_axis = torch.acos(l[:, 2].clamp(-1, 1))
    assert float(off_axis.max()) > 1.2                                  # the corners of the image: ~75 degrees off axis
    # the shim class the reference's Camera code calls
    from nr3d_lib.models.attributes import CameraMatrix3x3, FisheyeCameraMatHW, make_vector
    cam = FisheyeCameraMatHW(mat=CameraMatrix3x3(intr[fidx]), H=WH[fidx][:, 1].float(), W=WH[fidx][:, 0].float(),
                             distortion=make_vector(4)(dist[fidx]))
    lifted = cam.lift(wh[:, 0], wh[:, 1], torch.ones(N))
    assert torch.allclose(lifted / lifted.norm(dim=-1, keepdim=True), l / l.norm(dim=-1, keepdim=True), atol=2e-6)
    uu, vv, _ = cam.proj(lifted * 3.0)
    assert float(torch.maximum((uu - wh[:, 0]).abs(), (vv - wh[:, 1]).abs()).max()) < 5e-3
    # zero coefficients: the equidistant lens, pixel radius = f * theta
    z4 = torch.zeros(V, 4)
    _, d_e = fisheye_selected_rays(dv(xy), dv(fidx), dv(intr), dv(z4), dv(c2w), dv(WH))
    l_e = (Rm.transpose(1, 2) * d_e.cpu().unsqueeze(-2)).sum(-1)
    th = torch.acos(l_e[:, 2].clamp(-1, 1))
    rpx = torch.sqrt(((wh[:, 0] - K[0, 2]) / K[0, 0]) ** 2 + ((wh[:, 1] - K[1, 2]) / K[1, 1]) ** 2)
    assert torch.allclose(th, rpx, atol=2e-6)
    # pose gradient
    wo, wd = torch.randn(N, 3, generator=g), torch.randn(N, 3, generator=g)
    c_r = leaf(c2w)
    o_r, d_r = orr.pinhole_rays(xy, fidx, intr, c_r, WH, distortion=dist, n_iters=10, camera_model="fisheye")
    ((o_r * wo).sum() + (d_r * wd).sum()).backward()
    c_d = leaf(c2w, backend)
    o, d = fisheye_selected_rays(dv(xy), dv(fidx), dv(intr), dv(dist), c_d, dv(WH))
    ((o * dv(wo)).sum() + (d * dv(wd)).sum()).backward()
    assert torch.allclose(c_d.grad.cpu(), c_r.grad, rtol=2e-4, atol=2e-5)


def _sphere_occ(res=(64, 64, 64), shell=0.03):
    ax = [(torch.arange(r) + 0.5) / r * 2 - 1 for r in res]
    zz, yy, xx = torch.meshgrid(ax[2], ax[1], ax[0], indexing="ij")
    occ = ((xx ** 2 + yy ** 2 + zz ** 2).sqrt() - 0.5).abs() < shell
    return occ.reshape(-1)


def test_march_bit_exact(backend):
    xy, fidx, intr, c2w, WH = _rays(N=200, seed=3)
    o, d = orr.pinhole_rays(xy, fidx, intr, c2w, WH)
    near, far, hit = orr.aabb_ray_test(o, d, AABB[0], AABB[1], 0.01, None)
    o, d, near, far = o[hit], d[hit], near[hit], far[hit]
    R = o.shape[0]
    occ = _sphere_occ()
    res = torch.tensor([64, 64, 64])
    scale = res.float() / (AABB[1] - AABB[0])
    g = torch.Generator().manual_seed(0)
    for jitter in (None, torch.rand(R, generator=g)):
        jit = jitter if jitter is not None else torch.full((R,), 0.5)
        t_ref, ridx_ref, cnt_ref = orr.march_lattice(o, d, near, far, jit, occ, AABB[0], scale, res, 0.005, 4096)
        acc = OccGridAccel(AABB, device=backend)
        acc.occ_val.copy_(occ.float().to(backend))
        acc.pack_bits()
        dv = lambda t: t.to(backend).contiguous()
        counts = torch.zeros(R, dtype=torch.long, device=backend)
        jp = _lib.ptr(dv(jitter)) if jitter is not None else None
        args = (_lib.ptr(dv(o)), _lib.ptr(dv(d)), _lib.ptr(dv(near)), _lib.ptr(dv(far)), jp, R, _lib.ptr(acc.occ_bits),
                None, acc.meta, 0.005, 4096)
        _lib.call("nsim_march_count", *args, _lib.ptr(counts))
        assert torch.equal(counts.cpu(), cnt_ref)          # bit-exact sample membership
        assert int(cnt_ref.sum()) > 0
        pi = po.get_pack_infos_from_n(counts)
        t = torch.zeros(int(cnt_ref.sum()), device=backend)
        _lib.call("nsim_march_emit", *args, _lib.ptr(pi), _lib.ptr(t))
        assert torch.equal(t.cpu(), t_ref)
    # max_steps clamp
    t_ref, _, cnt_ref = orr.march_lattice(o, d, near, far, torch.full((R,), 0.5), torch.ones_like(occ), AABB[0], scale, res, 0.005, 100)
    acc.set_all_occupied()
    args = args[:4] + (None,) + args[5:-1]
    _lib.call("nsim_march_count", *args, 100, _lib.ptr(counts))
    assert torch.equal(counts.cpu(), cnt_ref) and int(cnt_ref.max()) == 100


def test_occ_update_and_bits(backend):
    g = torch.Generator().manual_seed(2)
    pts = torch.rand(5000, 3, generator=g) * 2.2 - 1.1     # some outside
    sdf = pts.norm(dim=-1) - 0.5
    res = torch.tensor([16, 8, 4])
    scale = res.float() / (AABB[1] - AABB[0])
    val0 = torch.rand(16 * 8 * 4, generator=g) * 0.5
    ref = orr.occ_update(val0, pts, sdf, AABB[0], scale, res, decay=0.95, inv_s=64.0)
    acc = OccGridAccel(AABB, resolution=(16, 8, 4), inv_s=64.0, device=backend)
    acc.occ_val.copy_(val0.to(backend))
    acc.update_from_samples(pts.to(backend), sdf.to(backend))
    assert torch.allclose(acc.occ_val.cpu(), ref, atol=1e-6)
    bits = acc.occ_bits.cpu().numpy().view("uint32")
    occ = (ref > 0.3).numpy()
    for v in range(occ.shape[0]):
        assert bool((bits[v >> 5] >> (v & 31)) & 1) == bool(occ[v])


def test_bench_oracle_marches_the_packed_bitfield(backend):
    """bench.oracle_of hands the oracle the occupancy the KERNELS march against -- the bitfield packed at the last refresh -- not
    the value grid thresholded now: between refreshes the sampling passes fold SDFs into the values (``collect``: max, bits
    untouched), and an oracle on the values marched voxels the kernels did not (round 6: one ray of the bench's 2048 at 0.06)."""
    import sys
    from pathlib import Path
    from types import SimpleNamespace
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    import bench
    from util import make_params, model_from_params
    p = make_params(sdf_D=2, small=True, sphere=True, seed=3)
    model = model_from_params(p, backend, precision="f32")
    acc = OccGridAccel(AABB, resolution=(16, 8, 4), inv_s=64.0, device=backend)
    g = torch.Generator().manual_seed(5)
    acc.occ_val.copy_((torch.rand(16 * 8 * 4, generator=g) * 0.6).to(backend))
    acc.pack_bits()
    at_pack = (acc.occ_val.cpu() > acc.occ_thre)
    # what a sampling pass does between refreshes: values go up, the bits stay
    pts = (torch.rand(400, 3, generator=g) * 2 - 1).to(backend)
    acc.collect(pts, torch.zeros(400, device=backend))
    assert int(((acc.occ_val.cpu() > acc.occ_thre) != at_pack).sum()) > 0
    model.accel = acc
    _, occ = bench.oracle_of(SimpleNamespace(model=model), table="master")
    assert torch.equal(occ, at_pack)


def test_coarse_upsample_merge(backend):
    g = torch.Generator().manual_seed(7)
    R, C = 23, 64
    near = torch.rand(R, generator=g) * 0.5 + 2.0
    far = near + 1.0 + torch.rand(R, generator=g)
    jc = torch.rand(R, C, generator=g)
    dv = lambda t: t.to(backend).contiguous()
    for j in (None, jc):
        out = torch.zeros(R, C, device=backend)
        _lib.call("nsim_coarse_depths", _lib.ptr(dv(near)), _lib.ptr(dv(far)), _lib.ptr(dv(j)) if j is not None else None,
                  R, C, _lib.ptr(out), None)
        assert torch.equal(out.cpu(), orr.coarse_depths(near, far, C, j))
    # ragged packs incl. > 64 samples; sdf of a sphere crossing
    n = torch.randint(2, 200, (R,), generator=g)
    n[0], n[1], n[2] = 2, 64, 65
    pi = opo.get_pack_infos_from_n(n)
    S = int(n.sum())
    ridx = opo.pack_ridx(pi, S)
    u = torch.rand(S, generator=g)
    t = near[ridx] + u * (far - near)[ridx]
    t, _ = opo.packed_sort(t, pi)
    sdf = (t - (near + 0.6 * (far - near))[ridx]) * -0.7 + 0.01 * torch.randn(S, generator=g)
    for use_est in (True, False):
        for inv_s, nf in ((64.0, 8), (1024.0, 32), (256.0, 70)):
            ref = orr.upsample_stage(t, sdf, pi, inv_s, nf, use_est)
            t_new = torch.zeros(R, nf, device=backend)
            scratch = torch.zeros(S, device=backend)
            ro = torch.randn(R, 3, generator=torch.Generator().manual_seed(1))
            rd = torch.nn.functional.normalize(torch.randn(R, 3, generator=torch.Generator().manual_seed(2)), dim=-1)
            x_new = torch.zeros(R, nf, 3, device=backend)
            _lib.call("nsim_upsample_stage", _lib.ptr(dv(t)), _lib.ptr(dv(sdf)), _lib.ptr(dv(pi)), R, inv_s, nf,
                      1 if use_est else 0, _lib.ptr(scratch), _lib.ptr(t_new), _lib.ptr(dv(ro)), _lib.ptr(dv(rd)),
                      _lib.ptr(x_new), None)
            assert torch.allclose(t_new.cpu(), ref, atol=2e-5), (use_est, inv_s, nf, (t_new.cpu() - ref).abs().max())
            assert torch.equal(x_new.cpu(), ro[:, None, :] + t_new.cpu()[..., None] * rd[:, None, :])   # bit-exact o + t d
            assert (t_new.cpu()[:, 1:] >= t_new.cpu()[:, :-1]).all()
    # merge
    nf = 8
    t_b = orr.upsample_stage(t, sdf, pi, 64.0, nf, True)
    t_b[3, 2] = t[int(pi[3, 0]) + 1]      # exact tie: a-first
    t_b, _ = t_b.sort(dim=1)
    v_b = torch.randn(R, nf, generator=g)
    t_ref, pi_ref, pa, pb = orr.merge_sorted(t, pi, t_b)
    v_ref = torch.empty_like(t_ref); v_ref[pa] = sdf; v_ref[pb.reshape(-1)] = v_b.reshape(-1)
    t_out = torch.zeros(S + R * nf, device=backend); v_out = torch.zeros_like(t_out)
    pi_out = torch.zeros(R, 2, dtype=torch.long, device=backend)
    ridx_out = torch.zeros(S + R * nf, dtype=torch.long, device=backend)
    x_out = torch.zeros(S + R * nf, 3, device=backend)
    _lib.call("nsim_merge_sorted", _lib.ptr(dv(t)), _lib.ptr(dv(sdf)), _lib.ptr(dv(pi)), _lib.ptr(dv(t_b)), _lib.ptr(dv(v_b)),
              R, nf, _lib.ptr(t_out), _lib.ptr(v_out), _lib.ptr(pi_out), _lib.ptr(ridx_out), _lib.ptr(dv(ro)),
              _lib.ptr(dv(rd)), _lib.ptr(x_out), None)
    assert torch.equal(x_out.cpu(), ro[ridx_out.cpu()] + t_out.cpu()[:, None] * rd[ridx_out.cpu()])
    assert torch.equal(pi_out.cpu(), pi_ref) and torch.equal(t_out.cpu(), t_ref) and torch.equal(v_out.cpu(), v_ref)
    assert torch.equal(ridx_out.cpu(), opo.pack_ridx(pi_ref, t_ref.shape[0]))


def test_neus_alpha_and_composite(backend):
    g = torch.Generator().manual_seed(11)
    P = 19
    n = torch.randint(1, 150, (P,), generator=g)
    n[2], n[5] = 1, 130
    pi = opo.get_pack_infos_from_n(n)
    S = int(n.sum())
    ridx = opo.pack_ridx(pi, S)
    t = opo.packed_sort(torch.rand(S, generator=g) * 3, pi)[0]
    sdf = 0.2 - 0.15 * (t - 1.0) + 0.01 * torch.randn(S, generator=g)
    rgb = torch.rand(S, 3, generator=g)
    nrm = torch.randn(S, 3, generator=g)
    ln_inv_s = torch.tensor([0.35])
    for fis in (0.0, 50.0):
        for nd in (False, True):
            s_o, l_o, r_o, n_o = leaf(sdf, dtype=torch.double), leaf(ln_inv_s, dtype=torch.double), leaf(rgb, dtype=torch.double), leaf(nrm, dtype=torch.double)
            inv_s = torch.exp(l_o * 10.0) if fis == 0.0 else torch.tensor(fis, dtype=torch.double)
            a_ref = orr.neus_alpha_packed(s_o, pi, inv_s)
            out_ref = orr.volume_integration(a_ref, t.double(), r_o, n_o, pi, nd)
            s_d, l_d, r_d, n_d = leaf(sdf, backend), leaf(ln_inv_s, backend), leaf(rgb, backend), leaf(nrm, backend)
            a = _NeusAlphaFn.apply(s_d, l_d, pi.to(backend), 10.0, fis)
            out = volume_integration(a, t.to(backend), r_d, n_d, pi.to(backend), nd)
            assert torch.allclose(a.cpu().double(), a_ref, atol=2e-6)
            ws = {k: torch.randn(out_ref[k].shape, generator=g, dtype=torch.double) for k in out_ref}
            loss_ref = sum((out_ref[k] * ws[k]).sum() for k in out_ref)
            loss = sum((out[k] * ws[k].float().to(backend)).sum() for k in out_ref)
            for k in out_ref:
                assert torch.allclose(out[k].cpu().double(), out_ref[k], atol=2e-5, rtol=1e-5), k
            loss_ref.backward()
            loss.backward()
            for a_, b_, nm in ((s_d, s_o, "sdf"), (r_d, r_o, "rgb"), (n_d, n_o, "nrm")):
                err = (a_.grad.cpu().double() - b_.grad).abs().max() / b_.grad.abs().max().clamp_min(1e-9)
                assert err < 2e-4, (nm, fis, nd, float(err))
            if fis == 0.0:
                assert abs(float(l_d.grad.cpu()) - float(l_o.grad)) / max(1e-9, abs(float(l_o.grad))) < 2e-3


def test_occupancy_collects_the_sampling_pass(backend):
    """``accel_cfg.update_from_samples_cfg: {}`` (dtu yaml:158): one armed training-step sampling pass max-folds the SDF of
    every sample it queries into the value grid (no decay, bits untouched until the next refresh) == the oracle's
    ``occ_collect`` on the oracle's own samples; evaluation passes (not armed) leave the grid alone."""
    from oracle import render as orr
    from util import look_at_cameras, make_params, model_from_params
    from neuralsim_amd.fields.neus import OccGridAccel
    p = make_params(sdf_D=2, small=True, sphere=True, seed=3, ln_inv_s=0.45, grid_bound=2e-2, noise_scale=1.0)
    g = torch.Generator().manual_seed(1)
    intr, c2w, WH = look_at_cameras(V=3, seed=3)
    N = 40
    o, d = orr.pinhole_rays(torch.rand(N, 2, generator=g) * 0.5 + 0.25, torch.randint(0, 3, (N,), generator=g), intr, c2w, WH)
    aabb = torch.tensor([[-1.0, -1, -1], [1.0, 1, 1]])
    res = [16, 16, 16]
    model = model_from_params(p, backend, precision="f32")
    model.accel = OccGridAccel(aabb, resolution=res, device=backend, update_from_samples_cfg={})
    val0, occ = orr.build_occ_grid(p, aabb[0], aabb[1], res, n_pts=2 ** 12, n_steps=2)
    val0 = val0 * 0.5                                           # so that fresh samples can raise values
    model.accel.occ_val.copy_(val0.to(backend))
    model.accel.pack_bits()
    bits0 = model.accel.occ_bits.clone()
    qp = dict(nablas_has_grad=True, num_coarse=16, num_fine=[4, 4, 8], upsample_inv_s=64.0, upsample_inv_s_factors=[1, 4, 16],
              upsample_use_estimate_alpha=True, march_cfg=dict(step_size=0.02, max_steps=512))
    cfg = dict(query_param=qp, with_rgb=True, query_mode="march_occ_multi_upsample")
    dv = lambda a: a.to(backend).contiguous()        # noqa: E731
    tested = model.ray_test(dv(o), dv(d), near=0.01)
    model.ray_query(ray_tested=tested, config=cfg)               # not armed: an evaluation render
    assert torch.equal(model.accel.occ_val.cpu(), val0)
    model.training_before_per_step(0)                            # arms ONE pass
    assert model.accel.collect_armed
    model.ray_query(ray_tested=tested, config=cfg)
    assert not model.accel.collect_armed
    ret_o = orr.ray_query(p, o, d, None, val0 > 0.3, aabb[0], aabb[1], res, near=0.01, num_coarse=16, num_fine=(4, 4, 8),
                          step_size=0.02, max_steps=512)
    res_t = torch.tensor(res)
    want = orr.occ_collect(val0, ret_o["debug"]["x_nograd"], ret_o["debug"]["sdf_nograd"], aabb[0],
                           res_t.float() / (aabb[1] - aabb[0]), res_t)
    got = model.accel.occ_val.cpu()
    assert float((want > val0).float().mean()) > 0.01           # the pass did raise values (samples on the marched rays only)
    # identical up to fine samples that sit within rounding of a voxel face (their positions differ by ~1e-6)
    assert float(((got - want).abs() > 1e-4).float().mean()) < 2e-3
    assert torch.equal(model.accel.occ_bits, bits0)              # thresholded only by the next refresh
    model.ray_query(ray_tested=tested, config=cfg)               # a second, un-armed pass: unchanged
    assert torch.equal(model.accel.occ_val.cpu(), got)


def test_refresh_points_are_a_stratified_sweep(backend):
    """``OccGridAccel.draw_points``: point i of a pass lies in voxel (sweep + i) mod n_voxels (storage order), the sweep
    continues across passes, offsets inside the voxel are uniform -- every voxel receives the same number of queries."""
    from neuralsim_amd.fields.neus import OccGridAccel
    aabb = torch.tensor([[-1.0, -0.5, -0.25], [1.0, 0.5, 0.75]])
    acc = OccGridAccel(aabb, resolution=(8, 4, 2), device=backend)
    nvox = 64
    g = torch.Generator(device=backend).manual_seed(0)
    seen = torch.zeros(nvox, dtype=torch.long)
    start = 0
    for n in (100, 64, 28, 192):
        pts = acc.draw_points(n, g).cpu()
        u = (pts - aabb[0]) / (aabb[1] - aabb[0]) * torch.tensor([8.0, 4.0, 2.0])
        assert bool((u >= 0).all()) and bool((u < torch.tensor([8.0, 4.0, 2.0])).all())
        ijk = u.floor().long()
        v = ijk[:, 0] + 8 * (ijk[:, 1] + 4 * ijk[:, 2])
        assert torch.equal(v, (torch.arange(n) + start) % nvox)
        frac = u - u.floor()
        assert 0.3 < float(frac.mean()) < 0.7
        seen += torch.bincount(v, minlength=nvox)
        start = (start + n) % nvox
    assert int(seen.min()) == int(seen.max()) == 6          # 384 points over 64 voxels


def _rand_packs(g, n_list):
    from oracle import pack_ops as opo
    n = torch.tensor(n_list)
    pi = opo.get_pack_infos_from_n(n)
    S = int(n.sum())
    ridx = opo.pack_ridx(pi, S)
    R = n.shape[0]
    near = torch.rand(R, generator=g) * 0.5 + 2.0
    far = near + 1.0 + torch.rand(R, generator=g)
    t = near[ridx] + torch.rand(S, generator=g) * (far - near)[ridx]
    t, _ = opo.packed_sort(t, pi)
    t = t + torch.arange(S).float() * 1e-5
    sdf = (near[ridx] + 0.4 + 0.4 * torch.rand(R, generator=g)[ridx] - t) * 0.6 + 0.01 * torch.randn(S, generator=g)
    return n, pi, S, ridx, t, sdf


def test_merge_upsample_equals_merge_then_upsample(backend):
    """``nsim_merge_upsample`` (merge of stage k + draws of stage k + 1, the default path) against the two launches
    it replaces -- ``nsim_merge_sorted`` then ``nsim_upsample_stage`` -- on random packs incl. rays on both sides of the
    per-wave LDS window (SMP_LDS_FLOATS = 512 floats): bit-equal t_out / v_out / pack_infos / ridx / t_new / x_new."""
    g = torch.Generator().manual_seed(21)
    n, pi, S, ridx, t, sdf = _rand_packs(g, [2, 3, 64, 65, 200, 505, 512, 513, 700, 17, 1, 129])
    R = n.shape[0]
    dv = lambda a: a.to(backend).contiguous()        # noqa: E731
    for nb, nf2, use_est in ((8, 8, 1), (8, 32, 0), (32, 70, 1)):
        t_b = torch.sort(t[pi[:, 0]][:, None] + torch.rand(R, nb, generator=g) * (t[pi[:, 0] + pi[:, 1] - 1] - t[pi[:, 0]])[:, None], dim=1).values
        v_b = torch.randn(R, nb, generator=g) * 0.1
        ro = torch.randn(R, 3, generator=g)
        rd = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
        T = S + R * nb
        outs = []
        for fused in (True, False):
            t_out, v_out = torch.zeros(T, device=backend), torch.zeros(T, device=backend)
            pi_out = torch.zeros(R, 2, dtype=torch.long, device=backend)
            ridx_out = torch.zeros(T, dtype=torch.long, device=backend)
            scratch = torch.zeros(T, device=backend)
            t_new = torch.zeros(R, nf2, device=backend)
            x_new = torch.zeros(R * nf2, 3, device=backend)
            if fused:
                _lib.call("nsim_merge_upsample", _lib.ptr(dv(t)), _lib.ptr(dv(sdf)), _lib.ptr(dv(pi)), _lib.ptr(dv(t_b)), _lib.ptr(dv(v_b)),
                          R, nb, _lib.ptr(t_out), _lib.ptr(v_out), _lib.ptr(pi_out), _lib.ptr(ridx_out), 256.0, nf2, use_est,
                          _lib.ptr(scratch), _lib.ptr(t_new), _lib.ptr(dv(ro)), _lib.ptr(dv(rd)), _lib.ptr(x_new), None)
            else:
                _lib.call("nsim_merge_sorted", _lib.ptr(dv(t)), _lib.ptr(dv(sdf)), _lib.ptr(dv(pi)), _lib.ptr(dv(t_b)), _lib.ptr(dv(v_b)),
                          R, nb, _lib.ptr(t_out), _lib.ptr(v_out), _lib.ptr(pi_out), _lib.ptr(ridx_out), None, None, None, None)
                _lib.call("nsim_upsample_stage", _lib.ptr(t_out), _lib.ptr(v_out), _lib.ptr(pi_out), R, 256.0, nf2, use_est,
                          _lib.ptr(scratch), _lib.ptr(t_new), _lib.ptr(dv(ro)), _lib.ptr(dv(rd)), _lib.ptr(x_new), None)
            outs.append([a.cpu() for a in (t_out, v_out, pi_out, ridx_out, t_new, x_new)])
        for a, b, nm in zip(outs[0], outs[1], ("t_out", "v_out", "pack_infos", "ridx", "t_new", "x_new")):
            assert torch.equal(a, b), (nb, nf2, use_est, nm)
        assert bool((outs[0][4][:, 1:] >= outs[0][4][:, :-1]).all())


def test_live_rank_beyond_2_to_the_19_live_rays(backend):
    """an 800x800 image queried in one chunk (``rayschunk: 0``) is 640 k rays; rounds 4-5 carried the live count in bits 44..63
    of the sample-count scan, which reached the sign bit at 2^19 live rays.  700 k rays, 5 in 6 live."""
    R = 700_001 if backend.type == "cuda" else 530_000
    g = torch.Generator().manual_seed(11)
    counts = torch.randint(0, 6, [R], generator=g)
    counts[R - 1] = 3
    live = counts > 0
    Rl = int(live.sum())
    assert Rl > 2 ** 19 or backend.type != "cuda"
    C, nfs = 16, [8, 8, 16, 0]
    lr = torch.full([R], 123, dtype=torch.long, device=backend)
    lidx = torch.full([R], 123, dtype=torch.long, device=backend)
    cnts = torch.zeros(8, dtype=torch.long, device=backend)
    pi_k = torch.full([R, 2], -7, dtype=torch.long, device=backend)
    tot_k = torch.zeros(1, dtype=torch.long, device=backend)
    _lib.call("nsim_live_rank", _lib.ptr(counts.to(backend)), R, C, *nfs, _lib.ptr(lr), _lib.ptr(lidx), _lib.ptr(cnts), None, 0,
              _lib.ptr(pi_k), _lib.ptr(tot_k), -1, None, 0)
    q = torch.cumsum(live.long(), 0) - live.long()
    assert torch.equal(lr.cpu(), torch.where(live, q, ~q))
    assert torch.equal(lidx.cpu()[:Rl], live.nonzero()[:, 0]) and bool((lidx.cpu()[Rl:] == 0).all())
    M = int(counts.sum())
    assert cnts.cpu().tolist() == [Rl, M + Rl * C, Rl * 8, Rl * 8, Rl * 16, 0, M, 0] and int(tot_k) == M
    off = torch.cumsum(counts, 0) - counts
    assert torch.equal(pi_k.cpu(), torch.stack([off, counts], -1))


def test_live_rank_and_compact_sampling(backend):
    """``upsample_on_marched_only``: ``nsim_live_rank`` (ranks, live list, device-side point counts) and the live-rank form of
    the sampling kernels -- only rays with marched samples get list-b / new samples, every per-live-ray array is indexed by the
    rank -- against the dense kernels run on the live rays alone."""
    g = torch.Generator().manual_seed(5)
    counts = torch.tensor([0, 5, 0, 0, 70, 1, 0, 600, 3, 0, 0, 64], dtype=torch.long)
    R = counts.shape[0]
    live = counts > 0
    Rl = int(live.sum())
    C, nfs = 6, [4, 8, 0, 0]
    dv = lambda a: a.to(backend).contiguous()        # noqa: E731
    lr = torch.full([R], 123, dtype=torch.long, device=backend)
    lidx = torch.full([R], 123, dtype=torch.long, device=backend)
    cnts = torch.zeros(8, dtype=torch.long, device=backend)
    from neuralsim_amd.graphics import pack_ops as ppo
    for cap in (-1, 100, 700):          # the same launch also emits the pack infos of the counts (nsim_pack_infos_from_n's rule for a capacity)
        pi_k = torch.full([R, 2], -7, dtype=torch.long, device=backend)
        tot_k = torch.zeros(1, dtype=torch.long, device=backend)
        _lib.call("nsim_live_rank", _lib.ptr(dv(counts)), R, C, *nfs, _lib.ptr(lr), _lib.ptr(lidx), _lib.ptr(cnts), None, 0,
                  _lib.ptr(pi_k), _lib.ptr(tot_k), cap, None, 0)
        pi_w, tot_w = ppo.get_pack_infos_from_n(dv(counts), return_total=True, cap=cap)
        assert torch.equal(pi_k, pi_w) and torch.equal(tot_k, tot_w), cap
    _lib.call("nsim_live_rank", _lib.ptr(dv(counts)), R, C, *nfs, _lib.ptr(lr), _lib.ptr(lidx), _lib.ptr(cnts), None, 0,
              None, None, -1, None, 0)
    q = torch.cumsum(live.long(), 0) - live.long()
    assert torch.equal(lr.cpu(), torch.where(live, q, ~q))
    assert torch.equal(lidx.cpu()[:Rl], live.nonzero()[:, 0]) and bool((lidx.cpu()[Rl:] == 0).all())
    M = int(counts.sum())
    assert cnts.cpu().tolist() == [Rl, M + Rl * C, Rl * 4, Rl * 8, 0, 0, M, 0]
    # marched samples of the live rays, coarse depths, merge, draws: live-rank kernels on all R rays == dense kernels on the live ones
    n, pi_l, S, ridx, t, sdf = _rand_packs(g, counts[live].tolist())
    from oracle import pack_ops as opo
    pi = opo.get_pack_infos_from_n(counts)
    near, far = torch.rand(R, generator=g) + 1.0, torch.rand(R, generator=g) + 3.0
    jc = torch.rand(R, C, generator=g)
    ro = torch.randn(R, 3, generator=g)
    rd = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)

    def run(rays, pi_, rank):
        Rr = rays.shape[0] if rank is None else R
        nl = rays.shape[0]
        sel = rays if rank is None else torch.arange(R)
        t_c = torch.full([nl, C], -1.0, device=backend)
        _lib.call("nsim_coarse_depths", _lib.ptr(dv(near[sel])), _lib.ptr(dv(far[sel])), _lib.ptr(dv(jc[sel])), Rr, C, _lib.ptr(t_c), _lib.ptr(rank))
        T = S + nl * C
        t_o, v_o = torch.zeros(T, device=backend), torch.zeros(T, device=backend)
        pi_o = torch.zeros(Rr, 2, dtype=torch.long, device=backend)
        r_o = torch.zeros(T, dtype=torch.long, device=backend)
        x_o = torch.zeros(T, 3, device=backend)
        _lib.call("nsim_merge_sorted", _lib.ptr(dv(t)), _lib.ptr(dv(sdf)), _lib.ptr(dv(pi_)), _lib.ptr(t_c), None, Rr, C, _lib.ptr(t_o),
                  _lib.ptr(v_o), _lib.ptr(pi_o), _lib.ptr(r_o), _lib.ptr(dv(ro[sel])), _lib.ptr(dv(rd[sel])), _lib.ptr(x_o), _lib.ptr(rank))
        t_n = torch.zeros(nl, 4, device=backend)
        x_n = torch.zeros(nl * 4, 3, device=backend)
        scratch = torch.zeros(T, device=backend)
        _lib.call("nsim_upsample_stage", _lib.ptr(t_o), _lib.ptr(v_o), _lib.ptr(pi_o), Rr, 64.0, 4, 1, _lib.ptr(scratch), _lib.ptr(t_n),
                  _lib.ptr(dv(ro[sel])), _lib.ptr(dv(rd[sel])), _lib.ptr(x_n), _lib.ptr(rank))
        v_n = (t_n * 0.5).contiguous()
        T2 = T + nl * 4
        t2, v2 = torch.zeros(T2, device=backend), torch.zeros(T2, device=backend)
        pi2 = torch.zeros(Rr, 2, dtype=torch.long, device=backend)
        t_n2 = torch.zeros(nl, 8, device=backend)
        scratch2 = torch.zeros(T2, device=backend)
        _lib.call("nsim_merge_upsample", _lib.ptr(t_o), _lib.ptr(v_o), _lib.ptr(pi_o), _lib.ptr(t_n), _lib.ptr(v_n), Rr, 4, _lib.ptr(t2),
                  _lib.ptr(v2), _lib.ptr(pi2), None, 256.0, 8, 1, _lib.ptr(scratch2), _lib.ptr(t_n2), _lib.ptr(dv(ro[sel])),
                  _lib.ptr(dv(rd[sel])), None, _lib.ptr(rank))
        return [a.cpu() for a in (t_c, t_o, v_o, pi_o, r_o, x_o, t_n, x_n, t2, v2, pi2, t_n2)]
    rays_live = live.nonzero()[:, 0]
    dense = run(rays_live, pi_l, None)
    comp = run(rays_live, pi, lr)
    names = ("t_c", "t_o", "v_o", "pi_o", "r_o", "x_o", "t_n", "x_n", "t2", "v2", "pi2", "t_n2")
    for a, b, nm in zip(comp, dense, names):
        if nm in ("pi_o", "pi2"):          # [R, 2] over all rays: the live rows are the dense ones, the others are empty packs
            assert torch.equal(a[live], b), nm
            assert bool((a[~live][:, 1] == 0).all()), nm
            assert torch.equal(a[1:, 0], (a[:, 0] + a[:, 1])[:-1]), nm          # packs tile the buffer in ray order
        elif nm == "r_o":                  # ray indices are those of the full ray list
            assert torch.equal(a, rays_live[b]), nm
        else:
            assert torch.equal(a, b), nm
