"""Matched PSNR (BASELINE.json north_star: "... at matched PSNR"; VERDICT r5 missing 4).

The HIP training step and the pure-PyTorch oracle are trained from ONE initialisation on IDENTICAL batches -- the same pixels,
targets, marching / coarse-depth jitter and uniform eikonal points, recorded from the product's run and replayed to the oracle --
each with its own Adam (the product's fused kernel; ``torch.optim.Adam`` with the reference's eps / betas for the oracle).  Then
both render the same HELD-OUT views (cameras of another seed, validation renderer settings) and the PSNR against the analytic
target image is compared, the way the reference's eval tool scores a run (code_single/tools/eval.py:241-316: PSNR of the
rendered rgb against the ground-truth image, per frame, then averaged).

The trajectories are NOT expected to stay bit-close over hundreds of steps (fp16 MFMA operands and float atomics on one side, f32
on the other: rounding differences of ~1e-3 grow along an Adam trajectory); what is claimed -- and asserted -- is that the two
land at the same image quality: |PSNR_hip - PSNR_oracle| <= 0.5 dB, both well above the starting point.
"""
import math

import pytest
import torch

from neuralsim_amd.eval import psnr, render_image
from neuralsim_amd.fields.neus import LoTDNeuSModel, marched_only
from neuralsim_amd.graphics.cameras import look_at_cameras
from neuralsim_amd.trainer import RenderTrainer
from oracle import field as ofield, render as orr
from util import SMALL_RES, oracle_of_neus

W_EIK = 0.1
RADIUS = 0.5
QP = dict(nablas_has_grad=True, num_coarse=16, num_fine=[4, 4, 8], upsample_inv_s=64.0, upsample_inv_s_factors=[1, 4, 16],
          upsample_use_estimate_alpha=True, march_cfg=dict(step_size=0.02, max_steps=512))


def _build(backend, log2_T, N, n_uni, lr, V):
    torch.manual_seed(0)
    m = LoTDNeuSModel(lod_res=SMALL_RES, log2_hashmap_size=log2_T, sdf_D=2, precision="fp16", ln_inv_s_init=0.45,
                      accel_cfg=dict(resolution=(32, 32, 32), update_from_net_cfg=dict(num_steps=1, num_pts=2 ** 14),
                                     update_from_samples_cfg={}, n_steps_between_update=16, n_steps_warmup=10 ** 9),
                      ray_query_cfg=dict(query_mode="march_occ_multi_upsample_compressed", query_param=dict(QP)), seed=7).to(backend)
    m.geometric_init_sphere(RADIUS)
    m.accel.init(m.query_sdf, num_steps=2, num_pts=2 ** 14)
    intr, c2w, WH = look_at_cameras(V=V, seed=1, device=backend)
    # the occupancy grid is held fixed (warm-up beyond the run: the target IS the initial sphere, the grid stays valid), so both
    # sides march the same bitfield for the whole run
    tr = RenderTrainer(m, intr, c2w, WH, num_rays=N, lr=lr, w_eikonal=W_EIK, num_uniform=n_uni, perturb=True, learn_inv_s=False,
                       target_sphere_radius=RADIUS)
    return m, tr


def _oracle_run(p, appear, occ, aabb, res, batches, lr, N, mo):
    params = [t for t in p.tensors() if t is not p.ln_inv_s] + [appear]
    for t in params:
        t.requires_grad_(True)
    opt = torch.optim.Adam(params, lr=lr, betas=(0.9, 0.99), eps=1e-15)         # optim.py: the reference's training_cfg
    losses = []
    for b in batches:
        ri, R = b["rays_inds"], int(b["rays_inds"].shape[0])
        jit, jit_c = torch.zeros(N), torch.zeros(N, QP["num_coarse"])
        jit[ri], jit_c[ri] = b["jitter"][:R], b["jitter_c"][:R]                    # rows 0..R-1 serve the R hit rays
        ha = appear[b["fidx"]]
        ret = orr.ray_query(p, b["rays_o"], b["rays_d"], ha, occ, aabb[0], aabb[1], res, near=0.01, num_coarse=QP["num_coarse"],
                            num_fine=tuple(QP["num_fine"]), upsample_inv_s=QP["upsample_inv_s"],
                            upsample_inv_s_factors=tuple(QP["upsample_inv_s_factors"]), step_size=QP["march_cfg"]["step_size"],
                            max_steps=QP["march_cfg"]["max_steps"], jitter=jit, jitter_c=jit_c, depth_use_normalized_vw=False,
                            compress=True, upsample_on_marched_only=mo)
        assert torch.equal(ret["rays_inds"], ri)
        loss, _ = orr.render_loss(ret, b["gt"], N, w_eikonal=W_EIK)
        _, nab_u = ofield.forward_sdf_nablas(b["x_uni"], p)
        loss = loss + W_EIK * ((nab_u.norm(dim=-1) - 1.0) ** 2).mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    return losses


def _oracle_views(p, occ, aabb, res, intr, c2w, WH, mo):
    """the held-out views through the oracle with the validation settings (no jitter, normalised depth weights)"""
    imgs = []
    W, H = int(WH[0, 0]), int(WH[0, 1])
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    xy = torch.stack([(xs.reshape(-1) + 0.5) / W, (ys.reshape(-1) + 0.5) / H], -1)
    with torch.no_grad():
        for f in range(intr.shape[0]):
            fidx = torch.full([xy.shape[0]], f, dtype=torch.long)
            o, d = orr.pinhole_rays(xy, fidx, intr, c2w, WH)
            ret = orr.ray_query(p, o, d, torch.zeros(xy.shape[0], 4), occ, aabb[0], aabb[1], res, near=0.01,
                                num_coarse=QP["num_coarse"], num_fine=tuple(QP["num_fine"]), upsample_inv_s=QP["upsample_inv_s"],
                                upsample_inv_s_factors=tuple(QP["upsample_inv_s_factors"]), step_size=QP["march_cfg"]["step_size"],
                                max_steps=QP["march_cfg"]["max_steps"], depth_use_normalized_vw=True, compress=True,
                                upsample_on_marched_only=mo)
            img = torch.zeros(xy.shape[0], 3)
            if ret["num_rays"] > 0 and "rendered" in ret:
                img[ret["rays_inds"]] = ret["rendered"]["rgb_volume"]
            imgs.append((img.view(H, W, 3), o, d))
    return imgs


def test_hip_and_oracle_converge_to_matched_psnr(backend):
    on_gpu = backend.type == "cuda"
    # emulator: the machinery only (a handful of steps of a toy size); MI355X: the claim
    N, K, log2_T, n_uni, HW = (384, 72, 15, 128, 48) if on_gpu else (48, 3, 10, 16, 12)
    lr = 5e-3
    m, tr = _build(backend, log2_T, N, n_uni, lr, V=12)
    p = oracle_of_neus(m)                                    # the SAME initial weights (table as its fp16 shadow holds it)
    p.grid = m.encoding.flattened_params.detach().cpu().float().clone()       # ... trained as an f32 master, like the product's
    appear_o = tr.appear.detach().cpu().clone()
    occ = m.accel.occ_val.detach().cpu() > m.accel.occ_thre
    aabb, res = m.accel.aabb.detach().cpu(), [32, 32, 32]
    mo = marched_only(QP)
    # held-out views: another seed of the rig, a small image
    intr_e, c2w_e, WH_e = look_at_cameras(V=3 if on_gpu else 1, seed=99, H=HW, W=HW, f=1111.1 * HW / 800.0, device=backend)
    ha0 = torch.zeros(1, 4, device=backend)

    def hip_views():
        return [render_image(tr.renderer, m, intr_e, c2w_e, WH_e, frame=f, rays_h_appear=ha0)["rgb_volume"].detach().cpu()
                for f in range(intr_e.shape[0])]

    imgs_o0 = _oracle_views(p, occ, aabb, res, intr_e.cpu(), c2w_e.cpu(), WH_e.cpu(), mo)
    targets = [RenderTrainer.sphere_image(o.to(backend), d.to(backend), RADIUS).cpu().view(HW, HW, 3) for _, o, d in imgs_o0]
    psnr_h0 = sum(psnr(a, t) for a, t in zip(hip_views(), targets)) / len(targets)
    psnr_o0 = sum(psnr(a, t) for (a, _, _), t in zip(imgs_o0, targets)) / len(targets)
    assert abs(psnr_h0 - psnr_o0) < 0.2, (psnr_h0, psnr_o0)               # same weights: same image

    # ---- the product's run; every batch it consumes is recorded for the oracle
    batches = []
    real_make = tr._make_batch

    def recording():
        b = real_make()
        R = int(b["tested"]["num_rays"])
        batches.append(dict(rays_o=b["rays_o"].detach().cpu(), rays_d=b["rays_d"].detach().cpu(), fidx=b["fidx"].cpu(),
                            gt=b["gt"].cpu(), rays_inds=b["tested"]["rays_inds"].cpu(), jitter=b["jitter"][:R].cpu(),
                            jitter_c=b["jitter_c"][:R].cpu(), x_uni=b["x_uni"].cpu()))
        return b
    tr._make_batch = recording
    tr._prefetched = None
    losses_h = [float(tr.train_step(it)) for it in range(K)]
    tr._make_batch = real_make
    batches = batches[:K]                                     # (the last step prefetched one more)
    assert all(l == l for l in losses_h)
    psnr_h = sum(psnr(a, t) for a, t in zip(hip_views(), targets)) / len(targets)

    # ---- the oracle's run on the recorded batches
    losses_o = _oracle_run(p, appear_o, occ, aabb, res, batches, lr, N, mo)
    imgs_o = _oracle_views(p, occ, aabb, res, intr_e.cpu(), c2w_e.cpu(), WH_e.cpu(), mo)
    psnr_o = sum(psnr(a, t) for (a, _, _), t in zip(imgs_o, targets)) / len(targets)
    rec = dict(steps=K, rays=N, psnr_start=round(psnr_o0, 2), psnr_hip=round(psnr_h, 2), psnr_oracle=round(psnr_o, 2),
               loss_first=(round(losses_h[0], 5), round(losses_o[0], 5)), loss_last=(round(losses_h[-1], 5), round(losses_o[-1], 5)))
    print("[matched psnr]", rec)
    try:
        import json
        from pathlib import Path
        out = Path(__file__).resolve().parent.parent / "gpurun_out"
        out.mkdir(exist_ok=True)
        (out / f"matched_psnr_{backend.type}.json").write_text(json.dumps(rec))
    except OSError:
        pass
    # the first step sees identical weights and identical randoms: the same loss to fp16 accuracy
    assert abs(losses_h[0] - losses_o[0]) < 2e-3 * (1 + abs(losses_o[0])), rec
    if not on_gpu:
        assert abs(psnr_h - psnr_o) < 0.5, rec
        return
    # both trained (measured on MI355X: 240 steps of 512 rays 15.89 -> 18.30 / 18.30 dB; 120 steps of 384 rays 15.94 -> 18.12 / 18.13 dB)
    assert psnr_o > psnr_o0 + 0.8 and psnr_h > psnr_h0 + 0.8, rec
    assert abs(psnr_h - psnr_o) <= 0.5, rec
