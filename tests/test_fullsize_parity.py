"""Oracle parity AT the BASELINE configuration (configs[1]: L = 16 / T = 2^19 table with resolutions 16..2048, 2x64 SDF
decoder, 2x64 radiance net, occupancy 64^3, num_coarse 64 + num_fine [8, 8, 32], step .005, the 100-camera 800x800
rig; model / hyper-parameters of code_single/configs/object_centric/lotd_neus.dtu.230814.yaml:94-190).

The model is ``bench.build_trainer``'s, its weights are copied verbatim into ``oracle.field.FieldParams`` and both sides
get identical rays, appearance codes and perturbation randoms:

* ``test_api_path_*``  -- ``ray_test`` + ``ray_query`` + autograd (the drop-in path a reference renderer drives) on 2048
  rays, both query modes.  precision f32 (exact-f32 MFMA): discrete decisions bit-exact (hit rays, march counts, merged /
  compressed sample counts), depths / sdf / colours / images and every gradient tight.  precision fp16 (the product
  default): images within the stated fp16 tolerance; gradients are compared on the ORACLE's sample set AND end to end --
  since round 3 the SDF queries of the sampling pass run in f32-equivalent arithmetic (hi + lo f16 operands on the matrix
  cores, ``sampling_precision``), so the fp16 step places and keeps the samples the f32 oracle does (in round 2 the
  up-sampler multiplied the ~1e-4 fp16 error of an SDF by inv_s = 1024 and individual fine samples moved).
* ``test_fused_step_*`` -- the bench's own launch chain (``RenderTrainer._train_render_fused``: 8192 rays + 4096 uniform
  eikonal points, compressed mode) against the oracle's loss and gradients of the same batch.

GPU only (``-m gpu``): the host emulator needs hours at this size; the same comparisons at emulator size are
tests/test_ray_query.py and tests/test_trainer.py.
"""
import json
import os
from pathlib import Path

import pytest
import torch

from oracle import field as ofield, render as orr
from util import leaf, oracle_flat_grads, rel_l2

pytestmark = pytest.mark.gpu

N_API = 2048
W_EIK = 0.1
REPORT_DIR = Path(__file__).resolve().parent.parent / "gpurun_out"

# ---- tolerances.  f32 = exact-f32 MFMA validation mode; fp16 = product default.
TOL = dict(
    # sample-level values (t, and sdf / nablas / rgb AT those t) carry the up-sampler's sensitivity: the inverse-CDF step
    # divides by CDF increments down to 1e-5, so 1e-6 differences of the no-grad SDFs move a fine sample by up to ~1e-3
    # -- the images and the fixed-sample-set values are the tight checks
    # (nablas: a sample that moves by 2e-4 crosses cells of the res-2048 level, whose features are hash noise)
    # measured (MI355X): t 2.3e-4, images <= 1e-5, sdf / rgb on the oracle's samples 4e-7 / 1e-7, gradients <= 5e-5
    f32=dict(t=2e-3, sdf=1e-3, rgb=2e-4, nablas=3e-2, fix=dict(sdf=2e-5, rgb=2e-5, nablas=1e-4),
             img=dict(mask_volume=5e-5, rgb_volume=5e-5, depth_volume=1e-4, normals_volume=5e-5),
             grad=2e-4, loss=1e-5, flips=2),
    # fp16 MFMA operands (measured at this config: images <= 2.3e-3, sdf 2.4e-4 / 5e-5 near the surface, nablas 5.5e-4).
    # Gradients on the oracle's sample set: 3e-2 on the compressed set (samples near the surface; measured 5e-3); the
    # un-compressed set is dominated by samples far from the surface where the eikonal residual |n| - 1 ~ 1e-3 is a
    # difference of cancelling terms -- the 5e-4 fp16 error of n is half of it (measured 6e-2 on the table gradient; the
    # f32 kernels give 1.5e-6 on the same set, i.e. the same conditioning x the f32 / fp16 epsilon ratio)
    fp16=dict(img=dict(mask_volume=5e-3, rgb_volume=5e-3, depth_volume=1e-2, normals_volume=1e-2),
              fix=dict(sdf=1e-3, rgb=1e-3, nablas=5e-3), grad=3e-2, grad_full=1e-1, loss=2e-3),
)


# fused chain, permutohedral model: end-to-end gradient bound (rel. L2 per parameter group); set from the measured record
PERMUTO_FUSED_GRAD = dict(f32=3e-2, fp16=0.2)      # measured: 5.2e-3 (table, 4 samples apart) / 6.5e-2 (table, fp16)


def _report(name, rec):
    try:
        REPORT_DIR.mkdir(exist_ok=True)
        (REPORT_DIR / f"parity_fullsize_{name}.json").write_text(json.dumps(rec, indent=1, sort_keys=True))
    except OSError:
        pass
    print(f"[parity {name}] " + json.dumps(rec, sort_keys=True))


_RIGS = {}


def _rig(precision, encoding="lotd"):
    """bench.build_trainer (BASELINE configs[1]) + the oracle carrying the same weights and occupancy grid.
    ``encoding="permuto"``: the same rig with the permutohedral-lattice model (bench ``variants.permuto_ms``; row f4)."""
    key = (precision, encoding)
    if key in _RIGS:
        return _RIGS[key]
    import bench
    assert torch.cuda.is_available()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    tr = bench.build_trainer(dev, 0, 1, precision=precision, encoding=encoding)
    m = tr.model
    cfg = m.encoding.cfg
    if encoding == "permuto":
        from oracle import permuto as operm
        pc = cfg.permuto
        assert tr.num_rays == 8192 and pc.num_levels == 16 and pc.hashmap_size == 2 ** 19 and m.sdf_D == 2
        p = ofield.params_from_flat([2] * 16, 4, torch.zeros(16 * 16), m.sdf_w, m.sdf_b, m.rad_w, m.rad_b, m.ln_inv_s, sdf_D=2,
                                    ln_inv_s_factor=m.ln_inv_s_factor)
        p.spec = operm.PermutoSpec(3, list(pc.res), 2, pc.hashmap_size, pc.shifts.clone())
        p.grid = m.encoding.flattened_params.detach().cpu().half().float().requires_grad_(True)      # the stored (fp16) values
        p.aabb = m.accel.aabb.detach().cpu().clone()
        occ = (m.accel.occ_val.detach().cpu() > m.accel.occ_thre)
        _RIGS[key] = (tr, p, occ, dev)
        return _RIGS[key]
    assert tr.num_rays == 8192 and cfg.n_params == 12196216 and list(cfg.lod_res)[-1] == 2048 and m.sdf_D == 2
    p = ofield.params_from_flat(cfg.lod_res, 19, m.encoding.flattened_params, m.sdf_w, m.sdf_b, m.rad_w, m.rad_b,
                                m.ln_inv_s, sdf_D=2, ln_inv_s_factor=m.ln_inv_s_factor)
    occ = (m.accel.occ_val.detach().cpu() > m.accel.occ_thre)
    _RIGS[key] = (tr, p, occ, dev)
    return _RIGS[key]


def _fresh(p):
    for t in p.tensors():
        t.grad = None
        t.requires_grad_(True)
    return p


def _zero_grads(tr):
    m = tr.model
    for q in (m.encoding.flattened_params, m.sdf_w, m.sdf_b, m.rad_w, m.rad_b, m.ln_inv_s, tr.appear):
        q.grad = None


def _product_grads(tr):
    m = tr.model
    return dict(grid=m.encoding.flattened_params.grad, sdf_w=m.sdf_w.grad, sdf_b=m.sdf_b.grad, rad_w=m.rad_w.grad,
                rad_b=m.rad_b.grad, ln_inv_s=m.ln_inv_s.grad)


def _batch(tr, N, seed):
    """N rays of the rig (+ targets = the analytic sphere image, per-ray appearance codes, perturbation randoms)."""
    g = torch.Generator().manual_seed(seed)
    V = tr.intr.shape[0]
    xy = torch.rand(N, 2, generator=g).clamp(1e-6, 1 - 1e-6)
    fidx = torch.randint(0, V, (N,), generator=g)
    o, d = orr.pinhole_rays(xy, fidx, tr.intr.cpu(), tr.c2w.cpu(), tr.WH.cpu())
    jit, jit_c = torch.rand(N, generator=g), torch.rand(N, 64, generator=g)
    ha = tr.appear.detach().cpu()[fidx].clone()
    dev = tr.model.device
    gt = tr.sphere_image(o.to(dev), d.to(dev), 0.75).cpu()
    return o, d, fidx, ha, jit, jit_c, gt


def _oracle_query(p, occ, tr, o, d, ha, jit, jit_c, compressed, **kw):
    a = tr.model.accel.aabb.detach().cpu()
    return orr.ray_query(p, o, d, ha, occ, a[0], a[1], tr.model.accel.resolution, near=0.01, far=None, jitter=jit,
                         jitter_c=jit_c, depth_use_normalized_vw=False, compress=compressed, compress_thre=1e-4, **kw)


@pytest.mark.parametrize("precision", ["f32", "fp16"])
def test_permuto_model_matches_oracle_at_baseline_size(precision):
    """The same comparison for the permutohedral-lattice model (PermutoNeuSObj, 16 levels x 2^19 entries, pre-trained to
    the sphere): ray_test + ray_query + autograd against the oracle's restatement of the lattice (oracle/permuto.py)."""
    _api_path_body(precision, True, "permuto")


@pytest.mark.parametrize("compressed", [True, False])
@pytest.mark.parametrize("precision", ["f32", "fp16"])
def test_api_path_matches_oracle_at_baseline_config(precision, compressed):
    _api_path_body(precision, compressed, "lotd")


def _api_path_body(precision, compressed, encoding):
    tr, p, occ, dev = _rig(precision, encoding)
    m = tr.model
    m._march_stat = None             # exact buffer sizes for the sampling pass (no speculative capacity)
    tol = TOL[precision]
    N = N_API
    # The permutohedral model is PRE-TRAINED on the device (300 Adam steps, float atomics: a different field every run).
    # Fitted by values alone it was rough at the scale of its finest levels (|d sdf / d x| ~ 20 over 5e-4: the up-sampler's
    # ~1e-3 sample sensitivity became ~1e-2 of SDF, 9-20 rays with another kept count); with the eikonal term of the
    # pre-training (w 0.1) it measures 2-3 rays, PSNR 101 dB in f32 mode.  The end-to-end leg keeps loose, run-independent
    # bounds; the kernel parity is what the fixed-sample-set leg (both sides on the ORACLE's samples) asserts tightly.
    e2e_tight = encoding == "lotd"
    flips_max = tol.get("flips", TOL["f32"]["flips"]) if e2e_tight else max(2, N // 50)
    o, d, fidx, ha, jit, jit_c, gt = _batch(tr, N, seed=11)
    _fresh(p)
    ha_o = leaf(ha)
    ret_o = _oracle_query(p, occ, tr, o, d, ha_o, jit, jit_c, compressed)
    loss_o, _ = orr.render_loss(ret_o, gt, N, w_eikonal=W_EIK)
    loss_o.backward()
    ref = oracle_flat_grads(p)
    vbo = ret_o["volume_buffer"]
    ri = ret_o["rays_inds"]
    dv = lambda a: a.to(dev).contiguous()        # noqa: E731
    mode = "march_occ_multi_upsample" + ("_compressed" if compressed else "")
    rec = dict(precision=precision, mode=mode, rays=N, hit=int(ri.shape[0]), samples_oracle=int(vbo["t"].shape[0]))

    # ---------------------------------------------------------------- end to end through the model's API
    _zero_grads(tr)
    ha_p = leaf(ha, dev)
    tested = m.ray_test(dv(o), dv(d), near=0.01, far=None, rays_h_appear=ha_p)
    assert tested["num_rays"] == ret_o["num_rays"] and torch.equal(tested["rays_inds"].cpu(), ri)
    cfg = dict(m.ray_query_cfg)
    cfg.update(query_mode=mode, with_rgb=True, with_normal=True, depth_use_normalized_vw=False, _render=True,
               _jitter=dv(jit[ri]), _jitter_c=dv(jit_c[ri]))
    ret = m.ray_query(ray_tested=tested, config=cfg, return_details=True)
    vb = ret["volume_buffer"]
    assert torch.equal(ret["details"]["march_counts"].cpu(), ret_o["debug"]["march_counts"])     # bit-exact, any precision
    rec["samples"] = int(vb["t"].shape[0])
    n_p, n_o = vb["pack_infos_hit"][:, 1].cpu(), vbo["pack_infos_hit"][:, 1]
    rec["rays_with_other_count"] = int((n_p != n_o).sum())
    # the no-grad SDFs of the sampling pass (un-compressed set): same points in f32 mode
    sdf_ng, sdf_ng_o = ret["details"]["sdf_nograd"].cpu(), ret_o["debug"]["sdf_nograd"]
    assert sdf_ng.shape == sdf_ng_o.shape            # march + 64 coarse + 48 fine per ray on both sides
    rec["sdf_nograd_max"] = float((sdf_ng - sdf_ng_o).abs().max())
    if not compressed:       # the un-compressed buffer IS the sampling set: samples at bit-identical depths (marched +
        tt_p, tt_o = vb["t"].cpu(), vbo["t"]             # coarse ones) are the same points -> the no-grad kernels' own error
        eq = tt_p == tt_o
        rec["bit_identical_depths"] = float(eq.float().mean())
        rec["march_coarse_sdf_max"] = float((sdf_ng - sdf_ng_o)[eq].abs().max())
    else:
        rec["march_coarse_sdf_max"] = 0.0
    for k in ("mask_volume", "depth_volume", "rgb_volume", "normals_volume"):
        rec["img_" + k] = float((ret["rendered"][k].cpu() - ret_o["rendered"][k]).abs().max())
    mse = float(((ret["rendered"]["rgb_volume"].cpu() - ret_o["rendered"]["rgb_volume"]) ** 2).mean())
    rec["psnr_rgb_db"] = round(-10.0 * torch.log10(torch.tensor(max(mse, 1e-20))).item(), 2)
    same = None
    if precision == "f32":
        # compress keep decisions (vw > 1e-4) sit on a threshold: a 1e-6 SDF difference may flip one of ~4e5
        assert rec["rays_with_other_count"] <= flips_max, rec
        same_ray = (n_p == n_o)
        same = torch.repeat_interleave(same_ray, n_o)               # oracle samples of rays with identical counts
        same_p = torch.repeat_interleave(same_ray, n_p)
        for k in ("t", "sdf", "rgb", "nablas"):
            a, b = vb[k].detach().cpu()[same_p], vbo[k].detach()[same]
            rec["smp_" + k] = float((a - b).abs().max())
    rgb_full = torch.zeros(N, 3, device=dev).index_put((tested["rays_inds"],), ret["rendered"]["rgb_volume"])
    loss = ((rgb_full - dv(gt)) ** 2).mean() + W_EIK * ((vb["nablas"].norm(dim=-1) - 1.0) ** 2).mean()
    loss.backward()
    rec["loss"], rec["loss_oracle"] = float(loss), float(loss_o)
    got = _product_grads(tr)
    got["h_appear"] = ha_p.grad
    ref["h_appear"] = ha_o.grad
    for k, v in got.items():
        rec["e2e_grad_" + k] = rel_l2(v.cpu(), ref[k])

    # ---------------------------------------------------------------- the differentiable part on the ORACLE's sample set
    from neuralsim_amd.fields.neus import _FieldFn, _NeusAlphaFn, volume_integration
    _zero_grads(tr)
    ha_p2 = leaf(ha[ri], dev)
    o_h, d_h = dv(o[ri]), dv(d[ri])
    t_o, pi_o = dv(vbo["t"]), dv(ret_o["pack_infos_tested"])          # packed per TESTED ray (the buffer lists the marched rays)
    ridx_o = torch.repeat_interleave(torch.arange(ri.shape[0]), ret_o["pack_infos_tested"][:, 1]).to(dev)
    sdf, nab, rgb = _FieldFn.apply(m, m.encoding.flattened_params, m.sdf_w, m.sdf_b, m.rad_w, m.rad_b, ha_p2, None, o_h,
                                   d_h, t_o, ridx_o, True)
    alpha = _NeusAlphaFn.apply(sdf, m.ln_inv_s, pi_o, m.ln_inv_s_factor, 0.0)
    rend = volume_integration(alpha, t_o, rgb, nab, pi_o, False)
    for k, a, b in (("sdf", sdf, vbo["sdf"]), ("rgb", rgb, vbo["rgb"]), ("nablas", nab, vbo["nablas"]),
                    ("alpha", alpha, vbo["opacity_alpha"])):
        rec["fix_" + k] = float((a.detach().cpu() - b.detach()).abs().max())
    for k in ("mask_volume", "depth_volume", "rgb_volume", "normals_volume"):
        rec["fix_img_" + k] = float((rend[k].detach().cpu() - ret_o["rendered"][k].detach()).abs().max())
    rgb_full = torch.zeros(N, 3, device=dev).index_put((dv(ri),), rend["rgb_volume"])
    loss2 = ((rgb_full - dv(gt)) ** 2).mean() + W_EIK * ((nab.norm(dim=-1) - 1.0) ** 2).mean()
    loss2.backward()
    rec["fix_loss"] = float(loss2)
    got = _product_grads(tr)
    got["h_appear"] = torch.zeros(N, 4, device=dev).index_put((dv(ri),), ha_p2.grad)
    for k, v in got.items():
        rec["fix_grad_" + k] = rel_l2(v.cpu(), ref[k])
    # the SDF decoder's gradient as ONE vector (weights + biases), and the norms the relative errors are quoted against
    rec["fix_grad_sdf_dec"] = rel_l2(torch.cat([got["sdf_w"].cpu().flatten(), got["sdf_b"].cpu().flatten()]),
                                     torch.cat([ref["sdf_w"].flatten(), ref["sdf_b"].flatten()]))
    rec["ref_norm_sdf_w"], rec["ref_norm_sdf_b"] = float(ref["sdf_w"].norm()), float(ref["sdf_b"].norm())
    if not e2e_tight:
        # ---- a WELL-CONDITIONED gradient check on the same sample set: random cotangents on the field's three outputs.
        # The render loss above is close to stationary for the pre-trained decoder -- its weight gradient is a sum of signed
        # terms that cancel to a norm which differs from one device pre-training to the next (float atomics), so a relative
        # error against that norm measures the conditioning of the run, not the kernels (flake hunt, round 6: 0.017 in two
        # runs, 0.104 in the third; round 4: 0.02 .. 0.48 on the bias part).  With random signs nothing cancels: the relative
        # error is the kernels' own fp16 / f32 error whatever field the pre-training ended at.
        S_o = int(vbo["t"].shape[0])
        gr = torch.Generator().manual_seed(123)
        c_sdf, c_nab, c_rgb = torch.randn(S_o, generator=gr), torch.randn(S_o, 3, generator=gr), torch.randn(S_o, 3, generator=gr)
        _fresh(p)
        ridx_c = ridx_o.cpu()
        x_c = (o[ri][ridx_c] + vbo["t"].detach()[:, None] * d[ri][ridx_c])
        sdf_r, nab_r, rgb_r = ofield.forward_field(x_c, d[ri][ridx_c], ha[ri][ridx_c], p)
        ((sdf_r * c_sdf).sum() + (nab_r * c_nab).sum() + (rgb_r * c_rgb).sum()).div(S_o).backward()
        ref_r = oracle_flat_grads(p)
        _zero_grads(tr)
        ha_p3 = leaf(ha[ri], dev)
        sdf3, nab3, rgb3 = _FieldFn.apply(m, m.encoding.flattened_params, m.sdf_w, m.sdf_b, m.rad_w, m.rad_b, ha_p3, None, o_h,
                                          d_h, t_o, ridx_o, True)
        ((sdf3 * dv(c_sdf)).sum() + (nab3 * dv(c_nab)).sum() + (rgb3 * dv(c_rgb)).sum()).div(S_o).backward()
        got_r = _product_grads(tr)
        for k in ("grid", "sdf_w", "sdf_b", "rad_w", "rad_b"):
            rec["rnd_grad_" + k] = rel_l2(got_r[k].cpu(), ref_r[k])
    _report(("permuto_" if encoding == "permuto" else "") + f"api_{precision}_{'compressed' if compressed else 'full'}", rec)

    # ---------------------------------------------------------------- assertions
    for k, lim in tol["img"].items():
        if e2e_tight:       # (permutohedral model: the end-to-end images are reported, PSNR and loss asserted below)
            assert rec["img_" + k] < lim, (k, rec["img_" + k])
        assert rec["fix_img_" + k] < lim * (1 if (e2e_tight or precision == "f32") else 4), ("fix", k, rec["fix_img_" + k])
    for k in ("sdf", "rgb", "nablas"):      # (the rough permutohedral field has normals of magnitude ~10: absolute fp16 bound x 4)
        assert rec["fix_" + k] < tol["fix"][k] * (1 if (e2e_tight or precision == "f32") else 4), (k, rec["fix_" + k])
    assert abs(rec["fix_loss"] - rec["loss_oracle"]) < tol["loss"] * (1 + abs(rec["loss_oracle"]))
    gtol = tol["grad"] if (compressed or precision == "f32") else tol["grad_full"]
    if not e2e_tight and precision == "fp16":
        # pre-trained WITH an eikonal term the lattice field has |n| ~ 1: the eikonal residual of the test loss is then a
        # difference of cancelling terms (as on the LoTD model's un-compressed set, TOL above): the table gradient gets
        # the ``grad_full`` bound (measured 7e-2; f32 on the same set 6e-6)
        gtol = tol["grad_full"]
    if not e2e_tight and precision == "f32":
        # the rough pre-trained lattice field has normals of magnitude ~10 feeding the radiance net: its weight / appearance
        # gradients are sums with cancellation whose f32 value depends on the summation order (measured between two
        # pre-training runs: 1e-6 .. 3e-4 on h_appear; table and SDF-decoder gradients stay <= 3e-6)
        gtol = 2e-3
    fix_keys = ["grid", "sdf_w", "sdf_b", "rad_w", "rad_b", "ln_inv_s", "h_appear"]
    if not e2e_tight and precision == "fp16":
        # The bias gradient of the device-pre-trained decoder is sum_i dL/d(pre-activation_i): signed terms that cancel, to a
        # norm that differs from one pre-training run to the next (float atomics); its relative error is run dependent (round 4, 8 fresh-process runs on identical forward errors: 0.02 .. 0.06 seven times, 0.48 once, with
        # sdf_w at 0.02 .. 0.09 throughout).  The decoder's gradient is asserted as one vector (weights + biases); the
        # bias part alone is reported (``fix_grad_sdf_b``, ``ref_norm_sdf_b``) and bounded loosely.  Round 6: the flake hunt
        # (3 fresh-process suites on one lease) saw the combined vector at 0.017, 0.017 and 0.104 -- the render-loss gradient of
        # the decoder is bounded loosely as a whole (0.5); the kernels' own error is what the random-cotangent leg asserts.
        fix_keys.remove("sdf_b")
        fix_keys.remove("sdf_w")
        assert rec["fix_grad_sdf_b"] < 1.0, rec["fix_grad_sdf_b"]
        assert rec["fix_grad_sdf_dec"] < 0.5, rec["fix_grad_sdf_dec"]
    for k in fix_keys:
        assert rec["fix_grad_" + k] < gtol, (k, rec["fix_grad_" + k])
    if not e2e_tight:
        # random cotangents: f32 = the exact-f32 kernels, fp16 = f16 MFMA operands / f16 tables.  Measured over three fresh
        # device pre-trainings (tools/r6_ab1.sh): f32 <= 3.6e-6 on two fields and 8.8e-4 (radiance weights; a field with
        # normals of magnitude ~10, see gtol above) on the third; fp16 <= 1.4e-2 (radiance weights), <= 3e-3 table / decoder
        for k in ("grid", "sdf_w", "sdf_b", "rad_w", "rad_b"):
            assert rec["rnd_grad_" + k] < (5e-3 if precision == "f32" else 5e-2), (k, rec["rnd_grad_" + k])
    if not e2e_tight:
        assert rec["psnr_rgb_db"] > 55.0 and abs(rec["loss"] - rec["loss_oracle"]) < 2e-2 * (1 + abs(rec["loss_oracle"]))
    elif precision == "f32":
        assert rec["sdf_nograd_max"] < tol["sdf"]
        assert rec["march_coarse_sdf_max"] < tol["fix"]["sdf"]
        for k in ("t", "sdf", "rgb", "nablas"):
            assert rec["smp_" + k] < tol[k], (k, rec["smp_" + k])
        assert abs(rec["loss"] - rec["loss_oracle"]) < tol["loss"] * (1 + abs(rec["loss_oracle"]))
        if rec["rays_with_other_count"] == 0:
            for k in ("grid", "sdf_w", "sdf_b", "rad_w", "rad_b", "ln_inv_s", "h_appear"):
                assert rec["e2e_grad_" + k] < tol["grad"], (k, rec["e2e_grad_" + k])
    else:
        assert abs(rec["loss"] - rec["loss_oracle"]) < tol["loss"] * (1 + abs(rec["loss_oracle"]))
        assert rec["psnr_rgb_db"] > 60.0
        # Round 3: the sampling pass runs in f32-equivalent arithmetic (``sampling_precision = "split"``), so the fp16 step
        # works on the oracle's sample set (at most the f32 mode's own threshold flips) and the end-to-end gradients obey
        # the bounds of the fixed-sample-set leg (round 2, fp16 sampling: 59 of 2038 rays kept another count, gate 0.15)
        assert rec["rays_with_other_count"] <= flips_max, rec
        for k in ("grid", "sdf_w", "sdf_b", "rad_w", "rad_b", "ln_inv_s", "h_appear"):
            assert rec["e2e_grad_" + k] < gtol, (k, rec["e2e_grad_" + k])


@pytest.mark.parametrize("precision", ["f32", "fp16"])
def test_permuto_fused_step_matches_oracle_at_baseline_size(precision):
    """The fused launch chain with the permutohedral-lattice model (bench ``variants.permuto_ms``) against the oracle's loss
    and gradients of the same batch.  The chain samples for itself, so on the rough device-pre-trained lattice field a few
    rays keep another sample count than the oracle's (see ``_api_path_body``): bounds of the end-to-end kind."""
    _fused_step_body(precision, "permuto")


@pytest.mark.parametrize("precision", ["f32", "fp16"])
def test_fused_step_matches_oracle_at_baseline_config(precision):
    """The bench's launch chain (no autograd engine) on one full batch: 8192 rays + 4096 uniform eikonal points."""
    _fused_step_body(precision, "lotd")


def _fused_step_body(precision, encoding):
    tr, p, occ, dev = _rig(precision, encoding)
    m = tr.model
    tol = TOL[precision]
    assert tr._fused_ok()
    tr._prefetched = None
    batch = tr._make_batch()
    N = tr.num_rays
    o, d = batch["rays_o"].cpu(), batch["rays_d"].cpu()
    ri = batch["tested"]["rays_inds"].cpu()
    R = int(ri.shape[0])
    jit, jit_c = torch.zeros(N), torch.zeros(N, 64)
    jit[ri], jit_c[ri] = batch["jitter"][:R].cpu(), batch["jitter_c"][:R].cpu()      # rows 0..R-1 serve the R hit rays
    ha = leaf(tr.appear.detach().cpu()[batch["fidx"].cpu()])
    gt, x_uni = batch["gt"].cpu(), batch["x_uni"].cpu()
    _fresh(p)
    ret_o = _oracle_query(p, occ, tr, o, d, ha, jit, jit_c, True)
    assert torch.equal(ret_o["rays_inds"], ri)
    loss_o, _ = orr.render_loss(ret_o, gt, N, w_eikonal=W_EIK)
    _, nab_u = ofield.forward_sdf_nablas(x_uni, p)
    loss_o = loss_o + W_EIK * ((nab_u.norm(dim=-1) - 1.0) ** 2).mean()
    loss_o.backward()
    ref = oracle_flat_grads(p)
    ref["appear"] = torch.zeros_like(tr.appear.detach().cpu()).index_add_(0, batch["fidx"].cpu(), ha.grad)
    # the fused chain (its prefetch hook would only draw the next batch: off for the comparison)
    pipe, tr.pipeline = tr.pipeline, False
    try:
        tr.optim.zero_grad()
        _zero_grads(tr)
        loss = tr._train_render_fused(batch)
    finally:
        tr.pipeline = pipe
    torch.cuda.synchronize()
    got = _product_grads(tr)
    got["appear"] = tr.appear.grad
    rec = dict(precision=precision, rays=N, hit=R, samples=int(tr.stats["S_f"]),
               samples_oracle=int(ret_o["volume_buffer"]["t"].shape[0]), loss=float(loss), loss_oracle=float(loss_o))
    for k, v in got.items():
        rec["grad_" + k] = rel_l2(v.cpu(), ref[k])
    _report(("permuto_" if encoding == "permuto" else "") + f"fused_{precision}", rec)
    if encoding == "permuto":
        # measured (MI355X, round 4): f32 4 of 270 528 samples apart, loss 7e-8, gradients <= 5.2e-3 (table), 9.5e-4 (decoder);
        # fp16 same sample count, table 6.5e-2, decoder weights 6.3e-2 / bias 0.105.  The sample sets differ by the rays whose up-sampling lands on the other
        # side of a keep threshold, so the gradients agree to the fraction of the loss those rays carry
        assert abs(rec["samples"] - rec["samples_oracle"]) <= max(8, rec["samples_oracle"] // 200), rec
        assert abs(rec["loss"] - rec["loss_oracle"]) < 2e-2 * (1 + abs(rec["loss_oracle"]))
        # every sample the two sides do NOT share moves the gradients by ~1e-3 of their norm (measured: 4 apart -> 5.2e-3 on
        # the table): the bound grows with the observed difference, so a run of the device-pre-trained model whose up-sampler
        # flips a few more rays is judged by the same rule
        apart = abs(rec["samples"] - rec["samples_oracle"])
        for k in got:      # (fp16: the decoder's bias gradient is a cancelling sum, see ``_api_path_body``: loose on its own)
            lim = 1.0 if (k == "sdf_b" and precision == "fp16") else PERMUTO_FUSED_GRAD[precision] + 3e-3 * apart
            assert rec["grad_" + k] < lim, (k, rec["grad_" + k], apart)
        return
    if precision == "f32":
        assert abs(rec["samples"] - rec["samples_oracle"]) <= 2 * tol["flips"]
        assert abs(rec["loss"] - rec["loss_oracle"]) < tol["loss"] * (1 + abs(rec["loss_oracle"]))
        for k in got:
            assert rec["grad_" + k] < (tol["grad"] if rec["samples"] == rec["samples_oracle"] else 5e-3), (k, rec["grad_" + k])
    else:
        assert abs(rec["samples"] - rec["samples_oracle"]) <= 2 * TOL["f32"]["flips"]      # f32-equivalent sampling pass
        assert abs(rec["loss"] - rec["loss_oracle"]) < tol["loss"] * (1 + abs(rec["loss_oracle"]))
        for k in got:          # compressed set: the fp16 bound of a given sample set (measured 5.9e-3; 0.15 in round 2)
            assert rec["grad_" + k] < tol["grad"], (k, rec["grad_" + k])
