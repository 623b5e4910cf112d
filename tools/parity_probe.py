"""Diagnostics of bench.py's fp16-vs-oracle rendering check: the rays with the largest colour error, their sample counts
in both pipelines, masks and depths (run on the GPU box).  usage: parity_probe.py [warmup steps]"""
import json
import os
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from oracle import render as orr  # noqa: E402


def main():
    warm = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    dev = torch.device("cuda", 0)
    tr = bench.build_trainer(dev, 0, 1)
    it = 257 - warm
    for _ in range(warm):
        tr.train_step(it)
        it += 1
    it = 257
    for _ in range(steps):
        tr.train_step(it)
        it += 1
    n_api = int(sys.argv[3]) if len(sys.argv) > 3 else 0       # then n_api steps through the autograd (API) path
    tr.fused_step = False
    for _ in range(n_api):
        tr.train_step(it)
        it += 1
    tr.fused_step = True
    # (bench.py also measures variants before the check; they do not touch this trainer's state)
    m = tr.model
    p, occ = bench.oracle_of(tr, table=os.environ.get("PROBE_TABLE", "stored"))
    if os.environ.get("PROBE_W16") == "1":       # decoder weights as the fp16 MFMA operands hold them
        for lst in (p.sdf_w, p.rad_w):
            for i, w in enumerate(lst):
                lst[i] = w.detach().half().float()
    aabb = m.accel.aabb.detach().cpu()
    intr, c2w, WH = tr.intr.cpu(), tr.c2w.cpu(), tr.WH.cpu()
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():
        n_par = bench.PARITY_RAYS
        xy = torch.rand(n_par, 2, generator=g).clamp(1e-6, 1 - 1e-6)
        fidx = torch.randint(0, intr.shape[0], (n_par,), generator=g)
        o, d = orr.pinhole_rays(xy, fidx, intr, c2w, WH)
        ha = tr.appear.detach().cpu()[fidx]
        ret = orr.ray_query(p, o, d, ha, occ, aabb[0], aabb[1], m.accel.resolution, near=0.01,
                            depth_use_normalized_vw=True, compress=True)
        from neuralsim_amd.renderers.single_volume_renderer import SingleVolumeRenderer
        rend = SingleVolumeRenderer(dict(with_rgb=True, with_normal=True, near=0.01, depth_use_normalized_vw=True)).eval()
        out = rend.render(m, rays=[o.to(dev), d.to(dev)], rays_h_appear=ha.to(dev), return_buffer=True)
        ri = ret["rays_inds"]
        z = lambda k: torch.zeros(n_par, *ret["rendered"][k].shape[1:]).index_put((ri,), ret["rendered"][k])  # noqa: E731
        rgb_o, mask_o, dep_o = z("rgb_volume"), z("mask_volume"), z("depth_volume")
        rgb_h, mask_h, dep_h = (out["rendered"][k].cpu() for k in ("rgb_volume", "mask_volume", "depth_volume"))
        nk = [k for k in ret["rendered"] if "normal" in k]
        nrm_o = z(nk[0]) if nk else None
        nrm_h = out["rendered"]["normals_volume"].cpu() if "normals_volume" in out["rendered"] else None
        err = (rgb_h - rgb_o).abs().max(dim=-1).values
        vb = out["volume_buffer"]
        n_h = torch.zeros(n_par, dtype=torch.long).index_put((vb["rays_inds_hit"].cpu(),), vb["pack_infos_hit"][:, 1].cpu())
        n_o = torch.zeros(n_par, dtype=torch.long).index_put((ri,), ret["volume_buffer"]["pack_infos_hit"][:, 1])
        n_q = torch.zeros(n_par, dtype=torch.long).index_put((ri,), ret["debug"]["pack_infos"][:, 1])
        rec = dict(psnr=float(-10 * torch.log10(((rgb_h - rgb_o) ** 2).mean())), max=float(err.max()),
                   q=[float(err.quantile(q)) for q in (0.5, 0.9, 0.99, 0.999)],
                   rays_with_different_sample_count=int((n_h != n_o).sum()), total_o=int(n_o.sum()), total_h=int(n_h.sum()))
        if nrm_o is not None and nrm_h is not None:
            ne = (nrm_h - nrm_o).abs().max(dim=-1).values
            rec["normals_err_q"] = [float(ne.quantile(q)) for q in (0.5, 0.99, 0.999, 1.0)]
        top = err.argsort(descending=True)[:8]
        rec["worst"] = [dict(ray=int(i), err=round(float(err[i]), 5), n_oracle=int(n_o[i]), n_hip=int(n_h[i]),
                             n_queried=int(n_q[i]), mask=(round(float(mask_o[i]), 4), round(float(mask_h[i]), 4)),
                             depth=(round(float(dep_o[i]), 4), round(float(dep_h[i]), 4)),
                             rgb_o=[round(float(x), 4) for x in rgb_o[i]], rgb_h=[round(float(x), 4) for x in rgb_h[i]],
                             nrm_o=[round(float(x), 4) for x in nrm_o[i]] if nrm_o is not None else None,
                             nrm_h=[round(float(x), 4) for x in nrm_h[i]] if nrm_h is not None else None)
                        for i in top]
    # conditioning of the normal on the worst ray: n = sum_k (d h_k / d x) (d sdf / d h_k) over the 32 features; a relative
    # error eps of the second factor (the fp16 decoder chain) moves n by <= eps * sum_k |J_k g_k|
    from oracle import field as ofield, lotd as olotd
    i = int(top[0])
    j = int((ri == i).nonzero()[0, 0]) if bool((ri == i).any()) else None
    if j is not None:
        vb_o = ret["volume_buffer"]
        st, n = [int(v) for v in vb_o["pack_infos_hit"][j]]
        x = vb_o["net_x"][st:st + n].detach().clone().requires_grad_(True)
        with torch.enable_grad():
            h = olotd.lotd_forward(x, p.grid.float(), p.spec)
            hd = h.detach().clone().requires_grad_(True)
            sdf = ofield.sdf_decoder(hd, p)
            g = torch.autograd.grad(sdf.sum(), hd)[0]                          # [n, 32]
            J = torch.stack([torch.autograd.grad(h[:, k].sum(), x, retain_graph=True)[0] for k in range(h.shape[1])], 1)
        contrib = J * g[:, :, None]                                            # [n, 32, 3]
        nab = contrib.sum(1)
        kappa = contrib.abs().sum(1) / nab.abs().clamp_min(1e-6)
        L = h.shape[1] // 2
        per_level = contrib.view(n, L, 2, 3).sum(2).abs().mean(0).max(-1).values
        rec["worst_ray_normal_conditioning"] = dict(
            sum_abs_terms_mean=[round(float(v), 2) for v in contrib.abs().sum(1).mean(0)],
            nablas_mean_abs=[round(float(v), 3) for v in nab.abs().mean(0)],
            kappa_median=[round(float(v), 1) for v in kappa.median(0).values],
            per_level_mean_abs_term=[round(float(v), 2) for v in per_level],
            g_abs_max=round(float(g.abs().max()), 3), J_abs_max=round(float(J.abs().max()), 1))
        # the product's field on the ORACLE's samples of that ray: per-sample differences
        from neuralsim_amd.fields.neus import _FieldFn
        with torch.no_grad():
            t_o = vb_o["t"][st:st + n].to(dev).contiguous()
            o1, d1 = o[i:i + 1].to(dev).contiguous(), d[i:i + 1].to(dev).contiguous()
            outs = _FieldFn.apply(m, m.encoding.flattened_params, m.sdf_w, m.sdf_b, m.rad_w, m.rad_b, ha[i:i + 1].to(dev), None,
                                  o1, d1, t_o, torch.zeros(n, dtype=torch.long, device=dev), True)
            sdf_h, nab_h, rgb_h1 = (v.cpu() for v in outs[:3])
        d_rgb = (rgb_h1 - vb_o["rgb"][st:st + n]).abs()
        d_nab = (nab_h - vb_o["nablas"][st:st + n]).abs()
        d_sdf = (sdf_h - vb_o["sdf"][st:st + n]).abs()
        vw = ofield_vw = ret["rendered"]["vw"][st:st + n] if "vw" in ret["rendered"] else None
        k = int(d_rgb.max(dim=-1).values.argmax())
        rec["worst_ray_same_samples"] = dict(
            max_d_sdf=float(d_sdf.max()), max_d_nablas=float(d_nab.max()), max_d_rgb=float(d_rgb.max()),
            nablas_norm_range=[float(vb_o["nablas"][st:st + n].norm(dim=-1).min()), float(vb_o["nablas"][st:st + n].norm(dim=-1).max())],
            at_sample=dict(k=k, sdf_o=float(vb_o["sdf"][st + k]), sdf_h=float(sdf_h[k]),
                           nab_o=[round(float(v), 4) for v in vb_o["nablas"][st + k]], nab_h=[round(float(v), 4) for v in nab_h[k]],
                           rgb_o=[round(float(v), 4) for v in vb_o["rgb"][st + k]], rgb_h=[round(float(v), 4) for v in rgb_h1[k]],
                           vw=float(vw[k]) if vw is not None else None),
            rgb_err_weighted=[float(v) for v in ((rgb_h1 - vb_o["rgb"][st:st + n]) * vw[:, None]).sum(0)] if vw is not None else None)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
