"""Diagnostics of bench.py's fp16-vs-oracle rendering check: the rays with the largest colour error, their sample counts
in both pipelines, masks and depths (run on the GPU box).  usage: parity_probe.py [warmup steps]"""
import json
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
    # (bench.py also measures variants before the check; they do not touch this trainer's state)
    m = tr.model
    p, occ = bench.oracle_of(tr)
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
        rend = SingleVolumeRenderer(dict(with_rgb=True, near=0.01, depth_use_normalized_vw=True)).eval()
        out = rend.render(m, rays=[o.to(dev), d.to(dev)], rays_h_appear=ha.to(dev), return_buffer=True)
        ri = ret["rays_inds"]
        z = lambda k: torch.zeros(n_par, *ret["rendered"][k].shape[1:]).index_put((ri,), ret["rendered"][k])  # noqa: E731
        rgb_o, mask_o, dep_o = z("rgb_volume"), z("mask_volume"), z("depth_volume")
        rgb_h, mask_h, dep_h = (out["rendered"][k].cpu() for k in ("rgb_volume", "mask_volume", "depth_volume"))
        err = (rgb_h - rgb_o).abs().max(dim=-1).values
        vb = out["volume_buffer"]
        n_h = torch.zeros(n_par, dtype=torch.long).index_put((vb["rays_inds_hit"].cpu(),), vb["pack_infos_hit"][:, 1].cpu())
        n_o = torch.zeros(n_par, dtype=torch.long).index_put((ri,), ret["volume_buffer"]["pack_infos_hit"][:, 1])
        n_q = torch.zeros(n_par, dtype=torch.long).index_put((ri,), ret["debug"]["pack_infos"][:, 1])
        rec = dict(psnr=float(-10 * torch.log10(((rgb_h - rgb_o) ** 2).mean())), max=float(err.max()),
                   q=[float(err.quantile(q)) for q in (0.5, 0.9, 0.99, 0.999)],
                   rays_with_different_sample_count=int((n_h != n_o).sum()), total_o=int(n_o.sum()), total_h=int(n_h.sum()))
        top = err.argsort(descending=True)[:8]
        rec["worst"] = [dict(ray=int(i), err=round(float(err[i]), 5), n_oracle=int(n_o[i]), n_hip=int(n_h[i]),
                             n_queried=int(n_q[i]), mask=(round(float(mask_o[i]), 4), round(float(mask_h[i]), 4)),
                             depth=(round(float(dep_o[i]), 4), round(float(dep_h[i]), 4)),
                             rgb_o=[round(float(x), 4) for x in rgb_o[i]], rgb_h=[round(float(x), 4) for x in rgb_h[i]])
                        for i in top]
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
