#!/usr/bin/env python
"""Which of bench.py's side measurements moves the fp16-vs-oracle rendering check of the SAME trainer (development aid):
the check after the fused steps, after API-path steps, after an 800x800 evaluation render."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import bench  # noqa: E402


def show(tag, tr):
    p = bench.parity_check(tr)
    print(tag, p["psnr_rgb_db"], p["psnr_rgb_db_without_worst_0p1pct"], p["max_abs_rgb"], p["rays_beyond_max_tol"], p["p99_abs_rgb"],
          p["p999_abs_rgb"], p["worst_ray"], p["ok"], flush=True)


def main():
    dev = torch.device("cuda", 0)
    tr = bench.build_trainer(dev, 0, 1)
    it = 241
    for _ in range(96):
        tr.train_step(it)
        it += 1
    show("after 96 fused steps        ", tr)
    for _ in range(30):
        tr.train_step(it)
        it += 1
    show("after 30 more fused steps   ", tr)
    tr.fused_step = False
    for _ in range(30):
        tr.train_step(it)
        it += 1
    tr.fused_step = True
    show("after 30 API-path steps     ", tr)
    for _ in range(30):
        tr.train_step(it)
        it += 1
    show("after 30 more fused steps   ", tr)


if __name__ == "__main__" and "--dump" not in sys.argv:
    main()


def dump_worst(tr):
    """the worst ray of the rendering check: what the oracle and the HIP path made of it (march count, kept samples, weights)"""
    import math
    from oracle import render as orr
    from neuralsim_amd.fields.neus import marched_only
    from neuralsim_amd.renderers.single_volume_renderer import SingleVolumeRenderer
    m = tr.model
    p, occ = bench.oracle_of(tr, table="stored")
    aabb = m.accel.aabb.detach().cpu()
    intr, c2w, WH = tr.intr.cpu(), tr.c2w.cpu(), tr.WH.cpu()
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():
        dev = m.device
        n_par = bench.PARITY_RAYS
        xy = torch.rand(n_par, 2, generator=g).clamp(1e-6, 1 - 1e-6)
        fidx = torch.randint(0, intr.shape[0], (n_par,), generator=g)
        o, d = orr.pinhole_rays(xy, fidx, intr, c2w, WH)
        ha = tr.appear.detach().cpu()[fidx]
        mode = m.ray_query_cfg.get("query_mode", "")
        ret = orr.ray_query(p, o, d, ha, occ, aabb[0], aabb[1], m.accel.resolution, near=0.01, depth_use_normalized_vw=True,
                            compress=mode.endswith("_compressed"),
                            upsample_on_marched_only=marched_only(m.ray_query_cfg.get("query_param", {})))
        rgb_o = torch.zeros(n_par, 3).index_put((ret["rays_inds"],), ret["rendered"]["rgb_volume"])
        rend = SingleVolumeRenderer(dict(with_rgb=True, near=0.01, depth_use_normalized_vw=True)).eval()
        out = rend.render(m, rays=[o.to(dev), d.to(dev)], rays_h_appear=ha.to(dev))
        rgb_h = out["rendered"]["rgb_volume"].cpu()
        err = (rgb_h - rgb_o).abs().max(dim=-1).values
        iw = int(err.argmax())
        print("worst ray", iw, "err", float(err[iw]), "hip", rgb_h[iw].tolist(), "oracle", rgb_o[iw].tolist())
        tested = ret["rays_inds"].tolist()
        if iw in tested:
            k = tested.index(iw)
            dbg = ret["debug"]
            print(" oracle: tested, march count", int(dbg["march_counts"][k]), "kept", int(dbg.get("compress_counts", dbg["pack_infos"][:, 1])[k]),
                  "mask", float(ret["rendered"]["mask_volume"][k]))
            pi = ret["pack_infos_tested"][k]
            a = ret["volume_buffer"]["opacity_alpha"][int(pi[0]):int(pi[0] + pi[1])]
            print(" oracle alphas of the kept samples:", [round(float(v), 4) for v in a][:20], " sdf:",
                  [round(float(v), 4) for v in ret["volume_buffer"]["sdf"][int(pi[0]):int(pi[0] + pi[1])]][:20])
        else:
            print(" oracle: ray failed the AABB test")
        print(" hip mask", float(out["rendered"]["mask_volume"][iw]), "keys", list(out.keys()))
        # the no-grad sampling of this ray alone through the model's API
        tested1 = m.ray_test(o[iw:iw + 1].to(dev), d[iw:iw + 1].to(dev), near=0.01)
        print(" hip ray_test num_rays", tested1["num_rays"])
        if tested1["num_rays"]:
            cfg = dict(m.ray_query_cfg)
            qp = dict(cfg.get("query_param", {}))
            _o, _d, t, pi, ridx, sdf_ng, mc, _g, _f = m._query_samples(tested1, cfg, qp)
            print(" hip alone: march count", mc.cpu().tolist(), "kept", pi.cpu().tolist(), "S_q", getattr(m, "_last_S_q", None))


if "--dump" in sys.argv:
    def main():      # noqa: F811
        dev = torch.device("cuda", 0)
        tr = bench.build_trainer(dev, 0, 1)
        it = 241
        for _ in range(126):
            tr.train_step(it)
            it += 1
        dump_worst(tr)
    main()
