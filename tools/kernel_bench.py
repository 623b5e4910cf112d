#!/usr/bin/env python
"""Per-kernel timing of the field kernels on one fixed render batch (8192 rays of the bench workload), with the
NSIM_ABLATE / NSIM_DEDUP_MAX_RES profiling knobs of csrc/field.hip.  Development aid, not part of the product."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import bench  # noqa: E402
from neuralsim_amd import _lib  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    tr = bench.build_trainer(dev, 0, 1)
    xy, fidx, gt = tr.sample_batch()
    variants = [("full", {}), ("no-scatter", {"NSIM_ABLATE": "1"}), ("no-dW", {"NSIM_ABLATE": "4"}),
                ("no-scatter,no-dW", {"NSIM_ABLATE": "5"}), ("no-dedup", {"NSIM_DEDUP_MAX_RES": "0"}),
                ("dedup-all", {"NSIM_DEDUP_MAX_RES": "4096"})]
    for name, env in variants:
        for k in ("NSIM_ABLATE", "NSIM_DEDUP_MAX_RES"):
            os.environ.pop(k, None)
        os.environ.update(env)
        for rep in range(3):
            _lib.TIMER = _lib.KernelTimer() if rep == 2 else None
            ret = tr.render(xy, fidx)
            loss, _ = tr.loss(ret, gt)
            tr.optim.zero_grad()
            loss.backward()
            torch.cuda.synchronize()
        s = _lib.TIMER.summary()
        _lib.TIMER = None
        S = ret["raw_per_obj_model"]["main"]["volume_buffer"]["t"].shape[0]
        print(f"{name:20s} S_f={S} " + " ".join(f"{k[5:]}={v['total_ms']:.3f}ms/{v['calls']}" for k, v in s.items()
                                              if k.startswith("nsim_field")), flush=True)


if __name__ == "__main__" and "--lotd" not in sys.argv:
    main()


def lotd_standalone():
    """How fast is the un-fused, high-occupancy gather / scatter (csrc/lotd.hip) on the same sample set?"""
    dev = torch.device("cuda", 0)
    for k in ("NSIM_ABLATE", "NSIM_DEDUP_MAX_RES"):
        os.environ.pop(k, None)
    tr = bench.build_trainer(dev, 0, 1)
    xy, fidx, gt = tr.sample_batch()
    from neuralsim_amd.graphics.cameras import pinhole_selected_rays
    rays_o, rays_d = pinhole_selected_rays(xy, fidx, tr.intr, tr.c2w, tr.WH)
    tested = tr.model.ray_test(rays_o, rays_d, near=0.01)
    ret = tr.model.ray_query(ray_tested=tested, config=dict(tr.model.ray_query_cfg, query_mode="march_occ_multi_upsample"),
                             return_details=True)
    vb = ret["volume_buffer"]
    ridx = ret["details"]["ridx"]
    x = (tested["rays_o"][ridx] + vb["t"][:, None] * tested["rays_d"][ridx]).contiguous()
    enc = tr.model.encoding
    S = x.shape[0]
    out = torch.zeros(S, 32, device=dev)
    dydx = torch.zeros(S, 32, 3, device=dev)
    dgrid = torch.zeros(enc.cfg.n_params, device=dev)
    g16 = enc.shadow()

    def timeit(fn, n=5):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    t1 = timeit(lambda: _lib.call("nsim_lotd_fwd", _lib.ptr(x), _lib.ptr(g16), enc.cfg.meta, S, _lib.ptr(out), None))
    t2 = timeit(lambda: _lib.call("nsim_lotd_fwd", _lib.ptr(x), _lib.ptr(g16), enc.cfg.meta, S, _lib.ptr(out), _lib.ptr(dydx)))
    t3 = timeit(lambda: _lib.call("nsim_lotd_bwd", _lib.ptr(x), _lib.ptr(out), None, enc.cfg.meta, S, _lib.ptr(dgrid)))
    t4 = timeit(lambda: _lib.call("nsim_lotd_bwd", _lib.ptr(x), _lib.ptr(out), _lib.ptr(dydx), enc.cfg.meta, S, _lib.ptr(dgrid)))
    print(f"standalone LoTD on S={S}: fwd {t1:.3f} ms, fwd+dydx {t2:.3f} ms, bwd {t3:.3f} ms, bwd+dydx {t4:.3f} ms", flush=True)


if __name__ == "__main__" and "--lotd" in sys.argv:
    lotd_standalone()
