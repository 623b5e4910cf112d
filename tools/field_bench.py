#!/usr/bin/env python
"""Micro-benchmark of the with-grad field query (gather -> decoders -> backward -> scatter) on synthetic samples, without a
scenario around it: per-entry-point HIP-event times of the C ABI.  Used to iterate on the 17..32-level (NC = 2) kernels of
the street configuration (19 cuboid levels, T = 2^20, 1x64 decoder) at its real point count.

    python tools/field_bench.py [--shape street|object] [--rays 16384] [--per-ray 85] [--iters 6]
    rocprofv3 --kernel-trace --stats -- python tools/field_bench.py ...        # kernel split of nsim_field_fwd
"""
import argparse
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="street", choices=("street", "object", "permuto", "vehicle", "vehicle_noembed"))
    ap.add_argument("--rays", type=int, default=16384)
    ap.add_argument("--per-ray", type=int, default=85)
    ap.add_argument("--iters", type=int, default=6)
    ap.add_argument("--precision", default="fp16")
    args = ap.parse_args()
    from neuralsim_amd import _lib
    from neuralsim_amd.fields.neus import LoTDNeuSModel, _FieldFn
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    if args.shape == "street":
        from neuralsim_amd.scenarios import STREET_AABB, STREET_SDF_SCALE, cuboid_ngp_res
        aabb = torch.tensor(STREET_AABB)
        res = cuboid_ngp_res((aabb[1] - aabb[0]).tolist(), 16, 2048, 19)
        m = LoTDNeuSModel(lod_res=res, log2_hashmap_size=20, sdf_D=1, precision=args.precision, sdf_scale=STREET_SDF_SCALE,
                          aabb=aabb, seed=1).to(dev)
    elif args.shape == "permuto":      # PermutoNeuSObj at the LoTD object model's table size: 16 levels x 2^19 entries, 2x64 decoder
        from neuralsim_amd.fields.permuto_neus import PermutoNeuSModel
        aabb = torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
        m = PermutoNeuSModel(permuto_auto_compute_cfg=dict(type="multi_res", n_levels=16, n_feats=2, log2_hashmap_size=19,
                                                           coarsest_res=16.0, finest_res=2000.0), sdf_D=2,
                             precision=args.precision, seed=1).to(dev)
    elif args.shape.startswith("vehicle"):
        # the decoder shape of the StyleLoTD Vehicle block (no_fg_occ.221218.yaml:307-357): 8 grown levels x 4 features = 16 plane
        # levels, relu 2x64 decoder, + sinusoidal_legacy-6 position embedding (71 inputs: csrc/wide_field.hip) | without it (MFMA)
        aabb = torch.tensor([[-0.7, -0.7, -0.7], [0.7, 0.7, 0.7]])
        res = [5, 5, 8, 8, 13, 13, 21, 21, 34, 34, 55, 55, 89, 89, 144, 144]
        m = LoTDNeuSModel(lod_res=res, log2_hashmap_size=22, sdf_D=2, precision=args.precision, softplus_beta=-1.0, aabb=aabb,
                          pos_embed_frequencies=6 if args.shape == "vehicle" else None, seed=1).to(dev)
    else:
        aabb = torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
        m = LoTDNeuSModel(sdf_D=2, precision=args.precision, seed=1).to(dev)
    with torch.no_grad():
        m.encoding.flattened_params.normal_(0, 1e-2, generator=g)
    R, K = args.rays, args.per_ray
    lo, hi = aabb[0].to(dev), aabb[1].to(dev)
    o = lo + (hi - lo) * (0.3 + 0.4 * torch.rand(R, 3, device=dev, generator=g))
    d = torch.nn.functional.normalize(torch.randn(R, 3, device=dev, generator=g), dim=-1)
    # K samples per ray inside a short stretch (as the up-sampled sets are): consecutive samples share fine cells
    t0 = torch.rand(R, 1, device=dev, generator=g) * 0.2 * float((hi - lo).min())
    t = (t0 + torch.sort(torch.rand(R, K, device=dev, generator=g), dim=-1).values * 0.05 * float((hi - lo).min())).reshape(-1)
    ridx = torch.arange(R, device=dev).repeat_interleave(K)
    ha = (torch.randn(R, 4, device=dev, generator=g) * 0.1).requires_grad_(True)
    S = R * K
    ws, wn, wr = torch.randn(S, device=dev, generator=g), torch.randn(S, 3, device=dev, generator=g) * 0.1, torch.randn(S, 3, device=dev, generator=g)
    _lib.TIMER = None

    def step():
        for p in (m.encoding.flattened_params, m.sdf_w, m.sdf_b, m.rad_w, m.rad_b):
            p.grad = None
        sdf, nab, rgb = _FieldFn.apply(m, m.encoding.flattened_params, m.sdf_w, m.sdf_b, m.rad_w, m.rad_b, ha, None, o, d, t, ridx, True)
        ((sdf * ws).sum() + (nab * wn).sum() + (rgb * wr).sum()).backward()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    _lib.TIMER = _lib.KernelTimer()
    for _ in range(args.iters):
        step()
    summ = _lib.TIMER.summary()
    _lib.TIMER = None
    out = dict(shape=args.shape, levels=m.encoding.cfg.num_levels, points=S, precision=args.precision,
               avg_ms={k: round(v["avg_ms"], 4) for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["total_ms"])},
               ns_per_point={k: round(v["avg_ms"] * 1e6 / S, 4) for k, v in summ.items() if v["avg_ms"] > 0.01})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
