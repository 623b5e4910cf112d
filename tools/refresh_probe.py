"""Where one occupancy refresh (4 x 2^20 SDF queries) goes, and what a voxel-ordered (stratified) draw of the same
number of points would cost -- measurement aid for DESIGN.md (run on the GPU box)."""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402


def timed(fn, n=5):
    torch.cuda.synchronize()
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    dev = torch.device("cuda", 0)
    tr = bench.build_trainer(dev, 0, 1)
    m = tr.model
    for it in range(20):
        tr.train_step(it)
    acc = m.accel
    lo, hi = acc.aabb[0], acc.aabb[1]
    N = 1 << 20
    g = torch.Generator(device=dev).manual_seed(1)
    out = {}
    pts_r = lo + torch.rand([N, 3], device=dev, generator=g) * (hi - lo)
    res = 64
    vid = torch.arange(N, device=dev) % (res ** 3)
    ijk = torch.stack([vid % res, (vid // res) % res, vid // (res * res)], -1).float()      # x fastest
    pts_s = lo + (ijk + torch.rand([N, 3], device=dev, generator=g)) / res * (hi - lo)
    def block_pts(bs, rep):
        i = torch.arange(N, device=dev)
        v = (i // rep) % (res ** 3)
        nb = res // bs
        blk, loc = v // (bs ** 3), v % (bs ** 3)
        x = (blk % nb) * bs + loc % bs
        y = ((blk // nb) % nb) * bs + (loc // bs) % bs
        z = (blk // (nb * nb)) * bs + loc // (bs * bs)
        ijk_ = torch.stack([x, y, z], -1).float()
        return lo + (ijk_ + torch.rand([N, 3], device=dev, generator=g)) / res * (hi - lo)
    for bs, rep in ((4, 1), (4, 4), (8, 1), (2, 1), (4, 2)):
        p_ = block_pts(bs, rep)
        out[f"query_block{bs}_rep{rep}_ms"] = timed(lambda: m.query_sdf(p_))
    out["refresh_total_ms"] = timed(lambda: acc.update_from_net(m.query_sdf, generator=g), 3)
    out["rand_points_ms"] = timed(lambda: lo + torch.rand([N, 3], device=dev, generator=g) * (hi - lo))
    out["query_random_ms"] = timed(lambda: m.query_sdf(pts_r))
    out["query_voxel_ordered_ms"] = timed(lambda: m.query_sdf(pts_s))
    sdf = m.query_sdf(pts_r)
    out["occ_update_ms"] = timed(lambda: acc.update_from_samples(pts_r, sdf, pack=False))
    out["pack_bits_ms"] = timed(lambda: acc.pack_bits())
    print(json.dumps({k: round(v, 4) for k, v in out.items()}))


if __name__ == "__main__":
    main()
