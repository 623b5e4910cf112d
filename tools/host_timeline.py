#!/usr/bin/env python
"""Host-side wall-clock timeline of one training step (perf_counter around the phases, no profiler, no extra syncs):
where does the host spend its time between the unavoidable syncs?  Development aid."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import bench  # noqa: E402

T = {}


def wrap(obj, name, key):
    fn = getattr(obj, name)

    def w(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        T[key] = T.get(key, 0.0) + time.perf_counter() - t0
        return r
    setattr(obj, name, w)


def main():
    dev = torch.device("cuda", 0)
    tr = bench.build_trainer(dev, 0, 1)
    for it in range(241, 300):
        tr.train_step(it)
    torch.cuda.synchronize()
    m = tr.model
    wrap(tr, "sample_batch", "sample_batch")
    wrap(tr, "_make_batch", "_make_batch(prefetch, incl. sync)")
    wrap(tr, "sample_uniform_x", "uniform_x")
    wrap(m, "ray_test", "ray_test(sync)")
    wrap(m, "_sample", "_sample(launch)")
    wrap(m, "_compress", "_compress(sync)")
    wrap(tr, "render", "render(total)")
    wrap(tr, "loss", "loss")
    wrap(tr.optim, "step", "optim.step")
    wrap(tr.optim, "zero_grad", "zero_grad")
    import neuralsim_amd.fields.neus as N
    fa = N._FieldFn.apply
    orig_bwd = torch.Tensor.backward

    def bwd(self, *a, **k):
        t0 = time.perf_counter()
        r = orig_bwd(self, *a, **k)
        T["backward"] = T.get("backward", 0.0) + time.perf_counter() - t0
        return r
    torch.Tensor.backward = bwd
    n = 48
    t0 = time.perf_counter()
    for it in range(305, 305 + n):
        tr.train_step(it)
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) / n * 1e3
    print(f"step {tot:.3f} ms")
    for k, v in T.items():
        print(f"  {k:20s} {v / n * 1e3:7.3f} ms")


if __name__ == "__main__":
    main()
