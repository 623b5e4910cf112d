"""Do the memory-bound and the MFMA/LDS-bound kernels of the with-grad half overlap when they run on two HIP streams?
Replays the launches of one real training step (arguments captured from ``_lib.call``): each kernel alone, pairs back to
back on one stream, the same pairs concurrently on two streams.  Measurement aid for DESIGN.md (run on the GPU box)."""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from neuralsim_amd import _lib  # noqa: E402

NAMES = ("nsim_field_fwd", "nsim_field_bwd_rad", "nsim_field_bwd_sdf", "nsim_lotd_scatter", "nsim_lotd_gather_lm",
         "nsim_field_sdf")


def main():
    dev = torch.device("cuda", 0)
    tr = bench.build_trainer(dev, 0, 1)
    tr.spec_forward = False
    for it in range(20):
        tr.train_step(it)
    rec = {}
    real = _lib.call

    def spy(name, *args):
        if name in NAMES:
            rec.setdefault(name, []).append(args)
        return real(name, *args)
    _lib.call = spy
    import neuralsim_amd.trainer as T
    import neuralsim_amd.fields.neus as N
    T._lib.call = spy
    tr.train_step(20)
    _lib.call = real
    torch.cuda.synchronize()
    calls = {k: v[0] for k, v in rec.items()}
    if "nsim_lotd_gather_lm" in rec:       # the largest no-grad gather of the step
        calls["nsim_lotd_gather_lm"] = max(rec["nsim_lotd_gather_lm"], key=lambda a: a[8])
    print({k: len(v) for k, v in rec.items()}, file=sys.stderr)

    def run(name):
        real(name, *calls[name])

    def timed(fn, n=20):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    s2 = torch.cuda.Stream(device=dev)
    out = {}
    for k in calls:
        out[k] = round(timed(lambda: run(k)), 4)

    def pair(a, b, conc):
        def f():
            if conc:
                ev = torch.cuda.current_stream().record_event()
                with torch.cuda.stream(s2):
                    s2.wait_event(ev)
                    run(b)
                    done = s2.record_event()
                run(a)
                torch.cuda.current_stream().wait_event(done)
            else:
                run(a)
                run(b)
        return f
    for a, b in (("nsim_lotd_scatter", "nsim_field_bwd_sdf"), ("nsim_lotd_scatter", "nsim_field_bwd_rad"),
                 ("nsim_lotd_scatter", "nsim_field_fwd"), ("nsim_lotd_gather_lm", "nsim_field_bwd_sdf"),
                 ("nsim_lotd_gather_lm", "nsim_field_sdf"), ("nsim_field_bwd_rad", "nsim_field_bwd_sdf"),
                 ("nsim_lotd_scatter", "nsim_lotd_gather_lm")):
        if a in calls and b in calls:
            out[f"{a[5:]}+{b[5:]} serial"] = round(timed(pair(a, b, False)), 4)
            out[f"{a[5:]}+{b[5:]} two streams"] = round(timed(pair(a, b, True)), 4)
    # the scatter level by level (its last two arguments are level_begin, level_count)
    sa = list(calls["nsim_lotd_scatter"])
    per = {}
    for l in range(16):
        sa[-2], sa[-1] = l, 1
        args = tuple(sa)
        per[l] = round(timed(lambda: real("nsim_lotd_scatter", *args), 10), 4)
    out["scatter_per_level_ms"] = per
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
