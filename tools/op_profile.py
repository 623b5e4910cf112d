#!/usr/bin/env python
"""aten-op level profile of steady-state training steps (torch.profiler, CPU + device activity, input shapes and the Python
source line of every op whose inputs are large).  Development aid:   python tools/op_profile.py [object|street|indoor|multi]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import bench  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "object"
    dev = torch.device("cuda", 0)
    tr = bench.build_trainer(dev, 0, 1) if cfg == "object" else bench.build_config_trainer(cfg, dev, 0, 1, 16384)
    for it in range(250, 262):
        tr.train_step(it)
    torch.cuda.synchronize()
    n = 6
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        for it in range(273, 273 + n):
            tr.train_step(it)
        torch.cuda.synchronize()
    ka = prof.key_averages(group_by_input_shape=True, group_by_stack_n=6)
    rows = sorted(ka, key=lambda e: -(e.self_device_time_total if hasattr(e, "self_device_time_total") else e.self_cuda_time_total))
    dt = lambda e: (e.self_device_time_total if hasattr(e, "self_device_time_total") else e.self_cuda_time_total)   # noqa: E731
    print(f"{cfg}: device self time {sum(dt(e) for e in rows) / n:.0f} us/step, ops/step {sum(e.count for e in rows) / n:.0f}")
    for e in rows[:45]:
        if dt(e) / n < 15:
            break
        stack = [s for s in (e.stack or []) if "neuralsim_amd" in s or "bench.py" in s][:3]
        print(f"{e.key[:44]:44s} n/step {e.count / n:5.1f} dev {dt(e) / n:8.1f} us/step shapes {str(e.input_shapes)[:70]:70s} {' <- '.join(x.split('/')[-1][:60] for x in stack)}")


if __name__ == "__main__":
    main()
