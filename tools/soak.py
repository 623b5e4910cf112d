"""Soak run of the bench workload (GPU box): N training steps, counting the speculative launches that had to be redone,
watching the loss, the sample counts and the allocator -- robustness evidence for DESIGN.md.  usage: soak.py [steps]"""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    mode = sys.argv[2] if len(sys.argv) > 2 else "fused"           # fused | api | distant
    dev = torch.device("cuda", 0)
    tr = bench.build_trainer(dev, 0, 1, distant=mode == "distant")
    if mode == "api":
        tr.fused_step = False
    m = tr.model
    it = 0
    redo = nospec = 0
    rec = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(steps):
        loss = tr.train_step(it)
        ok = bool(getattr(m, "_spec_ok", False))
        if not ok:
            if getattr(m, "_keep_stat", None) is None or it == 0:
                nospec += 1
            else:
                redo += 1
        if it % 500 == 0 or it == steps - 1:
            rec.append(dict(it=it, loss=round(float(loss), 6), S_f=tr.stats["S_f"], S_q=tr.stats.get("S_q"), R_hit=tr.stats["R_hit"],
                            R_live=tr.stats.get("R_live"),
                            occupied=round(m.accel.frac_occupied(), 4),
                            mem_MB=round(torch.cuda.max_memory_allocated() / 2 ** 20, 1),
                            reserved_MB=round(torch.cuda.memory_reserved() / 2 ** 20, 1)))
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(json.dumps(dict(mode=mode, steps=steps, ms_per_step=round(el / steps * 1e3, 4), speculative_redone=redo, not_speculated=nospec,
                          notify_alive=bool(m._host_notify() is not None), trace=rec)))


if __name__ == "__main__":
    main()
