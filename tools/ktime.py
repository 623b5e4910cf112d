#!/usr/bin/env python
"""Per-phase s_memtime breakdown of the joint backward kernels (needs a library built with -DNSIM_KTIME:
NSIM_EXTRA_HIPCC_FLAGS=-DNSIM_KTIME python -m neuralsim_amd.csrc.build --force).  Development aid."""
import ctypes
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import bench  # noqa: E402
from neuralsim_amd import _lib  # noqa: E402

LABELS = {0: ["loop top", "loads+rin", "rad fwd (2 dense)", "dout/scale", "barrier A3", "stage P3", "barrier B3", "dW3+rowsum",
              "dense R3T", "barrier A2", "stage P2", "barrier B2", "dW2+rowsum", "dense R2T", "barrier A1", "stage P1+barrier",
              "dW1+rowsum", "dense R1T + outputs"],
          1: ["loop top", "loads h, J (planes)", "fwd recompute (a1 a2 d2 e1 d1 g)", "gs/gn loads, gh", "dense dh1", "dW(d1,gh)",
              "dz1/eh1/d2", "dW(d2,eh1)", "dense dh2, whv/dz2", "dW(dz2,a1)", "dense da1", "dz1 +=", "rowsum whv",
              "dW(dz1,h)", "dense dh", "dh/g stores", "-", "-"],
          2: ["loop top", "loads h, J (planes)", "sdf fwd (a1 a2 sdf)", "g chain (d2 e1 d1 g)", "nablas", "radiance net", "stores"] + ["-"] * 11}


def main():
    dev = torch.device("cuda", 0)
    tr = bench.build_trainer(dev, 0, 1)
    it = 300
    for _ in range(8):
        tr.train_step(it)
        it += 1
    torch.cuda.synchronize()
    lib = _lib.get_lib()
    buf = (ctypes.c_longlong * (3 * 64 * 24))()
    rc = lib.nsim_debug_ktime(buf)
    assert rc == 0, rc
    t = torch.tensor(list(buf), dtype=torch.float64).view(3, 64, 24)
    out = {}
    for K in (0, 1, 2):
        x = t[K]
        if float(x.abs().sum()) == 0:
            continue
        d = (x[:, 1:18] - x[:, 0:17])
        lab = LABELS.get(K, [str(i) for i in range(18)])
        rec = {lab[i + 1] if i + 1 < len(lab) else str(i + 1): round(float(d[:, i].mean()), 1) for i in range(17)}
        rec["one group total"] = round(float((x[:, 17] - x[:, 0]).mean()), 1)
        rec["prologue (weights -> LDS)"] = round(float((x[:, 20] - x[:, 23]).mean()), 1)
        rec["all groups"] = round(float((x[:, 21] - x[:, 20]).mean()), 1)
        rec["flush"] = round(float((x[:, 22] - x[:, 21]).mean()), 1)
        out[f"kernel{K}"] = rec
        print(f"kernel {K} (s_memtime ticks = 100 MHz? see MICROARCH; mean over 64 workgroups)")
        for k, v in rec.items():
            print(f"  {k:32s} {v:10.1f}")
    (ROOT / "gpurun_out").mkdir(exist_ok=True)
    (ROOT / "gpurun_out" / "ktime.json").write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
