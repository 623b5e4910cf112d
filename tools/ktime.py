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

_KT = ROOT / "neuralsim_amd" / "csrc" / "_probe" / "libnsim_hip_ktime.so"      # a -DNSIM_KTIME build next to the product library
if _KT.exists():
    _lib.LIB_PATH = _KT

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
    t1 = None
    if hasattr(lib, "nsim_debug_ktime1"):
        buf1 = (ctypes.c_longlong * (3 * 64 * 24))()
        assert lib.nsim_debug_ktime1(buf1) == 0
        t1 = torch.tensor(list(buf1), dtype=torch.float64).view(3, 64, 24)
    if hasattr(lib, "nsim_debug_kiter"):        # loop-top stamps of every iteration: is the per-group time constant over a launch?
        bufi = (ctypes.c_longlong * (3 * 64 * 32))()
        assert lib.nsim_debug_kiter(bufi) == 0
        ti = torch.tensor(list(bufi), dtype=torch.float64).view(3, 64, 32)
        for K in (0, 1, 2):
            for b in (0, 1, 3, 31):
                row = ti[K, b]
                row = row[row > 0]
                if row.numel() > 1:
                    print(f"kernel {K} wg {b}: iteration lengths (ticks):", " ".join(f"{float(v):.0f}" for v in (row[1:] - row[:-1])))
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
        first = {}
        if t1 is not None and float(t1[K].abs().sum()) > 0:
            y = t1[K]
            d1 = (y[:, 1:18] - y[:, 0:17])
            first = {lab[i + 1] if i + 1 < len(lab) else str(i + 1): round(float(d1[:, i].mean()), 1) for i in range(17)}
            first["one group total"] = round(float((y[:, 17] - y[:, 0]).mean()), 1)
            first["prologue (weights -> LDS)"] = round(float((y[:, 0] - x[:, 20]).mean()), 1)      # loop entry -> first stamp
            rec["launch skew (first stamp of a workgroup - earliest)"] = round(float((x[:, 23] - x[:, 23].min()).mean()), 1)
        out[f"kernel{K}"] = dict(second_group=rec, first_group=first)
        if t1 is not None:      # raw timeline of a few workgroups, ticks since the earliest kernel-entry stamp
            y = t1[K]
            ent = x[:, 23]
            base = float(ent[ent > 0].min()) if bool((ent > 0).any()) else 0.0
            print("  timeline (ticks since the earliest entry):  wg  entry  prologue-done  g1-start  g1-end  g2-start  g2-end  loop-done  flush-done")
            for b in (0, 1, 2, 3, 31, 63):
                row = [x[b, 23], x[b, 20], y[b, 0], y[b, 17], x[b, 0], x[b, 17], x[b, 21], x[b, 22]]
                print("   ", b, " ".join(f"{float(v) - base:10.0f}" for v in row))
            out[f"kernel{K}"]["timeline"] = {str(b): [float(v) - base for v in (x[b, 23], x[b, 20], y[b, 0], y[b, 17], x[b, 0], x[b, 17], x[b, 21], x[b, 22])] for b in range(64)}
        print(f"kernel {K} (s_memtime ticks; mean over 64 workgroups)        second group   first group")
        for k, v in rec.items():
            f = first.get(k)
            print(f"  {k:52s} {v:10.1f} " + (f"{f:12.1f}" if f is not None else ""))
    (ROOT / "gpurun_out").mkdir(exist_ok=True)
    (ROOT / "gpurun_out" / "ktime.json").write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
