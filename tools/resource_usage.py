#!/usr/bin/env python
"""Per-kernel register / scratch / occupancy numbers of a .hip source as the gfx950 backend reports them
(-Rpass-analysis=kernel-resource-usage, device-only compile; no GPU needed).  Usage: tools/resource_usage.py field.hip [filter]"""
import os
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CS = ROOT / "neuralsim_amd" / "csrc"


def main():
    src = CS / sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    cmd = ["/opt/rocm/bin/hipcc", *os.environ.get("RU_FLAGS", "").split(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", f"-I{CS}",
           "-munsafe-fp-atomics", "--cuda-device-only", "-c", str(src), "-o", "/tmp/_ru.o",
           "-Rpass-analysis=kernel-resource-usage"]
    txt = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur, rec = None, {}
    for line in txt.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            rec[cur] = {}
        for key, short in (("VGPRs:", "vgpr"), ("AGPRs:", "agpr"), ("ScratchSize [bytes/lane]:", "scratch"),
                           ("Occupancy [waves/SIMD]:", "occ"), ("LDS Size [bytes/block]:", "lds")):
            if key in line and cur and "Spill" not in line:
                rec[cur][short] = line.split(key)[1].split()[0]
    for k, v in rec.items():
        if flt in k:
            name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
            print(f"{name[:70]:70s} {v}")


if __name__ == "__main__":
    main()
