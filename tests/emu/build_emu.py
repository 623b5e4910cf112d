"""Build libnsim_emu.so: the SAME kernel sources as the product, compiled for the host against the test-only
SIMT emulator (tests/emu/hip_emu.h).  TEST INFRASTRUCTURE -- the product loader cannot load this library."""
import hashlib
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE.parent.parent / "neuralsim_amd" / "csrc"
SOURCES = ["pack_ops.hip", "sampling.hip", "lotd.hip", "field.hip", "wide_field.hip", "permuto.hip", "nerf_field.hip", "sky.hip", "loss_ops.hip", "optim.hip", "misc.hip"]
import os
# NSIM_EMU_EXTRA_FLAGS="-fsanitize=address -g" (with LD_PRELOAD of the ASan runtime) builds an instrumented emulator
# into its own directory: out-of-bounds accesses of the kernels then fail the emulator tests
EXTRA = os.environ.get("NSIM_EMU_EXTRA_FLAGS", "").split()
LIB = HERE / ("_build" + ("_x" + hashlib.sha256(" ".join(EXTRA).encode()).hexdigest()[:8] if EXTRA else "")) / "libnsim_emu.so"
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
# -I tests/emu comes FIRST: its nsim_prims.h (the emulator's primitives) shadows the gfx950 one of the product tree
FLAGS = ["-x", "c++", "-std=c++17", "-O2", "-fPIC", "-ffp-contract=off", f"-I{HERE}", f"-I{CSRC}",
         "-Wno-unused-value", "-Wno-unknown-pragmas", "-Wno-pass-failed"]


def build(force=False):
    LIB.parent.mkdir(exist_ok=True)
    h = hashlib.sha256(" ".join(EXTRA).encode())
    for f in [CSRC / s for s in SOURCES] + [CSRC / "nsim_common.h", CSRC / "lotd_dev.h", CSRC / "mfma_mlp.h", CSRC / "occ_dev.h", HERE / "hip_emu.h", HERE / "nsim_prims.h",
                                            HERE / "hip_emu.cpp", CSRC.parent.parent / "include" / "nsim.h"]:
        h.update(f.read_bytes())
    stamp = LIB.parent / "stamp"
    if not force and LIB.exists() and stamp.exists() and stamp.read_text() == h.hexdigest():
        return LIB
    objs, procs = [], []
    for s in SOURCES + ["hip_emu.cpp"]:
        src = (CSRC / s) if s.endswith(".hip") else (HERE / s)
        obj = LIB.parent / (s + ".o")
        objs.append(str(obj))
        procs.append((s, subprocess.Popen([CLANG, *FLAGS, *EXTRA, "-c", str(src), "-o", str(obj)], stdout=subprocess.PIPE,
                                          stderr=subprocess.STDOUT, text=True)))
    bad = False
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            bad = True
            sys.stderr.write(f"[emu build] FAILED {s}\n{out}\n")
    if bad:
        raise RuntimeError("emulator build failed")
    subprocess.check_call([CLANG, "-shared", "-fPIC", *EXTRA, *objs, "-o", str(LIB)])
    stamp.write_text(h.hexdigest())
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
