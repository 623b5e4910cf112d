"""Build libnsim_hip.so (gfx950) from the .hip sources in this directory with hipcc.

In-tree build (the .so travels with the repo snapshot to the GPU box; it is git-ignored).
    python -m neuralsim_amd.csrc.build [--force]
"""
import hashlib
import os
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
SOURCES = ["pack_ops.hip", "sampling.hip", "lotd.hip", "field.hip", "wide_field.hip", "permuto.hip", "nerf_field.hip", "sky.hip", "loss_ops.hip", "optim.hip", "misc.hip"]
HEADERS = ["nsim_common.h", "nsim_prims.h", "lotd_dev.h", "mfma_mlp.h", "occ_dev.h", "../../include/nsim.h"]
LIB = HERE / "libnsim_hip.so"
BUILD = HERE / "_build"

# -ffp-contract=off: sample-membership arithmetic must round exactly like the oracle (mul then add);
# the hot arithmetic lives on the MFMA pipe, not in contracted VALU FMAs.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", f"-I{HERE}",
               "-munsafe-fp-atomics", "-Wno-unused-result"] + os.environ.get("NSIM_EXTRA_HIPCC_FLAGS", "").split()


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _digest():
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        h.update((HERE / f).read_bytes())
    h.update(" ".join(HIPCC_FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> Path:
    BUILD.mkdir(exist_ok=True)
    stamp = BUILD / "stamp"
    dig = _digest()
    if not force and LIB.exists() and stamp.exists() and stamp.read_text() == dig:
        return LIB
    hipcc = _hipcc()
    procs = []
    objs = []
    for src in SOURCES:
        obj = BUILD / (src + ".o")
        objs.append(str(obj))
        cmd = [hipcc, *HIPCC_FLAGS, "-c", str(HERE / src), "-o", str(obj)]
        if verbose:
            print("[nsim build]", " ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"[nsim build] FAILED {src}\n{out}\n")
        elif verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError("hipcc failed")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", str(LIB)]
    if verbose:
        print("[nsim build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    stamp.write_text(dig)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
