#!/usr/bin/env python
"""Development aid: build / run the bench on a VARIANT of the HIP library compiled with extra -D flags (A/B of compile-time
tunables on one GPU box in one gpurun call).  The product library is untouched; variants live under csrc/_probe/<name>/.

    python tools/variant.py build NAME "-DNSIM_SDF_NBUF=1 ..." [file.hip ...]   # here (hipcc cross-compiles); default: field.hip
    python tools/variant.py run NAME [bench args]                                  # on the GPU box
"""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CS = ROOT / "neuralsim_amd" / "csrc"


def build(name, flags, files):
    sys.path.insert(0, str(ROOT))
    from neuralsim_amd.csrc import build as b
    b.build(verbose=False)
    out = CS / "_probe" / name
    out.mkdir(parents=True, exist_ok=True)
    procs = []
    for f in files:
        procs.append(subprocess.Popen([b._hipcc(), *b.HIPCC_FLAGS, *flags.split(), "-c", str(CS / f), "-o", str(out / (f + ".o"))]))
    assert all(p.wait() == 0 for p in procs)
    objs = [str(out / (s + ".o")) if s in files else str(b.BUILD / (s + ".o")) for s in b.SOURCES]
    subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", str(out / "libnsim_hip.so")])
    for f in files:
        (out / (f + ".o")).unlink()
    print(out / "libnsim_hip.so")


def run(name, argv):
    sys.path.insert(0, str(ROOT))
    import neuralsim_amd._lib as L
    L.LIB_PATH = CS / "_probe" / name / "libnsim_hip.so"
    assert L.LIB_PATH.exists(), f"python tools/variant.py build {name} ... first"
    os.environ["NSIM_SKIP_BUILD"] = "1"
    import bench
    sys.argv = ["bench.py", *argv]
    bench.main()


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2], sys.argv[3], sys.argv[4:] or ["field.hip"])
    else:
        run(sys.argv[2], sys.argv[3:])
