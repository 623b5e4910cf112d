"""Measurement aid: how much of the with-grad decoder kernels is the softplus / sigmoid arithmetic?

Builds a SECOND library (neuralsim_amd/csrc/_probe/libnsim_hip.so, -DNSIM_PROBE_CHEAP_ACT: transcendental-free stand-in
activations in every decoder kernel but the sampling pass's k_field_sdf) and runs the bench on it; compare the
per-kernel times of its JSON line with the product build's.  The difference bounds what ANY cheaper activation
(packed-f16 minimax polynomial ...) could win.  Never part of the product path.

    python tools/act_probe.py build          # here (hipcc cross-compiles)
    python tools/act_probe.py run [bench args]   # on the GPU box
"""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CS = ROOT / "neuralsim_amd" / "csrc"
PROBE = CS / "_probe"


def build():
    sys.path.insert(0, str(ROOT))
    from neuralsim_amd.csrc import build as b
    b.build()                                   # product objects are reused for every file but field.hip
    PROBE.mkdir(exist_ok=True)
    obj = PROBE / "field.hip.o"
    subprocess.check_call([b._hipcc(), *b.HIPCC_FLAGS, "-DNSIM_PROBE_CHEAP_ACT", "-c", str(CS / "field.hip"), "-o", str(obj)])
    objs = [str(obj) if s == "field.hip" else str(b.BUILD / (s + ".o")) for s in b.SOURCES]
    subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", str(PROBE / "libnsim_hip.so")])
    print(PROBE / "libnsim_hip.so")


def run(argv):
    sys.path.insert(0, str(ROOT))
    import neuralsim_amd._lib as L
    L.LIB_PATH = PROBE / "libnsim_hip.so"
    assert L.LIB_PATH.exists(), "python tools/act_probe.py build first"
    os.environ["NSIM_SKIP_BUILD"] = "1"
    import bench
    sys.argv = ["bench.py", *argv]
    bench.main()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    else:
        run(sys.argv[2:] if len(sys.argv) > 1 and sys.argv[1] == "run" else sys.argv[1:])
