#!/usr/bin/env python
"""Development aid: run any script of this repo against a VARIANT of the HIP library built by tools/variant.py.

    python tools/variant_run.py NAME tools/field_bench.py --shape vehicle ...      # on the GPU box
"""
import runpy
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import neuralsim_amd._lib as L      # noqa: E402

L.LIB_PATH = ROOT / "neuralsim_amd" / "csrc" / "_probe" / sys.argv[1] / "libnsim_hip.so"
assert L.LIB_PATH.exists(), f"python tools/variant.py build {sys.argv[1]} ... first"
script = sys.argv[2]
sys.argv = [script, *sys.argv[3:]]
runpy.run_path(script, run_name="__main__")
