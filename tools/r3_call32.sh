#!/bin/bash
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
python bench.py --steps 64 --warmup 16 --no-cpu-baseline --no-parity --no-variants > $O/c32_object.json 2>$O/c32.err
python - <<PY
import json
d=json.loads(open("$O/c32_object.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("step_ms"), {k:v["avg_ms"] for k,v in d["kernels"].items()})
PY
