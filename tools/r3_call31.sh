#!/bin/bash
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
python bench.py --no-cpu-baseline > $O/c31_bench.json 2>$O/c31.err
python - <<PY
import json
d=json.loads(open("$O/c31_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("step_ms"), json.dumps(d.get("variants")))
PY
