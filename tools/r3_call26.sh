#!/bin/bash
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_fullsize_parity.py -m gpu -q -k "permuto" > $O/c26_tests.log 2>&1
grep -E "passed|failed|^FAILED" $O/c26_tests.log
python bench.py > $O/final_bench.json 2>$O/final_bench.err
python - <<PY
import json
d=json.loads(open("$O/final_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("step_ms"), json.dumps(d.get("variants")), json.dumps(d.get("cpu_baseline"))[:160])
PY
