#!/bin/bash
# GPU call 19: permuto full-size parity, permuto tests, bench with the permuto variant (no CPU baseline)
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_fullsize_parity.py -m gpu -x -q -k "permuto" > $O/c19_tests.log 2>&1
timeout 600 python -m pytest tests/test_permuto.py tests/test_shim.py -m gpu -x -q >> $O/c19_tests.log 2>&1
grep -E "passed|failed|Error|assert" $O/c19_tests.log | tail -8
python bench.py --no-cpu-baseline > $O/c19_bench.json 2>$O/c19_bench.err
python - <<PY
import json
d=json.loads(open("$O/c19_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], json.dumps(d.get("variants")))
PY
ls $O | grep parity_fullsize_permuto
