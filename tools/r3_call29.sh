#!/bin/bash
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_pack_ops.py tests/test_compose.py -m gpu -q > $O/c29_tests.log 2>&1
timeout 900 python -m pytest tests/test_fullsize_configs.py -m gpu -q -k "multi" >> $O/c29_tests.log 2>&1
grep -E "passed|failed|^FAILED" $O/c29_tests.log
python bench.py --config multi --steps 8 --warmup 4 --no-cpu-baseline --no-parity --no-variants > $O/c29_multi.json 2>$O/c29.err
python - <<PY
import json
d=json.loads(open("$O/c29_multi.json").read().strip().splitlines()[-1])
print("multi", d["value"], d["ms_per_step"], d.get("step_ms"))
PY
