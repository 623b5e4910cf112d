#!/bin/bash
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_field.py tests/test_permuto.py -m gpu -x -q -k "levels or 18" > $O/c24_tests.log 2>&1
timeout 900 python -m pytest tests/test_fullsize_configs.py -m gpu -x -q -k "street and fp16" >> $O/c24_tests.log 2>&1
grep -E "passed|failed" $O/c24_tests.log
A="--steps 8 --warmup 4 --no-cpu-baseline --no-parity --no-variants"
python bench.py --config street $A > $O/c24_street.json 2>$O/c24.err
python bench.py --config multi $A > $O/c24_multi.json 2>>$O/c24.err
for f in c24_street c24_multi; do python - <<PY
import json
d=json.loads(open("$O/$f.json").read().strip().splitlines()[-1])
print("$f", d["value"], d["ms_per_step"], {k:v["avg_ms"] for k,v in d["kernels"].items()})
PY
done
