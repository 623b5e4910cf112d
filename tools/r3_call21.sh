#!/bin/bash
# GPU call 21: LDS plane prefetch in the 17..32-level forward decoder: A/B on the street shape + tests
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_field.py tests/test_permuto.py -m gpu -x -q -k "levels or pose or relu or 18 or fwd_bwd" > $O/c21_tests.log 2>&1
timeout 900 python -m pytest tests/test_fullsize_parity.py -m gpu -x -q -k "permuto" >> $O/c21_tests.log 2>&1
grep -E "passed|failed" $O/c21_tests.log
rm -f $O/c21_fb.jsonl
for g in 1 0 1 0; do NSIM_FWD_GL2=$g python tools/field_bench.py --shape street >> $O/c21_fb.jsonl 2>>$O/c21.err; done
cut -c1-300 $O/c21_fb.jsonl
python bench.py --config street --steps 8 --warmup 4 --no-cpu-baseline --no-parity --no-variants > $O/c21_street.json 2>>$O/c21.err
python - <<PY
import json
d=json.loads(open("$O/c21_street.json").read().strip().splitlines()[-1])
print("street", d["value"], d["ms_per_step"], {k:v["avg_ms"] for k,v in d["kernels"].items()})
PY
