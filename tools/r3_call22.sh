#!/bin/bash
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_field.py -m gpu -x -q -k "relu or fwd_bwd or levels" > $O/c22_tests.log 2>&1
timeout 900 python -m pytest tests/test_fullsize_parity.py -m gpu -x -q -k "permuto" >> $O/c22_tests.log 2>&1
grep -E "passed|failed" $O/c22_tests.log
python tools/field_bench.py --shape street > $O/c22_fb.json 2>$O/c22.err; cut -c1-300 $O/c22_fb.json
A="--steps 8 --warmup 4 --no-cpu-baseline --no-parity --no-variants"
python bench.py --config street $A > $O/c22_street.json 2>>$O/c22.err
python bench.py --config multi $A > $O/c22_multi.json 2>>$O/c22.err
python bench.py --steps 32 --warmup 16 --no-cpu-baseline --no-parity --no-variants > $O/c22_object.json 2>>$O/c22.err
for f in c22_street c22_multi c22_object; do python - <<PY
import json
d=json.loads(open("$O/$f.json").read().strip().splitlines()[-1])
print("$f", d["value"], d["ms_per_step"], {k:v["avg_ms"] for k,v in d["kernels"].items()})
PY
done
