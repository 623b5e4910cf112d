#!/bin/bash
# GPU call 8: dead-plane-level skip + workgroup-joint SDF backward for 17..32-level pyramids (street / multi configs)
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_field.py -m gpu -x -q -k "levels or pose" > $O/c8_tests.log 2>&1
timeout 900 python -m pytest tests/test_fullsize_configs.py -m gpu -x -q -k "street" >> $O/c8_tests.log 2>&1
A="--config street --steps 8 --warmup 4 --no-cpu-baseline --no-parity --no-variants"
python bench.py $A > $O/c8_street_new.json 2>$O/c8.err
NSIM_SDF_BWD_OLD=1 python bench.py $A > $O/c8_street_oldbwd.json 2>>$O/c8.err
python bench.py --config multi --steps 8 --warmup 4 --no-cpu-baseline --no-parity --no-variants > $O/c8_multi_new.json 2>>$O/c8.err
python bench.py --steps 32 --warmup 16 --no-cpu-baseline --no-parity --no-variants > $O/c8_object.json 2>>$O/c8.err
grep -E "passed|failed|error" $O/c8_tests.log | tail -4
for f in c8_street_new c8_street_oldbwd c8_multi_new c8_object; do python - <<PY
import json
d=json.loads(open("$O/$f.json").read().strip().splitlines()[-1])
print("$f", d["value"], d["ms_per_step"], {k:v["avg_ms"] for k,v in d["kernels"].items()})
PY
done
