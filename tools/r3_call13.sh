#!/bin/bash
# GPU call 13: weight-gradient replicas (flush storm) A/B
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_field.py -m gpu -x -q -k "replicas or fwd_bwd or levels" > $O/c13_tests.log 2>&1
rm -f $O/c13_fb.jsonl
for r in 16 1 16 1; do
  NSIM_GRAD_REPLICAS=$r python tools/field_bench.py --shape object --rays 8192 --per-ray 38 --iters 12 >> $O/c13_fb.jsonl 2>>$O/c13.err
done
NSIM_GRAD_REPLICAS=16 python tools/field_bench.py --shape street >> $O/c13_fb.jsonl 2>>$O/c13.err
A="--steps 32 --warmup 16 --no-cpu-baseline --no-parity --no-variants"
NSIM_GRAD_REPLICAS=1 python bench.py $A > $O/c13_object_r1.json 2>>$O/c13.err
python bench.py $A > $O/c13_object_r16.json 2>>$O/c13.err
NSIM_GRAD_REPLICAS=1 python bench.py $A > $O/c13_object_r1b.json 2>>$O/c13.err
python bench.py $A > $O/c13_object_r16b.json 2>>$O/c13.err
tail -3 $O/c13_tests.log
cut -c1-330 $O/c13_fb.jsonl
for f in c13_object_r1 c13_object_r16 c13_object_r1b c13_object_r16b; do python - <<PY
import json
d=json.loads(open("$O/$f.json").read().strip().splitlines()[-1])
print("$f", d["value"], d["ms_per_step"], {k:v["avg_ms"] for k,v in d["kernels"].items()})
PY
done
