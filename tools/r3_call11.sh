#!/bin/bash
# GPU call 11: does pulling a kernel's own code into L2 in its prologue shorten the cold first pass?
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
rm -f $O/c11_*.jsonl
for pf in 0 16384 32768 0 32768; do
  NSIM_CODE_PREFETCH=$pf python tools/field_bench.py --shape object --rays 8192 --per-ray 38 --iters 12 >> $O/c11_pf.jsonl 2>>$O/c11.err
done
NSIM_CODE_PREFETCH=32768 python tools/field_bench.py --shape object --rays 1024 --per-ray 38 --iters 12 >> $O/c11_pf.jsonl 2>>$O/c11.err
A="--steps 32 --warmup 16 --no-cpu-baseline --no-parity --no-variants"
python bench.py $A > $O/c11_object_pf0.json 2>>$O/c11.err
NSIM_CODE_PREFETCH=32768 python bench.py $A > $O/c11_object_pf32k.json 2>>$O/c11.err
cat $O/c11_pf.jsonl | cut -c1-330
for f in c11_object_pf0 c11_object_pf32k; do python - <<PY
import json
d=json.loads(open("$O/$f.json").read().strip().splitlines()[-1])
print("$f", d["value"], d["ms_per_step"], {k:v["avg_ms"] for k,v in d["kernels"].items()})
PY
done
