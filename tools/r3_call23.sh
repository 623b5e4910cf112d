#!/bin/bash
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
rm -f $O/c23.jsonl
for g in 256 512 1024; do echo "fwd_grid $g" >> $O/c23.jsonl; NSIM_FWD_GRID=$g python tools/field_bench.py --shape street >> $O/c23.jsonl 2>>$O/c23.err; done
for g in 256 512 1024; do echo "bwd_grid $g" >> $O/c23.jsonl; NSIM_SDF_BWD_GRID=$g python tools/field_bench.py --shape street >> $O/c23.jsonl 2>>$O/c23.err; done
cut -c1-250 $O/c23.jsonl
python bench.py --config street --steps 8 --warmup 4 --no-cpu-baseline --no-parity --no-variants > $O/c23_street.json 2>>$O/c23.err
python - <<PY
import json
d=json.loads(open("$O/c23_street.json").read().strip().splitlines()[-1])
print("street", d["value"], d["ms_per_step"], {k:v["avg_ms"] for k,v in d["kernels"].items()})
PY
