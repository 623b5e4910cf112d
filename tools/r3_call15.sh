#!/bin/bash
# GPU call 15: sampling-decoder grid sweep + new forward grid, on the bench
R=/root/repo; O=$R/gpurun_out; mkdir -p $O; cd $R
A="--steps 32 --warmup 16 --no-cpu-baseline --no-parity --no-variants"
for g in 2048 1024 512 256; do
  NSIM_SDF_GRID=$g python bench.py $A > $O/c15_sdfgrid_$g.json 2>>$O/c15.err
done
for g in 2048 1024 512 256; do python - <<PY
import json
d=json.loads(open("$O/c15_sdfgrid_$g.json").read().strip().splitlines()[-1])
print("sdf_grid $g", d["value"], d["ms_per_step"], {k:v["avg_ms"] for k,v in d["kernels"].items()})
PY
done
